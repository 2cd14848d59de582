// Backward of the memory READ (first slice): gradients of the three `map_merge_projection` 1x1 convolutions and of
// the cascaded average pools of `CustomRecurrentFPN.forward` (Detic/detic/modeling/backbone/timm.py:142-192), the part of
// `forward_model` (custom_rcnn.py:584-679) that is specific to the spatial memory.  Forward, per level l = 3, 4, 5:
//
//   E_l = half(avg_pool2(float(E_{l-1})))          E_2 := avg_pool4(float(memory[proj]))  (fp32)
//   out_l = (conv1x1(float(E_l); W_l, b_l) * weight) + P_l
//
// Given G_l = dL/d out_l ([P_l, 256] rows, NHWC):
//   dW_l[co][ci] = weight * sum_pos G_l[pos][co] * E_l[pos][ci]      db_l[co] = weight * sum_pos G_l[pos][co]         (kernel 1)
//   dEc_l[pos][ci] = weight * sum_co G_l[pos][co] * W_l[co][ci]       a 1x1 convolution with the transposed weights: eod_conv2d
//   pools (autograd semantics of the fp16 casts: a gradient that flows into a half tensor is rounded to half, two contributions
//   to one half tensor are added in half):                                                                              (kernel 2)
//     gE_5 = half(dEc_5)
//     gE_4 = half(dEc_4) +h half(up2(float(gE_5)) / 4)
//     gE_3 = half(dEc_3) +h half(up2(float(gE_4)) / 4)
//     gE_2 = up2(float(gE_3)) / 4                                    (fp32, [H/4 * W/4, 512])
// The memory table itself is an input of the reference's training step (loaded from disk, loader.py:199-223), not a parameter: no
// gradient flows below E_2.
//
// Kernel 1 needs no LDS staging: the MFMA operand layout of v_mfma_f32_32x32x2_f32 (lane l supplies row l % 32, k = l / 32) reads
// 32 consecutive channels of one position per half wave -- coalesced as the rows lie in memory.  A workgroup owns one 32x32 tile
// of dW_l; its four waves take a quarter of the positions each and are added in wave order (deterministic).
#include "eod_common.h"
#include <algorithm>
#include "../../include/eod_hip.h"
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdlib>

namespace {

// pooled rows in the operand-fragment order eod_memory_gather_pool writes (include/eod_hip.h):
// [32-row tile][k-step s<32][hi<2][r<32][8] halves = element (row 32*tile + r, channel 16 s + 8 hi + j)
__device__ __forceinline__ size_t frag_half_offset(int tile32, int r, int c8) {
  return ((((size_t)tile32 * 32 + (c8 >> 1)) * 2 + (c8 & 1)) * 32 + r) * 8;
}

struct BwdArgs {
  const float* g[3];        // G_l [P_l, 256]
  const __half* pooled;     // E_3..E_5, fragment order, each level on a tile boundary
  float* dw[3];             // [256, 512]
  float* db[3];             // [256]
  int rows[3];              // P_l
  int tile_base[3];         // first 32-row tile of level l in `pooled`
  float weight;
  // splits > 1 (blockIdx.z): the positions of a level are cut into `splits` ranges, each writing its own partial dW / db into
  // part [level][splits][256 * 512] / bpart [level][splits][256]; wgrad_reduce_kernel adds them in range order.  Without it the 128
  // workgroups of the finest level each walk all of its 6 400 positions (290 us at 640x640).
  int splits;
  float* part;
  float* bpart;
};

__global__ __launch_bounds__(256) void proj_backward_weights_kernel(BwdArgs a) {
  const int level = blockIdx.y;
  const int tile = blockIdx.x;              // 8 (co) x 16 (ci) tiles of 32x32
  const int co0 = (tile >> 4) * 32, ci0 = (tile & 15) * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = a.rows[level];
  const float* __restrict__ G = a.g[level];
  const int col = lane & 31, kh = lane >> 5;
  // positions of this workgroup's range, then of this wave: a contiguous quarter, rounded to whole k-steps of 8 positions
  const int steps = (P + 7) / 8;
  const int sps = (steps + a.splits - 1) / a.splits;
  const int z_begin = (int)blockIdx.z * sps;
  const int z_end = min(z_begin + sps, steps);
  const int spw = (sps + 3) / 4;
  const int s_begin = z_begin + wave * spw;
  int s_end = s_begin + spw;
  if (s_end > z_end) s_end = z_end;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int ci = ci0 + col;
  for (int s = s_begin; s < s_end; ++s) {
    float av[4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = s * 8 + 2 * t + kh;       // instruction t contracts positions 8 s + 2 t and 8 s + 2 t + 1
      const bool ok = pos < P;
      av[t] = ok ? G[(size_t)pos * 256 + co0 + col] : 0.f;
      bv[t] = ok ? __half2float(a.pooled[frag_half_offset(a.tile_base[level] + (pos >> 5), pos & 31, ci >> 3) + (ci & 7)]) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
      bsum += av[t];
    }
  }
  __shared__ float red[3 * 16 * 64];
  __shared__ float bred[4 * 64];
  bred[wave * 64 + lane] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[((w - 1) * 16 + r) * 64 + lane];
  // C/D layout: column (ci) = lane & 31, row (co) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* dw = a.splits > 1 ? a.part + ((size_t)level * a.splits + blockIdx.z) * (256 * 512) : a.dw[level];
  float* db = a.splits > 1 ? a.bpart + ((size_t)level * a.splits + blockIdx.z) * 256 : a.db[level];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    dw[(size_t)co * 512 + ci] = acc[r] * a.weight;
  }
  if ((tile & 15) == 0 && lane < 32) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += bred[w * 64 + lane] + bred[w * 64 + 32 + lane];    // even + odd positions, wave order
    db[co0 + lane] = v * a.weight;
  }
}

struct PoolBwdArgs {
  const float* dec[3];      // dEc_3, dEc_4, dEc_5: [P_l, 512] fp32 rows
  __half* ge[3];            // gE_3, gE_4, gE_5: [P_l, 512] half rows
  float* ge2;               // [(H/4) * (W/4), 512] fp32
  int h2, w2;               // H / 4, W / 4
};

__device__ __forceinline__ float round_half(float v) { return __half2float(__float2half_rn(v)); }

// one thread per (position of the H/4 x W/4 grid, channel): it recomputes the chain of its P5 / P4 / P3 ancestors (pointwise) and
// stores the levels whose top-left corner it is
__global__ __launch_bounds__(256) void pool_backward_kernel(PoolBwdArgs a) {
  const size_t total = (size_t)a.h2 * a.w2 * 512;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i & 511);
    const int p = (int)(i >> 9);
    const int y = p / a.w2, x = p - y * a.w2;
    const int w3 = a.w2 >> 1, w4 = a.w2 >> 2, w5 = a.w2 >> 3;
    const size_t p3 = (size_t)(y >> 1) * w3 + (x >> 1), p4 = (size_t)(y >> 2) * w4 + (x >> 2), p5 = (size_t)(y >> 3) * w5 + (x >> 3);
    const float g5 = round_half(a.dec[2][p5 * 512 + c]);
    const float g4 = round_half(round_half(a.dec[1][p4 * 512 + c]) + round_half(g5 * 0.25f));
    const float g3 = round_half(round_half(a.dec[0][p3 * 512 + c]) + round_half(g4 * 0.25f));
    a.ge2[i] = g3 * 0.25f;
    if (((y | x) & 1) == 0) a.ge[0][p3 * 512 + c] = __float2half_rn(g3);
    if (((y | x) & 3) == 0) a.ge[1][p4 * 512 + c] = __float2half_rn(g4);
    if (((y | x) & 7) == 0) a.ge[2][p5 * 512 + c] = __float2half_rn(g5);
  }
}

// torch.optim.AdamW, single-tensor form (the reference trains with SOLVER.OPTIMIZER ADAMW, Base-C2_L_R5021k_640b64_4x_recurrent.yaml:69),
// preceded by detectron2's per-parameter clip_grad_value_ (SOLVER.CLIP_GRADIENTS.ENABLED, CLIP_TYPE "value"):
//   g = clamp(g, -clip, clip);  p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;
//   p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps)),   step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t) (host, double)
__global__ __launch_bounds__(256) void adamw_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, size_t n, float decay, float one_minus_b1, float b2,
                                                          float one_minus_b2, float step_size, float bc2_sqrt, float eps, float clip) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    float pi = p[i] * decay;
    float mi = m[i];
    mi = mi + (gi - mi) * one_minus_b1;
    const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi + (-step_size) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// The same update for up to ADAMW_MULTI tensors in ONE launch (the training step has 126 parameter tensors: 126 launches of a few
// microseconds each were 4 % of the iteration's kernel time and 126 launch boundaries).  The tensors' constants travel in the
// kernel arguments; a workgroup finds its tensor by the prefix of the tensors' workgroup counts.  Element for element the
// arithmetic of adamw_step_kernel.
#define ADAMW_MULTI 20
struct AdamWMulti {
  float* p[ADAMW_MULTI];
  const float* g[ADAMW_MULTI];
  float* m[ADAMW_MULTI];
  float* v[ADAMW_MULTI];
  unsigned long long n[ADAMW_MULTI];
  float decay[ADAMW_MULTI], step_size[ADAMW_MULTI], bc2_sqrt[ADAMW_MULTI];
  unsigned block_end[ADAMW_MULTI];     // exclusive end of tensor t's workgroups
  // optional: the stepped value times a per-row factor, written to a second matrix (a trunk conv's raw master -> the layer's weights
  // with its FrozenBatchNorm folded in: timm.py:277-299 keeps weight / bias of the norm as buffers, not parameters)
  float* folded[ADAMW_MULTI];
  const float* row_scale[ADAMW_MULTI];
  unsigned cols[ADAMW_MULTI], ld_out[ADAMW_MULTI];
  unsigned grad_of_folded;             // bit t: g is the gradient of the FOLDED weights (x row_scale = the master's, chain rule)
  int count;
  float one_minus_b1, b2, one_minus_b2, eps, clip;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamWMulti a) {
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.block_end[t]) ++t;
  const unsigned first = t ? a.block_end[t - 1] : 0u;
  const size_t nb = a.block_end[t] - first;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const size_t n = a.n[t];
  const float decay = a.decay[t], step_size = a.step_size[t], bc2_sqrt = a.bc2_sqrt[t];
  float* __restrict__ folded = a.folded[t];
  const float* __restrict__ rs = a.row_scale[t];
  const unsigned cols = a.cols[t], ldo = a.ld_out[t];
  const bool gscale = (a.grad_of_folded >> t) & 1u;
  for (size_t i = (size_t)(blockIdx.x - first) * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) {
    float gi = g[i];
    if (gscale) gi *= rs[(unsigned)(i / cols)];
    if (a.clip > 0.f) gi = fminf(fmaxf(gi, -a.clip), a.clip);
    float pi = p[i] * decay;
    float mi = m[i];
    mi = mi + (gi - mi) * a.one_minus_b1;
    const float vi = v[i] * a.b2 + a.one_minus_b2 * (gi * gi);
    const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
    pi = pi + (-step_size) * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (folded) {
      const unsigned r = (unsigned)(i / cols), c = (unsigned)(i - (size_t)r * cols);
      folded[(size_t)r * ldo + c] = pi * rs[r];
    }
  }
}

// Third slice: backward of a stride-1 'same' convolution layer (FPN output convs timm.py:118-136, CenterNet tower
// centernet_head.py:141-161, mask head convs: KH x KW taps, NHWC).  Weight gradient in the layout of the packed forward weights:
//   dW[co][(ky, kx, ci)] = sum over positions (n, oy, ox) of G[pos][co] * X[n][oy + ky - pad][ox + kx - pad][ci]      (0 outside the image)
//   db[co] = sum over positions of G[pos][co]
// Same scheme as proj_backward_weights_kernel: no LDS staging, the MFMA operand layout of v_mfma_f32_32x32x2_f32 reads 32 consecutive
// channels of one position per half wave (coalesced as the NHWC rows lie in memory); a workgroup owns one 32 (co) x 32 (ci) tile of
// one tap, its four waves take a quarter of the positions each and are added in wave order (deterministic).  The gradient with
// respect to the input is a convolution of G with the 180-degree rotated, in/out-transposed weights: eod_conv2d (ops.ConvBackward).
#define WGRAD_MAX_LEVELS 8
struct ConvBwdArgs {
  const float* x;   // [N,H,W,Cin]
  const float* g;   // [N,H,W,Cout]
  float* dw;        // [Cout][KH*KW*Cin]
  float* db;        // [Cout] or null
  int N, H, W, Cin, Cout, KH, KW, pad, stride, OH, OW;
  FastDiv div_w, div_h;     // by OW, OH
  // splits > 1 (blockIdx.z): the positions are cut into `splits` contiguous ranges, each writing its own partial dW / db into
  // part [splits][Cout * Ktot] / bpart [splits][Cout]; wgrad_reduce_kernel adds them in range order.  Layers with few channel tiles
  // and many positions (the trunk's first stages: 4 .. 64 workgroups otherwise) fill the chip this way.
  int splits;
  float* part;
  float* bpart;
  // pyramid mode of the LDS-tiled kernel (nlv > 0; stride 1, 'same' padding): x / g are row lists, rows [lv_off[l], lv_off[l+1])
  // are an lv_h[l] x lv_w[l] image, the weights are shared by the levels and dW / db are summed over all of them
  int nlv;
  int lv_off[WGRAD_MAX_LEVELS + 1], lv_h[WGRAD_MAX_LEVELS], lv_w[WGRAD_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void conv_backward_weights_kernel(ConvBwdArgs a) {
  const int ci_tiles = a.Cin >> 5;
  const int tile = blockIdx.x;                       // (co tile, ci tile)
  const int co0 = (tile / ci_tiles) * 32, ci0 = (tile % ci_tiles) * 32;
  const int tap = blockIdx.y;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int P = a.N * a.OH * a.OW;            // positions of the OUTPUT grid
  const int steps = (P + 7) / 8;
  const int sps = (steps + a.splits - 1) / a.splits;           // steps of this split
  const int z_begin = blockIdx.z * sps;
  const int z_end = min(z_begin + sps, steps);
  const int spw = (sps + 3) / 4;
  const int s_begin = z_begin + wave * spw;
  int s_end = s_begin + spw;
  if (s_end > z_end) s_end = z_end;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int s = s_begin; s < s_end; ++s) {
    float av[4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = s * 8 + 2 * t + kh;            // instruction t contracts positions 8 s + 2 t and 8 s + 2 t + 1
      float gv = 0.f, xv = 0.f;
      if (pos < P) {
        gv = a.g[(size_t)pos * a.Cout + co0 + col];
        const int row = (int)fdiv((unsigned)pos, a.div_w);           // n * OH + oy
        const int ox = pos - row * a.OW;
        const int n = (int)fdiv((unsigned)row, a.div_h);
        const int oy = row - n * a.OH;
        const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          xv = a.x[((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + ci0 + col];
      }
      av[t] = gv;
      bv[t] = xv;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
      bsum += av[t];
    }
  }
  __shared__ float red[3 * 16 * 64];
  __shared__ float bred[4 * 64];
  bred[wave * 64 + lane] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[((w - 1) * 16 + r) * 64 + lane];
  // C/D layout: column (ci) = lane & 31, row (co) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int Ktot = a.KH * a.KW * a.Cin;
  float* dw = a.splits > 1 ? a.part + (size_t)blockIdx.z * a.Cout * Ktot : a.dw;
  float* db = a.splits > 1 ? a.bpart + (size_t)blockIdx.z * a.Cout : a.db;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    dw[(size_t)co * Ktot + (size_t)tap * a.Cin + ci0 + col] = acc[r];
  }
  if (a.db && tap == 0 && (tile % ci_tiles) == 0 && lane < 32) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += bred[w * 64 + lane] + bred[w * 64 + 32 + lane];    // even + odd positions, wave order
    db[co0 + lane] = v;
  }
}

// The same contraction with a 64 (co) x 64 (ci) register block per wave (round 4).  The 32 x 32 form above issues one MFMA per pair
// of operand loads and redoes the position arithmetic (two divisions, the tap's bounds test) for every load: 35 TFLOP/s, bound by
// instruction issue and by the L1 operand path, not by the matrix cores.  Here a k-step loads G for two 32-channel blocks and X for
// two, with ONE position computation, and feeds four MFMAs: half the loads and a quarter of the address arithmetic per FLOP.  The
// four waves of a workgroup still split the range's positions and are added in wave order through LDS; a layer with 32 channels on
// one side (the 5-channel head padded to 32, bbox_pred.2) runs with the second block switched off.
__global__ __launch_bounds__(256) void conv_backward_weights_rb_kernel(ConvBwdArgs a) {
  const int ci_tiles = (a.Cin + 63) >> 6;
  const int tile = blockIdx.x;                       // (co tile, ci tile) of 64 x 64
  const int co0 = (tile / ci_tiles) * 64, ci0 = (tile % ci_tiles) * 64;
  const bool co2 = co0 + 32 < a.Cout, ci2 = ci0 + 32 < a.Cin;     // workgroup-uniform
  const int tap = blockIdx.y;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int P = a.N * a.OH * a.OW;
  const int steps = (P + 7) / 8;
  const int sps = (steps + a.splits - 1) / a.splits;
  const int z_begin = blockIdx.z * sps;
  const int z_end = min(z_begin + sps, steps);
  const int spw = (sps + 3) / 4;
  const int s_begin = z_begin + wave * spw;
  int s_end = s_begin + spw;
  if (s_end > z_end) s_end = z_end;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  for (int s = s_begin; s < s_end; ++s) {
    float g0[4], g1[4], x0[4], x1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = s * 8 + 2 * t + kh;            // instruction t contracts positions 8 s + 2 t and 8 s + 2 t + 1
      float ga = 0.f, gb = 0.f, xa = 0.f, xb = 0.f;
      if (pos < P) {
        const float* gp = a.g + (size_t)pos * a.Cout + co0 + col;
        ga = gp[0];
        if (co2) gb = gp[32];
        const int row = (int)fdiv((unsigned)pos, a.div_w);           // n * OH + oy
        const int ox = pos - row * a.OW;
        const int n = (int)fdiv((unsigned)row, a.div_h);
        const int oy = row - n * a.OH;
        const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
          const float* xp = a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + ci0 + col;
          xa = xp[0];
          if (ci2) xb = xp[32];
        }
      }
      g0[t] = ga; g1[t] = gb; x0[t] = xa; x1[t] = xb;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[t], x0[t], acc[0][0], 0, 0, 0);
      if (ci2) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0[t], x1[t], acc[0][1], 0, 0, 0);
      if (co2) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[t], x0[t], acc[1][0], 0, 0, 0);
      if (co2 && ci2) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[t], x1[t], acc[1][1], 0, 0, 0);
      bsum0 += g0[t];
      bsum1 += g1[t];
    }
  }
  // waves 1..3 hand their 64 x 64 block to wave 0 through LDS, one 32 x 32 quarter at a time (12 KB), added in wave order
  __shared__ float red[3 * 16 * 64];
  __shared__ float bred[2][4 * 64];
  bred[0][wave * 64 + lane] = bsum0;
  bred[1][wave * 64 + lane] = bsum1;
  const int Ktot = a.KH * a.KW * a.Cin;
  float* dw = a.splits > 1 ? a.part + (size_t)blockIdx.z * a.Cout * Ktot : a.dw;
  float* db = a.splits > 1 ? a.bpart + (size_t)blockIdx.z * a.Cout : a.db;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if ((i && !co2) || (j && !ci2)) continue;      // workgroup-uniform
      __syncthreads();                               // the previous quarter has been consumed
      if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (wave == 0) {
        f32x16 v = acc[i][j];
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += red[((w - 1) * 16 + r) * 64 + lane];
        // C/D layout: column (ci) = lane & 31, row (co) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
          dw[(size_t)co * Ktot + (size_t)tap * a.Cin + ci0 + 32 * j + col] = v[r];
        }
      }
    }
  }
  if (a.db && tap == 0 && (tile % ci_tiles) == 0 && wave == 0 && lane < 32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i && !co2) continue;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += bred[i][w * 64 + lane] + bred[i][w * 64 + 32 + lane];    // even + odd positions, wave order
      db[co0 + 32 * i + lane] = v;
    }
  }
}

// LDS-tiled form (round 4, second step): the contraction index of dW = G^T . X is the POSITION, the slow index of both operands, so
// the register-blocked kernel above still fetches one dword per lane and MFMA and computes an address per operand pair.  Here a
// workgroup (2 x 2 waves, 64 co x 64 ci of one tap) walks its position range in chunks of 32: the chunk's G rows [32][64] and the
// tap-shifted, border-masked X rows [32][64] are fetched with two 16-byte loads per thread each (ONE position computation per load)
// into registers, staged in LDS as they lie ([position][channel]: conflict-free stores), and every MFMA operand is one ds_read_b32
// (lane = channel, half wave = position parity).  The next chunk's global loads are in flight under the 16 MFMAs of the current one.
template <bool LEVELS>
__global__ __launch_bounds__(256) void conv_backward_weights_lds_kernel(ConvBwdArgs a) {
  constexpr int PK = 32, TC = 64;
  __shared__ __attribute__((aligned(16))) float As[PK][TC];
  __shared__ __attribute__((aligned(16))) float Bs[PK][TC];
  __shared__ float bred[16][TC];
  const int ci_tiles = (a.Cin + 63) >> 6;
  const int tile = blockIdx.x;
  const int co0 = (tile / ci_tiles) * 64, ci0 = (tile % ci_tiles) * 64;
  const int tap = blockIdx.y;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, kh = lane >> 5;
  const int P = LEVELS ? a.lv_off[a.nlv] : a.N * a.OH * a.OW;
  const int chunks = (P + PK - 1) / PK;
  const int cps = (chunks + a.splits - 1) / a.splits;
  const int c_begin = blockIdx.z * cps;
  const int c_end = min(c_begin + cps, chunks);
  // loader role: thread -> (position row lp + 16 i, four channels c4 .. c4 + 3)
  const int lp = tid >> 4, c4 = (tid & 15) * 4;
  const bool g_ok = co0 + c4 < a.Cout, x_ok = ci0 + c4 < a.Cin;          // Cout, Cin are multiples of 32 (and of 4)
  f32x4 gr[2], xr[2];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pos = c * PK + lp + 16 * i;
      f32x4 gv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
      if (pos < P) {
        if (g_ok) gv = *reinterpret_cast<const f32x4*>(a.g + (size_t)pos * a.Cout + co0 + c4);
        if (LEVELS) {
          int l = 0;
#pragma unroll
          for (int q = 1; q < WGRAD_MAX_LEVELS; ++q) l += (q < a.nlv && pos >= a.lv_off[q]) ? 1 : 0;
          const int base = a.lv_off[l], lw = a.lv_w[l], lh = a.lv_h[l];
          const int local = pos - base;
          const int oy = local / lw, ox = local - oy * lw;
          const int iy = oy + ky - a.pad, ix = ox + kx - a.pad;
          if (x_ok && (unsigned)iy < (unsigned)lh && (unsigned)ix < (unsigned)lw)
            xv = *reinterpret_cast<const f32x4*>(a.x + ((size_t)base + (size_t)iy * lw + ix) * a.Cin + ci0 + c4);
        } else {
          const int row = (int)fdiv((unsigned)pos, a.div_w);           // n * OH + oy
          const int ox = pos - row * a.OW;
          const int n = (int)fdiv((unsigned)row, a.div_h);
          const int oy = row - n * a.OH;
          const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
          if (x_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
            xv = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.Cin + ci0 + c4);
        }
      }
      gr[i] = gv;
      xr[i] = xv;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  if (c_begin < c_end) load_chunk(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&As[lp + 16 * i][c4]) = gr[i];
      *reinterpret_cast<f32x4*>(&Bs[lp + 16 * i][c4]) = xr[i];
      bsum += gr[i];
    }
    __syncthreads();
    if (c + 1 < c_end) load_chunk(c + 1);
    const float* ap = &As[kh][wm * 32 + r];
    const float* bp = &Bs[kh][wn * 32 + r];
#pragma unroll
    for (int kk = 0; kk < PK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * 2 * TC], bp[kk * 2 * TC], acc, 0, 0, 0);
    __syncthreads();
  }
  // C/D layout: column (ci) = lane & 31, row (co) = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
  const int Ktot = a.KH * a.KW * a.Cin;
  float* dw = a.splits > 1 ? a.part + (size_t)blockIdx.z * a.Cout * Ktot : a.dw;
  float* db = a.splits > 1 ? a.bpart + (size_t)blockIdx.z * a.Cout : a.db;
  const int ci = ci0 + wn * 32 + r;
  if (ci < a.Cin) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = co0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
      if (co < a.Cout) dw[(size_t)co * Ktot + (size_t)tap * a.Cin + ci] = acc[q];
    }
  }
  if (a.db && tap == 0 && (tile % ci_tiles) == 0) {
    // db[co] = sum over the positions: the 16 loader rows' sums, added in row order
    bred[lp][c4 + 0] = bsum.x; bred[lp][c4 + 1] = bsum.y; bred[lp][c4 + 2] = bsum.z; bred[lp][c4 + 3] = bsum.w;
    __syncthreads();
    if (tid < TC && co0 + tid < a.Cout) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += bred[q][tid];
      db[co0 + tid] = v;
    }
  }
}

// dW / db = the partial results of the position ranges added in range order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ dw,
                                                           float* __restrict__ db, size_t n, int Cout, int splits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n + (db ? Cout : 0); i += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < n) {
      for (int z = 0; z < splits; ++z) v += part[(size_t)z * n + i];
      dw[i] = v;
    } else {
      const size_t c = i - n;
      for (int z = 0; z < splits; ++z) v += bpart[(size_t)z * Cout + c];
      db[c] = v;
    }
  }
}

// Weight gradient of a 4-channel layer whose kernel rows are wider than 8 taps (fallback; the stem, timm.py:279, takes the MFMA
// kernel below): tap layout, packed k = (ky, kx, ci), ci < 4.  Workgroup = one tap x 16 output channels; its 16 waves take every 16th output position, lane =
// (co, ci); the 16 partial sums are added in wave order through LDS (deterministic).  Tap 0's workgroups also write db.
__global__ __launch_bounds__(1024) void conv_backward_weights_tap4_kernel(ConvBwdArgs a) {
  __shared__ float part[16][64];
  __shared__ float partb[16][16];
  const int tap = blockIdx.x, co0 = blockIdx.y * 16;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  const int t = threadIdx.x, ci = t & 3, col = (t >> 2) & 15, lane_p = t >> 6;
  const int P = a.N * a.OH * a.OW;
  const int pps = (P + a.splits - 1) / a.splits;
  const int p_begin = blockIdx.z * pps, p_end = min(p_begin + pps, P);
  float acc = 0.f, accb = 0.f;
#pragma unroll 4
  for (int p = p_begin + lane_p; p < p_end; p += 16) {
    const int ox = p % a.OW, r = p / a.OW;
    const int oy = r % a.OH, n = r / a.OH;
    const float gv = a.g[(size_t)p * a.Cout + co0 + col];
    accb += gv;
    const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) acc += gv * a.x[((size_t)(n * a.H + iy) * a.W + ix) * 4 + ci];
  }
  part[lane_p][t & 63] = acc;
  if (ci == 0) partb[lane_p][col] = accb;
  __syncthreads();
  if (t < 64) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += part[i][t];
    const int Ktot = a.KH * a.KW * 4;
    float* dw = a.splits > 1 ? a.part + (size_t)blockIdx.z * a.Cout * Ktot : a.dw;
    float* db = a.splits > 1 ? a.bpart + (size_t)blockIdx.z * a.Cout : a.db;
    dw[(size_t)(co0 + col) * Ktot + tap * 4 + ci] = s;
    if (tap == 0 && ci == 0 && a.db) {
      float sb = 0.f;
      for (int i = 0; i < 16; ++i) sb += partb[i][col];
      db[co0 + col] = sb;
    }
  }
}

// The stem's weight gradient on the matrix cores (KW * 4 <= 32): for one kernel row ky the packed columns (kx, ci) of an output
// position are KW * 4 CONSECUTIVE floats of the 4-channel image row, so a workgroup owns a 32 (co) x 32 (kx, ci) tile of one ky and
// contracts over the positions exactly like conv_backward_weights_kernel (columns >= KW * 4 are computed and dropped).
__global__ __launch_bounds__(256) void conv_backward_weights_tap4_mfma_kernel(ConvBwdArgs a) {
  const int tile = blockIdx.x;                       // (co tile, ky)
  const int co0 = (tile / a.KH) * 32, ky = tile % a.KH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kh = lane >> 5;
  const int kx = col >> 2;
  const bool col_live = kx < a.KW;
  const int P = a.N * a.OH * a.OW;
  const int steps = (P + 7) / 8;
  const int sps = (steps + a.splits - 1) / a.splits;
  const int z_begin = blockIdx.z * sps;
  const int z_end = min(z_begin + sps, steps);
  const int spw = (sps + 3) / 4;
  const int s_begin = z_begin + wave * spw;
  int s_end = s_begin + spw;
  if (s_end > z_end) s_end = z_end;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  for (int s = s_begin; s < s_end; ++s) {
    float av[4], bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pos = s * 8 + 2 * t + kh;
      float gv = 0.f, xv = 0.f;
      if (pos < P) {
        gv = a.g[(size_t)pos * a.Cout + co0 + col];
        const int row = (int)fdiv((unsigned)pos, a.div_w);
        const int ox = pos - row * a.OW;
        const int n = (int)fdiv((unsigned)row, a.div_h);
        const int oy = row - n * a.OH;
        const int iy = oy * a.stride + ky - a.pad, ix = ox * a.stride + kx - a.pad;
        if (col_live && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          xv = a.x[((size_t)(n * a.H + iy) * a.W + ix) * 4 + (col & 3)];
      }
      av[t] = gv;
      bv[t] = xv;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
      bsum += av[t];
    }
  }
  __shared__ float red[3 * 16 * 64];
  __shared__ float bred[4 * 64];
  bred[wave * 64 + lane] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[((w - 1) * 16 + r) * 64 + lane];
  const int Ktot = a.KH * a.KW * 4;
  float* dw = a.splits > 1 ? a.part + (size_t)blockIdx.z * a.Cout * Ktot : a.dw;
  float* db = a.splits > 1 ? a.bpart + (size_t)blockIdx.z * a.Cout : a.db;
  if (col_live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      dw[(size_t)co * Ktot + ky * a.KW * 4 + col] = acc[r];
    }
  }
  if (a.db && ky == 0 && lane < 32) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += bred[w * 64 + lane] + bred[w * 64 + 32 + lane];
    db[co0 + lane] = v;
  }
}

// Input gradient of a STRIDED convolution (P6 / P7, timm.py:359-364; the trunk's stride-2 layers), gather form: one workgroup per
// input position, thread = input channel; dX[n][iy][ix][ci] = sum over the taps (ky, kx) with (iy + pad - ky) and (ix + pad - kx)
// multiples of the stride of sum_co G[n][(iy + pad - ky) / s][(ix + pad - kx) / s][co] * W[co][(ky, kx, ci)].  Weight reads are
// coalesced over ci, G values are broadcasts.  (Stride-1 layers use eod_conv2d with the rotated weights: matrix cores.)
__global__ __launch_bounds__(256) void conv_backward_input_kernel(ConvBwdArgs a, const float* __restrict__ w, int Kpad, float* __restrict__ dx) {
  const int pos = blockIdx.x;                         // (n, iy, ix)
  const int ix = pos % a.W, t = pos / a.W;
  const int iy = t % a.H, n = t / a.H;
  for (int ci = threadIdx.x; ci < a.Cin; ci += blockDim.x) {
    float acc = 0.f;
    for (int ky = 0; ky < a.KH; ++ky) {
      const int ny = iy + a.pad - ky;
      if (ny < 0 || ny % a.stride != 0) continue;
      const int oy = ny / a.stride;
      if (oy >= a.OH) continue;
      for (int kx = 0; kx < a.KW; ++kx) {
        const int nx = ix + a.pad - kx;
        if (nx < 0 || nx % a.stride != 0) continue;
        const int ox = nx / a.stride;
        if (ox >= a.OW) continue;
        const float* gp = a.g + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.Cout;
        const float* wp = w + (size_t)(ky * a.KW + kx) * a.Cin + ci;
        for (int co = 0; co < a.Cout; ++co) acc += gp[co] * wp[(size_t)co * Kpad];
      }
    }
    dx[(size_t)pos * a.Cin + ci] = acc;
  }
}

// FPN top-down add (timm.py:128-133: lateral + nearest x2 of the coarser level): the coarser level's gradient is the sum over each
// 2x2 block of the finer level's gradient (the lateral branch gets the gradient itself).  g [N,2h,2w,C] -> out [N,h,w,C] (+= when
// accumulate: the coarser level also has its own output-conv branch).
__global__ __launch_bounds__(256) void upsample2_backward_kernel(const float* __restrict__ g, float* __restrict__ out, int N, int h, int w,
                                                                  int C4, int accumulate) {
  const size_t total = (size_t)N * h * w * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h), n = (int)(t / h);
    const f32x4* gp = reinterpret_cast<const f32x4*>(g);
    const size_t r0 = ((size_t)(n * 2 * h + 2 * y) * (2 * w) + 2 * x) * C4 + c, r1 = r0 + (size_t)2 * w * C4;
    f32x4 v = gp[r0];
    v += gp[r0 + C4];
    v += gp[r1];
    v += gp[r1 + C4];
    f32x4* op = reinterpret_cast<f32x4*>(out) + i;
    if (accumulate) v += *op;
    *op = v;
  }
}

// timm ResNet maxpool 3x3 s2 p1 (timm.py:281) backward: every input position collects the gradient of the output windows whose
// maximum it is -- the FIRST maximum of a window in row-major tap order, as torch's max_pool2d backward routes it.  Gather form
// (no atomics): x [N,H,W,C], y / g [N,OH,OW,C] -> dx [N,H,W,C].
__global__ __launch_bounds__(256) void maxpool3x3s2_backward_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                     const float* __restrict__ g, float* __restrict__ dx, int N, int H, int W,
                                                                     int C, int OH, int OW) {
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ix = (int)(t % W);
    t /= W;
    const int iy = (int)(t % H), n = (int)(t / H);
    const float xv = x[i];
    float acc = 0.f;
    // output windows that contain (iy, ix): oy in {ceil((iy - 1) / 2) .. floor((iy + 1) / 2)}: one for an even row, two for an odd one
    for (int oy = (iy + 1) / 2 - (iy & 1); oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= OH) continue;
      for (int ox = (ix + 1) / 2 - (ix & 1); ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= OW) continue;
        const size_t o = ((size_t)(n * OH + oy) * OW + ox) * C + c;
        if (y[o] != xv) continue;
        // am I the first tap of this window (row-major) that holds the maximum?
        bool first = true;
        for (int ky = 0; ky < 3 && first; ++ky) {
          const int yy = oy * 2 - 1 + ky;
          if (yy < 0 || yy >= H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = ox * 2 - 1 + kx;
            if (xx < 0 || xx >= W) continue;
            if (yy == iy && xx == ix) {
              ky = 3;
              break;
            }
            if (x[((size_t)(n * H + yy) * W + xx) * C + c] == xv) {
              first = false;
              break;
            }
          }
        }
        if (first) acc += g[o];
      }
    }
    dx[i] = acc;
  }
}

// dL/d(pre-activation) of a ReLU layer from dL/d(output): g where the output was positive
__global__ __launch_bounds__(256) void relu_backward_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out,
                                                             size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 o;
    o.x = yv.x > 0.f ? gv.x : 0.f;
    o.y = yv.y > 0.f ? gv.y : 0.f;
    o.z = yv.z > 0.f ? gv.z : 0.f;
    o.w = yv.w > 0.f ? gv.w : 0.f;
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

}  // namespace

static int conv_bwd_args(ConvBwdArgs& a, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 31) || (Cout & 31) || KH <= 0 || KW <= 0 || KH * KW > 64 ||
      pad < 0 || stride < 1 || H + 2 * pad < KH || W + 2 * pad < KW)
    return EOD_ERR_BAD_DIMS;
  if ((long)N * H * W >= (1L << 28)) return EOD_ERR_BAD_DIMS;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.pad = pad; a.stride = stride;
  a.OH = (H + 2 * pad - KH) / stride + 1;
  a.OW = (W + 2 * pad - KW) / stride + 1;
  a.div_w = eod_make_fastdiv((unsigned)a.OW);
  a.div_h = eod_make_fastdiv((unsigned)a.OH);
  return EOD_OK;
}

// position ranges of a weight-gradient launch: enough workgroups for the chip, ranges of at least 16 steps (128 positions)
// EOD_WGRAD_RB=0 selects the 32 x 32 kernel of round 3 (same-box A/B of the two; read once)
static bool wgrad_register_blocked() {
  static const bool on = [] {
    const char* e = getenv("EOD_WGRAD_RB");
    return !(e && e[0] == '0');
  }();
  return on;
}

// EOD_WGRAD_LDS=0 falls back to the register-blocked kernel (same-box A/B; read once)
static bool wgrad_lds() {
  static const bool on = [] {
    const char* e = getenv("EOD_WGRAD_LDS");
    return !(e && e[0] == '0');
  }();
  return on;
}

static int wgrad_splits(int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride) {
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  const long P = (long)N * OH * OW;
  const bool tap4_scalar = Cin == 4 && KW > 8;                    // wider kernel rows than a 32-column tile: the scalar kernel
  const long wgs = Cin == 4 ? (tap4_scalar ? (long)KH * KW * (Cout >> 4) : (long)(Cout >> 5) * KH)
                   : wgrad_register_blocked() ? (long)((Cout + 63) >> 6) * ((Cin + 63) >> 6) * KH * KW        // 64 x 64 register blocks
                                              : (long)(Cout >> 5) * (Cin >> 5) * KH * KW;
  long s = tap4_scalar ? 16 : (768 + wgs - 1) / wgs;
  const long cap = tap4_scalar ? P / 2048 : (Cin != 4 && wgrad_lds() ? (P + 31) / 32 / 4 : (P + 7) / 8 / 16);
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

extern "C" size_t eod_conv2d_backward_weights_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || pad < 0 || stride < 1) return 0;
  const int s = wgrad_splits(N, H, W, Cin, Cout, KH, KW, pad, stride);
  return s > 1 ? (size_t)s * ((size_t)Cout * KH * KW * Cin + Cout) * sizeof(float) : 0;
}

static int conv2d_backward_weights_impl(const float* x, const float* g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                                        int stride, float* dw, float* db, void* workspace, size_t workspace_bytes, eod_stream_t stream) {
  if (!x || !g || !dw) return EOD_ERR_NULL;
  ConvBwdArgs a{};
  const int st = conv_bwd_args(a, N, H, W, Cin == 4 ? 32 : Cin, Cout, KH, KW, pad, stride);     // Cin == 4: the stem's tap layout
  if (st != EOD_OK) return st;
  a.Cin = Cin;
  a.x = x; a.g = g; a.dw = dw; a.db = db;
  a.splits = 1;
  const size_t n = (size_t)Cout * KH * KW * Cin;
  if (workspace) {
    const size_t need = eod_conv2d_backward_weights_workspace_bytes(N, H, W, Cin, Cout, KH, KW, pad, stride);
    if (need > workspace_bytes) return EOD_ERR_CAPACITY;
    if (need) {
      a.splits = wgrad_splits(N, H, W, Cin, Cout, KH, KW, pad, stride);
      a.part = static_cast<float*>(workspace);
      a.bpart = a.part + (size_t)a.splits * n;
    }
  }
  if (Cin == 4 && KW <= 8)
    hipLaunchKernelGGL(conv_backward_weights_tap4_mfma_kernel, dim3((Cout >> 5) * KH, 1, a.splits), dim3(256), 0, (hipStream_t)stream, a);
  else if (Cin == 4)
    hipLaunchKernelGGL(conv_backward_weights_tap4_kernel, dim3(KH * KW, Cout >> 4, a.splits), dim3(1024), 0, (hipStream_t)stream, a);
  else if (wgrad_lds())
    hipLaunchKernelGGL(conv_backward_weights_lds_kernel<false>, dim3(((Cout + 63) >> 6) * ((Cin + 63) >> 6), KH * KW, a.splits), dim3(256), 0,
                       (hipStream_t)stream, a);
  else if (wgrad_register_blocked())
    hipLaunchKernelGGL(conv_backward_weights_rb_kernel, dim3(((Cout + 63) >> 6) * ((Cin + 63) >> 6), KH * KW, a.splits), dim3(256), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv_backward_weights_kernel, dim3((Cout >> 5) * (Cin >> 5), KH * KW, a.splits), dim3(256), 0, (hipStream_t)stream, a);
  if (a.splits > 1) {
    size_t blocks = (n + Cout + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a.part, a.bpart, dw, db, n, Cout, a.splits);
  }
  return eod_launch_status();
}

extern "C" int eod_conv2d_backward_weights(const float* x, const float* g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                                           int stride, float* dw, float* db, eod_stream_t stream) {
  return conv2d_backward_weights_impl(x, g, N, H, W, Cin, Cout, KH, KW, pad, stride, dw, db, nullptr, 0, stream);
}

extern "C" int eod_conv2d_backward_weights_ws(const float* x, const float* g, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                              int pad, int stride, float* dw, float* db, void* workspace, size_t workspace_bytes,
                                              eod_stream_t stream) {
  if (workspace_bytes && !workspace) return EOD_ERR_NULL;
  return conv2d_backward_weights_impl(x, g, N, H, W, Cin, Cout, KH, KW, pad, stride, dw, db, workspace, workspace_bytes, stream);
}

// Pyramid mode: one launch for a level-shared layer (CenterNet tower / head, centernet_head.py:141-161) over all levels' rows.
static int wgrad_levels_splits(long rows, int Cin, int Cout, int KH, int KW) {
  const long wgs = (long)((Cout + 63) >> 6) * ((Cin + 63) >> 6) * KH * KW;
  long s = (768 + wgs - 1) / wgs;
  const long cap = (rows + 31) / 32 / 4;
  if (s > cap) s = cap;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}

extern "C" size_t eod_conv2d_backward_weights_levels_workspace_bytes(int rows, int Cin, int Cout, int KH, int KW) {
  if (rows <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
  const int s = wgrad_levels_splits(rows, Cin, Cout, KH, KW);
  return s > 1 ? (size_t)s * ((size_t)Cout * KH * KW * Cin + Cout) * sizeof(float) : 0;
}

extern "C" int eod_conv2d_backward_weights_levels(const float* x, const float* g, int levels, const int32_t* level_off,
                                                  const int32_t* level_h, const int32_t* level_w, int Cin, int Cout, int KH, int KW, int pad,
                                                  float* dw, float* db, void* workspace, size_t workspace_bytes, eod_stream_t stream) {
  if (!x || !g || !dw || !level_off || !level_h || !level_w) return EOD_ERR_NULL;
  if (levels < 1 || levels > WGRAD_MAX_LEVELS || Cin <= 0 || Cout <= 0 || (Cin & 31) || (Cout & 31) || KH <= 0 || KW <= 0 || KH != KW ||
      pad * 2 != KH - 1 || level_off[0] != 0)
    return EOD_ERR_BAD_DIMS;
  for (int l = 0; l < levels; ++l)
    if (level_h[l] <= 0 || level_w[l] <= 0 || level_off[l + 1] - level_off[l] != level_h[l] * level_w[l]) return EOD_ERR_BAD_DIMS;
  if (workspace_bytes && !workspace) return EOD_ERR_NULL;
  if (!eod_aligned16(x) || !eod_aligned16(g) || !eod_aligned16(dw)) return EOD_ERR_ALIGN;
  ConvBwdArgs a{};
  a.x = x; a.g = g; a.dw = dw; a.db = db;
  a.N = 1; a.H = a.OH = level_h[0]; a.W = a.OW = level_w[0];
  a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW; a.pad = pad; a.stride = 1;
  a.nlv = levels;
  for (int l = 0; l < levels; ++l) {
    a.lv_off[l] = level_off[l]; a.lv_h[l] = level_h[l]; a.lv_w[l] = level_w[l];
  }
  a.lv_off[levels] = level_off[levels];
  a.splits = 1;
  const size_t n = (size_t)Cout * KH * KW * Cin;
  const size_t need = eod_conv2d_backward_weights_levels_workspace_bytes(level_off[levels], Cin, Cout, KH, KW);
  if (workspace && need) {
    if (need > workspace_bytes) return EOD_ERR_CAPACITY;
    a.splits = wgrad_levels_splits(level_off[levels], Cin, Cout, KH, KW);
    a.part = static_cast<float*>(workspace);
    a.bpart = a.part + (size_t)a.splits * n;
  }
  hipLaunchKernelGGL(conv_backward_weights_lds_kernel<true>, dim3(((Cout + 63) >> 6) * ((Cin + 63) >> 6), KH * KW, a.splits), dim3(256), 0,
                     (hipStream_t)stream, a);
  if (a.splits > 1) {
    size_t blocks = (n + Cout + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a.part, a.bpart, dw, db, n, Cout, a.splits);
  }
  return eod_launch_status();
}

extern "C" int eod_conv2d_backward_input(const float* g, const float* w, int Kpad, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                         int pad, int stride, float* dx, eod_stream_t stream) {
  if (!g || !w || !dx) return EOD_ERR_NULL;
  ConvBwdArgs a{};
  const int st = conv_bwd_args(a, N, H, W, Cin, Cout, KH, KW, pad, stride);
  if (st != EOD_OK) return st;
  if (Kpad < KH * KW * Cin) return EOD_ERR_BAD_DIMS;
  a.g = g;
  hipLaunchKernelGGL(conv_backward_input_kernel, dim3(N * H * W), dim3(256), 0, (hipStream_t)stream, a, w, Kpad, dx);
  return eod_launch_status();
}

extern "C" int eod_upsample2_sum_backward(const float* g, float* out, int N, int h, int w, int C, int accumulate, eod_stream_t stream) {
  if (!g || !out) return EOD_ERR_NULL;
  if (N <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(g) || !eod_aligned16(out)) return EOD_ERR_ALIGN;
  const size_t total = (size_t)N * h * w * (C >> 2);
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(upsample2_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, out, N, h, w, C >> 2, accumulate);
  return eod_launch_status();
}

extern "C" int eod_maxpool3x3s2_backward(const float* x, const float* y, const float* g, float* dx, int N, int H, int W, int C, int OH,
                                         int OW, eod_stream_t stream) {
  if (!x || !y || !g || !dx) return EOD_ERR_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1) return EOD_ERR_BAD_DIMS;
  const size_t total = (size_t)N * H * W * C;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(maxpool3x3s2_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, g, dx, N, H, W, C, OH, OW);
  return eod_launch_status();
}

extern "C" int eod_relu_backward(const float* g, const float* y, float* out, size_t n, eod_stream_t stream) {
  if (!g || !y || !out) return EOD_ERR_NULL;
  if (n == 0 || (n & 3)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(g) || !eod_aligned16(y) || !eod_aligned16(out)) return EOD_ERR_ALIGN;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, y, out, n / 4);
  return eod_launch_status();
}

extern "C" int eod_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int step, double clip_value, eod_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return EOD_ERR_NULL;
  if (n == 0 || step < 1 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return EOD_ERR_BAD_DIMS;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                     (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                     (float)sqrt(bc2), (float)eps, (float)clip_value);
  return eod_launch_status();
}

// The weights of a layer's input-gradient convolution (dX = conv of dY with the 180-degree rotated, in/out-transposed kernel), from the
// layer's packed forward weights: out[ci][(ky', kx', co)] = w[co][(KH-1-ky', KW-1-kx', ci)].  One launch per layer after an optimizer
// step (torch's flip + permute + copy were three).
__global__ __launch_bounds__(256) void rotate_weights_kernel(const float* __restrict__ w, int Cout, int KH, int KW, int Cin, int ld_in,
                                                              float* __restrict__ out, int ld_out) {
  const size_t n = (size_t)Cin * KH * KW * Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    size_t r = i / Cout;
    const int kx = (int)(r % KW);
    r /= KW;
    const int ky = (int)(r % KH);
    const int ci = (int)(r / KH);
    out[(size_t)ci * ld_out + ((size_t)ky * KW + kx) * Cout + co] =
        w[(size_t)co * ld_in + ((size_t)(KH - 1 - ky) * KW + (KW - 1 - kx)) * Cin + ci];
  }
}

extern "C" int eod_conv_rotate_weights(const float* w, int Cout, int KH, int KW, int Cin, int ld_in, float* out, int ld_out,
                                       eod_stream_t stream) {
  if (!w || !out) return EOD_ERR_NULL;
  if (Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0 || ld_in < KH * KW * Cin || ld_out < KH * KW * Cout) return EOD_ERR_BAD_DIMS;
  size_t blocks = ((size_t)Cin * KH * KW * Cout + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rotate_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, KH, KW, Cin, ld_in, out, ld_out);
  return eod_launch_status();
}

// The rotation for up to ROTATE_MULTI layers in one launch: after an optimizer step every layer with an input-gradient convolution
// needs it (74 layers of the recurrent detector: 74 launches of ~9 us were 4 % of the iteration).
#define ROTATE_MULTI 24
struct RotateMulti {
  const float* w[ROTATE_MULTI];
  float* out[ROTATE_MULTI];
  int Cout[ROTATE_MULTI], KH[ROTATE_MULTI], KW[ROTATE_MULTI], Cin[ROTATE_MULTI], ld_in[ROTATE_MULTI], ld_out[ROTATE_MULTI];
  unsigned block_end[ROTATE_MULTI];
  int count;
};

// Per tap the rotation is a (co, ci) -> (ci, co) transpose: a workgroup moves one 32 x 32 tile of one tap through LDS, reading rows of
// w along ci and writing rows of out along co (both coalesced; the element-per-thread form reads with a stride of a weight row).
__global__ __launch_bounds__(256) void rotate_weights_multi_kernel(RotateMulti a) {
  __shared__ float tile[32][33];
  int t = 0;
  while (t + 1 < a.count && blockIdx.x >= a.block_end[t]) ++t;
  const unsigned b = blockIdx.x - (t ? a.block_end[t - 1] : 0u);
  const float* __restrict__ w = a.w[t];
  float* __restrict__ out = a.out[t];
  const int Cout = a.Cout[t], KH = a.KH[t], KW = a.KW[t], Cin = a.Cin[t], ld_in = a.ld_in[t], ld_out = a.ld_out[t];
  const unsigned tiles_ci = (unsigned)(Cin + 31) >> 5, tiles_co = (unsigned)(Cout + 31) >> 5;
  const unsigned tap = b / (tiles_co * tiles_ci), rem = b - tap * tiles_co * tiles_ci;
  const int co0 = (int)(rem / tiles_ci) * 32, ci0 = (int)(rem % tiles_ci) * 32;
  const int ky = (int)tap / KW, kx = (int)tap - ky * KW;
  const int src_tap = (KH - 1 - ky) * KW + (KW - 1 - kx);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int co = co0 + ty + 8 * j, ci = ci0 + tx;
    tile[ty + 8 * j][tx] = (co < Cout && ci < Cin) ? w[(size_t)co * ld_in + (size_t)src_tap * Cin + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ci = ci0 + ty + 8 * j, co = co0 + tx;
    if (ci < Cin && co < Cout) out[(size_t)ci * ld_out + (size_t)tap * Cout + co] = tile[tx][ty + 8 * j];
  }
}

extern "C" int eod_conv_rotate_weights_multi(const EodRotateTensor* tensors, int count, eod_stream_t stream) {
  if (!tensors) return EOD_ERR_NULL;
  if (count < 1) return EOD_ERR_BAD_DIMS;
  for (int i = 0; i < count; ++i) {
    const EodRotateTensor& t = tensors[i];
    if (!t.w || !t.out) return EOD_ERR_NULL;
    if (t.Cout <= 0 || t.KH <= 0 || t.KW <= 0 || t.Cin <= 0 || t.ld_in < t.KH * t.KW * t.Cin || t.ld_out < t.KH * t.KW * t.Cout)
      return EOD_ERR_BAD_DIMS;
  }
  for (int i0 = 0; i0 < count; i0 += ROTATE_MULTI) {
    RotateMulti a{};
    a.count = std::min(ROTATE_MULTI, count - i0);
    unsigned blocks = 0;
    for (int k = 0; k < a.count; ++k) {
      const EodRotateTensor& t = tensors[i0 + k];
      a.w[k] = t.w; a.out[k] = t.out;
      a.Cout[k] = t.Cout; a.KH[k] = t.KH; a.KW[k] = t.KW; a.Cin[k] = t.Cin; a.ld_in[k] = t.ld_in; a.ld_out[k] = t.ld_out;
      blocks += (unsigned)((size_t)t.KH * t.KW * ((t.Cout + 31) >> 5) * ((t.Cin + 31) >> 5));     // one 32 x 32 tile each
      a.block_end[k] = blocks;
    }
    hipLaunchKernelGGL(rotate_weights_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  return eod_launch_status();
}

extern "C" int eod_adamw_step_multi(const EodAdamWTensor* tensors, int count, double beta1, double beta2, double eps, double clip_value,
                                    eod_stream_t stream) {
  if (!tensors) return EOD_ERR_NULL;
  if (count < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return EOD_ERR_BAD_DIMS;
  for (int i = 0; i < count; ++i) {
    const EodAdamWTensor& t = tensors[i];
    if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return EOD_ERR_NULL;
    if (t.n == 0 || t.step < 1 || !(t.lr >= 0.0)) return EOD_ERR_BAD_DIMS;
    if (t.folded_out && (!t.row_scale || t.cols <= 0 || t.ld_out < t.cols || t.n % (size_t)t.cols != 0)) return EOD_ERR_BAD_DIMS;
    if (t.grad_of_folded && (!t.row_scale || t.cols <= 0 || t.n % (size_t)t.cols != 0)) return EOD_ERR_BAD_DIMS;
  }
  for (int i0 = 0; i0 < count; i0 += ADAMW_MULTI) {
    AdamWMulti a{};
    a.count = std::min(ADAMW_MULTI, count - i0);
    a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.clip = (float)clip_value;
    unsigned blocks = 0;
    for (int k = 0; k < a.count; ++k) {
      const EodAdamWTensor& t = tensors[i0 + k];
      const double bc1 = 1.0 - pow(beta1, (double)t.step), bc2 = 1.0 - pow(beta2, (double)t.step);
      a.p[k] = t.param; a.g[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq; a.n[k] = t.n;
      a.folded[k] = t.folded_out; a.row_scale[k] = t.row_scale; a.cols[k] = (unsigned)t.cols; a.ld_out[k] = (unsigned)t.ld_out;
      if (t.grad_of_folded) a.grad_of_folded |= 1u << k;
      a.decay[k] = (float)(1.0 - t.lr * t.weight_decay);
      a.step_size[k] = (float)(t.lr / bc1);
      a.bc2_sqrt[k] = (float)sqrt(bc2);
      size_t nb = (t.n + 1023) / 1024;                  // four elements per thread and pass
      if (nb > 1024) nb = 1024;
      blocks += (unsigned)nb;
      a.block_end[k] = blocks;
    }
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  }
  return eod_launch_status();
}

#define PROJ_WGRAD_SPLITS 8
extern "C" size_t eod_memory_project_backward_weights_workspace_bytes(void) {
  return (size_t)3 * PROJ_WGRAD_SPLITS * (256 * 512 + 256) * sizeof(float);
}

static int memory_project_backward_weights_impl(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H,
                                                int W, float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5,
                                                float* db5, void* workspace, size_t workspace_bytes, eod_stream_t stream);

extern "C" int eod_memory_project_backward_weights(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H,
                                                   int W, float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5,
                                                   float* db5, eod_stream_t stream) {
  return memory_project_backward_weights_impl(g3, g4, g5, pooled_f16, H, W, weight, dw3, db3, dw4, db4, dw5, db5, nullptr, 0, stream);
}

extern "C" int eod_memory_project_backward_weights_ws(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H,
                                                      int W, float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5,
                                                      float* db5, void* workspace, size_t workspace_bytes, eod_stream_t stream) {
  if (!workspace || workspace_bytes < eod_memory_project_backward_weights_workspace_bytes()) return EOD_ERR_CAPACITY;
  return memory_project_backward_weights_impl(g3, g4, g5, pooled_f16, H, W, weight, dw3, db3, dw4, db4, dw5, db5, workspace,
                                              workspace_bytes, stream);
}

static int memory_project_backward_weights_impl(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H,
                                                int W, float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5,
                                                float* db5, void* workspace, size_t workspace_bytes, eod_stream_t stream) {
  if (!g3 || !g4 || !g5 || !pooled_f16 || !dw3 || !db3 || !dw4 || !db4 || !dw5 || !db5) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 31) || (W & 31)) return EOD_ERR_BAD_DIMS;
  BwdArgs a{};
  a.g[0] = g3; a.g[1] = g4; a.g[2] = g5;
  a.pooled = reinterpret_cast<const __half*>(pooled_f16);
  a.dw[0] = dw3; a.dw[1] = dw4; a.dw[2] = dw5;
  a.db[0] = db3; a.db[1] = db4; a.db[2] = db5;
  int base = 0;
  for (int l = 0; l < 3; ++l) {
    a.rows[l] = (H >> (3 + l)) * (W >> (3 + l));
    a.tile_base[l] = base;
    base += (a.rows[l] + 31) / 32;
  }
  a.weight = weight;
  a.splits = 1;
  if (workspace) {
    a.splits = PROJ_WGRAD_SPLITS;
    a.part = static_cast<float*>(workspace);
    a.bpart = a.part + (size_t)3 * PROJ_WGRAD_SPLITS * 256 * 512;
  }
  hipLaunchKernelGGL(proj_backward_weights_kernel, dim3(128, 3, a.splits), dim3(256), 0, (hipStream_t)stream, a);
  if (a.splits > 1) {
    for (int l = 0; l < 3; ++l)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(513), dim3(256), 0, (hipStream_t)stream, a.part + (size_t)l * a.splits * 256 * 512,
                         a.bpart + (size_t)l * a.splits * 256, a.dw[l], a.db[l], (size_t)256 * 512, 256, a.splits);
  }
  return eod_launch_status();
}

extern "C" int eod_memory_pool_backward(const float* dec3, const float* dec4, const float* dec5, int H, int W, uint16_t* ge3_f16,
                                        uint16_t* ge4_f16, uint16_t* ge5_f16, float* ge2, eod_stream_t stream) {
  if (!dec3 || !dec4 || !dec5 || !ge3_f16 || !ge4_f16 || !ge5_f16 || !ge2) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 31) || (W & 31)) return EOD_ERR_BAD_DIMS;
  PoolBwdArgs a{};
  a.dec[0] = dec3; a.dec[1] = dec4; a.dec[2] = dec5;
  a.ge[0] = reinterpret_cast<__half*>(ge3_f16); a.ge[1] = reinterpret_cast<__half*>(ge4_f16); a.ge[2] = reinterpret_cast<__half*>(ge5_f16);
  a.ge2 = ge2;
  a.h2 = H >> 2; a.w2 = W >> 2;
  const size_t total = (size_t)a.h2 * a.w2 * 512;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pool_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}
