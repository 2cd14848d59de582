// Shared by the convolution translation units: launch arguments, the fused epilogue and the launch entry points of the
// kernel families (fp32 MFMA: conv_fp32.hip, bf16x3 split: conv_bf16x3.hip).  Planning,
// argument checks and the C ABI live in conv_igemm.hip.
#pragma once
#include "eod_common.h"
#include "../../include/eod_hip.h"

namespace eodconv {


struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* gate;        // optional [M][Cout]: the stored value is 0 where gate <= 0 (EodConvDesc.gate: backward of a ReLU)
  float* y;
  float* partial;
  const int* m_count;
  int m_unit;
  int m_segs, seg_rows;     // m_segs > 1: m_count holds m_segs counts, one per run of seg_rows output rows (EodConvDesc.m_segments)
  const float* fuse_w;      // out_mode 2: predictor weights / bias / unit scatter
  const int* out_units;
  float fuse_b;
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, Kpad;
  int M, nchunks, splitk, cps;
  int relu, res_mode, in_relu, out_mode;
  int tiles_m, tiles_n;
  float out_scale;
  // multi-level mode (shared-weight head over the FPN pyramid): rows [lv_off[l], lv_off[l+1]) form an lv_h[l] x lv_w[l] image
  int nlv;
  int lv_off[EOD_MAX_LEVELS + 1], lv_h[EOD_MAX_LEVELS], lv_w[EOD_MAX_LEVELS];
  unsigned x_bytes, w_bytes;   // sizes of the two operand buffers (range of the buffer descriptors)
  const void* w3;              // optional pre-split weights of the bf16x3 kernels (eod_conv_split_weights_bf16x3)
  unsigned w3_bytes;
  double* gn_partial;          // optional (pyramid mode + split-K): GroupNorm partial sums written by the slab reduce (EodConvDesc)
  int gn_groups;
  float* y2;                   // optional second output: columns [split_n, Cout) go to y2 [M, Cout - split_n] (with the ReLU), columns
  int split_n;                 // [0, split_n) to y [M, split_n] (never with the ReLU): two linear layers on one input as one GEMM
  FastDiv div_ow, div_oh, div_cd, div_row;
};

// Rows that hold work under the device-side count.  One count: rows [0, count * m_unit).  m_segs > 1 (independent ROI lists back to
// back, seg_rows rows each): row m holds work iff (m mod seg_rows) < count[m / seg_rows] * m_unit; the row limit then stays p.M and
// whole tiles without work leave early (conv_tile_active).
__device__ __forceinline__ int conv_row_limit(const ConvArgs& p, int M) {
  if (p.m_count && p.m_segs <= 1) {
    const int lim = *p.m_count * p.m_unit;
    M = lim < M ? lim : M;
  }
  return M;
}

__device__ __forceinline__ bool conv_row_active(const ConvArgs& p, int m) {
  if (!p.m_count || p.m_segs <= 1) return true;      // the single count is applied through the row limit
  const int s = m / p.seg_rows;
  return m - s * p.seg_rows < p.m_count[s] * p.m_unit;
}

__device__ __forceinline__ bool conv_tile_active(const ConvArgs& p, int m0, int bm) {
  if (!p.m_count || p.m_segs <= 1) return true;
  const int last = (m0 + bm - 1 < p.M ? m0 + bm - 1 : p.M - 1);
  const int s0 = m0 / p.seg_rows, s1 = last / p.seg_rows;
  if (m0 - s0 * p.seg_rows < p.m_count[s0] * p.m_unit) return true;
  for (int s = s0 + 1; s <= s1; ++s)
    if (p.m_count[s] > 0) return true;
  return false;
}

// GATE: the kernel honours ConvArgs.gate (the 64x64 fp32 tile, the wave-K kernel and the slab reduces: the plans a gated layer is
// given).  The larger tiles and the bf16x3 kernels are compiled without the extra load: with it their register allocation
// spilled (320 bytes of scratch per lane, bf16x3 frame rate 352 -> 214).
template <bool GATE = false>
__device__ __forceinline__ float epilogue_store(const ConvArgs& p, float v, int m, int n) {
  int co = n;
  size_t oidx;
  if (p.out_mode == 1) {
    const int Cd = p.Cout >> 2;
    const int quad = (int)fdiv((unsigned)n, p.div_cd);
    co = n - quad * Cd;
    const int dy = quad >> 1, dx = quad & 1;
    const int t = (int)fdiv((unsigned)m, p.div_ow);
    const int ox = m - t * p.OW;
    const int img = (int)fdiv((unsigned)t, p.div_oh);
    const int oy = t - img * p.OH;
    oidx = ((size_t)(img * 2 * p.OH + 2 * oy + dy) * (2 * p.OW) + (2 * ox + dx)) * Cd + co;
  } else {
    oidx = (size_t)m * p.Cout + n;
  }
  if (p.bias) v += p.bias[co];
  v *= p.out_scale;
  if (p.res_mode == 1) {
    v += p.res[(size_t)m * p.Cout + n];
  } else if (p.res_mode == 2) {
    const int t = (int)fdiv((unsigned)m, p.div_ow);
    const int ox = m - t * p.OW;
    const int img = (int)fdiv((unsigned)t, p.div_oh);
    const int oy = t - img * p.OH;
    const int rh = p.OH >> 1, rw = p.OW >> 1;
    v += p.res[((size_t)(img * rh + (oy >> 1)) * rw + (ox >> 1)) * p.Cout + n];
  }
  if (p.split_n > 0) {
    if (n >= p.split_n) {
      if (p.relu) v = fmaxf(v, 0.0f);
      p.y2[(size_t)m * (p.Cout - p.split_n) + (n - p.split_n)] = v;
    } else {
      p.y[(size_t)m * p.split_n + n] = v;
    }
    return v;
  }
  if (p.relu) v = fmaxf(v, 0.0f);
  if (GATE && p.gate && !(p.gate[oidx] > 0.0f)) v = 0.0f;
  p.y[oidx] = v;
  return v;
}

// Bit tp = ky * KW + kx is set when tap (ky, kx) of the window anchored at (iy0, ix0) falls inside an hh x ww image.
__device__ __forceinline__ unsigned long long tap_mask(int iy0, int ix0, int hh, int ww, int KH, int KW) {
  unsigned long long mask = 0;
  int tp = 0;
  for (int ky = 0; ky < KH; ++ky) {
    const bool rok = (unsigned)(iy0 + ky) < (unsigned)hh;
    for (int kx = 0; kx < KW; ++kx, ++tp) {
      const bool ok = rok && ((unsigned)(ix0 + kx) < (unsigned)ww);
      mask |= (unsigned long long)ok << tp;
    }
  }
  return mask;
}

// Stores one wave's accumulators (TM x TN tiles of 32x32, MFMA C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
// through the fused epilogue, or as a split-K slab.
template <int TM, int TN, bool GATE = false>
__device__ __forceinline__ void store_wave_tiles(const ConvArgs& p, const f32x16 (&acc)[TM][TN], int m_base, int n_base, int M, int z,
                                                 int lane) {
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n_base + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m_base + i * 32 + row;
        if (m < M && n < p.Cout && conv_row_active(p, m)) {
          const float v = acc[i][j][r];
          if (p.splitk > 1) {
            p.partial[((size_t)z * p.M + m) * p.Cout + n] = v;
          } else {
            epilogue_store<GATE>(p, v, m, n);
          }
        }
      }
    }
  }
}

// tile: 1 = 128x128, 2 = 128x64, 3 = 64x64 (4 waves); bf16x3 also 4 = 256x128 (8 waves, dynamic LDS); fp32 also 5 = 64x256
// (one whole deconv quadrant per workgroup: the fused mask-head tail, out_mode 2)
void launch_conv_fp32(const ConvArgs& a, int tile, int bk, bool tap4, dim3 grid, hipStream_t s, int lds_reserve = 0, int prefetch2 = 0);
void launch_conv_bf16x3(const ConvArgs& a, int tile, dim3 grid, hipStream_t s);
void launch_conv_wavek(const ConvArgs& a, int nw, dim3 grid, hipStream_t s);
void launch_split_weights(const float* w, void* out, int Cout, int Kpad, hipStream_t s);

}  // namespace eodconv
