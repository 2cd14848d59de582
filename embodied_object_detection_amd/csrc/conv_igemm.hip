// Implicit-GEMM convolution / linear layer on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   y[m][n] = act( (sum_k A[m][k] * Wt[n][k] + bias[n]) * out_scale + res[m][n] )
//
// A[m][k] is the im2col view of the NHWC input: m = (img, oy, ox), k = (ky, kx, c) with c fastest, so a
// 32-wide K chunk is 128 contiguous bytes of one input pixel.  Wt is [Cout][Kpad] (K fastest), which makes
// both operand tiles "row = m or n, 32 contiguous k": they are staged through LDS as [rows][36] floats (the +4
// pad makes the 16-lane groups of ds_read_b128 conflict free) and each lane fetches 4 consecutive k of its
// row with ONE ds_read_b128.  The MFMA k-slot <-> k mapping is a free permutation as long as A and B agree:
// lane half h supplies k = kk*8 + 4h + t to instruction t (t = 0..3), for both operands.
//
// fp32 in / fp32 accumulate: the result is an exact fp32 fma chain (same numerics class as the CPU
// reference path), the peak is the fp32 MFMA rate 157 TFLOP/s.
//
// Block = 256 threads = 4 waves (2 x 2), block tile BM x BN in {128x128, 128x64, 64x64}, each wave owns
// (BM/2) x (BN/2) as 32x32 MFMA tiles.  Global -> register prefetch of chunk c+1 overlaps the MFMAs of chunk
// c; one LDS buffer, two barriers per chunk.  Tiles are dealt to workgroups through an XCD-aware bijective
// remap so that the workgroups sharing an L2 walk neighbouring tiles (same weight panel / same pixel rows).
// Small problems are split along K (grid.y) into fp32 slabs reduced by a second kernel that also applies the
// epilogue: deterministic, no atomics.
#include "eod_common.h"
#include "../../include/eod_hip.h"
#include <atomic>
#include <cstdlib>

namespace {

// Division by a launch-invariant integer with one mul_hi + shifts (Granlund-Montgomery, exact for every 32-bit n).
struct FastDiv {
  unsigned mp, sh1, sh2, d;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.mp, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct ConvArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  float* partial;
  const int* m_count;
  int m_unit;
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, Kpad;
  int M, nchunks, splitk, cps;
  int relu, res_mode, in_relu, out_mode;
  int tiles_m, tiles_n;
  float out_scale;
  // multi-level mode (shared-weight head over the FPN pyramid): rows [lv_off[l], lv_off[l+1]) form an lv_h[l] x lv_w[l] image
  int nlv;
  int lv_off[6], lv_h[5], lv_w[5];
  unsigned x_bytes, w_bytes;   // sizes of the two operand buffers (range of the buffer descriptors)
  FastDiv div_ow, div_oh, div_cd;
};

__device__ __forceinline__ void epilogue_store(const ConvArgs& p, float v, int m, int n) {
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
  if (p.relu) v = fmaxf(v, 0.0f);
  p.y[oidx] = v;
}

template <int BM, int BN, int BK, bool TAP4, bool MULTI>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
  constexpr int LS = BK + 4;  // LDS row stride in floats (+4: conflict-free 16-lane groups of ds_read_b128)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int QPR = BK / 4;        // float4 per tile row
  constexpr int RPP = 256 / QPR;     // tile rows staged per pass of the 256 threads
  constexpr int AR = BM / RPP, BR = BN / RPP;
  static_assert(!TAP4 || BK == 32, "the stem path stages one 7x7 tap per float4: BK must be 32");
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LS];
  float* As = lds;
  float* Bs = lds + BM * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  if (p.m_count) {
    const int c = *p.m_count;
    const int lim = c * p.m_unit;
    M = lim < M ? lim : M;
  }
  // Only the tiles that hold valid rows do work; the XCD remap is taken over THAT count so that a short dynamic
  // row count (e.g. 256 of 320 ROI slots) still spreads evenly over the 8 XCDs instead of idling the last ones.
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = tid / QPR, lq = tid % QPR;
  // Operand addressing.  Both tiles are fetched with SRSRC buffer loads (32-bit byte offsets + hardware range check):
  //  * every tile row gets ONE byte offset (its (ky,kx)=(0,0) tap position) and a bit mask of the taps that fall inside
  //    the image, both computed once per workgroup; per chunk a load costs an add, a bit test and a select -- no
  //    64-bit address arithmetic, no exec-mask branches; a masked-off / out-of-tile lane gets offset 0xFFFFFFFF, which
  //    the range check turns into zeros (the conv's zero padding);
  //  * a weight row's offset never changes: the K position goes into the scalar offset of the instruction.
  int a_off[AR], a_iy[AR], a_ix[AR];        // TAP4 (stem) path only
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
  const int ntaps = p.KH * p.KW;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lr + RPP * i;
    int iy0 = 0, ix0 = 0, off = 0, hh = 1, ww = 1;
    const bool rowok = m < M;
    if (rowok) {
      if (MULTI) {
        int l = 0;
        while (l + 1 < p.nlv && m >= p.lv_off[l + 1]) ++l;
        const int local = m - p.lv_off[l];
        ww = p.lv_w[l];
        hh = p.lv_h[l];
        const int oy = local / ww;
        iy0 = oy - p.pad;
        ix0 = (local - oy * ww) - p.pad;
        off = p.lv_off[l];
      } else {
        const int ox = m % p.OW;
        const int t2 = m / p.OW;
        const int oy = t2 % p.OH;
        const int img = t2 / p.OH;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
        off = img * p.H * p.W;
        hh = p.H;
        ww = p.W;
      }
    }
    a_iy[i] = rowok ? iy0 : -(1 << 28);
    a_ix[i] = ix0;
    a_off[i] = off;
    if (!TAP4) {
      unsigned long long mask = 0;
      if (rowok) {
        for (int tp = 0; tp < ntaps; ++tp) {
          const int ky = tp / p.KW, kx = tp - ky * p.KW;
          const bool ok = ((unsigned)(iy0 + ky) < (unsigned)hh) && ((unsigned)(ix0 + kx) < (unsigned)ww);
          mask |= (unsigned long long)ok << tp;
        }
      }
      a_mask[i] = mask;
      a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * lq) * 4);   // may wrap for padded taps: only used when the tap bit is set
      if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
    }
  }
  unsigned w_voff[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + RPP * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  f32x4 ar[AR], br[BR];
  auto load_chunk = [&](int chunk) {
    const int k0 = chunk * BK;
    if (!TAP4) {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int ky = tap / p.KW;
      const int kx = tap - ky * p.KW;
      const unsigned tap_off = MULTI ? (unsigned)((kx * p.Cin + c0) * 4) : (unsigned)(((ky * p.W + kx) * p.Cin + c0) * 4);
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = (a_mask[i] >> tap) & 1ull;
        unsigned vo = a_voff[i] + tap_off;
        if (MULTI) vo += (unsigned)ky * a_pitch[i];
        vo = ok ? vo : 0xFFFFFFFFu;
        ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, 0, 0));
      }
    } else {
      const int tap = chunk * 8 + lq;
      const int ky = tap / p.KW;
      const int kx = tap - ky * p.KW;
      const bool tv = tap < p.KH * p.KW;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        const bool ok = tv && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(a_off[i] + iy * p.W + ix) * 4);
        ar[i] = v;
      }
    }
    if (p.in_relu) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        ar[i].x = fmaxf(ar[i].x, 0.f);
        ar[i].y = fmaxf(ar[i].y, 0.f);
        ar[i].z = fmaxf(ar[i].z, 0.f);
        ar[i].w = fmaxf(ar[i].w, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < BR; ++j)
      br[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[j], k0 * 4, 0));
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // a 64x64 tile leaves one 32x32 accumulator per wave = one fully dependent MFMA chain: split it into two
  // independent chains (even / odd k-slots) that are added once at the end
  f32x16 acc_b;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_b[r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = 4 * (lane >> 5);
  const float* a_base = As + (wm * TM * 32 + frag_row) * LS + frag_k;
  const float* b_base = Bs + (wn * TN * 32 + frag_row) * LS + frag_k;

  load_chunk(c_begin);
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + (lr + RPP * i) * LS + 4 * lq) = ar[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(Bs + (lr + RPP * j) * LS + 4 * lq) = br[j];
    __syncthreads();
    if (chunk + 1 < c_end) load_chunk(chunk + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LS + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LS + kk * 8);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (TM * TN == 1 && (tt & 1))
              acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc_b, 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if (TM * TN == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_b[r];
  }
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + (wm * TM + i) * 32 + row;
        if (m < M && n < p.Cout) {
          const float v = acc[i][j][r];
          if (p.splitk > 1) {
            p.partial[((size_t)z * p.M + m) * p.Cout + n] = v;
          } else {
            epilogue_store(p, v, m, n);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the main loop (larger tiles, no register staging).
//   * both operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (out-of-range lanes land as zeros = the conv's
//     zero padding), two LDS stages, the DMA of chunk c+1 is in flight while chunk c is multiplied;
//   * LDS rows are 128 B (BK = 32) with no padding (a DMA wave-instruction writes 1 KiB linearly: 8 rows); the 16-byte slot
//     index is XOR-swizzled with (row >> 1) & 7 on the SOURCE address and on the READ so that the 16-lane groups of
//     ds_read_b128 are conflict-free;
//   * raw s_barrier + counted s_waitcnt vmcnt (a __syncthreads() would drain the DMA in flight).
template <int BM, int BN, bool MULTI>
__global__ __launch_bounds__(256) void conv_glds_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub
  constexpr int BK = 32;
  constexpr int ROWB = BK * 4;                 // bytes per tile row
  constexpr int STAGE = (BM + BN) * ROWB;      // bytes per stage
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AI = BM / 32, BI = BN / 32;    // DMA wave-instructions per wave and stage (8 rows each)
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  if (p.m_count) {
    const int c = *p.m_count;
    const int lim = c * p.m_unit;
    M = lim < M ? lim : M;
  }
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  // --- DMA side: this lane's rows -------------------------------------------------------------------
  const int lr8 = lane >> 3, sl = lane & 7;
  unsigned a_voff[AI];
  unsigned long long a_mask[AI];
  unsigned a_pitch[MULTI ? AI : 1];
  const int ntaps = p.KH * p.KW;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = 8 * (wave + 4 * i) + lr8;          // tile row written by this lane in DMA instruction i
    const int gslot = sl ^ ((row >> 1) & 7);           // global 16-byte slot that must land in LDS slot `sl`
    const int m = m0 + row;
    int iy0 = 0, ix0 = 0, off = 0, hh = 1, ww = 1;
    const bool rowok = m < M;
    if (rowok) {
      if (MULTI) {
        int l = 0;
        while (l + 1 < p.nlv && m >= p.lv_off[l + 1]) ++l;
        const int local = m - p.lv_off[l];
        ww = p.lv_w[l];
        hh = p.lv_h[l];
        const int oy = local / ww;
        iy0 = oy - p.pad;
        ix0 = (local - oy * ww) - p.pad;
        off = p.lv_off[l];
      } else {
        const int tq = (int)fdiv((unsigned)m, p.div_ow);
        const int ox = m - tq * p.OW;
        const int img = (int)fdiv((unsigned)tq, p.div_oh);
        const int oy = tq - img * p.OH;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
        off = img * p.H * p.W;
        hh = p.H;
        ww = p.W;
      }
    }
    unsigned long long mask = 0;
    if (rowok) {
      for (int tp = 0; tp < ntaps; ++tp) {
        const int ky = tp / p.KW, kx = tp - ky * p.KW;
        const bool ok = ((unsigned)(iy0 + ky) < (unsigned)hh) && ((unsigned)(ix0 + kx) < (unsigned)ww);
        mask |= (unsigned long long)ok << tp;
      }
    }
    a_mask[i] = mask;
    a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * gslot) * 4);
    if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
  }
  unsigned w_voff[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = 8 * (wave + 4 * j) + lr8;
    const int gslot = sl ^ ((row >> 1) & 7);
    const int n = n0 + row;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * gslot) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void lds_void;

  auto issue = [&](int chunk, int st) {
    const int k0 = chunk * BK;
    const int tap = k0 / p.Cin;
    const int c0 = k0 - tap * p.Cin;
    const int ky = tap / p.KW;
    const int kx = tap - ky * p.KW;
    const unsigned tap_off = MULTI ? (unsigned)((kx * p.Cin + c0) * 4) : (unsigned)(((ky * p.W + kx) * p.Cin + c0) * 4);
    char* sbase = lds + st * STAGE;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1ull;
      unsigned vo = a_voff[i] + tap_off;
      if (MULTI) vo += (unsigned)ky * a_pitch[i];
      vo = ok ? vo : 0xFFFFFFFFu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_void*)(sbase + 8 * (wave + 4 * i) * ROWB), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(sbase + BM * ROWB + 8 * (wave + 4 * j) * ROWB), 16, w_voff[j],
                                               k0 * 4, 0, 0);
  };

  // --- MFMA side: fragment addresses (loop invariant, stage offset added as an immediate) ---------------
  const int frow = lane & 31, fh = lane >> 5;
  unsigned a_addr[TM][4], b_addr[TN][4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wm * TM + i) * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a_addr[i][kk] = (unsigned)(row * ROWB + (((2 * kk + fh) ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = (wn * TN + j) * 32 + frow;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) b_addr[j][kk] = (unsigned)(BM * ROWB + row * ROWB + (((2 * kk + fh) ^ ((row >> 1) & 7)) << 4));
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int st) {
    const char* sbase = lds + st * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(sbase + a_addr[i][kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sbase + b_addr[j][kk]);
      if (p.in_relu) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          af[i].x = fmaxf(af[i].x, 0.f);
          af[i].y = fmaxf(af[i].y, 0.f);
          af[i].z = fmaxf(af[i].z, 0.f);
          af[i].w = fmaxf(af[i].w, 0.f);
        }
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
    }
  };

  issue(c_begin, 0);
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int st = (chunk - c_begin) & 1;
    if (chunk + 1 < c_end) {
      issue(chunk + 1, st ^ 1);
      if (AI + BI == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (AI + BI == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    if (st == 0) compute(0); else compute(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + (wm * TM + i) * 32 + row;
        if (m < M && n < p.Cout) {
          const float v = acc[i][j][r];
          if (p.splitk > 1) {
            p.partial[((size_t)z * p.M + m) * p.Cout + n] = v;
          } else {
            epilogue_store(p, v, m, n);
          }
        }
      }
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------
// fp32 convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) by operand splitting.
//   x = xh + xm + xl, three bf16 pieces of 8 significant bits each (round-to-nearest residuals): together they carry the
//   24-bit fp32 significand.  x*w = xh*wh + (xh*wm + xm*wh) + (xh*wl + xl*wh + xm*wm) + O(2^-24): six bf16 MFMAs per
//   K=16 step, every product exact in the fp32 accumulator.  The bf16 pipe is 16x the fp32-MFMA rate per MAC, so the
//   arithmetic ceiling is 16/6 = 2.67x the fp32-MFMA kernel at fp32-class accuracy (measured against an fp64 convolution in
//   tests/test_kernels_gpu.py).  Non-finite inputs turn into NaN (inf - inf in the residual) instead of propagating as inf.
//   * operands are fetched as fp32 exactly like conv_igemm_kernel (same buffer-load addressing, same epilogue), split
//     in registers at staging time and written to LDS as [row][xh(32) | xm(32) | xl(32)] bf16 + 16 B pad = 208 B rows
//     (13 slots of 16 B: odd pitch => the 16-lane groups of ds_read_b128 are conflict free);
//   * one ds_read_b128 per (32-row tile, piece, K=16 step) feeds the MFMA fragment directly: lane (r, h) takes
//     k = 8h .. 8h+7 of its row, the natural order of the staged bytes.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }

struct Split3 {
  uint2 h, m, l;
};
__device__ __forceinline__ Split3 split3(f32x4 v) {
  Split3 s;
  s.h.x = pk_bf16(v.x, v.y);
  s.h.y = pk_bf16(v.z, v.w);
  const float r0 = v.x - bf_lo(s.h.x), r1 = v.y - bf_hi(s.h.x), r2 = v.z - bf_lo(s.h.y), r3 = v.w - bf_hi(s.h.y);
  s.m.x = pk_bf16(r0, r1);
  s.m.y = pk_bf16(r2, r3);
  const float q0 = r0 - bf_lo(s.m.x), q1 = r1 - bf_hi(s.m.x), q2 = r2 - bf_lo(s.m.y), q3 = r3 - bf_hi(s.m.y);
  s.l.x = pk_bf16(q0, q1);
  s.l.y = pk_bf16(q2, q3);
  return s;
}

template <int BM, int BN, bool MULTI>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 32;
  constexpr int ROWB = 3 * 2 * BK + 16;     // 208 bytes per tile row
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AR = BM / 32, BR = BN / 32;  // float4 per thread and operand per chunk (8 threads per row, 32 rows per pass)
  __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * ROWB];
  char* As = lds;
  char* Bs = lds + BM * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  if (p.m_count) {
    const int c = *p.m_count;
    const int lim = c * p.m_unit;
    M = lim < M ? lim : M;
  }
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = tid >> 3, lq = tid & 7;
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
  const int ntaps = p.KH * p.KW;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lr + 32 * i;
    int iy0 = 0, ix0 = 0, off = 0, hh = 1, ww = 1;
    const bool rowok = m < M;
    if (rowok) {
      if (MULTI) {
        int l = 0;
        while (l + 1 < p.nlv && m >= p.lv_off[l + 1]) ++l;
        const int local = m - p.lv_off[l];
        ww = p.lv_w[l];
        hh = p.lv_h[l];
        const int oy = local / ww;
        iy0 = oy - p.pad;
        ix0 = (local - oy * ww) - p.pad;
        off = p.lv_off[l];
      } else {
        const int tq = (int)fdiv((unsigned)m, p.div_ow);
        const int ox = m - tq * p.OW;
        const int img = (int)fdiv((unsigned)tq, p.div_oh);
        const int oy = tq - img * p.OH;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
        off = img * p.H * p.W;
        hh = p.H;
        ww = p.W;
      }
    }
    unsigned long long mask = 0;
    if (rowok) {
      for (int tp = 0; tp < ntaps; ++tp) {
        const int ky = tp / p.KW, kx = tp - ky * p.KW;
        const bool ok = ((unsigned)(iy0 + ky) < (unsigned)hh) && ((unsigned)(ix0 + kx) < (unsigned)ww);
        mask |= (unsigned long long)ok << tp;
      }
    }
    a_mask[i] = mask;
    a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * lq) * 4);
    if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
  }
  unsigned w_voff[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + 32 * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  // Software pipeline: while chunk c is multiplied, the raw fp32 registers of chunk c+1 (fetched one iteration earlier) are
  // split into bf16 pieces between the MFMAs and immediately refilled with the loads of chunk c+2; the pieces go to LDS at
  // the top of the next iteration.  One staging unit (= one float4 of this thread) is attached to every group of MFMAs.
  f32x4 raw[AR + BR];
  Split3 sp[AR + BR];
  struct TapInfo { int tap, ky; unsigned tap_off, k0b; };
#ifdef ABL_NOGLOBAL
  bool chunk_guard = false;
#endif
  auto tap_info = [&](int chunk) {
    if (chunk > c_end - 1) chunk = c_end - 1;          // past-the-end prefetches re-read the last chunk (never used)
    const int k0 = chunk * BK;
    TapInfo ti;
    ti.tap = k0 / p.Cin;
    const int c0 = k0 - ti.tap * p.Cin;
    ti.ky = ti.tap / p.KW;
    const int kx = ti.tap - ti.ky * p.KW;
    ti.tap_off = MULTI ? (unsigned)((kx * p.Cin + c0) * 4) : (unsigned)(((ti.ky * p.W + kx) * p.Cin + c0) * 4);
    ti.k0b = (unsigned)(k0 * 4);
    return ti;
  };
  auto load_unit = [&](const TapInfo& ti, int u) {
#ifdef ABL_NOGLOBAL
    if (chunk_guard) return;
#endif
    if (u < AR) {
      const bool ok = (a_mask[u] >> ti.tap) & 1ull;
      unsigned vo = a_voff[u] + ti.tap_off;
      if (MULTI) vo += (unsigned)ti.ky * a_pitch[u];
      vo = ok ? vo : 0xFFFFFFFFu;
      raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, 0, 0));
    } else {
      raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[u - AR], ti.k0b, 0));
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  const char* a_base = As + ((wm * TM) * 32 + frow) * ROWB + fh * 16;
  const char* b_base = Bs + ((wn * TN) * 32 + frow) * ROWB + fh * 16;
  char* a_st = As + lr * ROWB + lq * 8;
  char* b_st = Bs + lr * ROWB + lq * 8;

  constexpr int UNITS = AR + BR;
  constexpr int GROUPS = 2 * TM * TN;
  constexpr int UPG = (UNITS + GROUPS - 1) / GROUPS;
  {
    const TapInfo t0 = tap_info(c_begin);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) load_unit(t0, u);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) sp[u] = split3(raw[u]);
    const TapInfo t1 = tap_info(c_begin + 1);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) load_unit(t1, u);
  }
#ifdef ABL_NOGLOBAL
  chunk_guard = true;
#endif
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
#ifdef ABL_NOLDSW
    if (chunk == c_begin)
#endif
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      char* dst = u < AR ? a_st + 32 * u * ROWB : b_st + 32 * (u - AR) * ROWB;
      *reinterpret_cast<uint2*>(dst) = sp[u].h;
      *reinterpret_cast<uint2*>(dst + 64) = sp[u].m;
      *reinterpret_cast<uint2*>(dst + 128) = sp[u].l;
    }
    __syncthreads();
    const TapInfo tn = tap_info(chunk + 2);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t af[TM][3], bfr[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          af[i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_base + i * 32 * ROWB + q * 64 + s * 32));
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          bfr[j][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b_base + j * 32 * ROWB + q * 64 + s * 32));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], acc[i][j], 0, 0, 0);
          const int g = (s * TM + i) * TN + j;
#pragma unroll
          for (int u = g * UPG; u < (g + 1) * UPG && u < UNITS; ++u) {
#ifdef ABL_NOSPLIT
            sp[u].h.x = __builtin_bit_cast(unsigned, raw[u].x); sp[u].h.y = __builtin_bit_cast(unsigned, raw[u].y);
            sp[u].m.x = __builtin_bit_cast(unsigned, raw[u].z); sp[u].m.y = __builtin_bit_cast(unsigned, raw[u].w);
            sp[u].l = sp[u].h;
#else
            sp[u] = split3(raw[u]);
#endif
            load_unit(tn, u);
          }
          // pin the group: one MFMA, then a slice of this group's split arithmetic; the refill load closes the group
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (UPG * 28 + 5) / 6, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, UPG, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
  }

  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + (wm * TM + i) * 32 + row;
        if (m < M && n < p.Cout) {
          const float v = acc[i][j][r];
          if (p.splitk > 1) {
            p.partial[((size_t)z * p.M + m) * p.Cout + n] = v;
          } else {
            epilogue_store(p, v, m, n);
          }
        }
      }
    }
  }
#endif
}

// 256 x 128 tile, 512 threads (8 waves as 4 x 2, 64 x 64 per wave), one workgroup per CU.
//   * two LDS stages of 384 rows x 208 B (159 744 B of the 160 KiB): the pieces of chunk c+1 are written into the other stage
//     WHILE chunk c is multiplied, so there is one barrier per chunk and no staging registers between iterations;
//   * every group of six MFMAs (one 32x32 tile, one K=16 step) carries one staging unit of this thread: split one float4
//     (22 VALU), three ds_write_b64, one buffer load that refills the raw register with chunk c+2;
//   * the fragments of the second K=16 step are read while the first one is multiplied.
template <bool MULTI>
__global__ __launch_bounds__(512) void conv_bf16x3_w8_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 128, BK = 32;
  constexpr int ROWB = 3 * 2 * BK + 16;     // 208
  constexpr int STAGE = (BM + BN) * ROWB;   // 79 872
  constexpr int AR = 4, BR = 2, UNITS = AR + BR;
  extern __shared__ __attribute__((aligned(16))) char lds_dyn[];
  char* lds = lds_dyn;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  if (p.m_count) {
    const int c = *p.m_count;
    const int lim = c * p.m_unit;
    M = lim < M ? lim : M;
  }
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = tid >> 3, lq = tid & 7;      // 64 rows per pass, 8 float4 per row
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
  const int ntaps = p.KH * p.KW;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lr + 64 * i;
    int iy0 = 0, ix0 = 0, off = 0, hh = 1, ww = 1;
    const bool rowok = m < M;
    if (rowok) {
      if (MULTI) {
        int l = 0;
        while (l + 1 < p.nlv && m >= p.lv_off[l + 1]) ++l;
        const int local = m - p.lv_off[l];
        ww = p.lv_w[l];
        hh = p.lv_h[l];
        const int oy = local / ww;
        iy0 = oy - p.pad;
        ix0 = (local - oy * ww) - p.pad;
        off = p.lv_off[l];
      } else {
        const int tq = (int)fdiv((unsigned)m, p.div_ow);
        const int ox = m - tq * p.OW;
        const int img = (int)fdiv((unsigned)tq, p.div_oh);
        const int oy = tq - img * p.OH;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
        off = img * p.H * p.W;
        hh = p.H;
        ww = p.W;
      }
    }
    unsigned long long mask = 0;
    if (rowok) {
      for (int tp = 0; tp < ntaps; ++tp) {
        const int ky = tp / p.KW, kx = tp - ky * p.KW;
        const bool ok = ((unsigned)(iy0 + ky) < (unsigned)hh) && ((unsigned)(ix0 + kx) < (unsigned)ww);
        mask |= (unsigned long long)ok << tp;
      }
    }
    a_mask[i] = mask;
    a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * lq) * 4);
    if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
  }
  unsigned w_voff[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + 64 * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  f32x4 raw[UNITS];
  bool in_loop = false;     // diagnostic builds (tools/ablate/run_bf16x3.py) drop parts of the loop body
  (void)in_loop;
  struct TapInfo { int tap, ky; unsigned tap_off, k0b; };
  auto tap_info = [&](int chunk) {
    if (chunk > c_end - 1) chunk = c_end - 1;          // past-the-end prefetches re-read the last chunk (never used)
    const int k0 = chunk * BK;
    TapInfo ti;
    ti.tap = k0 / p.Cin;
    const int c0 = k0 - ti.tap * p.Cin;
    ti.ky = ti.tap / p.KW;
    const int kx = ti.tap - ti.ky * p.KW;
    ti.tap_off = MULTI ? (unsigned)((kx * p.Cin + c0) * 4) : (unsigned)(((ti.ky * p.W + kx) * p.Cin + c0) * 4);
    ti.k0b = (unsigned)(k0 * 4);
    return ti;
  };
  auto load_unit = [&](const TapInfo& ti, int u) {
#ifdef ABL_NOGLOBAL
    if (in_loop) return;
#endif
    if (u < AR) {
      const bool ok = (a_mask[u] >> ti.tap) & 1ull;
      unsigned vo = a_voff[u] + ti.tap_off;
      if (MULTI) vo += (unsigned)ti.ky * a_pitch[u];
      vo = ok ? vo : 0xFFFFFFFFu;
      raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, 0, 0));
    } else {
      raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[u - AR], ti.k0b, 0));
    }
  };
  // staging unit u of this thread lands at st_off[u] inside a stage
  auto st_off = [&](int u) { return u < AR ? (lr + 64 * u) * ROWB + lq * 8 : (BM + lr + 64 * (u - AR)) * ROWB + lq * 8; };
  auto stage_unit = [&](char* stage, int u) {
#ifdef ABL_NOSPLIT
    Split3 s3;
    s3.h.x = __builtin_bit_cast(unsigned, raw[u].x); s3.h.y = __builtin_bit_cast(unsigned, raw[u].y);
    s3.m.x = __builtin_bit_cast(unsigned, raw[u].z); s3.m.y = __builtin_bit_cast(unsigned, raw[u].w);
    s3.l = s3.h;
    if (!in_loop) s3 = split3(raw[u]);
#else
    const Split3 s3 = split3(raw[u]);
#endif
#ifdef ABL_NOLDSW
    if (in_loop) {
      asm volatile("" ::"v"(s3.h.x), "v"(s3.h.y), "v"(s3.m.x), "v"(s3.m.y), "v"(s3.l.x), "v"(s3.l.y));
      return;
    }
#endif
    char* dst = stage + st_off(u);
    *reinterpret_cast<uint2*>(dst) = s3.h;
    *reinterpret_cast<uint2*>(dst + 64) = s3.m;
    *reinterpret_cast<uint2*>(dst + 128) = s3.l;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  const int a_fo = (wm * 64 + frow) * ROWB + fh * 16;
  const int b_fo = (BM + wn * 64 + frow) * ROWB + fh * 16;

  {
    const TapInfo t0 = tap_info(c_begin);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) load_unit(t0, u);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) stage_unit(lds, u);
    const TapInfo t1 = tap_info(c_begin + 1);
#pragma unroll
    for (int u = 0; u < UNITS; ++u) load_unit(t1, u);
  }
  in_loop = true;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int st = (chunk - c_begin) & 1;
    char* cur = lds + st * STAGE;
    char* nxt = lds + (st ^ 1) * STAGE;
    __syncthreads();      // stage `cur` fully written (previous iteration), stage `nxt` no longer read
    const TapInfo tn = tap_info(chunk + 2);
    bf16x8_t af[2][2][3], bfr[2][2][3];    // [K=16 step][tile][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        af[0][i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(cur + a_fo + i * 32 * ROWB + q * 64));
        bfr[0][i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(cur + b_fo + i * 32 * ROWB + q * 64));
      }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int g = (s * 2 + i) * 2 + j;
          if (s == 0) {     // prefetch a quarter of the second step's fragments per group
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              if (j == 0)
                af[1][i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(cur + a_fo + i * 32 * ROWB + q * 64 + 32));
              else
                bfr[1][i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(cur + b_fo + i * 32 * ROWB + q * 64 + 32));
            }
          }
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][2], bfr[s][j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][0], bfr[s][j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][1], bfr[s][j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][1], bfr[s][j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][0], bfr[s][j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i][0], bfr[s][j][0], acc[i][j], 0, 0, 0);
          if (g < UNITS) {
            stage_unit(nxt, g);
            load_unit(tn, g);
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            if (k < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k >= 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }

  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int m = m0 + (wm * 2 + i) * 32 + row;
        if (m < M && n < p.Cout) {
          const float v = acc[i][j][r];
          if (p.splitk > 1) {
            p.partial[((size_t)z * p.M + m) * p.Cout + n] = v;
          } else {
            epilogue_store(p, v, m, n);
          }
        }
      }
    }
  }
#endif
}

__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs p) {
  int M = p.M;
  if (p.m_count) {
    const int c = *p.m_count;
    const int lim = c * p.m_unit;
    M = lim < M ? lim : M;
  }
  const size_t total = (size_t)M * p.Cout;
  const size_t slab = (size_t)p.M * p.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int z = 0; z < p.splitk; ++z) v += p.partial[z * slab + idx];
    const int m = (int)(idx / p.Cout);
    const int n = (int)(idx - (size_t)m * p.Cout);
    epilogue_store(p, v, m, n);
  }
}

FastDiv make_fastdiv(unsigned d) {
  FastDiv f{};
  if (d == 0) d = 1;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mp = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 0 ? l - 1 : 0;
  f.d = d;
  return f;
}

inline int big_tile_env() {
  static const int v = [] {
    const char* e = getenv("EOD_CONV_BIG_TILE");
    return e ? atoi(e) : 0;
  }();
  return v;
}

inline int default_bk() {
  static const int v = [] {
    const char* e = getenv("EOD_CONV_BK");
    return (e && atoi(e) == 32) ? 32 : 64;   // measured: BK=64 (half the barriers) +16 % on the mask GEMM
  }();
  return v;
}

// Process-wide arithmetic mode of eod_conv2d (eod_set_conv_math / EOD_CONV_MATH): 0 fp32 MFMA, 1 bf16x3 split.
std::atomic<int>& math_mode() {
  static std::atomic<int> v([] {
    const char* e = getenv("EOD_CONV_MATH");
    return (e && e[0] == 'b') ? EOD_MATH_BF16X3 : EOD_MATH_FP32;
  }());
  return v;
}

struct Plan {
  int tile;  // 1=128x128 2=128x64 3=64x64
  int bk;    // K chunk staged per barrier pair: 32 or 64
  int glds;  // 1: LDS-DMA kernel (BK = 32); 2: bf16x3 split kernel (BK = 32)
  int bm, bn, tiles_m, tiles_n, splitk, cps, nchunks;
};

Plan make_plan(const EodConvDesc* d, int M, int nchunks32) {
  static const int cfg[4][2] = {{128, 128}, {128, 64}, {64, 64}, {256, 128}};   // the last one: bf16x3 8-wave kernel only
  Plan pl{};
  // Measured on MI355X (tools/conv_bench.py, profiles/r01_conv_bench.log): the 64x64 tile (7 waves/SIMD, finest
  // tile quantisation over 256 CUs) is the fastest or ties on every shape of this path, including the
  // 50k x 256 x 2304 mask-head GEMM (107 vs 92 TFLOP/s for 128x128).  The larger tiles stay selectable.
  int pick = 2;
  const int ft = d->force_tile % 10, fbk = d->force_tile / 10;   // force_tile = tile + 10 (BK 32) / + 20 (BK 64)
  if (ft >= 1 && ft <= 3) pick = ft - 1;
  else if (ft == 4 && fbk == 5 && !d->tap4 && !d->in_relu) pick = 3;
  else if (M >= 32768 && big_tile_env() >= 1 && big_tile_env() <= 3) pick = big_tile_env() - 1;   // experiment knob
  const bool bk64_ok = !d->tap4 && d->Cin % 64 == 0 && d->Kpad % 64 == 0;
  pl.glds = (fbk == 4 && !d->tap4) ? 1 : ((fbk == 5 && !d->tap4 && !d->in_relu) ? 2 : 0);
  if (d->force_tile == 0 && math_mode().load(std::memory_order_relaxed) == EOD_MATH_BF16X3 && !d->tap4 && !d->in_relu) {
    pl.glds = 2;
    const long t128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    const long t256 = (long)((M + 255) / 256) * ((d->Cout + 127) / 128);
    // measured (profiles/r01_bf16x3_accuracy_speed.log): 256x128 (8 waves, one workgroup per CU) once it fills the chip,
    // 128x128 once it fills two workgroups per CU, else the finest tile
    pick = t256 >= 256 ? 3 : (t128 >= 512 ? 0 : 2);
  }
  pl.bk = pl.glds ? 32 : ((fbk == 2 && bk64_ok) ? 64 : (fbk == 1 ? 32 : (bk64_ok && default_bk() == 64 ? 64 : 32)));
  const int nchunks = d->Kpad / pl.bk;
  (void)nchunks32;
  pl.nchunks = nchunks;
  pl.tile = pick + 1;
  pl.bm = cfg[pick][0];
  pl.bn = cfg[pick][1];
  pl.tiles_m = (M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (d->Cout + pl.bn - 1) / pl.bn;
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  int splitk = 1;
  if (d->force_splitk > 0) {
    splitk = d->force_splitk;
  } else if (tiles < 256 && nchunks >= 8) {
    int want = (int)((512 + tiles - 1) / tiles);
    int maxs = nchunks / 4;
    splitk = want < maxs ? want : maxs;
    if (splitk < 1) splitk = 1;
  }
  if (splitk > nchunks) splitk = nchunks;
  pl.cps = (nchunks + splitk - 1) / splitk;
  pl.splitk = (nchunks + pl.cps - 1) / pl.cps;
  return pl;
}

int total_rows(const EodConvDesc* d) { return d->levels > 0 ? d->level_off[d->levels] : d->N * d->OH * d->OW; }

int check_desc(const EodConvDesc* d) {
  if (!d || !d->x || !d->w || !d->y) return EOD_ERR_NULL;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return EOD_ERR_BAD_DIMS;
  if (d->levels <= 0 && (d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0)) return EOD_ERR_BAD_DIMS;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return EOD_ERR_BAD_DIMS;
  if (d->Kpad % 32 != 0 || d->Kpad < d->KH * d->KW * d->Cin) return EOD_ERR_BAD_DIMS;
  if (d->tap4) {
    if (d->Cin != 4) return EOD_ERR_BAD_DIMS;
  } else {
    if (d->Cin % 32 != 0 || d->Kpad != d->KH * d->KW * d->Cin) return EOD_ERR_BAD_DIMS;
  }
  if (d->levels > 0) {
    // pyramid mode: stride-1 'same' conv over up to 5 level images stored back to back
    if (d->levels > 5 || d->N != 1 || d->stride != 1 || d->KH != d->KW || d->pad != d->KH / 2 || d->tap4 || d->out_mode != 0 ||
        d->res_mode == 2 || d->level_off[0] != 0)
      return EOD_ERR_BAD_DIMS;
    for (int l = 0; l < d->levels; ++l)
      if (d->level_h[l] <= 0 || d->level_w[l] <= 0 || d->level_off[l + 1] - d->level_off[l] != d->level_h[l] * d->level_w[l])
        return EOD_ERR_BAD_DIMS;
  } else {
    if (d->OH != (d->H + 2 * d->pad - d->KH) / d->stride + 1) return EOD_ERR_BAD_DIMS;
    if (d->OW != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return EOD_ERR_BAD_DIMS;
  }
  // operand buffers are addressed with 32-bit byte offsets (buffer descriptors): < 2 GiB each
  if (d->levels <= 0 && (long)d->N * d->H * d->W * d->Cin >= (1L << 29)) return EOD_ERR_BAD_DIMS;
  if ((long)d->Cout * d->Kpad >= (1L << 29)) return EOD_ERR_BAD_DIMS;
  if (d->KH * d->KW > 64) return EOD_ERR_BAD_DIMS;
  if ((long)total_rows(d) * (d->Cout > d->Cin ? d->Cout : d->Cin) >= (1L << 31)) return EOD_ERR_BAD_DIMS;
  if (d->out_mode == 1 && (d->Cout % 4 != 0)) return EOD_ERR_BAD_DIMS;
  if (d->res_mode != 0 && !d->res) return EOD_ERR_NULL;
  if (d->res_mode == 2 && ((d->OH & 1) || (d->OW & 1))) return EOD_ERR_BAD_DIMS;
  if (d->m_count && d->m_unit <= 0) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(d->x) || !eod_aligned16(d->w)) return EOD_ERR_ALIGN;
  return EOD_OK;
}

template <int BM, int BN>
void launch_tile(const ConvArgs& a, bool tap4, int bk, int glds, dim3 grid, hipStream_t s) {
  if (glds == 2 && a.nlv > 0)
    hipLaunchKernelGGL((conv_bf16x3_kernel<BM, BN, true>), grid, dim3(256), 0, s, a);
  else if (glds == 2)
    hipLaunchKernelGGL((conv_bf16x3_kernel<BM, BN, false>), grid, dim3(256), 0, s, a);
  else if (glds && a.nlv > 0)
    hipLaunchKernelGGL((conv_glds_kernel<BM, BN, true>), grid, dim3(256), 0, s, a);
  else if (glds)
    hipLaunchKernelGGL((conv_glds_kernel<BM, BN, false>), grid, dim3(256), 0, s, a);
  else if (tap4)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, true, false>), grid, dim3(256), 0, s, a);
  else if (a.nlv > 0 && bk == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 64, false, true>), grid, dim3(256), 0, s, a);
  else if (a.nlv > 0)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, false, true>), grid, dim3(256), 0, s, a);
  else if (bk == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 64, false, false>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, false, false>), grid, dim3(256), 0, s, a);
}

}  // namespace

extern "C" int eod_set_conv_math(int mode) {
  if (mode != EOD_MATH_FP32 && mode != EOD_MATH_BF16X3) return EOD_ERR_BAD_DIMS;
  return math_mode().exchange(mode);
}

extern "C" int eod_get_conv_math(void) { return math_mode().load(); }

extern "C" size_t eod_conv2d_workspace_bytes(const EodConvDesc* d) {
  if (check_desc(d) != EOD_OK) return 0;
  const int M = total_rows(d);
  const int nchunks = d->Kpad / 32;
  const Plan pl = make_plan(d, M, nchunks);
  if (pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * M * d->Cout * sizeof(float);
}

extern "C" int eod_conv2d(const EodConvDesc* d, eod_stream_t stream) {
  const int st = check_desc(d);
  if (st != EOD_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvArgs a{};
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->res; a.y = d->y;
  a.partial = d->workspace;
  a.m_count = d->m_count; a.m_unit = d->m_unit;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.Kpad = d->Kpad;
  a.M = total_rows(d);
  a.div_ow = make_fastdiv((unsigned)(d->OW > 0 ? d->OW : 1));
  a.div_oh = make_fastdiv((unsigned)(d->OH > 0 ? d->OH : 1));
  a.div_cd = make_fastdiv((unsigned)((d->Cout >> 2) > 0 ? (d->Cout >> 2) : 1));
  {
    const size_t xe = d->levels > 0 ? (size_t)d->level_off[d->levels] * d->Cin : (size_t)d->N * d->H * d->W * d->Cin;
    a.x_bytes = (unsigned)(xe * sizeof(float));
    a.w_bytes = (unsigned)((size_t)d->Cout * d->Kpad * sizeof(float));
  }
  a.nlv = d->levels > 0 ? d->levels : 0;
  for (int l = 0; l < a.nlv; ++l) {
    a.lv_off[l] = d->level_off[l];
    a.lv_h[l] = d->level_h[l];
    a.lv_w[l] = d->level_w[l];
  }
  if (a.nlv) a.lv_off[a.nlv] = d->level_off[a.nlv];
  a.relu = d->relu; a.res_mode = d->res_mode; a.in_relu = d->in_relu; a.out_mode = d->out_mode;
  a.out_scale = d->out_scale;
  const Plan pl = make_plan(d, a.M, d->Kpad / 32);
  a.nchunks = pl.nchunks;
  a.splitk = pl.splitk; a.cps = pl.cps; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
  if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * a.M * a.Cout * sizeof(float);
    if (!d->workspace || d->workspace_bytes < need) return EOD_ERR_CAPACITY;
  }
  dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk);
  switch (pl.tile) {
    case 4: {
      constexpr int kLds = 2 * (256 + 128) * 208;
      static const bool attr = [] {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        return true;
      }();
      (void)attr;
      if (a.nlv > 0) hipLaunchKernelGGL((conv_bf16x3_w8_kernel<true>), grid, dim3(512), kLds, s, a);
      else hipLaunchKernelGGL((conv_bf16x3_w8_kernel<false>), grid, dim3(512), kLds, s, a);
      break;
    }
    case 1: launch_tile<128, 128>(a, d->tap4 != 0, pl.bk, pl.glds, grid, s); break;
    case 2: launch_tile<128, 64>(a, d->tap4 != 0, pl.bk, pl.glds, grid, s); break;
    default: launch_tile<64, 64>(a, d->tap4 != 0, pl.bk, pl.glds, grid, s); break;
  }
  if (pl.splitk > 1) {
    const size_t total = (size_t)a.M * a.Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
  }
  return eod_launch_status();
}
