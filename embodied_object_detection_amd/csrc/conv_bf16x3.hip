// fp32 convolution emulated on the bf16 matrix cores (three-way operand split): opt-in arithmetic mode, DESIGN.md §3.
#include "conv_common.h"

namespace eodconv {
namespace {

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
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

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
  M = conv_row_limit(p, M);
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (!conv_tile_active(p, m0, BM)) return;
  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = tid >> 3, lq = tid & 7;
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
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
      mask = tap_mask(iy0, ix0, hh, ww, p.KH, p.KW);
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
  // tap_info() is called for consecutive chunks (c_begin, c_begin + 1, ...): the (tap, channel) position is advanced instead of
  // re-derived with two divisions per chunk.  The one or two calls past c_end describe chunks that are fetched (range-checked
  // buffer loads) and never used.
  int nx_tap, nx_c0, nx_ky, nx_kx, nx_k0 = c_begin * BK;
  nx_tap = nx_k0 / p.Cin;
  nx_c0 = nx_k0 - nx_tap * p.Cin;
  nx_ky = nx_tap / p.KW;
  nx_kx = nx_tap - nx_ky * p.KW;
  auto tap_info = [&](int) {
    TapInfo ti;
    ti.tap = nx_tap < 63 ? nx_tap : 63;
    ti.ky = nx_ky;
    ti.tap_off = MULTI ? (unsigned)((nx_kx * p.Cin + nx_c0) * 4) : (unsigned)(((nx_ky * p.W + nx_kx) * p.Cin + nx_c0) * 4);
    ti.k0b = (unsigned)(nx_k0 * 4);
    nx_k0 += BK;
    nx_c0 += BK;
    if (nx_c0 >= p.Cin) {
      nx_c0 = 0;
      ++nx_tap;
      if (++nx_kx == p.KW) {
        nx_kx = 0;
        ++nx_ky;
      }
    }
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

  store_wave_tiles<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, z, lane);
#endif
}

// 256 x 128 tile, 512 threads (8 waves as 4 x 2, 64 x 64 per wave), one workgroup per CU.
//   * two LDS stages of 384 rows x 208 B (159 744 B of the 160 KiB): the pieces of chunk c+1 are written into the other stage
//     WHILE chunk c is multiplied, so there is one barrier per chunk and no staging registers between iterations;
//   * every group of six MFMAs (one 32x32 tile, one K=16 step) carries one staging unit of this thread: split one float4
//     (22 VALU), three ds_write_b64, one buffer load that refills the raw register with chunk c+2;
//   * the fragments of the second K=16 step are read while the first one is multiplied.
// WS: the weights come pre-split ([Cout][Kpad/32][xh(32)|xm(32)|xl(32)] bf16, eod_conv_split_weights_bf16x3): their LDS image is
// a plain copy (three 16-byte pieces per thread and chunk), no split arithmetic for the static operand.
template <bool MULTI, bool WS>
__global__ __launch_bounds__(512) void conv_bf16x3_w8_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 128, BK = 32;
  constexpr int ROWB = 3 * 2 * BK + 16;     // 208
  constexpr int STAGE = (BM + BN) * ROWB;   // 79 872
  constexpr int AR = 4, BR = WS ? 0 : 2, UNITS = AR + BR;
  constexpr int BP = 3;                     // WS: 16-byte weight pieces per thread and chunk (128 rows x 12 pieces / 512 threads)
  extern __shared__ __attribute__((aligned(16))) char lds_dyn[];
  char* lds = lds_dyn;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  M = conv_row_limit(p, M);
  const int ntiles = ((M + BM - 1) / BM) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (!conv_tile_active(p, m0, BM)) return;
  const int z = blockIdx.y;
  const int c_begin = z * p.cps;
  int c_end = c_begin + p.cps;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = tid >> 3, lq = tid & 7;      // 64 rows per pass, 8 float4 per row
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
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
      mask = tap_mask(iy0, ix0, hh, ww, p.KH, p.KW);
    }
    a_mask[i] = mask;
    a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * lq) * 4);
    if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
  }
  unsigned w_voff[BR > 0 ? BR : 1];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + 64 * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  // WS: piece q = tid + 512 u of the [128 rows][12 pieces] weight tile
  unsigned w3_voff[BP];
  int w3_lds[BP];
#pragma unroll
  for (int u = 0; u < BP; ++u) {
    const int q = tid + 512 * u;
    const int row = q / 12, pc = q - row * 12;
    const int n = n0 + row;
    w3_voff[u] = n < p.Cout ? (unsigned)(n * (p.Kpad / 32) * 192 + pc * 16) : 0xFFFFFFFFu;
    w3_lds[u] = (BM + row) * ROWB + pc * 16;
  }
  const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), 0, WS ? p.w3_bytes : 0u, 0x00020000);
  u32x4_t braw[BP];
  auto load_w3 = [&](int chunk) {
    if (chunk > c_end - 1) chunk = c_end - 1;
#pragma unroll
    for (int u = 0; u < BP; ++u) braw[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w3, w3_voff[u], chunk * 192, 0);
  };
  auto stage_w3 = [&](char* stage) {
#pragma unroll
    for (int u = 0; u < BP; ++u) *reinterpret_cast<u32x4_t*>(stage + w3_lds[u]) = braw[u];
  };
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  f32x4 raw[UNITS];
  bool in_loop = false;     // diagnostic builds (tools/ablate/run_bf16x3.py) drop parts of the loop body
  (void)in_loop;
  struct TapInfo { int tap, ky; unsigned tap_off, k0b; };
  // tap_info() is called for consecutive chunks (c_begin, c_begin + 1, ...): the (tap, channel) position is advanced instead of
  // re-derived with two divisions per chunk.  The one or two calls past c_end describe chunks that are fetched (range-checked
  // buffer loads) and never used.
  int nx_tap, nx_c0, nx_ky, nx_kx, nx_k0 = c_begin * BK;
  nx_tap = nx_k0 / p.Cin;
  nx_c0 = nx_k0 - nx_tap * p.Cin;
  nx_ky = nx_tap / p.KW;
  nx_kx = nx_tap - nx_ky * p.KW;
  auto tap_info = [&](int) {
    TapInfo ti;
    ti.tap = nx_tap < 63 ? nx_tap : 63;
    ti.ky = nx_ky;
    ti.tap_off = MULTI ? (unsigned)((nx_kx * p.Cin + nx_c0) * 4) : (unsigned)(((nx_ky * p.W + nx_kx) * p.Cin + nx_c0) * 4);
    ti.k0b = (unsigned)(nx_k0 * 4);
    nx_k0 += BK;
    nx_c0 += BK;
    if (nx_c0 >= p.Cin) {
      nx_c0 = 0;
      ++nx_tap;
      if (++nx_kx == p.KW) {
        nx_kx = 0;
        ++nx_ky;
      }
    }
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
    if (WS) {
      load_w3(c_begin);
      stage_w3(lds);
      load_w3(c_begin + 1);
    }
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
          if (WS && g == UNITS) {      // the weight pieces of chunk c+1 -> other stage, then refill with chunk c+2
            stage_w3(nxt);
            load_w3(chunk + 2);
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            if (k < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (k >= 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          if (WS && g == UNITS) __builtin_amdgcn_sched_group_barrier(0x020, BP, 0);
          else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }

  store_wave_tiles<2, 2>(p, acc, m0 + wm * 2 * 32, n0 + wn * 2 * 32, M, z, lane);
#endif
}

}  // namespace

// [Cout][Kpad] fp32 -> [Cout][Kpad/32][xh(32) | xm(32) | xl(32)] bf16: the LDS row image of the bf16x3 kernels, 192 B per K chunk
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, size_t quads,
                                                           int kquads) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / kquads;
    const int kq = (int)(i - n * kquads);               // float4 index inside the row
    const Split3 s3 = split3(*reinterpret_cast<const f32x4*>(w + i * 4));
    unsigned char* dst = out + (n * (kquads / 8) + kq / 8) * 192 + (kq & 7) * 8;
    *reinterpret_cast<uint2*>(dst) = s3.h;
    *reinterpret_cast<uint2*>(dst + 64) = s3.m;
    *reinterpret_cast<uint2*>(dst + 128) = s3.l;
  }
}

void launch_split_weights(const float* w, void* out, int Cout, int Kpad, hipStream_t s) {
  const size_t quads = (size_t)Cout * Kpad / 4;
  int blocks = (int)((quads + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<unsigned char*>(out), quads, Kpad / 4);
}

template <int BM, int BN>
static void launch_b3_tile(const ConvArgs& a, dim3 grid, hipStream_t s) {
  if (a.nlv > 0) hipLaunchKernelGGL((conv_bf16x3_kernel<BM, BN, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_bf16x3_kernel<BM, BN, false>), grid, dim3(256), 0, s, a);
}

void launch_conv_bf16x3(const ConvArgs& a, int tile, dim3 grid, hipStream_t s) {
  switch (tile) {
    case 4: {
      constexpr int kLds = 2 * (256 + 128) * 208;
      static const bool attr = [] {
        bool ok = true;
        for (const void* f : {reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<true, false>),
                              reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<false, false>),
                              reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<true, true>),
                              reinterpret_cast<const void*>(&conv_bf16x3_w8_kernel<false, true>)})
          ok = (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess) && ok;
        return ok;
      }();
      (void)attr;      // a refused attribute shows up as a launch error (EOD_ERR_LAUNCH) below
      if (a.w3) {
        if (a.nlv > 0) hipLaunchKernelGGL((conv_bf16x3_w8_kernel<true, true>), grid, dim3(512), kLds, s, a);
        else hipLaunchKernelGGL((conv_bf16x3_w8_kernel<false, true>), grid, dim3(512), kLds, s, a);
      } else {
        if (a.nlv > 0) hipLaunchKernelGGL((conv_bf16x3_w8_kernel<true, false>), grid, dim3(512), kLds, s, a);
        else hipLaunchKernelGGL((conv_bf16x3_w8_kernel<false, false>), grid, dim3(512), kLds, s, a);
      }
      break;
    }
    case 1: launch_b3_tile<128, 128>(a, grid, s); break;
    case 2: launch_b3_tile<128, 64>(a, grid, s); break;
    default: launch_b3_tile<64, 64>(a, grid, s); break;
  }
}

}  // namespace eodconv
