// 3x3 / stride 1 / pad 1 convolution on small images (the mask head: 14x14 ROI tiles, M = rois * 196, N = 256, K = 2304) with the
// input HALO staged once per channel chunk: fp32 MFMA (v_mfma_f32_32x32x2_f32), same arithmetic as conv_fp32.hip.
//
// conv_fp32.hip walks K as (ky, kx, c): every 32-wide chunk re-fetches the BM input pixels of ONE tap, so each input pixel is
// brought from L2 to LDS nine times (measured: operand fetch costs 22 % of the mask GEMM, profiles/r01_conv_ablation.log).  In an
// NHWC row list the nine taps of output row m are the rows m + (ky-1) W + (kx-1): a tile of BM consecutive output rows only ever
// touches the BM + 2 (W + 1) consecutive input rows around it.  So here K is walked as (c-chunk, ky, kx): per 32-channel chunk the
// slab of BM + 2 (W + 1) input rows is staged ONCE (94 rows for BM = 64, W = 14, instead of 9 x 64) and the nine taps read it at
// shifted row addresses; a tap that falls outside its 14x14 image reads a zero row instead (per-lane 9-bit mask), which also keeps
// the neighbouring ROI's rows inside the slab from leaking in.  Weights: one [BN x 32] chunk per (c-chunk, tap), double buffered;
// ONE barrier per step (the classic kernel needs two).  fp32 accumulation order differs from conv_fp32.hip (K permuted), the
// numerics class is the same.
#include "conv_common.h"

namespace eodconv {
namespace {

constexpr int HALO_MAX = 16;   // W + 1 <= 16

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_halo_kernel(ConvArgs p) {
  constexpr int LS = 36;                       // LDS row stride in floats (32 + 4: conflict-free 16-lane groups of ds_read_b128)
  constexpr int SLAB = BM + 2 * HALO_MAX;      // slab rows staged per channel chunk
  constexpr int ZROW = SLAB;                   // one extra all-zero row: the conv's zero padding
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AP = SLAB / 32;                // slab float4 per thread and chunk (32 rows per pass of the 256 threads)
  constexpr int BR = BN / 32;
  static_assert(SLAB % 32 == 0, "slab rows must be a multiple of 32");
  __shared__ __attribute__((aligned(16))) float lds[2 * (SLAB + 1) * LS + 2 * BN * LS];
  float* As = lds;                             // [2][SLAB + 1][LS]
  float* Bs = lds + 2 * (SLAB + 1) * LS;       // [2][BN][LS]

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
  const int halo = p.W + 1;

  // zero row(s) of both slab buffers
  if (tid < 2 * LS) As[(tid / LS) * (SLAB + 1) * LS + ZROW * LS + (tid % LS)] = 0.f;

  // ---- staging addresses -----------------------------------------------------------------------------------------------
  const int lr = tid >> 3, lq = tid & 7;       // 32 rows x 8 float4 per pass
  unsigned a_voff[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int j = lr + 32 * i;                 // slab row
    const long g = (long)m0 - halo + j;        // input row; rows outside [0, capacity) land as zeros (buffer range check)
    a_voff[i] = (g >= 0 && g < (long)p.M && j < BM + 2 * halo) ? (unsigned)((g * p.Cin + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  unsigned w_voff[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lr + 32 * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- MFMA side: per wave tile row the slab row of tap (1,1) and the 9-bit mask of taps inside the image ------------------
  const int frag_row = lane & 31;
  const int frag_k = 4 * (lane >> 5);
  int a_row[TM];
  unsigned a_mask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wm * TM + i) * 32 + frag_row;
    const int m = m0 + r;
    a_row[i] = r + halo;
    unsigned mask = 0;
    if (m < p.M) {
      const int t2 = (int)fdiv((unsigned)m, p.div_ow);
      const int ox = m - t2 * p.OW;
      const int img = (int)fdiv((unsigned)t2, p.div_oh);
      const int oy = t2 - img * p.OH;
      mask = (unsigned)tap_mask(oy - 1, ox - 1, p.H, p.W, 3, 3);
    }
    a_mask[i] = mask;
  }
  const float* b_base = Bs + (wn * TN * 32 + frag_row) * LS + frag_k;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 acc_b;     // second chain of a 64x64 tile (one 32x32 accumulator per wave): even / odd k-slots, added at the end
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_b[r] = 0.f;

  const int cchunks = p.Cin >> 5;
  const int nsteps = cchunks * 9;
  f32x4 ar[AP], br[2][BR];     // weights are fetched TWO steps ahead (two register sets): ~2 x 1 024 MFMA cycles to cover the L2 latency

  auto load_slab = [&](int cc) {
#pragma unroll
    for (int i = 0; i < AP; ++i) ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, a_voff[i], cc * 128, 0));
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(As + buf * (SLAB + 1) * LS + (lr + 32 * i) * LS + 4 * lq) = ar[i];
  };
  auto load_w = [&](int set, int step) {
    const int c = step / 9, tp = step - 9 * c;
    const int k0 = tp * p.Cin + c * 32;
#pragma unroll
    for (int j = 0; j < BR; ++j) br[set][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[j], k0 * 4, 0));
  };
  auto store_w = [&](int set, int buf) {
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(Bs + buf * BN * LS + (lr + 32 * j) * LS + 4 * lq) = br[set][j];
  };

  // prologue: slab 0 and weight chunk of step 0 into LDS, weight chunk of step 1 into registers
  load_slab(0);
  load_w(0, 0);
  store_slab(0);
  store_w(0, 0);
  if (nsteps > 1) load_w(1, 1);
  __syncthreads();

  int cc = 0, tap = 0;
#pragma unroll 2
  for (int step = 0; step < nsteps; ++step) {
    // next step's coordinates
    int ntap = tap + 1, ncc = cc;
    if (ntap == 9) {
      ntap = 0;
      ++ncc;
    }
    const bool has_next = step + 1 < nsteps;
    // the slab of c-chunk cc+1 is fetched during tap 0 of cc and written to the other slab buffer after tap 2 (that buffer was last
    // read during c-chunk cc-1)
    const bool slab_load = tap == 0 && cc + 1 < cchunks;
    const bool slab_store = tap == 2 && cc + 1 < cchunks;
    if (step + 2 < nsteps) load_w(step & 1, step + 2);      // register set (step & 1) was stored to LDS at the end of step - 1
    if (slab_load) load_slab(cc + 1);

    // ---- MFMAs of this step out of slab buffer cc & 1 and weight buffer step & 1 ----
    const float* a_buf = As + (cc & 1) * (SLAB + 1) * LS + frag_k;
    const float* b_buf = b_base + (step & 1) * BN * LS;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int off = (ky - 1) * p.W + (kx - 1);
    const float* a_ptr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1u;
      a_ptr[i] = a_buf + (ok ? a_row[i] + off : ZROW) * LS;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b_buf + j * 32 * LS + kk * 8);
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
    if (has_next) store_w((step + 1) & 1, (step + 1) & 1);   // the chunk of step + 1 was loaded during step - 1 into set (step + 1) & 1
    if (slab_store) store_slab((cc + 1) & 1);
    __syncthreads();
    tap = ntap;
    cc = ncc;
  }

  if (TM * TN == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_b[r];
  }
  store_wave_tiles<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, 0, lane);
}

}  // namespace

void launch_conv_halo(const ConvArgs& a, int tile, dim3 grid, hipStream_t s) {
  switch (tile) {
    case 2: hipLaunchKernelGGL((conv_halo_kernel<128, 64>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_halo_kernel<64, 64>), grid, dim3(256), 0, s, a); break;
  }
}

}  // namespace eodconv
