// Experimental LDS-DMA staging of the fp32 implicit-GEMM convolution (force_tile 41-43; measured, not the default: DESIGN.md §3).
#include "conv_common.h"

namespace eodconv {
namespace {

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
      mask = tap_mask(iy0, ix0, hh, ww, p.KH, p.KW);
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

  store_wave_tiles<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, z, lane);
#endif
}

}  // namespace

template <int BM, int BN>
static void launch_glds_tile(const ConvArgs& a, dim3 grid, hipStream_t s) {
  if (a.nlv > 0) hipLaunchKernelGGL((conv_glds_kernel<BM, BN, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_glds_kernel<BM, BN, false>), grid, dim3(256), 0, s, a);
}

void launch_conv_glds(const ConvArgs& a, int tile, dim3 grid, hipStream_t s) {
  switch (tile) {
    case 1: launch_glds_tile<128, 128>(a, grid, s); break;
    case 2: launch_glds_tile<128, 64>(a, grid, s); break;
    default: launch_glds_tile<64, 64>(a, grid, s); break;
  }
}

}  // namespace eodconv
