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
#include "conv_common.h"

#ifndef EOD_MFMA_PRIO
#define EOD_MFMA_PRIO 1
#endif
#ifndef EOD_LDS_PIPE
#define EOD_LDS_PIPE 1
#endif

namespace eodconv {
namespace {

// PF2: two operand sets in registers -- chunks c+1 and c+2 are in flight while chunk c is multiplied.  For launches that leave only
// 2-5 workgroups on a CU (the mask head on ~40 or ~90 ROIs) the global-load latency is no longer hidden by other workgroups'
// MFMAs; one more chunk of prefetch hides it.
// PIPE 2: the operand tiles are double buffered in LDS -- chunk c+1 is written to the other buffer while chunk c is multiplied: ONE
// workgroup barrier per chunk instead of two, at twice the LDS (36 KB for 64x64: 4 workgroups per CU).
template <int BM, int BN, int BK, bool TAP4, bool MULTI, int PIPE = 0>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
  constexpr bool PF2 = PIPE == 1;
  constexpr bool DB = PIPE == 2;
  constexpr int LS = BK + 4;  // LDS row stride in floats (+4: conflict-free 16-lane groups of ds_read_b128)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int QPR = BK / 4;        // float4 per tile row
  constexpr int RPP = 256 / QPR;     // tile rows staged per pass of the 256 threads
  constexpr int AR = BM / RPP, BR = BN / RPP;
  static_assert(!TAP4 || BK == 32, "the stem path stages one 7x7 tap per float4: BK must be 32");
  constexpr int BUF = (BM + BN) * LS;
  __shared__ __attribute__((aligned(16))) float lds[(DB ? 2 : 1) * BUF];
  float* As = lds;
  float* Bs = lds + BM * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int M = p.M;
  M = conv_row_limit(p, M);
  // Only the tiles that hold valid rows do work; the XCD remap is taken over THAT count so that a short dynamic
  // row count (e.g. 256 of 320 ROI slots) still spreads evenly over the 8 XCDs instead of idling the last ones.
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
        const int t2 = (int)fdiv((unsigned)m, p.div_ow);      // invariant divisors: one mul_hi instead of a division sequence
        const int ox = m - t2 * p.OW;
        const int img = (int)fdiv((unsigned)t2, p.div_oh);
        const int oy = t2 - img * p.OH;
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
        mask = tap_mask(iy0, ix0, hh, ww, p.KH, p.KW);
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

  f32x4 ar0[AR], br0[BR];
  f32x4 ar1[PF2 ? AR : 1], br1[PF2 ? BR : 1];
  // (tap, channel offset) of the NEXT chunk to fetch: chunks are fetched in order, so the position is advanced instead of
  // re-derived with two divisions per chunk (Cin is a multiple of BK on this path)
  int nx_tap = 0, nx_c0 = 0, nx_ky = 0, nx_kx = 0;
  if (!TAP4) {
    const int k0 = c_begin * BK;
    nx_tap = k0 / p.Cin;
    nx_c0 = k0 - nx_tap * p.Cin;
    nx_ky = nx_tap / p.KW;
    nx_kx = nx_tap - nx_ky * p.KW;
  }
  auto load_chunk = [&](int chunk, auto& ar, auto& br) {
    const int k0 = chunk * BK;
    if (!TAP4) {
      const int tap = nx_tap, c0 = nx_c0, ky = nx_ky, kx = nx_kx;
      nx_c0 += BK;
      if (nx_c0 >= p.Cin) {
        nx_c0 = 0;
        ++nx_tap;
        if (++nx_kx == p.KW) {
          nx_kx = 0;
          ++nx_ky;
        }
      }
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

  auto stage_and_multiply = [&](auto& ar, auto& br, int next_chunk) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + (lr + RPP * i) * LS + 4 * lq) = ar[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(Bs + (lr + RPP * j) * LS + 4 * lq) = br[j];
    __syncthreads();
    if (next_chunk < c_end) load_chunk(next_chunk, ar, br);
    if constexpr (EOD_LDS_PIPE && TM * TN == 1) {
      // 64x64 tile: the fragment reads of k-step kk + 1 are issued before the four MFMAs of k-step kk (two fragment sets in
      // registers), so that the LDS latency runs under the MFMAs of the same wave instead of after them
      f32x4 af[2], bf[2];
      af[0] = *reinterpret_cast<const f32x4*>(a_base);
      bf[0] = *reinterpret_cast<const f32x4*>(b_base);
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 8) {
          af[nxt] = *reinterpret_cast<const f32x4*>(a_base + (kk + 1) * 8);
          bf[nxt] = *reinterpret_cast<const f32x4*>(b_base + (kk + 1) * 8);
        }
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (tt & 1)
            acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][tt], bf[cur][tt], acc_b, 0, 0, 0);
          else
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][tt], bf[cur][tt], acc[0][0], 0, 0, 0);
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LS + kk * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LS + kk * 8);
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
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
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();
  };
  load_chunk(c_begin, ar0, br0);
  if constexpr (DB) {
    auto store_tiles = [&](int buf) {
#pragma unroll
      for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + buf * BUF + (lr + RPP * i) * LS + 4 * lq) = ar0[i];
#pragma unroll
      for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(Bs + buf * BUF + (lr + RPP * j) * LS + 4 * lq) = br0[j];
    };
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
      const bool more = chunk + 1 < c_end;
      if (more) load_chunk(chunk + 1, ar0, br0);
      const float* ab = a_base + cur * BUF;
      const float* bb = b_base + cur * BUF;
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(ab + i * 32 * LS + kk * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bb + j * 32 * LS + kk * 8);
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(1);
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
        if (EOD_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
      }
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  } else if constexpr (PF2) {
    if (c_begin + 1 < c_end) load_chunk(c_begin + 1, ar1, br1);
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
      stage_and_multiply(ar0, br0, chunk + 2);
      if (chunk + 1 < c_end) stage_and_multiply(ar1, br1, chunk + 3);
    }
  } else {
    for (int chunk = c_begin; chunk < c_end; ++chunk) stage_and_multiply(ar0, br0, chunk + 1);
  }

  if (TM * TN == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_b[r];
  }
  if constexpr (BM == 64 && BN == 256) {
    if (p.out_mode == 2) {
      // Fused mask-head tail: this workgroup holds deconv quadrant `tile_n` (dy, dx) of 64 input pixels for all 256 channels.
      // Per pixel: sum_co relu(acc + b[co]) * pw[co]; reduced in the lane (4 tiles), across the 32 lanes of a half-wave
      // (butterfly), then across the two column waves through LDS -- a fixed order, so the result is reproducible.
      float sred[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sred[r] = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = (wn * TN + j) * 32 + (lane & 31);
        const float bw = p.bias ? p.bias[co] : 0.f;
        const float pw = p.fuse_w[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = (acc[0][j][r] + bw) * p.out_scale;
          if (p.relu) v = fmaxf(v, 0.f);
          sred[r] += v * pw;
        }
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) sred[r] += __shfl_xor(sred[r], off, 64);
      float* red = lds;      // [wm][wn][32 rows]; the tiles are dead after the loop's final barrier
      if ((lane & 31) == 0) {
        const int half = lane >> 5;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wm * 2 + wn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = sred[r];
      }
      __syncthreads();
      if (tid < BM) {
        const int wr = tid >> 5, rr = tid & 31;
        const float v = red[(wr * 2 + 0) * 32 + rr] + red[(wr * 2 + 1) * 32 + rr] + p.fuse_b;
        const int m = m0 + tid;
        if (m < M) {
          const int tq = (int)fdiv((unsigned)m, p.div_ow);
          const int ox = m - tq * p.OW;
          const int img = (int)fdiv((unsigned)tq, p.div_oh);
          const int oy = tq - img * p.OH;
          const int unit = p.out_units ? p.out_units[img] : img;
          const int dy = tile_n >> 1, dx = tile_n & 1;
          p.y[((size_t)unit * 2 * p.OH + 2 * oy + dy) * (2 * p.OW) + 2 * ox + dx] = eod_sigmoid_precise(v);
        }
      }
      return;
    }
  }
  store_wave_tiles<TM, TN, BM == 64 && BN == 64>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, M, z, lane);
}


// ------------------------------------------------------------------------------------------------------
// Few-row / deep-K layers without split-K slabs: the K range is split over the WAVES of a workgroup.
//
// A layer with fewer than 256 tiles of 64x64 (ResNet layer3/4 at batch 1, FPN laterals, the box heads' 1024-wide FC layers,
// P6/P7) used to be split along K into fp32 slabs in global memory + a reduce launch.  Here a workgroup owns a 32x32 output
// tile and each of its NW waves (4 or 8) walks a contiguous 1/NW of K with wave-PRIVATE operand staging: no workgroup barrier
// inside the K loop, no slabs, no second launch.  The NW partial accumulators are added in wave order through LDS (fixed
// order: deterministic) and wave 0 applies the fused epilogue.  The K walk does not depend on the row count, so a batch of
// images gives bitwise the rows of the single-image call without a shared plan.
// ------------------------------------------------------------------------------------------------------
template <int NW, bool MULTI>
__global__ __launch_bounds__(NW * 64) void conv_wavek_kernel(ConvArgs p) {
  constexpr int BK = 32;
  constexpr int LS = BK + 4;
  constexpr int AR = 4;          // tile rows per lane and operand: row = (lane >> 3) + 8 i, float4 column = lane & 7
  __shared__ __attribute__((aligned(16))) float lds[NW * 2 * 32 * LS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  float* As = lds + wave * (2 * 32 * LS);
  float* Bs = As + 32 * LS;

  int M = p.M;
  M = conv_row_limit(p, M);
  const int ntiles = ((M + 31) / 32) * p.tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  const int t = xcd_remap(blockIdx.x, ntiles);
  const int tile_m = t / p.tiles_n;
  const int tile_n = t - tile_m * p.tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  if (!conv_tile_active(p, m0, 32)) return;

  // this wave's chunks
  const int cpw = (p.nchunks + NW - 1) / NW;
  const int c_begin = wave * cpw;
  int c_end = c_begin + cpw;
  if (c_end > p.nchunks) c_end = p.nchunks;

  const int lr = lane >> 3, lq = lane & 7;
  unsigned a_voff[AR];
  unsigned long long a_mask[AR];
  unsigned a_pitch[MULTI ? AR : 1];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lr + 8 * i;
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
        const int t2 = (int)fdiv((unsigned)m, p.div_ow);
        const int ox = m - t2 * p.OW;
        const int img = (int)fdiv((unsigned)t2, p.div_oh);
        const int oy = t2 - img * p.OH;
        iy0 = oy * p.stride - p.pad;
        ix0 = ox * p.stride - p.pad;
        off = img * p.H * p.W;
        hh = p.H;
        ww = p.W;
      }
    }
    a_mask[i] = rowok ? tap_mask(iy0, ix0, hh, ww, p.KH, p.KW) : 0ull;
    a_voff[i] = (unsigned)(((off + iy0 * ww + ix0) * p.Cin + 4 * lq) * 4);
    if (MULTI) a_pitch[i] = (unsigned)(ww * p.Cin * 4);
  }
  unsigned w_voff[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int n = n0 + lr + 8 * j;
    w_voff[j] = n < p.Cout ? (unsigned)((n * p.Kpad + 4 * lq) * 4) : 0xFFFFFFFFu;
  }
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

  // two operand sets in registers: while chunk c is multiplied, chunks c+1 and c+2 are in flight (a wave walks its K share alone:
  // with 1-2 waves per SIMD the global-load latency is hidden by prefetch depth, not by occupancy)
  f32x4 ar0[AR], br0[AR], ar1[AR], br1[AR];
  int nx_tap, nx_c0, nx_ky, nx_kx;
  {
    const int k0 = c_begin * BK;
    nx_tap = k0 / p.Cin;
    nx_c0 = k0 - nx_tap * p.Cin;
    nx_ky = nx_tap / p.KW;
    nx_kx = nx_tap - nx_ky * p.KW;
  }
  auto load_chunk = [&](int chunk, f32x4 (&ar)[AR], f32x4 (&br)[AR]) {
    const int k0 = chunk * BK;
    const int tap = nx_tap, c0 = nx_c0, ky = nx_ky, kx = nx_kx;
    nx_c0 += BK;
    if (nx_c0 >= p.Cin) {
      nx_c0 = 0;
      ++nx_tap;
      if (++nx_kx == p.KW) {
        nx_kx = 0;
        ++nx_ky;
      }
    }
    const unsigned tap_off = MULTI ? (unsigned)((kx * p.Cin + c0) * 4) : (unsigned)(((ky * p.W + kx) * p.Cin + c0) * 4);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = (a_mask[i] >> tap) & 1ull;
      unsigned vo = a_voff[i] + tap_off;
      if (MULTI) vo += (unsigned)ky * a_pitch[i];
      vo = ok ? vo : 0xFFFFFFFFu;
      ar[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, vo, 0, 0));
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
    for (int j = 0; j < AR; ++j)
      br[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[j], k0 * 4, 0));
  };

  f32x16 acc, acc_b;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[r] = 0.f;
    acc_b[r] = 0.f;
  }
  const int frag_row = lane & 31;
  const int frag_k = 4 * (lane >> 5);
  const float* a_base = As + frag_row * LS + frag_k;
  const float* b_base = Bs + frag_row * LS + frag_k;

  // wave-private staging: the wave's own LDS writes are ordered before its reads by the waitcnt the compiler inserts; the fences
  // keep the compiler from moving accesses across the hand-over
  auto stage_and_multiply = [&](f32x4 (&ar)[AR], f32x4 (&br)[AR], int next_chunk) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(As + (lr + 8 * i) * LS + 4 * lq) = ar[i];
#pragma unroll
    for (int j = 0; j < AR; ++j) *reinterpret_cast<f32x4*>(Bs + (lr + 8 * j) * LS + 4 * lq) = br[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (next_chunk < c_end) load_chunk(next_chunk, ar, br);       // this set's registers are free again: two chunks ahead
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(a_base + kk * 8);
      const f32x4 bf = *reinterpret_cast<const f32x4*>(b_base + kk * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
      acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc_b, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
      acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc_b, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  if (c_begin < c_end) load_chunk(c_begin, ar0, br0);
  if (c_begin + 1 < c_end) load_chunk(c_begin + 1, ar1, br1);
  for (int chunk = c_begin; chunk < c_end; chunk += 2) {
    stage_and_multiply(ar0, br0, chunk + 2);
    if (chunk + 1 < c_end) stage_and_multiply(ar1, br1, chunk + 3);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc_b[r];

  // partial sums of waves 1..NW-1 -> LDS (the operand tiles are dead), added by wave 0 in wave order
  __syncthreads();
  float* red = lds;     // [NW-1][16][64]
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < NW; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[((w - 1) * 16 + r) * 64 + lane];
  f32x16 out[1][1];
  out[0][0] = acc;
  store_wave_tiles<1, 1, true>(p, out, m0, n0, M, 0, lane);
}

}  // namespace

template <int BM, int BN>
static void launch_fp32_tile(const ConvArgs& a, bool tap4, int bk, dim3 grid, hipStream_t s, int dyn) {
  if (tap4)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, true, false>), grid, dim3(256), dyn, s, a);
  else if (a.nlv > 0 && bk == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 64, false, true>), grid, dim3(256), dyn, s, a);
  else if (a.nlv > 0)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, false, true>), grid, dim3(256), dyn, s, a);
  else if (bk == 64)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 64, false, false>), grid, dim3(256), dyn, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, 32, false, false>), grid, dim3(256), dyn, s, a);
}

// `lds_reserve`: dynamic LDS the launch allocates and the kernel never touches (EodConvDesc.lds_reserve): caps the workgroups per CU
void launch_conv_fp32(const ConvArgs& a, int tile, int bk, bool tap4, dim3 grid, hipStream_t s, int lds_reserve, int prefetch2) {
  const int dyn = lds_reserve > 0 ? lds_reserve : 0;
  if (prefetch2 == 1 && !tap4 && a.nlv == 0 && bk == 32 && tile == 3) {      // 64x64 only: the 64x256 tail would need 256 VGPRs
    hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 32, false, false, 1>), grid, dim3(256), dyn, s, a);
    return;
  }
  if (prefetch2 == 2 && !tap4 && bk == 32 && tile == 3) {                     // double-buffered LDS, 64x64
    if (a.nlv > 0) hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 32, false, true, 2>), grid, dim3(256), dyn, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<64, 64, 32, false, false, 2>), grid, dim3(256), dyn, s, a);
    return;
  }
  switch (tile) {
    case 5: hipLaunchKernelGGL((conv_igemm_kernel<64, 256, 32, false, false>), grid, dim3(256), dyn, s, a); break;
    case 1: launch_fp32_tile<128, 128>(a, tap4, bk, grid, s, dyn); break;
    case 2: launch_fp32_tile<128, 64>(a, tap4, bk, grid, s, dyn); break;
    default: launch_fp32_tile<64, 64>(a, tap4, bk, grid, s, dyn); break;
  }
}

// NW waves (4 or 8) per 32x32 tile; grid = ceil(M/32) * ceil(Cout/32) workgroups
void launch_conv_wavek(const ConvArgs& a, int nw, dim3 grid, hipStream_t s) {
  if (a.nlv > 0) {
    if (nw == 8) hipLaunchKernelGGL((conv_wavek_kernel<8, true>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_wavek_kernel<4, true>), grid, dim3(256), 0, s, a);
  } else {
    if (nw == 8) hipLaunchKernelGGL((conv_wavek_kernel<8, false>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_wavek_kernel<4, false>), grid, dim3(256), 0, s, a);
  }
}

}  // namespace eodconv
