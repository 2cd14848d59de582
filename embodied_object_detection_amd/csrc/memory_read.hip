// Memory READ path of the spatial feature memory (SURVEY §8 rows a4 + a8), second generation.
//
//   a4  create_implicit_memory + fp16 cast      custom_rcnn.py:762-774,1036   -> normalize_dirty_f16_kernel
//   a8  mem_fp16[proj] -> avg_pool 4,2,2,2      backbone/timm.py:147-168       -> gather_pool_kernel
//   a8  Conv2d(512->256,1x1) x3, x weight, sum  backbone/timm.py:174-189       -> project_fuse_kernel
//
// What changed against the first generation (memory.hip of round 1) and why (profiles/r01_bench_640_kernel_stats.csv):
//  * the fp16 table is kept INCREMENTALLY: the write path marks the cells whose accumulator row or observation count changed
//    (`dirty`), and only those rows are re-normalised.  The reference clones and divides the whole map every frame (O(N):
//    123 MB at 40 000 cells, 805 MB at 262 144) for a few thousand changed rows.
//  * the gather fetches every DISTINCT cell of a 16x16 pixel quadrant once into LDS (wave-level ballot de-duplication, rows
//    brought in by global -> LDS DMA; round 1 issued one 1 KiB row read per pixel: 420 MB of L2 traffic for <= 60 MB
//    compulsory) and pools out of LDS.  Two summation orders (ABI argument `torch_order`): 1 = pixel by pixel in
//    F.avg_pool2d's row-major order, bit-identical to the reference on every input (what the product path uses); 0 = per distinct
//    cell of a 4x4 block, count x row with one v_fma_mix_f32 per element -- the same 16 numbers with fewer roundings, identical
//    to the sequential sum only when that sum is exact (exponents inside the block within 9 bits per channel), ~9 us faster at
//    640x640 (13.6 against 22.6 us on the synthetic scene, profiles/r02_mem_bench_rocprof.txt).
//  * the three 1x1 projections + "x MAP_FEATURE_WEIGHT" + fusion are ONE launch on the f16 matrix cores: the pooled operand is
//    exactly fp16 (timm.py:168 casts it), each fp32 weight is split into TWO f16 pieces of a per-row power-of-two scaling
//    (round-to-nearest residual: h + m carries 11 + 1 + 11 + 1 = 24 significant bits, i.e. |w - (h + m)| <= 2^-24 |w|, the size
//    of one fp32 rounding), every f16 x f16 product is exact in the fp32 accumulator: fp32-class result at 16/2 of the
//    fp32-MFMA rate.  Out-of-range cell indices are clamped and flagged.
#include "eod_common.h"
#include "memory_rows.h"
#include <mutex>
#include "../../include/eod_hip.h"
#include <hip/hip_fp16.h>
#include <type_traits>

// diagnostics hook (tools/experiments/gp_timing.hip defines it to record s_memtime per phase); nothing in the product build
#ifndef GP_STAMP
#define GP_STAMP(k)
#endif

namespace {

typedef unsigned long long u64;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------
// a4, incremental: rows flagged dirty -> out_f16[row] = half(obs > 1 ? mem / obs : mem); flags cleared
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_dirty_f16_kernel(const float* __restrict__ mem, const float* __restrict__ obs,
                                                                   int* __restrict__ dirty, __half* __restrict__ out, int n_cells) {
  EOD_CHAIN_PRIO();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_groups = (n_cells + 63) >> 6;
  // The 4 waves of a workgroup share one 64-cell group and split its dirty rows (bit index mod 4): dirty cells come in runs
  // (neighbouring map cells), so a run is spread over 4 waves x 4 rows in flight instead of serialising in one wave.
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int c = (g << 6) + lane;
    int f = 0;
    if (c < n_cells) f = dirty[c];
    __syncthreads();                       // every wave has read the flags before wave 0 clears them
    if (wave == 0 && f) dirty[c] = 0;
    const u64 bal = __ballot(f != 0) & (0x1111111111111111ull << wave);
    eod_snapshot_rows(mem, out, g, bal, lane, [&](int bit) { return obs[(g << 6) + bit]; });
  }
}

// ------------------------------------------------------------------------------------------------------
// a8: gather + cascaded average pooling, one workgroup per 32x32 pixel tile
// ------------------------------------------------------------------------------------------------------
constexpr int GP_LIST = 8;      // a 4x4 block with up to this many distinct cells is summed as a (row, count) list
constexpr int GP_CAP = 16;      // distinct rows cached per 16x16 quadrant (16 KiB of LDS per wave); pixels beyond it read L2/HBM directly

// acc += float(half): ONE v_fma_mix_f32 (f16 source operand, * 1.0, f32 accumulate; a single rounding, identical to convert + add).
// The compiler's own choice for this pattern is v_cvt_f32_f16 x2 + v_pk_add_f32, 1.5 instructions per element on a VALU-bound loop.
__device__ __forceinline__ void mix_lo(float& acc, unsigned pk) {
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pk));
}
__device__ __forceinline__ void mix_hi(float& acc, unsigned pk) {
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pk));
}
// acc += float(half) * n  (n a small integer count as f32: the product is exact, one rounding in the add)
__device__ __forceinline__ void mixn_lo(float& acc, unsigned pk, float n) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pk), "v"(n));
}
__device__ __forceinline__ void mixn_hi(float& acc, unsigned pk, float n) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(pk), "v"(n));
}
// acc = float(half) * n  (no read of acc: saves the zero fill of a fresh accumulator)
__device__ __forceinline__ void mixn0_lo(float& acc, unsigned pk, float n) {
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(pk), "v"(n));
}
__device__ __forceinline__ void mixn0_hi(float& acc, unsigned pk, float n) {
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(acc) : "v"(pk), "v"(n));
}
__device__ __forceinline__ void setn8(float (&acc)[8], const uint4& raw, float n) {
  mixn0_lo(acc[0], raw.x, n); mixn0_hi(acc[1], raw.x, n);
  mixn0_lo(acc[2], raw.y, n); mixn0_hi(acc[3], raw.y, n);
  mixn0_lo(acc[4], raw.z, n); mixn0_hi(acc[5], raw.z, n);
  mixn0_lo(acc[6], raw.w, n); mixn0_hi(acc[7], raw.w, n);
}
__device__ __forceinline__ void addn8(float (&acc)[8], const uint4& raw, float n) {
  mixn_lo(acc[0], raw.x, n); mixn_hi(acc[1], raw.x, n);
  mixn_lo(acc[2], raw.y, n); mixn_hi(acc[3], raw.y, n);
  mixn_lo(acc[4], raw.z, n); mixn_hi(acc[5], raw.z, n);
  mixn_lo(acc[6], raw.w, n); mixn_hi(acc[7], raw.w, n);
}
__device__ __forceinline__ void add8(float (&acc)[8], const uint4& raw) {
  mix_lo(acc[0], raw.x); mix_hi(acc[1], raw.x);
  mix_lo(acc[2], raw.y); mix_hi(acc[3], raw.y);
  mix_lo(acc[4], raw.z); mix_hi(acc[5], raw.z);
  mix_lo(acc[6], raw.w); mix_hi(acc[7], raw.w);
}

__device__ __forceinline__ float round_f16(float v) { return __half2float(__float2half_rn(v)); }

__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
  __half2 h0 = __floats2half2_rn(v[0], v[1]), h1 = __floats2half2_rn(v[2], v[3]);
  __half2 h2 = __floats2half2_rn(v[4], v[5]), h3 = __floats2half2_rn(v[6], v[7]);
  uint4 pk;
  pk.x = *reinterpret_cast<unsigned*>(&h0);
  pk.y = *reinterpret_cast<unsigned*>(&h1);
  pk.z = *reinterpret_cast<unsigned*>(&h2);
  pk.w = *reinterpret_cast<unsigned*>(&h3);
  return pk;
}

// Pooled rows are stored in the A-fragment order of v_mfma_f32_32x32x16_f16 so that project_fuse reads every operand fragment
// as ONE contiguous 1 KiB wave load: [level][32-row tile][k-step s = 0..31][lane = 32 hi + r][8 halves], element
// (row = 32 tile + r, channel = 16 s + 8 hi + j).  Each level starts on a tile boundary (its last tile may be partly unused).
__device__ __forceinline__ size_t frag_half_offset(int tile32, int r, int c8) {
  // c8 = channel / 8 = 2 s + hi
  return ((((size_t)tile32 * 32 + (c8 >> 1)) * 2 + (c8 & 1)) * 32 + r) * 8;
}

// TORCH_ORDER: every 4x4 block is summed pixel by pixel in torch's row-major order (bit-identical to F.avg_pool2d on every
// input).  Default (false): a block with <= 8 distinct cells is summed as sum count x row in cache-slot order -- the same 16
// numbers, <= 8 roundings instead of 15, identical to the sequential sum whenever that sum is exact (exponents inside the block
// span <= 9 bits per channel: 11-bit values + 4 bits of count in a 24-bit accumulator).  The kernel is instruction-issue bound
// (in-kernel stamps: ~8 cycles per instruction and wave, two waves per SIMD): the per-block (cell, count) lists are built once per
// wave in vector code, 16 blocks in parallel, and a block then costs ~12 instructions per distinct cell instead of ~180.
template <bool TORCH_ORDER>
__global__ __launch_bounds__(256) void gather_pool_kernel(const __half* mem, const int* proj, int H, int W, int n_cells, __half* pooled,
                                                           int* __restrict__ err, size_t pooled_halves) {
  EOD_CHAIN_PRIO();
  static_assert(GP_CAP <= 16, "slot ids must fit a nibble");
  {
    const size_t b = blockIdx.y;        // scene of a batch: its own table, index image and pooled rows
    mem += b * (size_t)n_cells * 512;
    proj += b * (size_t)H * W;
    pooled += b * pooled_halves;
  }
  // LDS: per wave GP_CAP rows of 1 KiB, then the stride-16 exchange buffer
  extern __shared__ __align__(1024) unsigned char smem_raw[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tiles_x = W >> 5;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int qy = wave >> 1, qx = wave & 1;                 // the wave's 16x16 quadrant = one stride-16 cell
  unsigned char* my_rows = smem_raw + (size_t)wave * GP_CAP * 1024;
  const uint4* rows = reinterpret_cast<const uint4*>(my_rows);                          // [GP_CAP][64] x 16 B
  float* s16 = reinterpret_cast<float*>(smem_raw + (size_t)4 * GP_CAP * 1024);          // [4][512]

  GP_STAMP(0);
  // phase 1: lane l holds the 4 pixels (y = l >> 2, x = 4 (l & 3) ..+3) of its quadrant: one 16-byte index load
  int c[4];
  {
    const int y = ty * 32 + qy * 16 + (lane >> 2), x = tx * 32 + qx * 16 + 4 * (lane & 3);
    const int4 v = *reinterpret_cast<const int4*>(proj + (size_t)y * W + x);
    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((unsigned)c[j] >= (unsigned)n_cells) {
        bad = true;
        c[j] = c[j] < 0 ? 0 : n_cells - 1;
      }
    }
    if (bad && err) atomicOr(err, EOD_FLAG_BAD_CELL_INDEX);
  }
  // phase 2: de-duplicate inside the wave with ballots (no LDS atomics: an LDS hash over the tile cost 13 us of CAS contention,
  // the pixels of a tile hit ~15 distinct cells).  Each round takes a cell of the first lane with pixels left as key, starts the key
  // row's global -> LDS DMA (1 KiB, `buffer_load ... lds`: no VGPR staging) and assigns its slot to every pixel with that cell.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__half*>(mem), 0, 0xFFFFFFFFu, 0x00020000);
  GP_STAMP(1);
  // Normally ONE pass over the quadrant.  A quadrant with more distinct cells than the cache holds (1-3 % of them at 640x640 with
  // 0.2 m cells, most of them at 960x960 with 0.08 m cells) is redone as two passes over its upper and lower 16x8 half -- each a
  // de-duplication of its own, each pooling its own two stride-8 cells -- instead of sending the overflowing pixels to the table
  // one 4x4 block at a time; a half that still overflows falls back to that.  (Splitting further, half -> 8x8 cells, was measured
  // and gains nothing: what is left of the tail are tiles whose pixels are nearly all distinct cells -- far background --
  // and those need ~1 MB of rows per tile whichever way they are fetched.)
  const int c_all[4] = {c[0], c[1], c[2], c[3]};
  float acc16[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc16[q] = 0.f;
  const int w8 = W >> 3, w16 = W >> 4, w32 = W >> 5, h8 = H >> 3, h16 = H >> 4;
  const int t16 = (h8 * w8 + 31) >> 5;                 // first 32-row tile of the stride-16 level
  const int t32 = t16 + ((h16 * w16 + 31) >> 5);       // ... of the stride-32 level
  int npass = 1;
#pragma unroll 1
  for (int pass = 0; pass < npass; ++pass) {
  if (npass == 2) {
    const bool mine = (lane >> 5) == pass;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = mine ? c_all[q] : -1;
  }
  int sl[4] = {255, 255, 255, 255};      // cache slot of each pixel; 255 = not cached (read the table directly)
  int n_rows = 0;
#pragma unroll 1
  while (n_rows < GP_CAP) {
    // a pixel that has its slot holds -1: the largest remaining cell of the first lane with any left is the round's key
    const int cand = max(max(c[0], c[1]), max(c[2], c[3]));
    const u64 bal = __ballot(cand >= 0);
    if (bal == 0) break;
    const int key = __builtin_amdgcn_readlane(cand, __ffsll((long long)bal) - 1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(my_rows + n_rows * 1024), 16, (unsigned)key * 1024u + lane * 16u, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool hit = c[q] == key;
      sl[q] = hit ? n_rows : sl[q];
      c[q] = hit ? -1 : c[q];
    }
    ++n_rows;
  }
  if (npass == 1) {
    const u64 left = __ballot(max(max(c[0], c[1]), max(c[2], c[3])) >= 0);
    if (left != 0) {
      npass = 2;
      // The rounds take their keys from the lowest lane with pixels left, i.e. from the upper rows first: when every pixel of the
      // upper half has its slot, this attempt IS pass 0 (the lower half's lanes are simply not pooled now) and only the lower
      // half is de-duplicated again.  Otherwise both halves start over.
      if ((left & 0xFFFFFFFFull) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows in flight target the slots that are about to be reused
        pass = -1;
        continue;
      }
    }
  }
  const unsigned slots = (unsigned)sl[0] | (unsigned)sl[1] << 8 | (unsigned)sl[2] << 16 | (unsigned)sl[3] << 24;
  GP_STAMP(2);

  // phase 3: pooling.  Each lane owns 8 consecutive channels.  Order mirrors torch: avg_pool2d(4) sums 16 pixels row-major in
  // f32, /16; each avg_pool2d(2) sums 4 values row-major, /4, rounds to fp16 (timm.py:152,168).
  // Per 4x4 block (16 per quadrant) a descriptor, computed ONCE in vector code, all 64 lanes at work (in-kernel stamps: unpacking
  // 16 slots per block with ~100 scalar instructions, not the adds, was 800 cycles per block; a first vector version that walked
  // the 16 pixels of a block in one lane was 1 200 instructions, this one is ~70):
  //   binfo  = counts of list entries 0-3 (5 bits each, ascending slot order, 0 for unused entries) | kind << 20 | n << 22
  //   binfo2 = counts of list entries 4-7
  //           kind 0: the block is that list (a block of one row is {(row, 16)}: 16 v is exact and (16 v) / 16 = v)
  //           kind 1: pixel by pixel (more than 8 distinct rows, or TORCH_ORDER)   kind 2: some pixel reads the table directly
  //   blist  = the rows' slots (8 x 4 bits; unused entries repeat entry 0, their count 0 adds an exact zero)
  int binfo, binfo2;
  unsigned blist;
  {
    // lane l holds pixel row (l >> 2) & 3 of block (row l >> 4, column l & 3): the four lanes of a block are l, l + 4, l + 8, l + 12
    // inside one 16-lane row, so two row rotations combine them (all four lanes end up with the block's descriptor).
    auto ror = [](unsigned v, auto ctrl) {
      return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, false);
    };
    using ror4 = std::integral_constant<int, 0x124>;      // row_ror:4
    using ror8 = std::integral_constant<int, 0x128>;      // row_ror:8
    unsigned m = 0;                                        // bit s: slot s occurs in the block; bit 31: a pixel without a slot
#pragma unroll
    for (int q = 0; q < 4; ++q) m |= 1u << (sl[q] & 31);
    m |= ror(m, ror4{});
    m |= ror(m, ror8{});
    const bool direct = (m >> 31) != 0;
    const unsigned mm = m & 0xFFFFu;
    const int nu = __popc(mm);
    // the list is the block's slots in ascending order; a pixel's list position is the number of smaller slots present
    unsigned cnt = 0, cnt2 = 0;                            // positions 0-3 / 4-7, 5 bits each
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned below = mm & ((1u << (sl[q] & 31)) - 1u);
      const int pos = __popc(below);
      const unsigned one = 1u << (5 * (pos & 3));         // (positions >= 8 only occur in blocks that are not summed as a list)
      cnt += (pos & 4) ? 0u : one;
      cnt2 += (pos & 4) ? one : 0u;
    }
    cnt += ror(cnt, ror4{});
    cnt += ror(cnt, ror8{});
    cnt2 += ror(cnt2, ror4{});
    cnt2 += ror(cnt2, ror8{});
    const unsigned us0 = (unsigned)(__ffs((int)mm) - 1) & 15u;
    blist = us0;
    unsigned rest = mm;
#pragma unroll
    for (int k = 1; k < GP_LIST; ++k) {
      rest &= rest - 1u;
      blist |= (rest ? (unsigned)(__ffs((int)rest) - 1) : us0) << (4 * k);
    }
    const bool overflow = nu > GP_LIST, uniform = nu == 1;
    const unsigned kind = direct ? 2u : ((overflow || (TORCH_ORDER && !uniform)) ? 1u : 0u);
    binfo = (int)((cnt & 0xFFFFFu) | kind << 20 | (unsigned)min(nu, 15) << 22);
    binfo2 = (int)(cnt2 & 0xFFFFFu);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMA'd rows are in LDS (only this wave reads them)
  GP_STAMP(3);
#pragma unroll 1
  for (int cy8 = (npass == 2 ? pass : 0); cy8 < (npass == 2 ? pass + 1 : 2); ++cy8) {
#pragma unroll 1
    for (int cx8 = 0; cx8 < 2; ++cx8) {
      float acc8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc8[q] = 0.f;
      // descriptors of the cell's four 4x4 blocks
      int info[4], info2[4];
      unsigned bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = (cy8 * 2 + (j >> 1)) * 16 + cx8 * 2 + (j & 1);          // a lane of block (row, column)
        info[j] = __builtin_amdgcn_readlane(binfo, b);
        info2[j] = __builtin_amdgcn_readlane(binfo2, b);
        bl[j] = (unsigned)__builtin_amdgcn_readlane((int)blist, b);
      }
      const unsigned kinds = ((unsigned)(info[0] | info[1] | info[2] | info[3]) >> 20) & 3u;
      int nmax = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) nmax = max(nmax, (info[j] >> 22) & 15);
      auto count_of = [&](int j, int k) {
        return (float)(int)((k < 4 ? ((unsigned)info[j] >> (5 * k)) : ((unsigned)info2[j] >> (5 * (k - 4)))) & 31u);
      };
      // common case: every block is a short list.  Branch-free inside: the (8 or 16) row reads are issued first, then the
      // multiply-adds block by block; unused list entries re-read entry 0 with a count of 0 (an exact + 0).
      auto listed = [&](auto vtag) {
        constexpr int V = decltype(vtag)::value;
        constexpr int G = V > 4 ? 2 : 4;             // blocks whose reads are in flight together (64 VGPRs of row data at most)
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += G) {
          uint4 raw[G][V];
#pragma unroll
          for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < V; ++k) raw[j][k] = rows[((bl[j0 + j] >> (4 * k)) & 15u) * 64 + lane];
#pragma unroll
          for (int j = 0; j < G; ++j) {
            float acc4[8];
            setn8(acc4, raw[j][0], count_of(j0 + j, 0));
#pragma unroll
            for (int k = 1; k < V; ++k) addn8(acc4, raw[j][k], count_of(j0 + j, k));
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] += acc4[q] * 0.0625f;
          }
        }
      };
      if (kinds == 0 && nmax <= 2) {
        listed(std::integral_constant<int, 2>{});
      } else if (kinds == 0 && nmax == 3) {
        listed(std::integral_constant<int, 3>{});
      } else if (kinds == 0 && nmax == 4) {
        listed(std::integral_constant<int, 4>{});
      } else if (kinds == 0 && nmax <= 6) {
        listed(std::integral_constant<int, 6>{});
      } else if (kinds == 0) {
        listed(std::integral_constant<int, 8>{});
      } else {
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
          const int by = j >> 1, bx = j & 1;
          const int yq = cy8 * 8 + by * 4, xq4 = cx8 * 2 + bx;          // first row / 4-pixel column group inside the quadrant
          float acc4[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) acc4[q] = 0.f;
          const int inf = __builtin_amdgcn_readlane(binfo, (cy8 * 2 + by) * 16 + xq4);      // (j is a run-time index here)
          const int kind = (inf >> 20) & 3;
          if (kind == 0) {
            const unsigned bw = (unsigned)__builtin_amdgcn_readlane((int)blist, (cy8 * 2 + by) * 16 + xq4);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              addn8(acc4, rows[((bw >> (4 * k)) & 15u) * 64 + lane], (float)(int)(((unsigned)inf >> (5 * k)) & 31u));
            if (((inf >> 22) & 15) > 4) {
              const unsigned inf2 = (unsigned)__builtin_amdgcn_readlane(binfo2, (cy8 * 2 + by) * 16 + xq4);
#pragma unroll
              for (int k = 4; k < 8; ++k)
                addn8(acc4, rows[((bw >> (4 * k)) & 15u) * 64 + lane], (float)(int)((inf2 >> (5 * (k - 4))) & 31u));
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] += acc4[q] * 0.0625f;
            continue;
          }
          // the block's 16 slots sit in 4 lanes (one per pixel row): wave-uniform after readlane
          int sl[16];
#pragma unroll
          for (int dy = 0; dy < 4; ++dy) {
            const unsigned pk = (unsigned)__builtin_amdgcn_readlane((int)slots, (yq + dy) * 4 + xq4);
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) sl[4 * dy + dx] = (int)((pk >> (8 * dx)) & 0xFFu);
          }
          if (kind == 1) {
            // all 16 rows are in LDS: 16 independent reads, then the adds in row-major order
            uint4 raw[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) raw[i] = rows[sl[i] * 64 + lane];
#pragma unroll
            for (int i = 0; i < 16; ++i) add8(acc4, raw[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] += acc4[q] * 0.0625f;
          } else {
            // rare (1-3 % of the quadrants of a frame, but a workgroup that hits it sets the kernel's tail): more distinct cells in
            // the quadrant than the LDS cache holds.  All 16 reads of the block -- LDS or table -- are issued before the first add:
            // one memory round trip per block, not one per pixel row.
            uint4 raw[16];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
              const int src_lane = (yq + dy) * 4 + xq4;
#pragma unroll
              for (int dx = 0; dx < 4; ++dx) {
                if (sl[4 * dy + dx] != 255) {
                  raw[4 * dy + dx] = rows[sl[4 * dy + dx] * 64 + lane];
                } else {
                  const int cell = __builtin_amdgcn_readlane(c[dx], src_lane);
                  raw[4 * dy + dx] = *reinterpret_cast<const uint4*>(mem + (size_t)cell * 512 + lane * 8);
                }
              }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) add8(acc4, raw[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc8[q] += acc4[q] * 0.0625f;
          }
        }
      }
      float v8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v8[q] = round_f16(acc8[q] * 0.25f);
        acc16[q] += v8[q];
      }
      const int oy = ty * 4 + qy * 2 + cy8, ox = tx * 4 + qx * 2 + cx8;
      {
        const int row = oy * w8 + ox;
        *reinterpret_cast<uint4*>(pooled + frag_half_offset(row >> 5, row & 31, lane)) = pack8(v8);
      }
    }
  }
  }      // pass
  GP_STAMP(4);
  float v16[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v16[q] = round_f16(acc16[q] * 0.25f);
    s16[wave * 512 + lane * 8 + q] = v16[q];
  }
  {
    const int oy = ty * 2 + qy, ox = tx * 2 + qx;
    const int row = oy * w16 + ox;
    *reinterpret_cast<uint4*>(pooled + frag_half_offset(t16 + (row >> 5), row & 31, lane)) = pack8(v16);
  }
  __syncthreads();
  if (tid < 64) {
    float v32[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = tid * 8 + q;
      v32[q] = (((s16[c] + s16[512 + c]) + s16[1024 + c]) + s16[1536 + c]) * 0.25f;
    }
    const int row = ty * w32 + tx;
    *reinterpret_cast<uint4*>(pooled + frag_half_offset(t32 + (row >> 5), row & 31, tid)) = pack8(v32);
  }
  GP_STAMP(5);
}

// ------------------------------------------------------------------------------------------------------
// a8: three 1x1 projections + "x MAP_FEATURE_WEIGHT" + fusion into P3..P5, one launch, f16 matrix cores
// ------------------------------------------------------------------------------------------------------
// weights: per level and output channel n, scaled by 2^S_n so that max_k |w| lands in [2^13, 2^14), then split
//   w * 2^S ~= h + m  (h = f16(w'), m = f16(w' - h); the residual w' - h is exact in f32, m rounds it to 11 more bits)
// layout [level][piece][32-column tile][k-step s][lane = 32 hi + r][8 halves]: the B fragment B[k = 16 s + 8 hi + j][col 32 tile + r]
// of a wave is ONE contiguous 1 KiB load, like the A fragments gather_pool writes.
__global__ __launch_bounds__(64) void project_prepare_kernel(const float* __restrict__ w /*[256][512]*/, _Float16* __restrict__ out,
                                                               float* __restrict__ sinv, int piece_stride) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float mx = 0.f;
  for (int k = lane; k < 512; k += 64) mx = fmaxf(mx, fabsf(w[n * 512 + k]));
  mx = wave_reduce_max(mx);
  int S = 0;
  if (mx > 0.f && mx < INFINITY) {
    int e;
    frexpf(mx, &e);      // mx = f * 2^e, f in [0.5, 1)
    S = 14 - e;          // mx * 2^S in [2^13, 2^14)
  }
  const float sc = ldexpf(1.0f, S);
  for (int k = lane; k < 512; k += 64) {
    const float v = w[n * 512 + k] * sc;          // exact (power of two)
    const _Float16 h = (_Float16)v;
    const float r1 = v - (float)h;
    const _Float16 m = (_Float16)r1;
    const size_t o = ((((size_t)(n >> 5) * 32 + (k >> 4)) * 2 + ((k >> 3) & 1)) * 32 + (n & 31)) * 8 + (k & 7);
    out[(size_t)0 * piece_stride + o] = h;
    out[(size_t)1 * piece_stride + o] = m;
  }
  if (lane == 0) sinv[n] = ldexpf(1.0f, -S);
}

struct ProjArgs {
  int level_off[4];     // row offsets of P3, P4, P5 in the fp32 row list `feats` (+ end)
  int tile_off[4];      // first 64-row workgroup tile of each level (+ end)
  int frag_tile_off[3]; // first 32-row fragment tile of each level in the pooled buffer
  float weight;         // MODEL.MAP_FEATURE_WEIGHT
  int mode;             // 0: P = (x.W + b) * weight + P   (sum)     1: P = (x.W + b) * weight   (mem_only)
  int batch;            // scenes in lock-step (grid.y)
  size_t pooled_halves; // one scene's pooled buffer
};

// Workgroup = 64 rows x 128 columns, 4 waves; wave w owns columns [128 half + 32 w, +32) for both 32-row tiles: per k-step
// 2 A + 2 B fragment loads (1 KiB each, contiguous) feed 4 MFMAs.  Small tiles on purpose: the op is 2.2 GFLOP, what matters
// is how many bytes are in flight per CU (264 workgroups at 640x640, 2 per CU fit).
__global__ __launch_bounds__(256) void project_fuse_kernel(const _Float16* X, const _Float16* __restrict__ Wsplit,
                                                            const float* __restrict__ sinv, const float* __restrict__ bias,
                                                            float* P, ProjArgs a) {
  EOD_CHAIN_PRIO();
  // Workgroups that share an XCD (id % 8 under the round-robin dispatch) take CONSECUTIVE (tile, half) pairs: both halves of a row
  // tile read the same pooled fragments through one L2, and a level's 0.5 MB weight panel is fetched by the XCDs that hold its
  // tiles only (P4 by two, P5 by one) instead of by all eight -- the 8 L2s are not coherent with each other, every one that sees a
  // panel fetches its own copy (PMC FETCH_SIZE: 32.3 MB per launch against 18.8 MB of operands before this mapping).
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int t = vb >> 1, half = vb & 1;
  const int lvl = t >= a.tile_off[2] ? 2 : (t >= a.tile_off[1] ? 1 : 0);
  if (a.batch > 1) {
    // scene b of a batch: its own pooled rows; `feats` is level major over the scenes, level l of scene b starts at row
    // batch * level_off[l] + b * rows_l
    const size_t b = blockIdx.y;
    X += b * a.pooled_halves;
    P += ((size_t)(a.batch - 1) * a.level_off[lvl] + b * (size_t)(a.level_off[lvl + 1] - a.level_off[lvl])) * 256;
  }
  const int row0 = a.level_off[lvl] + (t - a.tile_off[lvl]) * 64;
  const int row_end = a.level_off[lvl + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 31, hi = lane >> 5;
  const _Float16* Wl = Wsplit + (size_t)lvl * 2 * 256 * 512;

  // fragment blocks: 1 KiB per (32-row or 32-column tile, k-step); lane l reads bytes [16 l, 16 l + 16)
  const int local_tile = (t - a.tile_off[lvl]) * 2;                 // first 32-row tile of this workgroup inside its level
  const int level_tiles = (row_end - a.level_off[lvl] + 31) >> 5;
  const _Float16* ap[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    int lt = local_tile + m;
    lt = lt < level_tiles ? lt : level_tiles - 1;                   // a level with an odd tile count: re-read the last one, masked store
    ap[m] = X + ((size_t)(a.frag_tile_off[lvl] + lt) * 32 * 64 + lane) * 8;
  }
  const int col_tile = half * 4 + wave;
  const _Float16* bp = Wl + ((size_t)col_tile * 32 * 64 + lane) * 8;

  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

  // The residual P (mode "sum") and the per-column scale / bias are requested FIRST: they arrive while the k-loop runs, instead of
  // costing a load round trip after it (written as load-modify-store per element the compiler must also keep every load behind
  // the previous store -- the addresses may alias -- and the epilogue becomes serial round trips).
  const int col = col_tile * 32 + r;
  const float si = sinv[lvl * 256 + col], b = bias[lvl * 256 + col];
  float old[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int rbase = row0 + 32 * m + 4 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int row = rbase + (i & 3) + 8 * (i >> 2);
      row = row < row_end ? row : row_end - 1;            // clamped address instead of a divergent branch per element
      old[m][i] = a.mode == 0 ? P[(size_t)row * 256 + col] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // Software pipeline, written out: the 4 fragment loads of k-step s + 3 are issued before the 4 MFMAs of k-step s (four
  // register sets).  Left to the compiler the loop became load -> s_waitcnt vmcnt(0) -> MFMA with ONE load in flight (50 us).
  constexpr int DEPTH = 4;
  f16x8 af[DEPTH][2], bf[DEPTH][2];
  auto load_step = [&](int buf, int s) {
#pragma unroll
    for (int m = 0; m < 2; ++m) af[buf][m] = *reinterpret_cast<const f16x8*>(ap[m] + (size_t)s * 64 * 8);
#pragma unroll
    for (int p = 0; p < 2; ++p) bf[buf][p] = *reinterpret_cast<const f16x8*>(bp + (size_t)p * 256 * 512 + (size_t)s * 64 * 8);
  };
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) load_step(s, s);
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    if (s + DEPTH - 1 < 32) load_step((s + DEPTH - 1) % DEPTH, s + DEPTH - 1);
    __builtin_amdgcn_sched_barrier(0);
    // the small piece first: the low-order products enter the fp32 accumulator before the large ones
#pragma unroll
    for (int p = 1; p >= 0; --p)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s % DEPTH][m], bf[s % DEPTH][p], acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // epilogue: C/D layout col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 hi; the residuals were fetched before the loop.
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int rbase = row0 + 32 * m + 4 * hi;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = rbase + (i & 3) + 8 * (i >> 2);
      // same rounding steps as conv -> "* weight" -> "+ P_l" in the reference (timm.py:174,177,182): no contraction
      const float v = __fmul_rn(__fadd_rn(__fmul_rn(acc[m][i], si), b), a.weight);
      if (row < row_end) P[(size_t)row * 256 + col] = a.mode == 0 ? __fadd_rn(v, old[m][i]) : v;
    }
  }
}

}  // namespace

extern "C" int eod_memory_normalize_dirty_f16(const float* mem, const float* obs, int32_t* dirty, uint16_t* out_f16, int n_cells, int D,
                                              eod_stream_t stream) {
  if (!mem || !obs || !dirty || !out_f16) return EOD_ERR_NULL;
  if (n_cells <= 0 || D != 512) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(mem) || !eod_aligned16(out_f16)) return EOD_ERR_ALIGN;
  int blocks = (n_cells + 63) / 64;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(normalize_dirty_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mem, obs, dirty,
                     reinterpret_cast<__half*>(out_f16), n_cells);
  return eod_launch_status();
}

extern "C" int eod_memory_gather_pool(const uint16_t* mem_f16, const int32_t* proj, int H, int W, int D, int n_cells, uint16_t* pooled_f16,
                                      int32_t* err_flags, int torch_order, int batch, eod_stream_t stream) {
  if (!mem_f16 || !proj || !pooled_f16) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 31) || (W & 31) || D != 512 || n_cells <= 0 || n_cells > (1 << 22) || batch > EOD_MAX_BATCH)
    return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(mem_f16) || !eod_aligned16(pooled_f16) || !eod_aligned16(proj)) return EOD_ERR_ALIGN;
  const size_t lds = (size_t)4 * GP_CAP * 1024 + 4 * 512 * sizeof(float);
  // the attribute belongs to the (device, function) pair: one flag per device ordinal, set under a mutex (a second device in the
  // process, or a concurrent first call, must not launch a 72 KB dynamic-LDS kernel without it)
  static std::mutex attr_mutex;
  static bool attr_set_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return EOD_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(attr_mutex);
  bool& attr_set = attr_set_dev[dev];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gather_pool_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gather_pool_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
      return EOD_ERR_LAUNCH;
    attr_set = true;
  }
  const dim3 grid((H >> 5) * (W >> 5), batch > 1 ? batch : 1);
  const size_t ph = eod_memory_pooled_halves(H, W);
  if (torch_order)
    hipLaunchKernelGGL(gather_pool_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, reinterpret_cast<const __half*>(mem_f16), proj, H,
                       W, n_cells, reinterpret_cast<__half*>(pooled_f16), err_flags, ph);
  else
    hipLaunchKernelGGL(gather_pool_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, reinterpret_cast<const __half*>(mem_f16), proj, H,
                       W, n_cells, reinterpret_cast<__half*>(pooled_f16), err_flags, ph);
  return eod_launch_status();
}

extern "C" size_t eod_memory_project_weights_bytes(void) { return (size_t)3 * 2 * 256 * 512 * 2 + (size_t)2 * 3 * 256 * 4; }

extern "C" int eod_memory_project_prepare(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                          const float* b3, void* prepared, eod_stream_t stream) {
  if (!w1 || !w2 || !w3 || !b1 || !b2 || !b3 || !prepared) return EOD_ERR_NULL;
  if (!eod_aligned16(prepared)) return EOD_ERR_ALIGN;
  _Float16* ws = static_cast<_Float16*>(prepared);
  float* sinv = reinterpret_cast<float*>(ws + (size_t)3 * 2 * 256 * 512);
  float* bias = sinv + 3 * 256;
  const float* w[3] = {w1, w2, w3};
  const float* b[3] = {b1, b2, b3};
  for (int l = 0; l < 3; ++l) {
    hipLaunchKernelGGL(project_prepare_kernel, dim3(256), dim3(64), 0, (hipStream_t)stream, w[l], ws + (size_t)l * 2 * 256 * 512,
                       sinv + l * 256, 256 * 512);
    if (hipMemcpyAsync(bias + l * 256, b[l], 256 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return EOD_ERR_LAUNCH;
  }
  return eod_launch_status();
}

extern "C" int eod_memory_project_fuse(const uint16_t* pooled_f16, const void* prepared, float* feats, int H, int W, float weight, int mode,
                                       int batch, eod_stream_t stream) {
  if (!pooled_f16 || !prepared || !feats) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 31) || (W & 31) || (mode != 0 && mode != 1) || batch > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(pooled_f16) || !eod_aligned16(prepared)) return EOD_ERR_ALIGN;
  ProjArgs a{};
  const int rows[3] = {(H >> 3) * (W >> 3), (H >> 4) * (W >> 4), (H >> 5) * (W >> 5)};
  a.level_off[0] = 0;
  a.tile_off[0] = 0;
  int ft = 0;
  for (int l = 0; l < 3; ++l) {
    a.level_off[l + 1] = a.level_off[l] + rows[l];
    a.tile_off[l + 1] = a.tile_off[l] + (rows[l] + 63) / 64;
    a.frag_tile_off[l] = ft;
    ft += (rows[l] + 31) / 32;
  }
  a.weight = weight;
  a.mode = mode;
  a.batch = batch > 1 ? batch : 1;
  a.pooled_halves = eod_memory_pooled_halves(H, W);
  const _Float16* ws = static_cast<const _Float16*>(prepared);
  const float* sinv = reinterpret_cast<const float*>(ws + (size_t)3 * 2 * 256 * 512);
  const float* bias = sinv + 3 * 256;
  hipLaunchKernelGGL(project_fuse_kernel, dim3(a.tile_off[3] * 2, a.batch), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(pooled_f16), ws, sinv, bias, feats, a);
  return eod_launch_status();
}

extern "C" size_t eod_memory_pooled_halves(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const int rows[3] = {(H >> 3) * (W >> 3), (H >> 4) * (W >> 4), (H >> 5) * (W >> 5)};
  size_t tiles = 0;
  for (int l = 0; l < 3; ++l) tiles += (size_t)((rows[l] + 31) / 32);
  return tiles * 32 * 512;
}
