// SMNet-style spatial feature memory on device: depth un-projection + integer grid-cell indexing, the memory
// READ (full observation normalise -> fp16; the incremental normalise, the gather + pooling and the projection live in
// memory_read.hip) and the memory
// WRITE (instance CLIP features -> per-pixel mean over covering instances -> every 8th observed pixel ->
// per-cell mean -> accumulate, observation counters).
//
// All of it is HBM / cache-bandwidth work on bytes and indices; nothing here is shaped into a GEMM.  The
// reference materialises [H,W,512] fp16 + 2x f32 copies on the read side and a [1,512,H,W] f32 image plus a
// dense [Npix/8, N] one-hot on the write side (Detic/detic/modeling/backbone/timm.py:147-152,
// Detic/detic/modeling/meta_arch/custom_rcnn.py:884-936); here the gather is fused with the pooling and the
// write works on the sparse set of selected pixels only.
#include "eod_common.h"
#include "memory_rows.h"
#include "../../include/eod_hip.h"
#include <hip/hip_fp16.h>

namespace {

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------------
// a1 + a2
// ------------------------------------------------------------------------------------------------------
struct UnprojArgs {
  float T[16];
  float fx, fy, cx, cy;
  float ps[3], ms[3];
  float cell;
  int map_w, map_h, order;
};

__global__ __launch_bounds__(256) void unproject_kernel(const float* __restrict__ depth, int H, int W, UnprojArgs a,
                                                         float* __restrict__ xyz, int* __restrict__ idx) {
  const int total = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    const int v = p / W, u = p - v * W;
    // explicit _rn intrinsics: no FMA contraction, IEEE divide -> bit-identical to oracle/projector.c
    const float xs = __fdiv_rn(__fsub_rn(__fadd_rn((float)u, 0.5f), a.cx), a.fx);
    const float ys = __fdiv_rn(__fsub_rn(__fadd_rn((float)v, 0.5f), a.cy), a.fy);
    const float z = depth[p];
    const float x = __fmul_rn(z, xs);
    const float y = __fmul_rn(z, ys);
    float w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float t0 = __fmul_rn(a.T[i * 4 + 0], x);
      const float t1 = __fmul_rn(a.T[i * 4 + 1], y);
      const float t2 = __fmul_rn(a.T[i * 4 + 2], z);
      float s = __fadd_rn(__fadd_rn(__fadd_rn(t0, t1), t2), a.T[i * 4 + 3]);
      s = __fsub_rn(s, a.ps[i]);
      w[i] = s;
    }
    if (xyz) {
      xyz[(size_t)p * 3 + 0] = w[0];
      xyz[(size_t)p * 3 + 1] = w[1];
      xyz[(size_t)p * 3 + 2] = w[2];
    }
    const float qx = rintf(__fdiv_rn(__fsub_rn(w[0], a.ms[0]), a.cell));
    const float qz = rintf(__fdiv_rn(__fsub_rn(w[2], a.ms[2]), a.cell));
    long ix = (qx != qx) ? 0 : (qx < -1e9f ? -1000000000L : (qx > 1e9f ? 1000000000L : (long)qx));
    long iz = (qz != qz) ? 0 : (qz < -1e9f ? -1000000000L : (qz > 1e9f ? 1000000000L : (long)qz));
    ix = ix < 0 ? 0 : (ix > a.map_w - 1 ? a.map_w - 1 : ix);
    iz = iz < 0 ? 0 : (iz > a.map_h - 1 ? a.map_h - 1 : iz);
    idx[p] = (int)(a.order == 0 ? iz * a.map_w + ix : ix * a.map_h + iz);
  }
}

// ------------------------------------------------------------------------------------------------------
// a4 + fp16 cast
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_f16_kernel(const float* __restrict__ mem, const float* __restrict__ obs,
                                                             __half* __restrict__ out, int n_cells, int D) {
  const int d4 = D >> 2;
  const size_t total = (size_t)n_cells * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cell = (int)(i / d4);
    const float o = obs[cell];
    f32x4 v = *reinterpret_cast<const f32x4*>(mem + i * 4);
    if (o > 1.0f) {
      v.x = __fdiv_rn(v.x, o);
      v.y = __fdiv_rn(v.y, o);
      v.z = __fdiv_rn(v.z, o);
      v.w = __fdiv_rn(v.w, o);
    }
    __half2 h0 = __floats2half2_rn(v.x, v.y);
    __half2 h1 = __floats2half2_rn(v.z, v.w);
    uint2 pk;
    pk.x = *reinterpret_cast<unsigned*>(&h0);
    pk.y = *reinterpret_cast<unsigned*>(&h1);
    *reinterpret_cast<uint2*>(out + i * 4) = pk;
  }
}

// ------------------------------------------------------------------------------------------------------
// a16-a19 write path
// ------------------------------------------------------------------------------------------------------
struct MwWs {
  int* inst_rows;   // [R_cap] unique proposal rows, ascending
  int* k_u;         // [1]
  unsigned char* cover;  // [P]
  int* sel_pix;     // [P/8+1]
  int* n_sel;       // [1]
  int* cell_flag;   // [N] any pixel of the frame hit the cell
  int* cell_mark;   // [N] a selected pixel hit the cell
  int* cell_slot;   // [N]
  int* slot_cell;   // [U_max]
  int* n_slots;     // [1]
  long long* wtab;  // [U_max, K_cap] fixed point 2^-32: sum over the slot's sampled pixels of 1/cover for every instance
  int* slot_cnt;    // [U_max]
  int* blk_pix;     // [ceil(P/4096)]
  int* blk_cell;    // [ceil(N/4096)]
  size_t bytes;
};

inline size_t up(size_t v) { return (v + 255) / 256 * 256; }

MwWs mw_carve(void* base, int H, int W, int D, int n_cells, int R_cap, int K_cap) {
  (void)D;
  MwWs w{};
  char* b = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = b ? b + off : nullptr;
    off += up(bytes);
    return p;
  };
  const size_t P = (size_t)H * W;
  const size_t smax = P / 8 + 1;
  const size_t umax = smax < (size_t)n_cells ? smax : (size_t)n_cells;
  w.inst_rows = (int*)take((size_t)R_cap * 4);
  w.k_u = (int*)take(4);
  w.cover = (unsigned char*)take(P);
  w.sel_pix = (int*)take(smax * 4);
  w.n_sel = (int*)take(4);
  w.cell_flag = (int*)take((size_t)n_cells * 4);
  w.cell_mark = (int*)take((size_t)n_cells * 4);
  w.cell_slot = (int*)take((size_t)n_cells * 4);
  w.slot_cell = (int*)take(umax * 4);
  w.n_slots = (int*)take(4);
  w.wtab = (long long*)take(umax * (size_t)K_cap * 8);
  w.slot_cnt = (int*)take(umax * 4);
  w.blk_pix = (int*)take(((P + 4095) / 4096 + 1) * 4);
  w.blk_cell = (int*)take((((size_t)n_cells + 4095) / 4096 + 1) * 4);
  w.bytes = off;
  return w;
}

// unique(det_rows) ascending (custom_rcnn.py:875); single block
__global__ __launch_bounds__(512) void mw_unique_rows_kernel(const int* __restrict__ det_rows, const int* __restrict__ det_count, int K_cap,
                                                              int R_cap, int* __restrict__ inst_rows, int* __restrict__ k_u,
                                                              int* __restrict__ k_out) {
  EOD_CHAIN_PRIO();
  __shared__ int flag[512];
  const int t = threadIdx.x;
  flag[t] = 0;
  __syncthreads();
  int K = *det_count;
  K = K < K_cap ? K : K_cap;
  for (int i = t; i < K; i += blockDim.x) {
    const int r = det_rows[i];
    if (r >= 0 && r < R_cap) flag[r] = 1;
  }
  __syncthreads();
  if (t == 0) {
    int n = 0;
    for (int r = 0; r < R_cap; ++r)
      if (flag[r]) inst_rows[n++] = r;
    *k_u = n;
    if (k_out) *k_out = n;
  }
}

// mask test of one instance at pixel centre (x+0.5, y+0.5): same arithmetic as paste_masks_kernel
__device__ __forceinline__ bool mask_hit(const float* __restrict__ m, float x0, float y0, float x1, float y1, int x, int y, float thr) {
  const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * 28.0f - 1.0f) / 2.0f;
  if (!(ix > -1.0f && ix < 28.0f && iy > -1.0f && iy < 28.0f)) return false;
  const float fx = floorf(ix), fy = floorf(iy);
  const int xw = (int)fx, yn = (int)fy;
  const int xe = xw + 1, ys = yn + 1;
  const float nw = ((float)xe - ix) * ((float)ys - iy);
  const float ne = (ix - (float)xw) * ((float)ys - iy);
  const float sw = ((float)xe - ix) * (iy - (float)yn);
  const float se = (ix - (float)xw) * (iy - (float)yn);
  const bool xwv = (unsigned)xw < 28u, xev = (unsigned)xe < 28u, ynv = (unsigned)yn < 28u, ysv = (unsigned)ys < 28u;
  float v = 0.f;
  if (xwv && ynv) v += m[yn * 28 + xw] * nw;
  if (xev && ynv) v += m[yn * 28 + xe] * ne;
  if (xwv && ysv) v += m[ys * 28 + xw] * sw;
  if (xev && ysv) v += m[ys * 28 + xe] * se;
  return v >= thr;
}

// per pixel: number of covering instances; marks every cell the frame hits (for the observation counters)
__global__ __launch_bounds__(256) void mw_coverage_kernel(const float* __restrict__ boxes, const float* __restrict__ masks,
                                                           const int* __restrict__ inst_rows, const int* __restrict__ k_u,
                                                           const int* __restrict__ proj, int H, int W, int n_cells, float thr,
                                                           unsigned char* __restrict__ cover, int* __restrict__ cell_flag,
                                                           int* __restrict__ err) {
  EOD_CHAIN_PRIO();
  const int K = *k_u;
  if (K == 0) return;
  bool bad = false;
  const int total = H * W;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    int cnt = 0;
    for (int k = 0; k < K; ++k) {
      const int r = inst_rows[k];
      const float x0 = boxes[r * 4 + 0], y0 = boxes[r * 4 + 1], x1 = boxes[r * 4 + 2], y1 = boxes[r * 4 + 3];
      // quick reject: a sample more than one mask pixel outside the box is zero
      const float mx = (x1 - x0) * (1.0f / 14.0f), my = (y1 - y0) * (1.0f / 14.0f);
      const float fxp = (float)x + 0.5f, fyp = (float)y + 0.5f;
      if (fxp < x0 - mx || fxp > x1 + mx || fyp < y0 - my || fyp > y1 + my) continue;
      if (mask_hit(masks + (size_t)r * 784, x0, y0, x1, y1, x, y, thr)) ++cnt;
    }
    cover[p] = (unsigned char)cnt;
    int cell = proj[p];
    if ((unsigned)cell >= (unsigned)n_cells) {      // an index image written for another map size: clamp and flag, never fault
      bad = true;
      cell = cell < 0 ? 0 : n_cells - 1;
    }
    cell_flag[cell] = 1;
  }
  if (bad && err) atomicOr(err, EOD_FLAG_BAD_CELL_INDEX);
}

__device__ __forceinline__ int clamp_cell(int cell, int n_cells) { return cell < 0 ? 0 : (cell >= n_cells ? n_cells - 1 : cell); }

// Stream compaction in two launches: per-block counts, then per-block local scan + prefix of the block counts.
// Block = 1024 threads x 4 consecutive elements.  Used for (a) observed pixels -> every 8th in row-major order
// (custom_rcnn.py:913-914) and (b) marked cells -> slot ids in ascending cell order.
#define SCAN_ELEMS 4096

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* total) {
  __shared__ int wsum[16];
  __shared__ int wtot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 16; ++i) {
      const int t = wsum[i];
      wsum[i] = run;
      run += t;
    }
    wtot = run;
  }
  __syncthreads();
  *total = wtot;
  return wsum[wave] + inc - v;
}

__global__ __launch_bounds__(1024) void mw_count_pixels_kernel(const unsigned char* __restrict__ cover, const int* __restrict__ k_u, int P,
                                                                int* __restrict__ block_cnt) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) return;
  const int base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < P) c += cover[base + j] > 0;
  int total;
  block_exclusive_scan_1024(c, &total);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void mw_select_kernel(const unsigned char* __restrict__ cover, const int* __restrict__ k_u, int P,
                                                          const int* __restrict__ proj, int n_cells, const int* __restrict__ block_cnt,
                                                          int* __restrict__ sel_pix, int* __restrict__ n_sel, int* __restrict__ cell_mark) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_sel = 0;
    return;
  }
  __shared__ int sh_off;
  if (threadIdx.x < 64) {
    int s = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) s += block_cnt[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) sh_off = s;
  }
  const int base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
  int c = 0;
  bool ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ob[j] = (base + j < P) && cover[base + j] > 0;
    c += ob[j];
  }
  int total;
  int rank = block_exclusive_scan_1024(c, &total) + sh_off;   // the scan's barriers also publish sh_off
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (ob[j]) {
      if ((rank & 7) == 0) {
        sel_pix[rank >> 3] = base + j;
        cell_mark[clamp_cell(proj[base + j], n_cells)] = 1;
      }
      ++rank;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_sel = (sh_off + total + 7) >> 3;
}

__global__ __launch_bounds__(1024) void mw_count_cells_kernel(const int* __restrict__ cell_mark, const int* __restrict__ k_u, int N,
                                                               int* __restrict__ block_cnt) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) return;
  const int base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < N) c += cell_mark[base + j] != 0;
  int total;
  block_exclusive_scan_1024(c, &total);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void mw_slots_kernel(const int* __restrict__ cell_mark, const int* __restrict__ k_u, int N,
                                                         const int* __restrict__ block_cnt, int* __restrict__ cell_slot,
                                                         int* __restrict__ slot_cell, int* __restrict__ n_slots) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_slots = 0;
    return;
  }
  __shared__ int sh_off;
  if (threadIdx.x < 64) {
    int s = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) s += block_cnt[b];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) sh_off = s;
  }
  const int base = blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
  int c = 0;
  bool mk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mk[j] = (base + j < N) && cell_mark[base + j] != 0;
    c += mk[j];
  }
  int total;
  int rank = block_exclusive_scan_1024(c, &total) + sh_off;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (mk[j]) {
      cell_slot[base + j] = rank;
      slot_cell[rank] = base + j;
      ++rank;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_slots = sh_off + total;
}

__global__ __launch_bounds__(256) void mw_zero_slots_kernel(long long* __restrict__ wtab, int* __restrict__ slot_cnt,
                                                             const int* __restrict__ n_slots, int K_cap) {
  EOD_CHAIN_PRIO();
  const size_t total = (size_t)(*n_slots) * K_cap;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) wtab[i] = 0;
  const int ns = *n_slots;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) slot_cnt[i] = 0;
}

// The per-cell mean of the per-pixel means (custom_rcnn.py:884-936) is linear in the instance features:
//   mean_cell = (1 / n_cell) * sum_k W[cell][k] * f_k,   W[cell][k] = sum over the cell's sampled pixels covered by k of 1 / cover(p)
// so a sampled pixel contributes ONE scalar per covering instance (2^-32 fixed point, integer atomics: order independent,
// bitwise reproducible) instead of 512 channel atomics.  One wave per sampled pixel, lanes over the instances.
__global__ __launch_bounds__(256) void mw_accumulate_kernel(const float* __restrict__ boxes, const float* __restrict__ masks,
                                                             const int* __restrict__ inst_rows, const int* __restrict__ k_u,
                                                             const int* __restrict__ sel_pix, const int* __restrict__ n_sel,
                                                             const unsigned char* __restrict__ cover, const int* __restrict__ proj,
                                                             int n_cells, const int* __restrict__ cell_slot, int W, int K_cap, float thr,
                                                             long long* __restrict__ wtab, int* __restrict__ slot_cnt) {
  EOD_CHAIN_PRIO();
  const int K = *k_u;
  const int S = *n_sel;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < S; s += gridDim.x * wpb) {
    const int p = sel_pix[s];
    const int y = p / W, x = p - y * W;
    const long long share = (long long)llrint(4294967296.0 / (double)cover[p]);
    const int slot = cell_slot[clamp_cell(proj[p], n_cells)];
    long long* dst = wtab + (size_t)slot * K_cap;
    for (int k = lane; k < K; k += 64) {
      const int r = inst_rows[k];
      const float x0 = boxes[r * 4 + 0], y0 = boxes[r * 4 + 1], x1 = boxes[r * 4 + 2], y1 = boxes[r * 4 + 3];
      if (mask_hit(masks + (size_t)r * 784, x0, y0, x1, y1, x, y, thr))
        atomicAdd(reinterpret_cast<unsigned long long*>(dst + k), (unsigned long long)share);
    }
    if (lane == 0) atomicAdd(slot_cnt + slot, 1);
  }
}

// one wave per slot: mean = (sum_k W_k f_k) / n in f64 (instance order), mem[cell] += mean (custom_rcnn.py:738-743)
__global__ __launch_bounds__(256) void mw_apply_kernel(const long long* __restrict__ wtab, const int* __restrict__ slot_cnt,
                                                        const int* __restrict__ slot_cell, const int* __restrict__ n_slots,
                                                        const float* __restrict__ featn, const int* __restrict__ inst_rows,
                                                        const int* __restrict__ k_u, int K_cap, int D, float* __restrict__ mem) {
  EOD_CHAIN_PRIO();
  const int K = *k_u;
  const int NS = *n_slots;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int slot = blockIdx.x * wpb + (threadIdx.x >> 6); slot < NS; slot += gridDim.x * wpb) {
    double a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = 0.0;
    const long long* wt = wtab + (size_t)slot * K_cap;
    for (int k = 0; k < K; ++k) {
      const long long wk = wt[k];
      if (wk == 0) continue;                      // wave-uniform
      const double w = (double)wk * (1.0 / 4294967296.0);
      const float* f = featn + (size_t)inst_rows[k] * D + lane;
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += w * (double)f[q * 64];
    }
    const double inv = 1.0 / (double)slot_cnt[slot];
    float* m = mem + (size_t)slot_cell[slot] * D + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q * 64] = m[q * 64] + (float)(a[q] * inv);
  }
}

// observation counters (custom_rcnn.py:699-701,743) + reset of the per-frame cell flags
// `dirty` (caller-owned, may be NULL): every cell whose observation count (hence its normalised row) changed; consumed and
// cleared by eod_memory_normalize_dirty_f16.  Cells written by mw_apply are a subset (sampled pixels are pixels of the frame).
__global__ __launch_bounds__(256) void mw_obs_kernel(int* __restrict__ cell_flag, int* __restrict__ cell_mark, const int* __restrict__ k_u,
                                                      int N, float* __restrict__ obs, int* __restrict__ dirty) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    if (cell_flag[i]) {
      obs[i] += 1.0f;
      cell_flag[i] = 0;
      if (dirty) dirty[i] = 1;
    }
    cell_mark[i] = 0;
  }
}

// The same counters, and in the same pass the fp16 snapshot rows of exactly the cells whose count changed (`snapshot`: the
// table the next frame's gather reads; a cell written by mw_apply is a sampled pixel's cell, hence among them): what
// eod_memory_normalize_dirty_f16 would do at the start of the next frame, without its launch and its second scan of the flags.
// The 4 waves of a workgroup share one 64-cell group and split its rows (bit index mod 4).
__global__ __launch_bounds__(256) void mw_obs_snapshot_kernel(int* __restrict__ cell_flag, int* __restrict__ cell_mark,
                                                               const int* __restrict__ k_u, int N, float* __restrict__ obs,
                                                               const float* __restrict__ mem, __half* __restrict__ snapshot) {
  EOD_CHAIN_PRIO();
  if (*k_u == 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_groups = (N + 63) >> 6;
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int c = (g << 6) + lane;
    int f = 0;
    float o = 0.f;
    if (c < N) {
      f = cell_flag[c];
      o = obs[c];
    }
    __syncthreads();                       // every wave has read the flags and counts before wave 0 updates them
    if (f) o += 1.0f;
    if (wave == 0 && c < N) {
      if (f) {
        obs[c] = o;
        cell_flag[c] = 0;
      }
      cell_mark[c] = 0;
    }
    const unsigned long long bal = __ballot(f != 0) & (0x1111111111111111ull << wave);
    eod_snapshot_rows(mem, snapshot, g, bal, lane, [&](int bit) { return __shfl(o, bit, 64); });
  }
}

inline int blocks_for(size_t work, int per = 256, int cap = 4096) {
  size_t b = (work + per - 1) / per;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int eod_unproject_grid_index(const float* depth, int H, int W, const float* T16, float fx, float fy, float cx, float cy,
                                        const float* proj_shift3, const float* map_shift3, float cell, int map_w, int map_h, int order,
                                        float* xyz_or_null, int32_t* idx, eod_stream_t stream) {
  if (!depth || !T16 || !proj_shift3 || !map_shift3 || !idx) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || map_w <= 0 || map_h <= 0 || !(cell > 0.f) || (order != 0 && order != 1)) return EOD_ERR_BAD_DIMS;
  if ((long)map_w * map_h >= (1L << 31)) return EOD_ERR_BAD_DIMS;
  UnprojArgs a{};
  for (int i = 0; i < 16; ++i) a.T[i] = T16[i];
  a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
  for (int i = 0; i < 3; ++i) {
    a.ps[i] = proj_shift3[i];
    a.ms[i] = map_shift3[i];
  }
  a.cell = cell; a.map_w = map_w; a.map_h = map_h; a.order = order;
  hipLaunchKernelGGL(unproject_kernel, dim3(blocks_for((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, depth, H, W, a, xyz_or_null,
                     idx);
  return eod_launch_status();
}

extern "C" int eod_memory_normalize_f16(const float* mem, const float* obs, uint16_t* out_f16, int n_cells, int D, eod_stream_t stream) {
  if (!mem || !obs || !out_f16) return EOD_ERR_NULL;
  if (n_cells <= 0 || D % 4 != 0) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(mem) || !eod_aligned16(out_f16)) return EOD_ERR_ALIGN;
  hipLaunchKernelGGL(normalize_f16_kernel, dim3(blocks_for((size_t)n_cells * (D / 4))), dim3(256), 0, (hipStream_t)stream, mem, obs,
                     reinterpret_cast<__half*>(out_f16), n_cells, D);
  return eod_launch_status();
}

extern "C" size_t eod_memory_write_workspace_bytes(int H, int W, int D, int n_cells, int K_cap, int R_cap) {
  return mw_carve(nullptr, H, W, D, n_cells, R_cap, K_cap).bytes;
}

extern "C" int eod_memory_write_init(void* workspace, size_t workspace_bytes, int H, int W, int D, int n_cells, int K_cap, int R_cap,
                                     eod_stream_t stream) {
  // the per-frame cell flags must start at zero; every eod_memory_write leaves them zero again
  if (!workspace) return EOD_ERR_NULL;
  if (K_cap <= 0) return EOD_ERR_BAD_DIMS;
  const MwWs w = mw_carve(workspace, H, W, D, n_cells, R_cap, K_cap);
  if (workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  if (hipMemsetAsync(w.cell_flag, 0, (size_t)n_cells * 4, (hipStream_t)stream) != hipSuccess) return EOD_ERR_LAUNCH;
  if (hipMemsetAsync(w.cell_mark, 0, (size_t)n_cells * 4, (hipStream_t)stream) != hipSuccess) return EOD_ERR_LAUNCH;
  return eod_launch_status();
}

extern "C" int eod_unique_rows(const int32_t* rows, const int32_t* count, int K_cap, int R_cap, int32_t* out_rows, int32_t* out_count,
                               eod_stream_t stream) {
  if (!rows || !count || !out_rows || !out_count) return EOD_ERR_NULL;
  if (K_cap <= 0 || R_cap <= 0 || R_cap > 512) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(mw_unique_rows_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, rows, count, K_cap, R_cap, out_rows, out_count,
                     (int*)nullptr);
  return eod_launch_status();
}

extern "C" int eod_memory_write(const EodMemWriteDesc* d, eod_stream_t stream) {
  if (!d || !d->featn || !d->prop_boxes || !d->prop_masks || !d->det_rows || !d->det_count || !d->proj || !d->mem || !d->obs ||
      !d->workspace)
    return EOD_ERR_NULL;
  if (d->H <= 0 || d->W <= 0 || d->D != 512 || d->n_cells <= 0 || d->R_cap <= 0 || d->R_cap > 512 || d->K_cap <= 0)
    return EOD_ERR_BAD_DIMS;
  const MwWs w = mw_carve(d->workspace, d->H, d->W, d->D, d->n_cells, d->R_cap, d->K_cap);
  if (d->workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  if (d->snapshot_f16 && (!eod_aligned16(d->snapshot_f16) || !eod_aligned16(d->mem))) return EOD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int P = d->H * d->W;
  hipLaunchKernelGGL(mw_unique_rows_kernel, dim3(1), dim3(512), 0, s, d->det_rows, d->det_count, d->K_cap, d->R_cap, w.inst_rows, w.k_u,
                     d->k_out);
  hipLaunchKernelGGL(mw_coverage_kernel, dim3(blocks_for((size_t)P)), dim3(256), 0, s, d->prop_boxes, d->prop_masks, w.inst_rows, w.k_u,
                     d->proj, d->H, d->W, d->n_cells, d->mask_thresh, w.cover, w.cell_flag, d->err_flags);
  const int pb = (P + SCAN_ELEMS - 1) / SCAN_ELEMS, cb = (d->n_cells + SCAN_ELEMS - 1) / SCAN_ELEMS;
  hipLaunchKernelGGL(mw_count_pixels_kernel, dim3(pb), dim3(1024), 0, s, w.cover, w.k_u, P, w.blk_pix);
  hipLaunchKernelGGL(mw_select_kernel, dim3(pb), dim3(1024), 0, s, w.cover, w.k_u, P, d->proj, d->n_cells, w.blk_pix, w.sel_pix, w.n_sel,
                     w.cell_mark);
  hipLaunchKernelGGL(mw_count_cells_kernel, dim3(cb), dim3(1024), 0, s, w.cell_mark, w.k_u, d->n_cells, w.blk_cell);
  hipLaunchKernelGGL(mw_slots_kernel, dim3(cb), dim3(1024), 0, s, w.cell_mark, w.k_u, d->n_cells, w.blk_cell, w.cell_slot, w.slot_cell,
                     w.n_slots);
  hipLaunchKernelGGL(mw_zero_slots_kernel, dim3(512), dim3(256), 0, s, w.wtab, w.slot_cnt, w.n_slots, d->K_cap);
  hipLaunchKernelGGL(mw_accumulate_kernel, dim3(2048), dim3(256), 0, s, d->prop_boxes, d->prop_masks, w.inst_rows, w.k_u, w.sel_pix,
                     w.n_sel, w.cover, d->proj, d->n_cells, w.cell_slot, d->W, d->K_cap, d->mask_thresh, w.wtab, w.slot_cnt);
  hipLaunchKernelGGL(mw_apply_kernel, dim3(1024), dim3(256), 0, s, w.wtab, w.slot_cnt, w.slot_cell, w.n_slots, d->featn, w.inst_rows,
                     w.k_u, d->K_cap, d->D, d->mem);
  if (d->snapshot_f16) {
    int groups = (d->n_cells + 63) / 64;
    hipLaunchKernelGGL(mw_obs_snapshot_kernel, dim3(groups < 4096 ? groups : 4096), dim3(256), 0, s, w.cell_flag, w.cell_mark, w.k_u,
                       d->n_cells, d->obs, d->mem, reinterpret_cast<__half*>(d->snapshot_f16));
  } else {
    hipLaunchKernelGGL(mw_obs_kernel, dim3(blocks_for((size_t)d->n_cells)), dim3(256), 0, s, w.cell_flag, w.cell_mark, w.k_u, d->n_cells,
                       d->obs, d->dirty);
  }
  return eod_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// a20: explicit semantic map from the implicit memory (custom_rcnn.py:745-756, 938-1017); evaluated lazily
// ------------------------------------------------------------------------------------------------------
namespace {

// one wave per cell: label = argmax_c<C of (temp * mem/|mem|) . zs[:,c] (softmax is monotonic), intensity = mean|mem| (/obs if obs>1)
__global__ __launch_bounds__(256) void semmap_cell_kernel(const float* __restrict__ mem, const float* __restrict__ obs,
                                                           const float* __restrict__ zs, int n_cells, int D, int C1,
                                                           float* __restrict__ intensity, int* __restrict__ labels,
                                                           unsigned* __restrict__ minmax) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int cell = blockIdx.x * wpb + (threadIdx.x >> 6); cell < n_cells; cell += gridDim.x * wpb) {
    float x[8];
    float ss = 0.f, sa = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      x[q] = mem[(size_t)cell * D + q * 64 + lane];
      ss += x[q] * x[q];
      sa += fabsf(x[q]);
    }
    ss = wave_reduce_sum(ss);
    sa = wave_reduce_sum(sa);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    float best = -INFINITY;
    int besti = 0;
    for (int c = 0; c < C1 - 1; ++c) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += (50.0f * (x[q] / denom)) * zs[(size_t)(q * 64 + lane) * C1 + c];
      s = wave_reduce_sum(s);
      if (s > best) {
        best = s;
        besti = c;
      }
    }
    if (lane == 0) {
      float inten = sa / (float)D;
      const float o = obs[cell];
      if (o > 1.0f) inten = inten / o;
      intensity[cell] = inten;
      labels[cell] = besti;
      // non-negative floats order like their bit patterns
      atomicMin(minmax + 0, __float_as_uint(inten));
      atomicMax(minmax + 1, __float_as_uint(inten));
    }
  }
}

__global__ __launch_bounds__(256) void semmap_threshold_kernel(const float* __restrict__ intensity, const unsigned* __restrict__ minmax,
                                                                int n_cells, float thresh, int* __restrict__ labels) {
  const float lo = __uint_as_float(minmax[0]), hi = __uint_as_float(minmax[1]);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += gridDim.x * blockDim.x) {
    const float v = (intensity[i] - lo) / (hi - lo);   // NaN when hi == lo: nothing is thresholded (reference quirk, :751)
    if (v < thresh) labels[i] = -1;
  }
}

}  // namespace

extern "C" int eod_semmap_labels(const float* mem, const float* obs, const float* zs, int n_cells, int D, int C1, float thresh,
                                 int32_t* labels, float* workspace, eod_stream_t stream) {
  if (!mem || !obs || !zs || !labels || !workspace) return EOD_ERR_NULL;
  if (n_cells <= 0 || D != 512 || C1 < 2) return EOD_ERR_BAD_DIMS;
  hipStream_t s = (hipStream_t)stream;
  unsigned* minmax = reinterpret_cast<unsigned*>(workspace);
  float* intensity = workspace + 4;
  static const unsigned init[2] = {0x7F800000u, 0u};   // +inf, 0 (static: outlives the async copy)
  if (hipMemcpyAsync(minmax, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return EOD_ERR_LAUNCH;
  int blocks = (n_cells + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(semmap_cell_kernel, dim3(blocks), dim3(256), 0, s, mem, obs, zs, n_cells, D, C1, intensity, labels, minmax);
  hipLaunchKernelGGL(semmap_threshold_kernel, dim3((n_cells + 255) / 256 > 1024 ? 1024 : (n_cells + 255) / 256), dim3(256), 0, s, intensity,
                     minmax, n_cells, thresh, labels);
  return eod_launch_status();
}
