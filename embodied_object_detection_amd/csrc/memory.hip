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
#include <cstdlib>
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
// a16-a19 write path: three launches
//   mw_cover_kernel    unique instance rows (custom_rcnn.py:875), per-pixel cover count, observed pixels per SCAN_ELEMS-pixel block,
//                      cells the frame hits
//   mw_scatter_kernel  every 8th observed pixel in row-major order (custom_rcnn.py:913-914) adds its 1/cover share to the
//                      (cell, instance) weight table
//   mw_commit_kernel   per hit cell: mean of the per-pixel means from the weight table (f64, instance order), accumulate into the
//                      memory, observation counter, fp16 snapshot row (or dirty mark); leaves every per-frame table zero again
// ------------------------------------------------------------------------------------------------------
#define SCAN_ELEMS 1024        // pixels per workgroup of the two pixel passes: one pixel per thread (the mask gathers of a pixel are a
                               // dependent chain: occupancy, not instruction count, is what hides it)

struct MwWs {
  int* inst_rows;   // [R_cap] unique proposal rows, ascending
  int* k_u;         // [1]
  unsigned char* cover;  // [P] 1 = at least one instance covers the pixel ("observed", custom_rcnn.py:897)
  int* cell_flag;   // [N] any pixel of the frame hit the cell                                   (zero between calls)
  int* cell_cnt;    // [N] number of sampled pixels that hit the cell                             (zero between calls)
  long long* wtab;  // [N, K_cap] fixed point 2^-32: sum over the cell's sampled pixels of 1/cover for every instance (zero between calls)
  int* blk_pix;     // [ceil(P/SCAN_ELEMS)]
  size_t bytes;
};

inline size_t up(size_t v) { return (v + 255) / 256 * 256; }

// scene b of a batch (blockIdx.y of the three write kernels): a workspace pointer of scene 0 moved to scene b's workspace
template <class T>
__device__ __forceinline__ T* scene_ws(T* p, size_t ws_stride) {
  return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)blockIdx.y * ws_stride);
}

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
  w.inst_rows = (int*)take((size_t)R_cap * 4);
  w.k_u = (int*)take(4);
  w.cover = (unsigned char*)take(P);
  w.cell_flag = (int*)take((size_t)n_cells * 4);
  w.cell_cnt = (int*)take((size_t)n_cells * 4);
  w.wtab = (long long*)take((size_t)n_cells * (size_t)K_cap * 8);
  w.blk_pix = (int*)take(((P + SCAN_ELEMS - 1) / SCAN_ELEMS + 1) * 4);
  w.bytes = off;
  return w;
}

#define MW_MAX_R 512
#define MW_MAX_K 128

// unique(det_rows) ascending (custom_rcnn.py:875) by the whole workgroup (any multiple of 64 threads): flags in LDS, ballot
// compaction.  Returns the count; `rows_s[0..count)` holds the rows.  Every thread of the block must call it.
__device__ __forceinline__ int block_unique_rows(const int* __restrict__ det_rows, const int* __restrict__ det_count, int K_cap, int R_cap,
                                                 int* flag_s /*[512]*/, int* wcnt_s /*[8]*/, int* rows_s /*[MW_MAX_K]*/) {
  const int t = threadIdx.x;
  for (int i = t; i < MW_MAX_R; i += blockDim.x) flag_s[i] = 0;
  __syncthreads();
  int K = *det_count;
  K = K < K_cap ? K : K_cap;
  for (int i = t; i < K; i += blockDim.x) {
    const int r = det_rows[i];
    if (r >= 0 && r < R_cap) flag_s[r] = 1;
  }
  __syncthreads();
  const int lane = t & 63;
  for (int i = t; i < MW_MAX_R; i += blockDim.x) {        // i >> 6 is wave-uniform
    const unsigned long long bal = __ballot(flag_s[i] != 0);
    if (lane == 0) wcnt_s[i >> 6] = __popcll(bal);
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int w = 0; w < MW_MAX_R / 64; ++w) total += wcnt_s[w];
  for (int i = t; i < MW_MAX_R; i += blockDim.x) {
    const int f = flag_s[i];
    const unsigned long long bal = __ballot(f != 0);
    if (f) {
      int before = 0;
      for (int w = 0; w < (i >> 6); ++w) before += wcnt_s[w];
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < MW_MAX_K) rows_s[pos] = i;
    }
  }
  __syncthreads();
  return total < MW_MAX_K ? total : MW_MAX_K;
}

__global__ __launch_bounds__(512) void mw_unique_rows_kernel(const int* __restrict__ det_rows, const int* __restrict__ det_count, int K_cap,
                                                              int R_cap, int* __restrict__ inst_rows, int* __restrict__ k_u) {
  EOD_CHAIN_PRIO();
  __shared__ int flag_s[MW_MAX_R], wcnt_s[8], rows_s[MW_MAX_K];
  const int n = block_unique_rows(det_rows, det_count, K_cap, R_cap, flag_s, wcnt_s, rows_s);
  for (int i = threadIdx.x; i < n; i += blockDim.x) inst_rows[i] = rows_s[i];
  if (threadIdx.x == 0) *k_u = n;
}

__device__ __forceinline__ int clamp_cell(int cell, int n_cells) { return cell < 0 ? 0 : (cell >= n_cells ? n_cells - 1 : cell); }


// The instances whose box (grown by one mask pixel: a sample further out is exactly zero) reaches the image rows of this block's
// SCAN_ELEMS pixels, in instance order: cand_s[i] = index k into the unique list, box_s[i] = its box.  Returns their number.
__device__ __forceinline__ int block_band_candidates(const float* __restrict__ boxes, const int* rows_s, int K, int W, int P,
                                                     int* cand_s, float* box_s, int* ncand_s) {
  const int p0 = blockIdx.x * SCAN_ELEMS;
  int p1 = p0 + SCAN_ELEMS - 1;
  if (p1 > P - 1) p1 = P - 1;
  const float ylo = (float)(p0 / W) + 0.5f, yhi = (float)(p1 / W) + 0.5f;
  if (threadIdx.x == 0) *ncand_s = 0;
  __syncthreads();
  if (threadIdx.x < 64) {                      // one wave, instance order kept by ballot compaction
    int base = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + (int)threadIdx.x;
      bool in = false;
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      if (k < K) {
        const int r = rows_s[k];
        b0 = boxes[r * 4 + 0]; b1 = boxes[r * 4 + 1]; b2 = boxes[r * 4 + 2]; b3 = boxes[r * 4 + 3];
        const float my = (b3 - b1) * (1.0f / 14.0f);
        in = !(yhi < b1 - my || ylo > b3 + my);
      }
      const unsigned long long bal = __ballot(in);
      if (in) {
        const int pos = base + __popcll(bal & ((1ull << threadIdx.x) - 1ull));
        cand_s[pos] = k;
        box_s[pos * 4 + 0] = b0; box_s[pos * 4 + 1] = b1; box_s[pos * 4 + 2] = b2; box_s[pos * 4 + 3] = b3;
      }
      base += __popcll(bal);
    }
    if (threadIdx.x == 0) *ncand_s = base;
  }
  __syncthreads();
  return *ncand_s;
}

// exclusive scan over the SCAN_ELEMS threads of a workgroup
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  constexpr int NW = SCAN_ELEMS / 64;
  __shared__ int wsum[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int c = wsum[w];
    if (w < wave) before += c;
    all += c;
  }
  *total = all;
  return before + inc - v;
}

// Mask test of one instance at pixel centre (x + 0.5, y + 0.5), the arithmetic of paste_masks_kernel (heads.hip), split into the part
// that depends on the image row only and the per-pixel part
struct RowSample {
  bool ok;
  int yn, ys;
  float wy_n, wy_s;        // (ys - iy), (iy - yn)
  bool ynv, ysv;
};
__device__ __forceinline__ RowSample row_sample(float y0, float y1, int y) {
  RowSample r;
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
  const float iy = ((gy + 1.0f) * 28.0f - 1.0f) / 2.0f;
  r.ok = iy > -1.0f && iy < 28.0f;
  const float fy = floorf(iy);
  r.yn = (int)fy;
  r.ys = r.yn + 1;
  r.wy_n = (float)r.ys - iy;
  r.wy_s = iy - (float)r.yn;
  r.ynv = (unsigned)r.yn < 28u;
  r.ysv = (unsigned)r.ys < 28u;
  return r;
}
__device__ __forceinline__ bool mask_hit_row(const float* __restrict__ m, const RowSample& r, float x0, float x1, int x, float thr) {
  const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
  if (!(ix > -1.0f && ix < 28.0f && r.ok)) return false;
  const float fx = floorf(ix);
  const int xw = (int)fx;
  const int xe = xw + 1;
  const float nw = ((float)xe - ix) * r.wy_n;
  const float ne = (ix - (float)xw) * r.wy_n;
  const float sw = ((float)xe - ix) * r.wy_s;
  const float se = (ix - (float)xw) * r.wy_s;
  const bool xwv = (unsigned)xw < 28u, xev = (unsigned)xe < 28u;
  float v = 0.f;
  if (xwv && r.ynv) v += m[r.yn * 28 + xw] * nw;
  if (xev && r.ynv) v += m[r.yn * 28 + xe] * ne;
  if (xwv && r.ysv) v += m[r.ys * 28 + xw] * sw;
  if (xev && r.ysv) v += m[r.ys * 28 + xe] * se;
  return v >= thr;
}

// Launch 1.  Block = SCAN_ELEMS threads = SCAN_ELEMS consecutive pixels.  (256-thread workgroups -- which fit into a CU wherever one
// 4-wave implicit-GEMM workgroup of the concurrent detection pass fits -- were measured in the frame: the write took 0.31 ms beside
// the detection pass instead of 0.26 ms, 280 against 283 frames/s: four times the workgroups repeat the per-block row list and
// band set-up, and the stretch of the write beside the GEMMs is not a matter of wave slots.)
__global__ __launch_bounds__(SCAN_ELEMS) void mw_cover_kernel(const float* boxes, const float* masks, const int* det_rows, const int* det_count,
                                                         int K_cap, int R_cap, const int* proj, int H, int W, int n_cells, float thr,
                                                         unsigned char* cover, int* cell_flag, int* blk_pix, int* inst_rows, int* k_u,
                                                         int* k_out, int* __restrict__ err, size_t ws_stride) {
  EOD_CHAIN_PRIO();
  __shared__ int flag_s[MW_MAX_R], wcnt_s[8], rows_s[MW_MAX_K], cand_s[MW_MAX_K], ncand_s;
  __shared__ float box_s[MW_MAX_K * 4];
  if (blockIdx.y) {
    const size_t b = blockIdx.y;
    boxes += b * R_cap * 4; masks += b * (size_t)R_cap * 784; det_rows += b * K_cap; det_count += b; proj += b * (size_t)H * W;
    cover = scene_ws(cover, ws_stride); cell_flag = scene_ws(cell_flag, ws_stride); blk_pix = scene_ws(blk_pix, ws_stride);
    inst_rows = scene_ws(inst_rows, ws_stride); k_u = scene_ws(k_u, ws_stride);
    if (k_out) k_out += b;
  }
  const int K = block_unique_rows(det_rows, det_count, K_cap, R_cap, flag_s, wcnt_s, rows_s);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < K; i += blockDim.x) inst_rows[i] = rows_s[i];
    if (threadIdx.x == 0) {
      *k_u = K;
      if (k_out) *k_out = K;
    }
  }
  if (K == 0) return;                      // update_implicit_memory returns before touching the state (custom_rcnn.py:689-690)
  const int P = H * W;
  const int nc = block_band_candidates(boxes, rows_s, K, W, P, cand_s, box_s, &ncand_s);
  const int p = blockIdx.x * SCAN_ELEMS + threadIdx.x;
  bool bad = false;
  int cnt = 0;
  if (p < P) {
    const int y = p / W, x = p - y * W;
    const float fxp = (float)x + 0.5f, fyp = (float)y + 0.5f;
    // only "observed or not" is needed here (the cover COUNT matters for the sampled pixels alone, every 8th observed one: the
    // scatter pass counts it for those): stop at the first instance that covers the pixel
    for (int i = 0; i < nc && cnt == 0; ++i) {
      const float x0 = box_s[i * 4 + 0], y0 = box_s[i * 4 + 1], x1 = box_s[i * 4 + 2], y1 = box_s[i * 4 + 3];
      // quick reject: a sample more than one mask pixel outside the box is zero
      const float mx = (x1 - x0) * (1.0f / 14.0f), my = (y1 - y0) * (1.0f / 14.0f);
      if (fxp < x0 - mx || fxp > x1 + mx || fyp < y0 - my || fyp > y1 + my) continue;
      const RowSample rs = row_sample(y0, y1, y);
      cnt = mask_hit_row(masks + (size_t)rows_s[cand_s[i]] * 784, rs, x0, x1, x, thr) ? 1 : 0;
    }
    cover[p] = (unsigned char)cnt;
    int cell = proj[p];
    if ((unsigned)cell >= (unsigned)n_cells) {      // an index image written for another map size: clamp and flag, never fault
      bad = true;
      cell = cell < 0 ? 0 : n_cells - 1;
    }
    cell_flag[cell] = 1;
  }
  int total;
  block_exclusive_scan(cnt > 0 ? 1 : 0, &total);
  if (threadIdx.x == 0) blk_pix[blockIdx.x] = total;
  if (bad && err) atomicOr(err, EOD_FLAG_BAD_CELL_INDEX);
}

// Entry `key` of the frame's tables: key >= 0 is weight-table entry wtab[key] (= cell * K_cap + instance), key <= -2 the sample
// count of cell -key - 2.
#define MW_AGG 512
__device__ __forceinline__ void mw_table_add(int key, unsigned long long v, long long* wtab, int* cell_cnt) {
  if (key >= 0) atomicAdd(reinterpret_cast<unsigned long long*>(wtab + key), v);
  else atomicAdd(cell_cnt + (-key - 2), (int)v);
}
// ... summed in the workgroup's LDS table first (open addressing, 8 probes; a crowded table sends the share straight to memory)
__device__ __forceinline__ void mw_agg_add(int* key_s, unsigned long long* val_s, int key, unsigned long long v, long long* wtab,
                                           int* cell_cnt) {
  unsigned h = ((unsigned)key * 2654435761u) >> (32 - 9);
  for (int probe = 0; probe < 8; ++probe) {
    const int old = atomicCAS(&key_s[h], -1, key);
    if (old == -1 || old == key) {
      atomicAdd(&val_s[h], v);
      return;
    }
    h = (h + 1) & (MW_AGG - 1);
  }
  mw_table_add(key, v, wtab, cell_cnt);
}

// Launch 2.  The per-cell mean of the per-pixel means (custom_rcnn.py:884-936) is linear in the instance features:
//   mean_cell = (1 / n_cell) * sum_k W[cell][k] * f_k,   W[cell][k] = sum over the cell's sampled pixels covered by k of 1 / cover(p)
// so a sampled pixel contributes ONE scalar per covering instance (2^-32 fixed point, integer atomics: order independent,
// bitwise reproducible) instead of 512 channel atomics.
__global__ __launch_bounds__(SCAN_ELEMS) void mw_scatter_kernel(const float* boxes, const float* masks, const int* inst_rows, const int* k_u,
                                                           const unsigned char* cover, const int* blk_pix, const int* proj, int H, int W,
                                                           int n_cells, int K_cap, float thr, long long* wtab, int* cell_cnt, int R_cap,
                                                           size_t ws_stride) {
  EOD_CHAIN_PRIO();
  if (blockIdx.y) {
    const size_t b = blockIdx.y;
    boxes += b * R_cap * 4; masks += b * (size_t)R_cap * 784; proj += b * (size_t)H * W;
    inst_rows = scene_ws(inst_rows, ws_stride); k_u = scene_ws(k_u, ws_stride); cover = scene_ws(cover, ws_stride);
    blk_pix = scene_ws(blk_pix, ws_stride); wtab = scene_ws(wtab, ws_stride); cell_cnt = scene_ws(cell_cnt, ws_stride);
  }
  const int K = *k_u;
  if (K == 0) return;
  __shared__ int rows_s[MW_MAX_K], cand_s[MW_MAX_K], ncand_s, sh_off;
  __shared__ float box_s[MW_MAX_K * 4];
  const int P = H * W;
  const int p = blockIdx.x * SCAN_ELEMS + threadIdx.x;
  const int cv = p < P ? (int)cover[p] : 0;
  int total;
  const int local = block_exclusive_scan(cv != 0 ? 1 : 0, &total);
  if (total == 0) return;                                    // no observed pixel in this block (block-uniform)
  // observed pixels in the blocks before this one (one wave; its barrier follows)
  if (threadIdx.x < 64) {
    int sum = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 64) sum += blk_pix[b];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (threadIdx.x == 0) sh_off = sum;
  }
  for (int i = threadIdx.x; i < K; i += blockDim.x) rows_s[i] = inst_rows[i];
  __syncthreads();
  // does any pixel of this block get sampled?  ranks sh_off .. sh_off + total - 1 contain a multiple of 8?
  const int first = sh_off;
  if (((first + 7) & ~7) >= first + total) return;           // block-uniform
  const int nc = block_band_candidates(boxes, rows_s, K, W, P, cand_s, box_s, &ncand_s);
  const int rank = first + local;
  // every 8th observed pixel, row-major (custom_rcnn.py:913-914): at most SCAN_ELEMS / 8 of the block's pixels.  They are compacted
  // into a list, and EIGHT lanes share one sampled pixel: lane j tests band candidates j, j + 8, ... (the mask gathers of one pixel
  // against ~40 candidates were a serial chain on one lane in eight), the hit sets are OR-ed over the eight lanes.
  __shared__ int samp_s[SCAN_ELEMS / 8];
  __shared__ int nsamp_s;
  // The block's shares are summed in LDS first, keyed by their table entry: its <= 128 sampled pixels lie on one or two image rows
  // and neighbours fall into the same cell, so the ~250 shares of a block are a few dozen distinct entries -- and the scatter pass
  // is bound by its integer atomics on ~800 cells' rows in memory.  Integer sums: the order does not matter, bitwise as before.
  __shared__ int key_s[MW_AGG];
  __shared__ unsigned long long val_s[MW_AGG];
  for (int i = threadIdx.x; i < MW_AGG; i += blockDim.x) {
    key_s[i] = -1;
    val_s[i] = 0ull;
  }
  if (threadIdx.x == 0) nsamp_s = 0;
  __syncthreads();
  {
    const bool mine = cv != 0 && (rank & 7) == 0;
    const unsigned long long bal = __ballot(mine);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && bal) base = atomicAdd(&nsamp_s, __popcll(bal));
    base = __shfl(base, 0, 64);
    if (mine) samp_s[base + __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull))] = p;
  }
  __syncthreads();
  const int slot = threadIdx.x >> 3, sub = threadIdx.x & 7;
  if (slot < nsamp_s) {                     // uniform over the 8 lanes of a slot
  const int sp = samp_s[slot];
  const int y = sp / W, x = sp - y * W;
  const float fxp = (float)x + 0.5f, fyp = (float)y + 0.5f;
  const int cell = clamp_cell(proj[sp], n_cells);
  // bit i = band candidate i covers this pixel (MW_MAX_K <= 128 candidates)
  unsigned long long hit0 = 0, hit1 = 0;
  for (int i = sub; i < nc; i += 8) {
    const float x0 = box_s[i * 4 + 0], y0 = box_s[i * 4 + 1], x1 = box_s[i * 4 + 2], y1 = box_s[i * 4 + 3];
    const float mx = (x1 - x0) * (1.0f / 14.0f), my = (y1 - y0) * (1.0f / 14.0f);
    if (fxp < x0 - mx || fxp > x1 + mx || fyp < y0 - my || fyp > y1 + my) continue;
    const RowSample rs = row_sample(y0, y1, y);
    if (mask_hit_row(masks + (size_t)rows_s[cand_s[i]] * 784, rs, x0, x1, x, thr)) {
      if (i < 64) hit0 |= 1ull << i;
      else hit1 |= 1ull << (i - 64);
    }
  }
  const unsigned long long mine0 = hit0, mine1 = hit1;      // the hits this lane found: it also adds their shares
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    hit0 |= __shfl_xor(hit0, off, 64);
    hit1 |= __shfl_xor(hit1, off, 64);
  }
  const int ncov = __popcll(hit0) + __popcll(hit1);          // >= 1: the pixel is observed
  const unsigned long long share = (unsigned long long)llrint(4294967296.0 / (double)ncov);
  if (sub == 0) mw_agg_add(key_s, val_s, -(cell + 2), 1ull, wtab, cell_cnt);
  unsigned long long m0 = mine0, m1 = mine1;
  while (m0) {
    const int i = (int)__ffsll((long long)m0) - 1;
    m0 &= m0 - 1;
    mw_agg_add(key_s, val_s, cell * K_cap + cand_s[i], share, wtab, cell_cnt);
  }
  while (m1) {
    const int i = (int)__ffsll((long long)m1) - 1;
    m1 &= m1 - 1;
    mw_agg_add(key_s, val_s, cell * K_cap + cand_s[64 + i], share, wtab, cell_cnt);
  }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MW_AGG; i += blockDim.x) {
    const int k = key_s[i];
    if (k != -1) mw_table_add(k, val_s[i], wtab, cell_cnt);
  }
}

// Launch 3.  The NW (4 or 16) waves of a workgroup share one 64-cell group and split its hit cells (bit index mod NW): the cells a
// frame writes are a few compact regions of the map, i.e. a few groups with many hit cells each, and a cell is a dependent chain
// (weight row -> instance rows -> memory row) of ~1.5 us: 16 waves per group shorten the longest chain 4x.  One wave per cell,
// 8 consecutive channels per lane.  For a cell with sampled pixels: mean = (sum_k W_k f_k) / n in f64 (instance order),
// mem[cell] += mean (custom_rcnn.py:738-743).  Every hit cell: observation counter + 1 (custom_rcnn.py:699-701,743) and either its
// row of the fp16 snapshot (`snapshot`: the table the next frame's gather reads -- what eod_memory_normalize_dirty_f16 would do
// at the start of the next frame, without its launch and its scan of the flags) or its `dirty` mark.  Resets the per-frame
// tables it consumed (flags, counts, weight-table entries).
template <bool SNAPSHOT, int NW>
__global__ __launch_bounds__(64 * NW) void mw_commit_kernel(int* cell_flag, int* cell_cnt, long long* wtab, const int* k_u, const int* inst_rows,
                                                         const float* featn, int K_cap, int N, float* obs, float* mem, __half* snapshot,
                                                         int* dirty, int R_cap, size_t ws_stride) {
  EOD_CHAIN_PRIO();
  if (blockIdx.y) {
    const size_t b = blockIdx.y;
    cell_flag = scene_ws(cell_flag, ws_stride); cell_cnt = scene_ws(cell_cnt, ws_stride); wtab = scene_ws(wtab, ws_stride);
    k_u = scene_ws(k_u, ws_stride); inst_rows = scene_ws(inst_rows, ws_stride);
    featn += b * (size_t)R_cap * 512; obs += b * (size_t)N; mem += b * (size_t)N * 512;
    if (SNAPSHOT) snapshot += b * (size_t)N * 512;
    if (dirty) dirty += b * (size_t)N;
  }
  const int K = *k_u;
  if (K == 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_groups = (N + 63) >> 6;
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int c = (g << 6) + lane;
    int f = 0, n = 0;
    float o = 0.f;
    if (c < N) {
      f = cell_flag[c];
      o = obs[c];
      n = cell_cnt[c];
    }
    __syncthreads();                       // every wave has read the flags and counts before wave 0 updates them
    if (f) o += 1.0f;
    if (wave == 0 && c < N && f) {
      obs[c] = o;
      cell_flag[c] = 0;
      if (n) cell_cnt[c] = 0;
      if (!SNAPSHOT && dirty) dirty[c] = 1;
    }
    // wave w takes the hit cells whose bit index is w modulo NW
    unsigned long long bal = __ballot(f != 0) & ((NW == 4 ? 0x1111111111111111ull : 0x0001000100010001ull) << wave);
    while (bal) {
      const int bit = (int)__ffsll((long long)bal) - 1;
      bal &= bal - 1;
      const int cell = (g << 6) + bit;
      const float ob = __shfl(o, bit, 64);
      const int nb = __shfl(n, bit, 64);
      float* mrow = mem + (size_t)cell * 512 + lane * 8;
      if (nb == 0 && !SNAPSHOT) continue;
      f32x4 x = *reinterpret_cast<const f32x4*>(mrow);
      f32x4 y = *reinterpret_cast<const f32x4*>(mrow + 4);
      if (nb > 0) {
        long long* wt = wtab + (size_t)cell * K_cap;
        const long long w0 = lane < K ? wt[lane] : 0;
        const long long w1 = (lane + 64) < K ? wt[lane + 64] : 0;
        if (w0) wt[lane] = 0;
        if (w1) wt[lane + 64] = 0;
        double a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = 0.0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          unsigned long long nz = __ballot((half ? w1 : w0) != 0);
          while (nz) {
            const int kl = (int)__ffsll((long long)nz) - 1;
            nz &= nz - 1;
            const long long wk = __shfl(half ? w1 : w0, kl, 64);
            const double w = (double)wk * (1.0 / 4294967296.0);
            const float* fr = featn + (size_t)inst_rows[kl + 64 * half] * 512 + lane * 8;
            const f32x4 fa = *reinterpret_cast<const f32x4*>(fr);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(fr + 4);
            a[0] += w * (double)fa.x; a[1] += w * (double)fa.y; a[2] += w * (double)fa.z; a[3] += w * (double)fa.w;
            a[4] += w * (double)fb.x; a[5] += w * (double)fb.y; a[6] += w * (double)fb.z; a[7] += w * (double)fb.w;
          }
        }
        const double inv = 1.0 / (double)nb;
        x.x = x.x + (float)(a[0] * inv); x.y = x.y + (float)(a[1] * inv); x.z = x.z + (float)(a[2] * inv); x.w = x.w + (float)(a[3] * inv);
        y.x = y.x + (float)(a[4] * inv); y.y = y.y + (float)(a[5] * inv); y.z = y.z + (float)(a[6] * inv); y.w = y.w + (float)(a[7] * inv);
        *reinterpret_cast<f32x4*>(mrow) = x;
        *reinterpret_cast<f32x4*>(mrow + 4) = y;
      }
      if (SNAPSHOT) {
        if (ob > 1.0f) {
          x.x = __fdiv_rn(x.x, ob); x.y = __fdiv_rn(x.y, ob); x.z = __fdiv_rn(x.z, ob); x.w = __fdiv_rn(x.w, ob);
          y.x = __fdiv_rn(y.x, ob); y.y = __fdiv_rn(y.y, ob); y.z = __fdiv_rn(y.z, ob); y.w = __fdiv_rn(y.w, ob);
        }
        __half2 h0 = __floats2half2_rn(x.x, x.y), h1 = __floats2half2_rn(x.z, x.w);
        __half2 h2 = __floats2half2_rn(y.x, y.y), h3 = __floats2half2_rn(y.z, y.w);
        uint4 pk;
        pk.x = *reinterpret_cast<unsigned*>(&h0);
        pk.y = *reinterpret_cast<unsigned*>(&h1);
        pk.z = *reinterpret_cast<unsigned*>(&h2);
        pk.w = *reinterpret_cast<unsigned*>(&h3);
        *reinterpret_cast<uint4*>(snapshot + (size_t)cell * 512 + lane * 8) = pk;
      }
    }
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
  return mw_carve(nullptr, H, W, D, n_cells, R_cap, K_cap).bytes;      // one scene; a batch needs `batch` times this
}

extern "C" int eod_memory_write_init(void* workspace, size_t workspace_bytes, int H, int W, int D, int n_cells, int K_cap, int R_cap,
                                     eod_stream_t stream) {
  // the per-frame cell tables must start at zero; every eod_memory_write leaves them zero again.  A workspace of B scenes
  // (workspace_bytes >= B x the single-scene size) is initialised scene by scene.
  if (!workspace) return EOD_ERR_NULL;
  if (K_cap <= 0 || K_cap > MW_MAX_K || R_cap <= 0 || R_cap > MW_MAX_R) return EOD_ERR_BAD_DIMS;
  if (n_cells <= 0 || (long long)n_cells * K_cap >= (1ll << 31)) return EOD_ERR_BAD_DIMS;      // weight-table entries are int keys
  const size_t one = mw_carve(nullptr, H, W, D, n_cells, R_cap, K_cap).bytes;
  if (workspace_bytes < one) return EOD_ERR_CAPACITY;
  for (size_t b = 0; (b + 1) * one <= workspace_bytes && b < EOD_MAX_BATCH; ++b) {
    const MwWs w = mw_carve(static_cast<char*>(workspace) + b * one, H, W, D, n_cells, R_cap, K_cap);
    if (hipMemsetAsync(w.cell_flag, 0, (size_t)n_cells * 4, (hipStream_t)stream) != hipSuccess) return EOD_ERR_LAUNCH;
    if (hipMemsetAsync(w.cell_cnt, 0, (size_t)n_cells * 4, (hipStream_t)stream) != hipSuccess) return EOD_ERR_LAUNCH;
    if (hipMemsetAsync(w.wtab, 0, (size_t)n_cells * (size_t)K_cap * 8, (hipStream_t)stream) != hipSuccess) return EOD_ERR_LAUNCH;
  }
  return eod_launch_status();
}

extern "C" int eod_unique_rows(const int32_t* rows, const int32_t* count, int K_cap, int R_cap, int32_t* out_rows, int32_t* out_count,
                               eod_stream_t stream) {
  if (!rows || !count || !out_rows || !out_count) return EOD_ERR_NULL;
  if (K_cap <= 0 || K_cap > MW_MAX_K || R_cap <= 0 || R_cap > MW_MAX_R) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(mw_unique_rows_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, rows, count, K_cap, R_cap, out_rows, out_count);
  return eod_launch_status();
}

extern "C" int eod_memory_write(const EodMemWriteDesc* d, eod_stream_t stream) {
  if (!d || !d->featn || !d->prop_boxes || !d->prop_masks || !d->det_rows || !d->det_count || !d->proj || !d->mem || !d->obs ||
      !d->workspace)
    return EOD_ERR_NULL;
  if (d->H <= 0 || d->W <= 0 || d->D != 512 || d->n_cells <= 0 || d->R_cap <= 0 || d->R_cap > MW_MAX_R || d->K_cap <= 0 ||
      d->K_cap > MW_MAX_K)
    return EOD_ERR_BAD_DIMS;
  const MwWs w = mw_carve(d->workspace, d->H, d->W, d->D, d->n_cells, d->R_cap, d->K_cap);
  const int nb = d->batch > 1 ? d->batch : 1;
  if (nb > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  if (d->workspace_bytes < w.bytes * nb) return EOD_ERR_CAPACITY;
  if (!eod_aligned16(d->mem) || !eod_aligned16(d->featn) || (d->snapshot_f16 && !eod_aligned16(d->snapshot_f16))) return EOD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int P = d->H * d->W;
  const int pb = (P + SCAN_ELEMS - 1) / SCAN_ELEMS;
  const size_t wss = w.bytes;      // scene b's workspace starts b * wss bytes further (every carved piece is 256-byte aligned)
  hipLaunchKernelGGL(mw_cover_kernel, dim3(pb, nb), dim3(SCAN_ELEMS), 0, s, d->prop_boxes, d->prop_masks, d->det_rows, d->det_count, d->K_cap, d->R_cap,
                     d->proj, d->H, d->W, d->n_cells, d->mask_thresh, w.cover, w.cell_flag, w.blk_pix, w.inst_rows, w.k_u, d->k_out,
                     d->err_flags, wss);
  hipLaunchKernelGGL(mw_scatter_kernel, dim3(pb, nb), dim3(SCAN_ELEMS), 0, s, d->prop_boxes, d->prop_masks, w.inst_rows, w.k_u, w.cover, w.blk_pix,
                     d->proj, d->H, d->W, d->n_cells, d->K_cap, d->mask_thresh, w.wtab, w.cell_cnt, d->R_cap, wss);
  // one workgroup per CU at most: the kernel runs beside dense launches of other streams, where every workgroup dispatch waits
  // for a slot; a workgroup walks its 64-cell groups in a grid-stride loop
  int groups = (d->n_cells + 63) / 64;
  if (groups > 256) groups = 256;
  if (nb > 1 && groups > 256 / nb) groups = 256 / nb;
  static const int commit_waves = [] {
    const char* e = getenv("EOD_MW_COMMIT_WAVES");
    return (e && atoi(e) == 4) ? 4 : 16;
  }();
  const dim3 cg(groups, nb);
  __half* snap = reinterpret_cast<__half*>(d->snapshot_f16);
  if (d->snapshot_f16 && commit_waves == 16)
    hipLaunchKernelGGL((mw_commit_kernel<true, 16>), cg, dim3(1024), 0, s, w.cell_flag, w.cell_cnt, w.wtab, w.k_u, w.inst_rows, d->featn,
                       d->K_cap, d->n_cells, d->obs, d->mem, snap, (int*)nullptr, d->R_cap, wss);
  else if (d->snapshot_f16)
    hipLaunchKernelGGL((mw_commit_kernel<true, 4>), cg, dim3(256), 0, s, w.cell_flag, w.cell_cnt, w.wtab, w.k_u, w.inst_rows, d->featn,
                       d->K_cap, d->n_cells, d->obs, d->mem, snap, (int*)nullptr, d->R_cap, wss);
  else if (commit_waves == 16)
    hipLaunchKernelGGL((mw_commit_kernel<false, 16>), cg, dim3(1024), 0, s, w.cell_flag, w.cell_cnt, w.wtab, w.k_u, w.inst_rows, d->featn,
                       d->K_cap, d->n_cells, d->obs, d->mem, (__half*)nullptr, d->dirty, d->R_cap, wss);
  else
    hipLaunchKernelGGL((mw_commit_kernel<false, 4>), cg, dim3(256), 0, s, w.cell_flag, w.cell_cnt, w.wtab, w.k_u, w.inst_rows, d->featn,
                       d->K_cap, d->n_cells, d->obs, d->mem, (__half*)nullptr, d->dirty, d->R_cap, wss);
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
