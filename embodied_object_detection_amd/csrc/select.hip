// Data-dependent selection on device, with no host synchronisation:
//   * CenterNet proposal decode: sigmoid / threshold / per-level top-k / box decode / class-agnostic NMS 0.9 /
//     post-NMS top-256 keeping ties  (Detic/third_party/CenterNet2/centernet/modeling/dense_heads/centernet.py:603-745,
//     compute_grids :321-339, layers/ml_nms.py:4-31)
//   * detectron2 fast_rcnn_inference (threshold, per-class NMS, top-k) used for the final detections
//     (Detic/detic/modeling/roi_heads/detic_roi_heads.py:214-221) and for the memory update
//     (Detic/detic/modeling/meta_arch/custom_rcnn.py:862-869)
//
// Every list has a fixed capacity and a device-side count.  Ordering is made total and deterministic with
// 64-bit keys  (float bits of the score) << 32 | ~index : descending key order == descending score, ties ->
// lower original index first (the rule the CPU oracle uses where upstream is implementation-defined).
// Sorting is an in-LDS bitonic network (<= 16384 keys = 128 KiB of the 160 KiB LDS); NMS is the classic
// 64x64-bit suppression matrix followed by a single-workgroup scan that resolves each 64-box diagonal block
// in registers of one wave and stops as soon as the requested number of boxes has been kept.
#include "eod_common.h"
#include "../../include/eod_hip.h"
#include <algorithm>

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 make_key(float score, unsigned idx) {
  return ((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned key_index(u64 k) { return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_score(u64 k) { return __uint_as_float((unsigned)(k >> 32)); }

// Descending bitonic sort of 1024*E keys held E per thread: thread t owns elements [t*E, t*E + E) of the
// sequence.  Of the log2(N)(log2(N)+1)/2 compare-exchange stages only those with partner distance j >= 64*E need
// the LDS exchange buffer and barriers (10 of 91 stages for N = 8192); distances E <= j < 64*E are wave shuffles
// and j < E stay inside the thread's registers.  Block size must be 1024.
template <int E, int J>
__device__ __forceinline__ void sort_stage_intra(u64 (&v)[E], int base_i, int k) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if ((e & J) == 0) {
      const bool desc = ((base_i + e) & k) == 0;
      const u64 a = v[e], b = v[e | J];
      const bool sw = desc ? (a < b) : (a > b);
      v[e] = sw ? b : a;
      v[e | J] = sw ? a : b;
    }
  }
}

template <int E>
__device__ void block_sort_desc_reg(u64 (&v)[E], u64* xch) {
  const int t = threadIdx.x;
  const int base_i = t * E;
  constexpr int N = 1024 * E;
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
        if (j == 1) sort_stage_intra<E, 1>(v, base_i, k);
        else if (j == 2) sort_stage_intra<E, 2>(v, base_i, k);
        else if (j == 4) sort_stage_intra<E, (E > 4 ? 4 : 1)>(v, base_i, k);
        else sort_stage_intra<E, (E > 8 ? 8 : 1)>(v, base_i, k);
      } else {
        const int tj = j / E;  // partner thread = t ^ tj, same element slot
        const bool lower = (t & tj) == 0;
        if (tj >= 64) {
#pragma unroll
          for (int e = 0; e < E; ++e) xch[e * 1024 + t] = v[e];
          __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          u64 b;
          if (tj < 64) {
            const unsigned lo = __shfl_xor((unsigned)(v[e] & 0xFFFFFFFFull), tj, 64);
            const unsigned hi = __shfl_xor((unsigned)(v[e] >> 32), tj, 64);
            b = ((u64)hi << 32) | lo;
          } else {
            b = xch[e * 1024 + (t ^ tj)];
          }
          const bool desc = ((base_i + e) & k) == 0;
          const u64 a = v[e];
          v[e] = ((a > b) == (lower == desc)) ? a : b;   // keep the max iff (lower == desc); equal keys: either
        }
        if (tj >= 64) __syncthreads();
      }
    }
  }
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}

// ------------------------------------------------------------------------------------------------------
// NMS on a score-sorted list
// ------------------------------------------------------------------------------------------------------
// Suppression matrix, stored TRANSPOSED: maskT[bj*(nb*64) + i] bit t: box (bj*64+t) is suppressed by box i (j > i, same
// label, IoU > thr).  Word bj of 64 consecutive rows is then 512 contiguous bytes, which is what the scan reads.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ labels,
                                                       const int* __restrict__ n_ptr, int nb, float thr, u64* __restrict__ mask) {
  EOD_CHAIN_PRIO();
  const int n = *n_ptr;
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi || bi * 64 >= n || bj * 64 >= n) return;
  __shared__ float cb[64 * 4];
  __shared__ int cl[64];
  const int t = threadIdx.x;
  const int j = bj * 64 + t;
  if (j < n) {
    cb[t * 4 + 0] = boxes[j * 4 + 0];
    cb[t * 4 + 1] = boxes[j * 4 + 1];
    cb[t * 4 + 2] = boxes[j * 4 + 2];
    cb[t * 4 + 3] = boxes[j * 4 + 3];
    cl[t] = labels ? labels[j] : 0;
  }
  __syncthreads();
  const int i = bi * 64 + t;
  if (i >= n) return;
  const float x1 = boxes[i * 4 + 0], y1 = boxes[i * 4 + 1], x2 = boxes[i * 4 + 2], y2 = boxes[i * 4 + 3];
  const float area_i = (x2 - x1) * (y2 - y1);
  const int li = labels ? labels[i] : 0;
  u64 bits = 0;
  const int jmax = min(64, n - bj * 64);
  for (int c = 0; c < jmax; ++c) {
    const int jj = bj * 64 + c;
    if (jj <= i || cl[c] != li) continue;
    const float a1 = cb[c * 4 + 0], b1 = cb[c * 4 + 1], a2 = cb[c * 4 + 2], b2 = cb[c * 4 + 3];
    const float w = fmaxf(fminf(x2, a2) - fmaxf(x1, a1), 0.f);
    const float h = fmaxf(fminf(y2, b2) - fmaxf(y1, b1), 0.f);
    const float inter = w * h;
    const float area_j = (a2 - a1) * (b2 - b1);
    const float iou = inter / (area_i + area_j - inter);
    if (iou > thr) bits |= (1ull << c);
  }
  mask[(size_t)bj * (nb * 64) + i] = bits;
}

struct ScanOut {
  // gathered outputs (any may be null)
  float* out_boxes;
  float* out_scores;
  int* out_labels;
  int* out_rows;
  int* out_count;
  int cap;
};

// Single workgroup of nb (<=128) threads.  keep_ties: keep every kept box whose score equals the score of
// kept box #max_keep (centernet.py:733-741, ">= kth"), else truncate at max_keep.
__global__ __launch_bounds__(128) void nms_scan_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        const int* __restrict__ labels, const int* __restrict__ rows,
                                                        const int* __restrict__ n_ptr, int nb, const u64* __restrict__ mask,
                                                        int max_keep, int keep_ties, int* __restrict__ keep_idx, ScanOut o) {
  EOD_CHAIN_PRIO();
  const int n = *n_ptr;
  const int tid = threadIdx.x;
  __shared__ u64 sh_removed;
  __shared__ u64 sh_kept;
  __shared__ int sh_total;
  __shared__ int sh_stop;
  u64 removed = 0;  // thread w owns word w
  if (tid == 0) {
    sh_total = 0;
    sh_stop = 0;
  }
  __syncthreads();
  const int nchunk = (n + 63) >> 6;
  float kth = -1.0f;
  // the diagonal block of chunk c+1 does not depend on the scan state: fetch it one chunk ahead
  u64 diag_next = 0;
  if (tid < 64 && tid < n) diag_next = mask[(size_t)tid];
  for (int c = 0; c < nchunk; ++c) {
    if (tid == c) sh_removed = removed;
    __syncthreads();
    if (tid < 64) {
      const u64 diag = diag_next;
      {
        const int nrow = (c + 1) * 64 + tid;
        diag_next = (c + 1 < nchunk && nrow < n) ? mask[(size_t)(c + 1) * (nb * 64) + nrow] : 0ull;
      }
      u64 cur = sh_removed;
      u64 kept = 0;
      const int lim = min(64, n - c * 64);
      for (int i = 0; i < lim; ++i) {
        const unsigned lo = __shfl((unsigned)(diag & 0xFFFFFFFFull), i, 64);
        const unsigned hi = __shfl((unsigned)(diag >> 32), i, 64);
        if (!((cur >> i) & 1ull)) {
          kept |= (1ull << i);
          cur |= ((u64)hi << 32) | lo;
        }
      }
      const int total0 = sh_total;
      const int nk = __popcll(kept);
      if (total0 + nk < max_keep) {
        // fast path (every chunk but the last one): all lanes append their kept index in parallel
        if ((kept >> tid) & 1ull) {
          const int pos = total0 + __popcll(kept & ((1ull << tid) - 1ull));
          if (pos < o.cap) keep_idx[pos] = c * 64 + tid;
        }
        if (tid == 0) {
          sh_total = total0 + nk;
          sh_kept = kept;
          sh_stop = 0;
        }
      } else
      if (tid == 0) {
        // append kept indices, honouring max_keep / ties
        int total = sh_total;
        int stop = 0;
        for (int i = 0; i < lim; ++i) {
          if (!((kept >> i) & 1ull)) continue;
          const int idx = c * 64 + i;
          if (total < max_keep) {
            if (total < o.cap) keep_idx[total] = idx;
            ++total;
            if (total == max_keep) kth = scores[idx];
          } else if (keep_ties && scores[idx] >= kth) {
            if (total < o.cap) keep_idx[total] = idx;
            ++total;
          } else {
            stop = 1;
            // boxes after this one have lower-or-equal score; with ties they may still be equal only if
            // scores[idx] >= kth, which failed -> everything later is strictly lower: stop.
            kept &= ((1ull << i) - 1ull);
            break;
          }
        }
        sh_total = total;
        sh_kept = kept;
        // if the list is full and the next chunk starts below kth, stop
        if (!stop && total >= max_keep) {
          const int nxt = (c + 1) * 64;
          if (!keep_ties || nxt >= n || scores[nxt] < kth) stop = 1;
        }
        sh_stop = stop;
      }
    }
    __syncthreads();
    if (sh_stop) break;
    const u64 kept = sh_kept;
    if (tid > c && tid < nb) {
      // word `tid` of the chunk's 64 rows: 512 contiguous bytes, loaded unconditionally as 32 independent 16-byte loads
      // (all in flight together) and masked by the kept bits -- no dependent load chain.
      const ulonglong2* col = reinterpret_cast<const ulonglong2*>(mask + (size_t)tid * (nb * 64) + c * 64);
      ulonglong2 v[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) v[q] = col[q];
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if ((kept >> (2 * q)) & 1ull) removed |= v[q].x;
        if ((kept >> (2 * q + 1)) & 1ull) removed |= v[q].y;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  int total = sh_total;
  if (total > o.cap) total = o.cap;
  if (tid == 0 && o.out_count) *o.out_count = total;
  for (int r = tid; r < total; r += blockDim.x) {
    const int idx = keep_idx[r];
    if (o.out_boxes) {
      o.out_boxes[r * 4 + 0] = boxes[idx * 4 + 0];
      o.out_boxes[r * 4 + 1] = boxes[idx * 4 + 1];
      o.out_boxes[r * 4 + 2] = boxes[idx * 4 + 2];
      o.out_boxes[r * 4 + 3] = boxes[idx * 4 + 3];
    }
    if (o.out_scores) o.out_scores[r] = scores[idx];
    if (o.out_labels) o.out_labels[r] = labels ? labels[idx] : 0;
    if (o.out_rows) o.out_rows[r] = rows ? rows[idx] : idx;
  }
}

// ------------------------------------------------------------------------------------------------------
// CenterNet proposals
// ------------------------------------------------------------------------------------------------------
struct CnArgs {
  const float* head;
  int head_stride;
  int levels;
  int level_off[6];
  int level_w[5];
  int level_stride[5];
  float level_scale[5];
  float score_thresh;
  int topk;
  u64* cand_keys;  // packed per level: level l owns slots [pk_off[l], pk_off[l+1]), min(level size, topk) each
  int* cand_cnt;   // [levels]
  int pk_off[6];
};

#define EOD_SORT_MAX 16384

// one block per level: per-level top-k by score (E = 8: levels up to 8192 positions, E = 16: up to 16384)
template <int E>
__global__ __launch_bounds__(1024) void cn_level_topk_kernel(CnArgs p) {
  EOD_CHAIN_PRIO();
  __shared__ u64 xch[1024 * E];
  __shared__ int sh_cnt;
  const int level = blockIdx.x;
  const int r0 = p.level_off[level];
  const int n = p.level_off[level + 1] - r0;
  if (threadIdx.x == 0) sh_cnt = 0;
  __syncthreads();
  u64 v[E];
  int local = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    u64 k = 0;
    if (i < n) {
      const float heat = eod_sigmoid_precise(p.head[(size_t)(r0 + i) * p.head_stride]);
      if (heat > p.score_thresh) {
        k = make_key(heat, (unsigned)i);
        ++local;
      }
    }
    v[e] = k;
  }
  if (local) atomicAdd(&sh_cnt, local);
  block_sort_desc_reg<E>(v, xch);
  __syncthreads();
  const int cnt = sh_cnt;
  const int take = cnt < p.topk ? cnt : p.topk;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = threadIdx.x * E + e;
    if (r < p.pk_off[level + 1] - p.pk_off[level]) {
      u64 k = 0;
      if (r < take) {
        const float heat = key_score(v[e]);
        const unsigned i = key_index(v[e]);
        k = make_key(sqrtf(heat), (unsigned)(r0 + (int)i));
      }
      p.cand_keys[p.pk_off[level] + r] = k;
    }
  }
  if (threadIdx.x == 0) p.cand_cnt[level] = take;
}

// single block: merge the per-level lists, sort by sqrt-score, decode boxes (E*1024 >= packed slots: E = 4 covers the usual
// 1000 + 1000 + <=1000 + ... <= 4096 candidates with half the sort of E = 8)
template <int E>
__global__ __launch_bounds__(1024) void cn_merge_decode_kernel(CnArgs p, float* sorted_boxes, float* sorted_scores, int* n_sorted) {
  EOD_CHAIN_PRIO();
  __shared__ u64 xch[1024 * E];
  const int total_slots = p.pk_off[p.levels];
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    v[e] = i < total_slots ? p.cand_keys[i] : 0ull;
  }
  block_sort_desc_reg<E>(v, xch);
  int n = 0;
  for (int l = 0; l < p.levels; ++l) n += p.cand_cnt[l];
  if (threadIdx.x == 0) *n_sorted = n;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = threadIdx.x * E + e;
    if (r >= n) continue;
    const u64 k = v[e];
    const int g = (int)key_index(k);
    int level = 0;
    while (level + 1 < p.levels && g >= p.level_off[level + 1]) ++level;
    const int i = g - p.level_off[level];
    const int w = p.level_w[level];
    const int stride = p.level_stride[level];
    const int gy_i = i / w, gx_i = i - gy_i * w;
    const float gx = (float)(gx_i * stride + stride / 2);
    const float gy = (float)(gy_i * stride + stride / 2);
    const float* h = p.head + (size_t)g * p.head_stride;
    const float sc = p.level_scale[level];
    const float st = (float)stride;
    const float r0 = fmaxf(h[1] * sc, 0.f) * st;
    const float r1 = fmaxf(h[2] * sc, 0.f) * st;
    const float r2 = fmaxf(h[3] * sc, 0.f) * st;
    const float r3 = fmaxf(h[4] * sc, 0.f) * st;
    const float x1 = gx - r0, y1 = gy - r1;
    float x2 = gx + r2, y2 = gy + r3;
    x2 = fmaxf(x2, x1 + 0.01f);
    y2 = fmaxf(y2, y1 + 0.01f);
    sorted_boxes[r * 4 + 0] = x1;
    sorted_boxes[r * 4 + 1] = y1;
    sorted_boxes[r * 4 + 2] = x2;
    sorted_boxes[r * 4 + 3] = y2;
    sorted_scores[r] = key_score(k);
  }
}

// ------------------------------------------------------------------------------------------------------
// fast_rcnn_inference candidates: threshold + sort
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void det_candidates_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                               const int* __restrict__ count, int R_cap, int C1, float img_w,
                                                               float img_h, float thr, float* sorted_boxes, float* sorted_scores,
                                                               int* sorted_labels, int* sorted_rows, int* n_sorted) {
  EOD_CHAIN_PRIO();
  constexpr int E = 8;
  __shared__ u64 xch[1024 * E];
  __shared__ int sh_cnt;
  __shared__ unsigned char row_ok[1024];
  int R = R_cap;
  if (count) {
    const int c = *count;
    R = c < R ? c : R;
  }
  const int C = C1 - 1;
  const int slots = R_cap * C;
  if (threadIdx.x == 0) sh_cnt = 0;
  // a row takes part only if its box and ALL its scores are finite (d2 fast_rcnn_inference drops the others)
  for (int r = threadIdx.x; r < R_cap; r += blockDim.x) {
    bool fin = r < R;
    if (fin) {
      for (int q = 0; q < 4; ++q) fin = fin & (bool)isfinite(boxes[r * 4 + q]);
      for (int q = 0; q < C1; ++q) fin = fin & (bool)isfinite(scores[r * C1 + q]);
    }
    row_ok[r] = fin ? 1 : 0;
  }
  __syncthreads();
  u64 v[E];
  int local = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    u64 k = 0;
    if (i < slots) {
      const int r = i / C, c = i - r * C;
      if (row_ok[r]) {
        const float s = scores[r * C1 + c];
        if (s > thr) {
          k = make_key(s, (unsigned)i);
          ++local;
        }
      }
    }
    v[e] = k;
  }
  if (local) atomicAdd(&sh_cnt, local);
  block_sort_desc_reg<E>(v, xch);
  __syncthreads();
  const int n = sh_cnt;
  if (threadIdx.x == 0) *n_sorted = n;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int q = threadIdx.x * E + e;
    if (q >= n) continue;
    const u64 k = v[e];
    const int i = (int)key_index(k);
    const int r = i / C, c = i - r * C;
    sorted_boxes[q * 4 + 0] = fminf(fmaxf(boxes[r * 4 + 0], 0.f), img_w);
    sorted_boxes[q * 4 + 1] = fminf(fmaxf(boxes[r * 4 + 1], 0.f), img_h);
    sorted_boxes[q * 4 + 2] = fminf(fmaxf(boxes[r * 4 + 2], 0.f), img_w);
    sorted_boxes[q * 4 + 3] = fminf(fmaxf(boxes[r * 4 + 3], 0.f), img_h);
    sorted_scores[q] = key_score(k);
    sorted_labels[q] = c;
    sorted_rows[q] = r;
  }
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct SelWs {
  float* sorted_boxes;
  float* sorted_scores;
  int* sorted_labels;
  int* sorted_rows;
  int* n_sorted;
  int* keep_idx;
  u64* mask;
  u64* cand_keys;
  int* cand_cnt;
  size_t bytes;
};

SelWs carve(void* base, int cap_sort, int keep_cap, int cand_slots) {
  SelWs w{};
  size_t off = 0;
  char* b = static_cast<char*>(base);
  const int nb = (cap_sort + 63) / 64;
  auto take = [&](size_t bytes) {
    char* p = b ? b + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  w.sorted_boxes = reinterpret_cast<float*>(take((size_t)cap_sort * 4 * sizeof(float)));
  w.sorted_scores = reinterpret_cast<float*>(take((size_t)cap_sort * sizeof(float)));
  w.sorted_labels = reinterpret_cast<int*>(take((size_t)cap_sort * sizeof(int)));
  w.sorted_rows = reinterpret_cast<int*>(take((size_t)cap_sort * sizeof(int)));
  w.n_sorted = reinterpret_cast<int*>(take(sizeof(int)));
  w.keep_idx = reinterpret_cast<int*>(take((size_t)keep_cap * sizeof(int)));
  w.mask = reinterpret_cast<u64*>(take((size_t)cap_sort * nb * sizeof(u64)));
  w.cand_keys = reinterpret_cast<u64*>(take((size_t)(cand_slots > 0 ? cand_slots : 1) * sizeof(u64)));
  w.cand_cnt = reinterpret_cast<int*>(take(8 * sizeof(int)));
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" size_t eod_proposals_workspace_bytes(int total_positions, int levels, int pre_nms_topk) {
  (void)total_positions;
  const int slots = levels * pre_nms_topk;
  return carve(nullptr, slots, slots, slots).bytes;
}

extern "C" int eod_centernet_proposals(const EodProposalDesc* d, eod_stream_t stream) {
  if (!d || !d->head_out || !d->out_boxes || !d->out_scores || !d->out_count || !d->workspace) return EOD_ERR_NULL;
  if (d->levels < 1 || d->levels > 5 || d->pre_nms_topk < 1 || d->head_stride < 5) return EOD_ERR_BAD_DIMS;
  const int slots = d->levels * d->pre_nms_topk;
  if (slots > 8192) return EOD_ERR_CAPACITY;
  for (int l = 0; l < d->levels; ++l) {
    const int n = d->level_off[l + 1] - d->level_off[l];
    if (n <= 0 || n > EOD_SORT_MAX || d->level_w[l] <= 0 || n % d->level_w[l] != 0) return EOD_ERR_CAPACITY;
  }
  if (d->cap < d->post_nms_topk) return EOD_ERR_CAPACITY;
  const SelWs w = carve(d->workspace, slots, slots, slots);
  if (d->workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  CnArgs a{};
  a.head = d->head_out; a.head_stride = d->head_stride; a.levels = d->levels;
  for (int l = 0; l <= d->levels; ++l) a.level_off[l] = d->level_off[l];
  for (int l = 0; l < d->levels; ++l) {
    a.level_w[l] = d->level_w[l];
    a.level_stride[l] = d->level_stride[l];
    a.level_scale[l] = d->level_scale[l];
  }
  a.score_thresh = d->score_thresh; a.topk = d->pre_nms_topk; a.cand_keys = w.cand_keys; a.cand_cnt = w.cand_cnt;
  // the per-level candidate lists are packed: level l holds at most min(level size, topk) entries
  a.pk_off[0] = 0;
  for (int l = 0; l < d->levels; ++l) a.pk_off[l + 1] = a.pk_off[l] + std::min(d->level_off[l + 1] - d->level_off[l], d->pre_nms_topk);
  int max_level = 0;
  for (int l = 0; l < d->levels; ++l) max_level = std::max(max_level, d->level_off[l + 1] - d->level_off[l]);
  if (max_level <= 8192)
    hipLaunchKernelGGL(cn_level_topk_kernel<8>, dim3(d->levels), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(cn_level_topk_kernel<16>, dim3(d->levels), dim3(1024), 0, s, a);
  if (a.pk_off[d->levels] <= 4096)
    hipLaunchKernelGGL(cn_merge_decode_kernel<4>, dim3(1), dim3(1024), 0, s, a, w.sorted_boxes, w.sorted_scores, w.n_sorted);
  else
    hipLaunchKernelGGL(cn_merge_decode_kernel<8>, dim3(1), dim3(1024), 0, s, a, w.sorted_boxes, w.sorted_scores, w.n_sorted);
  const int nb = (slots + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, s, w.sorted_boxes, (const int*)nullptr, w.n_sorted, nb, d->nms_thresh,
                     w.mask);
  ScanOut o{d->out_boxes, d->out_scores, nullptr, nullptr, d->out_count, d->cap};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(128), 0, s, w.sorted_boxes, w.sorted_scores, (const int*)nullptr,
                     (const int*)nullptr, w.n_sorted, nb, w.mask, d->post_nms_topk, 1, w.keep_idx, o);
  return eod_launch_status();
}

extern "C" size_t eod_detections_workspace_bytes(int R_cap, int C1) {
  const int slots = R_cap * (C1 - 1);
  return carve(nullptr, slots, slots, 0).bytes;
}

extern "C" int eod_fast_rcnn_inference(const EodDetDesc* d, eod_stream_t stream) {
  if (!d || !d->boxes || !d->scores || !d->out_boxes || !d->out_scores || !d->out_classes || !d->out_rows || !d->out_count ||
      !d->workspace)
    return EOD_ERR_NULL;
  if (d->R_cap <= 0 || d->R_cap > 1024 || d->C1 < 2 || d->topk <= 0) return EOD_ERR_BAD_DIMS;
  const int slots = d->R_cap * (d->C1 - 1);
  if (slots > 8192) return EOD_ERR_CAPACITY;
  const SelWs w = carve(d->workspace, slots, slots, 0);
  if (d->workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(det_candidates_kernel, dim3(1), dim3(1024), 0, s, d->boxes, d->scores, d->count, d->R_cap, d->C1, d->img_w, d->img_h,
                     d->score_thresh, w.sorted_boxes, w.sorted_scores, w.sorted_labels, w.sorted_rows, w.n_sorted);
  const int nb = (slots + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, s, w.sorted_boxes, (const int*)w.sorted_labels, w.n_sorted, nb,
                     d->nms_thresh, w.mask);
  ScanOut o{d->out_boxes, d->out_scores, d->out_classes, d->out_rows, d->out_count, d->topk};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(128), 0, s, w.sorted_boxes, w.sorted_scores, (const int*)w.sorted_labels,
                     (const int*)w.sorted_rows, w.n_sorted, nb, w.mask, d->topk, 0, w.keep_idx, o);
  return eod_launch_status();
}
