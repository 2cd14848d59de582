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
// Sorting is an in-LDS bitonic network (<= 16384 keys = 128 KiB of the 160 KiB LDS).  NMS runs in the SAME workgroup right behind
// the sort (one launch per selection): the sorted list is walked in chunks of 64; a chunk's boxes are tested against the boxes kept
// so far (16 waves share the kept list) and against each other (64x64-bit diagonal block, resolved in the registers of one wave),
// and the walk stops as soon as the requested number of boxes has been kept -- exactly the greedy NMS of the suppression-matrix
// form, without the [n, n/64] matrix, its launch and the scan's launch.
#include "eod_common.h"
#include "../../include/eod_hip.h"
#include <algorithm>

// In-kernel phase stamps (diagnostics; compiled in only with -DEOD_STAMPS, see tools/select_stamps.py): thread 0 of workgroup 0 writes
// the 100 MHz wall clock at named points of the two single-workgroup selection kernels.
#ifdef EOD_STAMPS
__device__ unsigned long long eod_stamps[64];
#define EOD_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) eod_stamps[i] = wall_clock64(); } while (0)
extern "C" int eod_debug_read_stamps(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(eod_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#else
#define EOD_STAMP(i)
#endif

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
// NMS on a score-sorted list, by the workgroup that sorted it
// ------------------------------------------------------------------------------------------------------
struct ScanOut {
  // gathered outputs (any may be null)
  float* out_boxes;
  float* out_scores;
  int* out_labels;
  int* out_rows;
  int* out_count;
  int cap;
  // optional: torch.unique of out_rows (ascending, custom_rcnn.py:875) and its length
  int* uniq_rows;
  int* uniq_count;
  int uniq_cap;
  // optional: groups of kept entries that share one source row (= one class-agnostic box): rep_of[r] = first kept entry with
  // the row of entry r; rep_list = the entries that are their own representative, ascending; rep_count = their number
  int* rep_of;
  int* rep_list;
  int* rep_count;
};

// scene b of a batch: every output is `batch` single-scene buffers back to back
__device__ __forceinline__ ScanOut scene_outputs(ScanOut o, int b) {
  if (o.out_boxes) o.out_boxes += (size_t)b * o.cap * 4;
  if (o.out_scores) o.out_scores += (size_t)b * o.cap;
  if (o.out_labels) o.out_labels += (size_t)b * o.cap;
  if (o.out_rows) o.out_rows += (size_t)b * o.cap;
  if (o.out_count) o.out_count += b;
  if (o.uniq_rows) o.uniq_rows += (size_t)b * o.uniq_cap;
  if (o.uniq_count) o.uniq_count += b;
  if (o.rep_of) o.rep_of += (size_t)b * o.cap;
  if (o.rep_list) o.rep_list += (size_t)b * o.cap;
  if (o.rep_count) o.rep_count += b;
  return o;
}

#define NMS_KEPT_MAX 512

struct NmsSmem {
  float kb[NMS_KEPT_MAX * 4];   // kept boxes, in keep order
  int kl[NMS_KEPT_MAX];         // their labels
  int kidx[NMS_KEPT_MAX];       // their position in the sorted list
  float cb[64 * 4];             // the chunk's boxes
  float cs[64];                 // the chunk's scores (the tie rule of the last chunks reads them one by one)
  int cl[64];
  u64 diag[64];                 // bit c of word i: chunk box c (c > i) is suppressed by chunk box i
  u64 supp;                     // bit i: chunk box i is suppressed by a box kept in an earlier chunk
  int total, stop;
  float kth;
  int flag[NMS_KEPT_MAX];       // unique rows
  int wcnt[8];
};

__device__ __forceinline__ bool iou_over(float x1, float y1, float x2, float y2, float area_i, float a1, float b1, float a2, float b2,
                                         float thr) {
  const float w = fmaxf(fminf(x2, a2) - fmaxf(x1, a1), 0.f);
  const float h = fmaxf(fminf(y2, b2) - fmaxf(y1, b1), 0.f);
  const float inter = w * h;
  const float area_j = (a2 - a1) * (b2 - b1);
  const float iou = inter / (area_i + area_j - inter);
  return iou > thr;
}

// Greedy NMS over the first n entries of a score-sorted list (boxes / scores / labels / rows in global memory, written by this
// workgroup before the call) + gather of the kept entries.  keep_ties: keep every kept box whose score equals the score of kept
// box #max_keep (centernet.py:733-741, ">= kth"), else truncate at max_keep.  Block = 1024 threads.
__device__ bool block_greedy_nms(const float* __restrict__ boxes, const float* __restrict__ scores, const int* __restrict__ labels,
                                 const int* __restrict__ rows, int n, float thr, int max_keep, int keep_ties, const ScanOut& o,
                                 NmsSmem* S) {
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  if (tid == 0) {
    S->total = 0;
    S->stop = 0;
    S->kth = -1.0f;
  }
  __syncthreads();
  const int nchunk = (n + 63) >> 6;
  // wave 0 holds the NEXT chunk's boxes / scores / labels in registers: its global loads are issued a whole chunk ahead instead of
  // at the top of the chunk that needs them (one L2 round trip per chunk off the serial path)
  float nb0 = 0.f, nb1 = 0.f, nb2 = 0.f, nb3 = 0.f, nsc = 0.f;
  int nlb = 0;
  auto fetch = [&](int c) {
    const int j = c * 64 + tid;
    if (tid < 64 && j < n) {
      nb0 = boxes[j * 4 + 0]; nb1 = boxes[j * 4 + 1]; nb2 = boxes[j * 4 + 2]; nb3 = boxes[j * 4 + 3];
      nsc = scores[j];
      nlb = labels ? labels[j] : 0;
    }
  };
  fetch(0);
  for (int c = 0; c < nchunk; ++c) {
    const int lim = min(64, n - c * 64);
    if (tid < 64) {
      if (tid < lim) {
        S->cb[tid * 4 + 0] = nb0;
        S->cb[tid * 4 + 1] = nb1;
        S->cb[tid * 4 + 2] = nb2;
        S->cb[tid * 4 + 3] = nb3;
        S->cl[tid] = nlb;
        S->cs[tid] = nsc;
      }
      S->diag[tid] = 0;
      if (tid == 0) S->supp = 0;
    }
    if (c + 1 < nchunk) fetch(c + 1);
    __syncthreads();
    const int total0 = S->total;
    const int nk = total0 < NMS_KEPT_MAX ? total0 : NMS_KEPT_MAX;
    const bool valid = lane < lim;
    float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;
    int lj = 0;
    if (valid) {
      a1 = S->cb[lane * 4 + 0]; b1 = S->cb[lane * 4 + 1]; a2 = S->cb[lane * 4 + 2]; b2 = S->cb[lane * 4 + 3];
      lj = S->cl[lane];
    }
    // (A) against the boxes kept in earlier chunks: wave `grp` takes kept boxes grp, grp + 16, ...
    bool sup = false;
    if (valid) {
      for (int j = grp; j < nk; j += 16) {
        if (S->kl[j] != lj) continue;
        const float x1 = S->kb[j * 4 + 0], y1 = S->kb[j * 4 + 1], x2 = S->kb[j * 4 + 2], y2 = S->kb[j * 4 + 3];
        const float area_i = (x2 - x1) * (y2 - y1);
        if (iou_over(x1, y1, x2, y2, area_i, a1, b1, a2, b2, thr)) {
          sup = true;
          break;
        }
      }
    }
    const u64 bal = __ballot(sup);
    if (lane == 0 && bal) atomicOr(&S->supp, bal);
    // (B) inside the chunk: lane = suppressor i, wave `grp` takes the later boxes grp*4 .. grp*4+3
    if (valid) {
      u64 bits = 0;
      const float area_i = (a2 - a1) * (b2 - b1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c2 = grp * 4 + q;
        if (c2 > lane && c2 < lim && S->cl[c2] == lj) {
          if (iou_over(a1, b1, a2, b2, area_i, S->cb[c2 * 4 + 0], S->cb[c2 * 4 + 1], S->cb[c2 * 4 + 2], S->cb[c2 * 4 + 3], thr))
            bits |= 1ull << c2;
        }
      }
      if (bits) atomicOr(&S->diag[lane], bits);
    }
    __syncthreads();
    if (tid < 64) {
      const u64 diag = S->diag[tid];
      u64 cur = S->supp;
      u64 kept = 0;
      for (int i = 0; i < lim; ++i) {
        // i is wave-uniform: v_readlane into scalar registers instead of an LDS-crossbar shuffle per step
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag & 0xFFFFFFFFull), i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), i);
        if (!((cur >> i) & 1ull)) {
          kept |= (1ull << i);
          cur |= ((u64)hi << 32) | lo;
        }
      }
      const int nkept = __popcll(kept);
      if (total0 + nkept < max_keep) {
        // fast path (every chunk but the last one): all lanes append their kept box in parallel
        if ((kept >> tid) & 1ull) {
          const int pos = total0 + __popcll(kept & ((1ull << tid) - 1ull));
          if (pos < NMS_KEPT_MAX) {
            S->kb[pos * 4 + 0] = a1; S->kb[pos * 4 + 1] = b1; S->kb[pos * 4 + 2] = a2; S->kb[pos * 4 + 3] = b2;
            S->kl[pos] = lj;
            S->kidx[pos] = c * 64 + tid;
          }
        }
        if (tid == 0) S->total = total0 + nkept;
      } else if (tid == 0) {
        // append kept boxes one by one, honouring max_keep / ties
        int total = total0;
        int stop = 0;
        float kth = S->kth;
        for (int i = 0; i < lim; ++i) {
          if (!((kept >> i) & 1ull)) continue;
          const int idx = c * 64 + i;
          bool take = false;
          if (total < max_keep) {
            take = true;
          } else if (keep_ties && S->cs[i] >= kth) {
            take = true;
          } else {
            // boxes after this one have lower-or-equal score; with ties they may still be equal only if scores[idx] >= kth,
            // which failed -> everything later is strictly lower: stop
            stop = 1;
            break;
          }
          if (take) {
            if (total < NMS_KEPT_MAX) {
              S->kb[total * 4 + 0] = S->cb[i * 4 + 0]; S->kb[total * 4 + 1] = S->cb[i * 4 + 1];
              S->kb[total * 4 + 2] = S->cb[i * 4 + 2]; S->kb[total * 4 + 3] = S->cb[i * 4 + 3];
              S->kl[total] = S->cl[i];
              S->kidx[total] = idx;
            }
            ++total;
            if (total == max_keep) kth = S->cs[i];
          }
        }
        S->total = total;
        S->kth = kth;
        // if the list is full and the next chunk starts below kth, stop
        if (!stop && total >= max_keep) {
          const int nxt = (c + 1) * 64;
          if (!keep_ties || nxt >= n || scores[nxt] < kth) stop = 1;
        }
        S->stop = stop;
      }
    }
    __syncthreads();
#ifdef EOD_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) eod_stamps[6] = (unsigned long long)(c + 1);      // chunks walked
#endif
    if (S->stop) break;
  }
  __syncthreads();
  EOD_STAMP(5);
  const bool stopped = S->stop != 0;      // the walk ended on its own rule, not because the list ran out
  int total = S->total;
  if (total > o.cap) total = o.cap;
  if (total > NMS_KEPT_MAX) total = NMS_KEPT_MAX;
  if (tid == 0 && o.out_count) *o.out_count = total;
  for (int r = tid; r < total; r += blockDim.x) {
    const int idx = S->kidx[r];
    if (o.out_boxes) {
      o.out_boxes[r * 4 + 0] = S->kb[r * 4 + 0];
      o.out_boxes[r * 4 + 1] = S->kb[r * 4 + 1];
      o.out_boxes[r * 4 + 2] = S->kb[r * 4 + 2];
      o.out_boxes[r * 4 + 3] = S->kb[r * 4 + 3];
    }
    if (o.out_scores) o.out_scores[r] = scores[idx];
    if (o.out_labels) o.out_labels[r] = S->kl[r];
    if (o.out_rows) o.out_rows[r] = rows ? rows[idx] : idx;
  }
  if (o.rep_of && rows) {
    // entries of one source row carry the same box: the first of them represents the group (the mask head is class agnostic)
    if (tid < NMS_KEPT_MAX) S->flag[tid] = 0x7FFFFFFF;
    __syncthreads();
    int my_row = -1;
    if (tid < total) {
      my_row = rows[S->kidx[tid]];
      if (my_row >= 0 && my_row < NMS_KEPT_MAX) atomicMin(&S->flag[my_row], tid);
    }
    __syncthreads();
    int is_rep = 0;
    if (tid < total) {
      const int rep = (my_row >= 0 && my_row < NMS_KEPT_MAX) ? S->flag[my_row] : tid;
      o.rep_of[tid] = rep;
      is_rep = rep == tid;
    }
    u64 rb = 0;
    if (tid < NMS_KEPT_MAX) {
      rb = __ballot(is_rep != 0);
      if (lane == 0) S->wcnt[grp] = __popcll(rb);
    }
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NMS_KEPT_MAX / 64; ++w) {
      const int cw = S->wcnt[w];
      if (w < grp) before += cw;
      all += cw;
    }
    if (tid < NMS_KEPT_MAX && is_rep) o.rep_list[before + __popcll(rb & ((1ull << lane) - 1ull))] = tid;
    if (tid == 0) *o.rep_count = all;
    __syncthreads();
  }
  if (o.uniq_rows) {
    // torch.unique of the kept rows: flags over the row ids (< NMS_KEPT_MAX), ballot compaction, ascending
    if (tid < NMS_KEPT_MAX) S->flag[tid] = 0;
    __syncthreads();
    for (int r = tid; r < total; r += blockDim.x) {
      const int row = rows ? rows[S->kidx[r]] : S->kidx[r];
      if (row >= 0 && row < NMS_KEPT_MAX) S->flag[row] = 1;
    }
    __syncthreads();
    int f = 0;
    u64 fb = 0;
    if (tid < NMS_KEPT_MAX) {
      f = S->flag[tid];
      fb = __ballot(f != 0);
      if (lane == 0) S->wcnt[grp] = __popcll(fb);
    }
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NMS_KEPT_MAX / 64; ++w) {
      const int cw = S->wcnt[w];
      if (w < grp) before += cw;
      all += cw;
    }
    if (tid < NMS_KEPT_MAX && f) {
      const int pos = before + __popcll(fb & ((1ull << lane) - 1ull));
      if (pos < o.uniq_cap) o.uniq_rows[pos] = tid;
    }
    if (tid == 0 && o.uniq_count) *o.uniq_count = all < o.uniq_cap ? all : o.uniq_cap;
  }
  __syncthreads();
  return stopped;
}

__device__ __forceinline__ int score_bin(unsigned bits) {
  int bin = (int)(bits >> 14) - (int)(0x3C000000u >> 14);
  return bin < 0 ? 0 : (bin > 4095 ? 4095 : bin);
}

// sort the n keys of `buf` (n <= 1024 E) descending and leave them there in sorted order
template <int E>
__device__ __forceinline__ void sort_in_place(u64* buf, int n) {
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    v[e] = i < n ? buf[i] : 0ull;
  }
  __syncthreads();                     // every thread holds its keys: the buffer becomes the sort's exchange buffer
  block_sort_desc_reg<E>(v, buf);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int q = threadIdx.x * E + e;
    if (q < n) buf[q] = v[e];
  }
  __syncthreads();
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
  // batch > 1: `head` is level major over the scenes (level l of scene b = rows batch * level_off[l] + b * n_l ...), candidate keys
  // carry the scene's own position level_off[l] + i; cand_keys / cand_cnt hold `batch` sets back to back
  int batch;
};

__device__ __forceinline__ size_t cn_head_row(const CnArgs& p, int scene, int level, int i) {
  const int n = p.level_off[level + 1] - p.level_off[level];
  return (size_t)p.batch * p.level_off[level] + (size_t)scene * n + i;
}

#define EOD_SORT_MAX 16384

// Sorts the n <= 1024 E keys of `buf` (descending) and writes the best `take` of them as (sqrt(score), global position) keys into
// the level's packed candidate slots; the remaining slots are zeroed.
template <int E>
__device__ __forceinline__ void cn_sort_emit(u64* buf, int n, int take, int r0, int slots, u64* __restrict__ out) {
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    v[e] = i < n ? buf[i] : 0ull;
  }
  __syncthreads();
  block_sort_desc_reg<E>(v, buf);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = threadIdx.x * E + e;
    if (r < slots) {
      u64 k = 0;
      if (r < take) k = make_key(sqrtf(key_score(v[e])), (unsigned)(r0 + (int)key_index(v[e])));
      out[r] = k;
    }
  }
}

// one block per level: per-level top-k by score.  Only the candidates that can be among the best `topk` are sorted: a 4096-bin
// histogram of the score bits gives the lowest bin b* whose suffix holds >= topk candidates (scores are sigmoids in (0, 1): the
// bins are 2^14-wide ranges of the float bit pattern, a monotone function of the score), the keys of the bins >= b* are compacted
// and a sort of just their number (1024 / 2048 / 4096 / ... keys) follows -- at 640x640 the 6400 positions of the finest level
// pass the threshold almost everywhere, and a full 8192-key bitonic sort was 70 us on the frame's critical chain.
// EMAX = 8: levels up to 8192 positions, 16: up to 16384.
template <int EMAX>
__global__ __launch_bounds__(1024) void cn_level_topk_kernel(CnArgs p) {
  EOD_CHAIN_PRIO();
  __shared__ u64 xch[1024 * EMAX];
  __shared__ int hist[4096];
  __shared__ int wsum[16];
  __shared__ int sh_cnt, sh_cut, sh_n2;
  const int level = blockIdx.x, scene = blockIdx.y;
  const int r0 = p.level_off[level];
  const int n = p.level_off[level + 1] - r0;
  const size_t head0 = cn_head_row(p, scene, level, 0);
  // (the argument struct is not written to: a modified copy lives in scratch memory -- 168 bytes per lane -- and every indexed
  // read of the level tables goes through it)
  u64* const cand_keys = p.cand_keys + (size_t)scene * p.pk_off[p.levels];
  int* const cand_cnt = p.cand_cnt + scene * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
  if (tid == 0) {
    sh_cnt = 0;
    sh_cut = 0;
    sh_n2 = 0;
  }
  __syncthreads();
  u64 v[EMAX];
  int local = 0;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = e * 1024 + tid;
    u64 k = 0;
    if (i < n) {
      const float heat = eod_sigmoid_precise(p.head[(head0 + i) * p.head_stride]);
      if (heat > p.score_thresh) {
        k = make_key(heat, (unsigned)i);
        int bin = (int)(__float_as_uint(heat) >> 14) - (int)(0x3C000000u >> 14);
        bin = bin < 0 ? 0 : (bin > 4095 ? 4095 : bin);
        atomicAdd(&hist[bin], 1);
        ++local;
      }
    }
    v[e] = k;
  }
  if (local) atomicAdd(&sh_cnt, local);
  __syncthreads();
  const int cnt = sh_cnt;
  const int take = cnt < p.topk ? cnt : p.topk;
  // suffix sums over the bins: thread t owns bins 4 t .. 4 t + 3
  {
    const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    const int mine = h0 + h1 + h2 + h3;
    int inc = mine;                                  // inclusive suffix inside the wave (towards higher lanes)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_down(inc, off, 64);
      if (lane + off < 64) inc += t;
    }
    if (lane == 0) wsum[wave] = inc;
    __syncthreads();
    int after = 0;                                   // candidates in the bins of higher threads' waves
    for (int w = wave + 1; w < 16; ++w) after += wsum[w];
    const int above = after + inc - mine;            // candidates in bins > 4 t + 3
    // the lowest bin whose suffix reaches `take`: exactly one thread finds it
    const int s3 = above + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
    if (take > 0) {
      if (above < take && s3 >= take) sh_cut = 4 * tid + 3;
      else if (s3 < take && s2 >= take) sh_cut = 4 * tid + 2;
      else if (s2 < take && s1 >= take) sh_cut = 4 * tid + 1;
      else if (s1 < take && s0 >= take) sh_cut = 4 * tid;
    }
  }
  __syncthreads();
  const int cut = sh_cut;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    bool in = false;
    if (v[e]) {
      int bin = (int)((unsigned)(v[e] >> 32) >> 14) - (int)(0x3C000000u >> 14);
      bin = bin < 0 ? 0 : (bin > 4095 ? 4095 : bin);
      in = bin >= cut;
    }
    const u64 bal = __ballot(in);
    int base = 0;
    if (lane == 0 && bal) base = atomicAdd(&sh_n2, __popcll(bal));
    base = __shfl(base, 0, 64);
    if (in) xch[base + __popcll(bal & ((1ull << lane) - 1ull))] = v[e];
  }
  __syncthreads();
  const int n2 = sh_n2;                              // take <= n2 <= cnt
  const int slots = p.pk_off[level + 1] - p.pk_off[level];
  u64* out = cand_keys + p.pk_off[level];
  const int ns = n2 > slots ? n2 : slots;            // the sort's size must also cover the slots it zero-fills
  if (ns <= 1024) cn_sort_emit<1>(xch, n2, take, r0, slots, out);
  else if (ns <= 2048) cn_sort_emit<2>(xch, n2, take, r0, slots, out);
  else if (ns <= 4096) cn_sort_emit<4>(xch, n2, take, r0, slots, out);
  else if (EMAX <= 8 || ns <= 8192) cn_sort_emit<8>(xch, n2, take, r0, slots, out);
  else cn_sort_emit<(EMAX > 8 ? 16 : 8)>(xch, n2, take, r0, slots, out);
  if (tid == 0) cand_cnt[level] = take;
}

// single block: merge the per-level lists, sort by sqrt-score, decode boxes (E*1024 >= packed slots: E = 4 covers the usual
// 1000 + 1000 + <=1000 + ... <= 4096 candidates with half the sort of E = 8), then class-agnostic NMS + post-NMS cut with ties
template <int E>
__global__ __launch_bounds__(1024) void cn_merge_nms_kernel(CnArgs p, float* sorted_boxes, float* sorted_scores, float nms_thresh,
                                                             int post_topk, ScanOut o) {
  EOD_CHAIN_PRIO();
  __shared__ u64 xch[1024 * E];
  static_assert(sizeof(NmsSmem) <= sizeof(u64) * 1024 * E, "the NMS state reuses the sort's exchange buffer");
  const int total_slots = p.pk_off[p.levels];
  const int scene = blockIdx.x;
  const u64* const cand_keys = p.cand_keys + (size_t)scene * total_slots;
  const int* const cand_cnt = p.cand_cnt + scene * 8;
  sorted_boxes += (size_t)scene * total_slots * 4;
  sorted_scores += (size_t)scene * total_slots;
  o = scene_outputs(o, scene);
  EOD_STAMP(0);
  u64 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int i = threadIdx.x * E + e;
    v[e] = i < total_slots ? cand_keys[i] : 0ull;
  }
  EOD_STAMP(1);
  int n = 0;
  for (int l = 0; l < p.levels; ++l) n += cand_cnt[l];
  // Decode the boxes of the sorted keys held in `keys` (registers of a full sort, or the sorted LDS list of the best candidates)
  auto decode = [&](int r, u64 k) {
    const int g = (int)key_index(k);
    int level = 0;
    while (level + 1 < p.levels && g >= p.level_off[level + 1]) ++level;
    const int i = g - p.level_off[level];
    const int w = p.level_w[level];
    const int stride = p.level_stride[level];
    const int gy_i = i / w, gx_i = i - gy_i * w;
    const float gx = (float)(gx_i * stride + stride / 2);
    const float gy = (float)(gy_i * stride + stride / 2);
    const float* h = p.head + cn_head_row(p, scene, level, i) * p.head_stride;
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
  };
  // Fast path: NMS 0.9 keeps nearly every candidate, so the post-NMS cut (256 + ties) is reached inside the best few hundred of
  // the up to 4096.  A histogram of the score bits (as in cn_level_topk_kernel) picks the lowest bin whose suffix holds >= 1024
  // candidates; only those are sorted (1024-2048 keys instead of 4096: 35 us of this kernel were the sort) and walked.  If the
  // walk runs out of them before it stops by its own rule, the full list is sorted and walked instead (same results either way).
  __shared__ int hist[4096];
  __shared__ int wsum[16];
  __shared__ int sh_cut, sh_n2;
  if (n > 1024) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
    if (tid == 0) {
      sh_cut = 0;
      sh_n2 = 0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (v[e]) atomicAdd(&hist[score_bin((unsigned)(v[e] >> 32))], 1);
    __syncthreads();
    {
      const int want = 1024;
      const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const int mine = h0 + h1 + h2 + h3;
      int inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(inc, off, 64);
        if (lane + off < 64) inc += t;
      }
      if (lane == 0) wsum[wave] = inc;
      __syncthreads();
      int after = 0;
      for (int w = wave + 1; w < 16; ++w) after += wsum[w];
      const int above = after + inc - mine;
      const int s3 = above + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
      if (above < want && s3 >= want) sh_cut = 4 * tid + 3;
      else if (s3 < want && s2 >= want) sh_cut = 4 * tid + 2;
      else if (s2 < want && s1 >= want) sh_cut = 4 * tid + 1;
      else if (s1 < want && s0 >= want) sh_cut = 4 * tid;
    }
    __syncthreads();
    const int cut = sh_cut;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool in = v[e] != 0 && score_bin((unsigned)(v[e] >> 32)) >= cut;
      const u64 bal = __ballot(in);
      int base = 0;
      if (lane == 0 && bal) base = atomicAdd(&sh_n2, __popcll(bal));
      base = __shfl(base, 0, 64);
      if (in) xch[base + __popcll(bal & ((1ull << lane) - 1ull))] = v[e];
    }
    __syncthreads();
    const int n2 = sh_n2;                            // 1024 <= n2 <= n
    if (n2 < n && n2 <= 4096) {
      if (n2 <= 1024) sort_in_place<1>(xch, n2);
      else if (n2 <= 2048) sort_in_place<2>(xch, n2);
      else sort_in_place<(E >= 4 ? 4 : E)>(xch, n2);
      EOD_STAMP(2);
      for (int r = tid; r < n2; r += 1024) decode(r, xch[r]);
      __syncthreads();
      EOD_STAMP(3);
      const bool stopped = block_greedy_nms(sorted_boxes, sorted_scores, nullptr, nullptr, n2, nms_thresh, post_topk, 1, o,
                                            reinterpret_cast<NmsSmem*>(xch));
      EOD_STAMP(4);
      if (stopped) return;                           // workgroup-uniform
    }
  }
  block_sort_desc_reg<E>(v, xch);
  EOD_STAMP(2);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = threadIdx.x * E + e;
    if (r < n) decode(r, v[e]);
  }
  __syncthreads();          // the sorted list (global) and the end of the sort's use of xch
  EOD_STAMP(3);
  block_greedy_nms(sorted_boxes, sorted_scores, nullptr, nullptr, n, nms_thresh, post_topk, 1, o, reinterpret_cast<NmsSmem*>(xch));
  EOD_STAMP(4);
}

// ------------------------------------------------------------------------------------------------------
// CenterNet proposals, wide lists (the TRAINING thresholds: PRE / POST_NMS_TOPK_TRAIN 4000 / 2000, NMS_TH_TRAIN 0.9,
// Base-C2_L_R5021k_640b64_4x_recurrent.yaml:45-49 -> centernet.py:214-219,603-745): up to 16384 merged candidates and 4096 kept
// boxes, beyond what one workgroup's LDS holds.  Three launches behind cn_level_topk_kernel:
//   cn_rank_decode_kernel   every candidate finds its rank in the merged (sqrt-score desc, position asc) order by binary searches
//                           in the per-level lists (they are sorted already) and writes its decoded box there: a merge without a sort
//   nms_matrix_kernel       the upper triangle of the [n, n/64] suppression bit matrix, one 64 x 64 block per wave, whole chip
//   nms_scan_kernel         one workgroup per scene walks the list in chunks of 64: diagonal block resolved in one wave's registers,
//                           the kept rows OR-ed into the removed set by all 16 waves (next chunk's rows prefetched unconditionally),
//                           stops by the '>= kth' rule of centernet.py:733-741 and writes the kept boxes in order
// ------------------------------------------------------------------------------------------------------
#define WIDE_MAX_SLOTS 16384
#define WIDE_MAX_WORDS (WIDE_MAX_SLOTS / 64)
#define WIDE_MAX_CAP 4096

// number of keys of a (descending by score) level list whose score is > s (strict = 1) or >= s (strict = 0)
__device__ __forceinline__ int cn_count_above(const u64* __restrict__ keys, int n, float s, int strict) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float v = key_score(keys[mid]);
    const bool above = strict ? (v > s) : (v >= s);
    if (above) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void cn_rank_decode_kernel(CnArgs p, float* sorted_boxes, float* sorted_scores, int* n_out) {
  const int total_slots = p.pk_off[p.levels];
  const int scene = blockIdx.y;
  const u64* const cand_keys = p.cand_keys + (size_t)scene * total_slots;
  const int* const cand_cnt = p.cand_cnt + scene * 8;
  sorted_boxes += (size_t)scene * total_slots * 4;
  sorted_scores += (size_t)scene * total_slots;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {
    int n = 0;
    for (int l = 0; l < p.levels; ++l) n += cand_cnt[l];
    n_out[scene] = n;
  }
  if (s >= total_slots) return;
  int level = 0;
  while (level + 1 < p.levels && s >= p.pk_off[level + 1]) ++level;
  if (s - p.pk_off[level] >= cand_cnt[level]) return;
  const u64 k = cand_keys[s];
  const float sc = key_score(k);
  const unsigned g = key_index(k);
  // rank = number of candidates that precede k in (score desc, position asc) order.  A level list is sorted by the HEAT; sqrt is
  // monotone, so the keys of one sqrt-score form a run [a, b) -- inside it the positions are counted one by one (two heats can share
  // a sqrt in fp32, and quantised logits give long runs)
  int rank = 0;
  for (int l = 0; l < p.levels; ++l) {
    const u64* lk = cand_keys + p.pk_off[l];
    const int nl = cand_cnt[l];
    const int a = cn_count_above(lk, nl, sc, 1), b = cn_count_above(lk, nl, sc, 0);
    rank += a;
    for (int q = a; q < b; ++q) rank += key_index(lk[q]) < g ? 1 : 0;
  }
  const int i = (int)g - p.level_off[level];
  const int w = p.level_w[level];
  const int stride = p.level_stride[level];
  const int gy_i = i / w, gx_i = i - gy_i * w;
  const float gx = (float)(gx_i * stride + stride / 2);
  const float gy = (float)(gy_i * stride + stride / 2);
  const float* h = p.head + cn_head_row(p, scene, level, i) * p.head_stride;
  const float scl = p.level_scale[level];
  const float st = (float)stride;
  const float r0 = fmaxf(h[1] * scl, 0.f) * st;
  const float r1 = fmaxf(h[2] * scl, 0.f) * st;
  const float r2 = fmaxf(h[3] * scl, 0.f) * st;
  const float r3 = fmaxf(h[4] * scl, 0.f) * st;
  const float x1 = gx - r0, y1 = gy - r1;
  float x2 = gx + r2, y2 = gy + r3;
  x2 = fmaxf(x2, x1 + 0.01f);
  y2 = fmaxf(y2, y1 + 0.01f);
  reinterpret_cast<float4*>(sorted_boxes)[rank] = make_float4(x1, y1, x2, y2);
  sorted_scores[rank] = sc;
}

// bit (j & 63) of M[i][j >> 6], j > i: IoU(box i, box j) > thr.  Block = 4 waves = 4 column words of one 64-row block.
__global__ __launch_bounds__(256) void nms_matrix_kernel(const float* __restrict__ sorted_boxes, const int* __restrict__ n_ptr,
                                                          int total_slots, int words, float thr, u64* __restrict__ M) {
  const int scene = blockIdx.z;
  const int n = n_ptr[scene];
  const int rb = blockIdx.y, cw = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cw < rb || rb * 64 >= n || cw * 64 >= n) return;              // wave-uniform
  const float4* boxes = reinterpret_cast<const float4*>(sorted_boxes) + (size_t)scene * total_slots;
  M += (size_t)scene * total_slots * words;
  const int i = rb * 64 + lane, j = cw * 64 + lane;
  const float4 bi = i < n ? boxes[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 bj = j < n ? boxes[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
  const int jn = min(64, n - cw * 64);
  u64 bits = 0;
  for (int c = 0; c < jn; ++c) {
    const float a1 = __shfl(bj.x, c, 64), b1 = __shfl(bj.y, c, 64), a2 = __shfl(bj.z, c, 64), b2 = __shfl(bj.w, c, 64);
    if (cw * 64 + c > i && iou_over(bi.x, bi.y, bi.z, bi.w, area_i, a1, b1, a2, b2, thr)) bits |= 1ull << c;
  }
  if (i < n) M[(size_t)i * words + cw] = bits;
}

__global__ __launch_bounds__(1024) void nms_scan_kernel(const float* __restrict__ sorted_boxes, const float* __restrict__ sorted_scores,
                                                         const int* __restrict__ n_ptr, int total_slots, int words,
                                                         const u64* __restrict__ M, int max_keep, ScanOut o) {
  __shared__ u64 removed[WIDE_MAX_WORDS];
  __shared__ u64 sh_kept;
  __shared__ int sh_total, sh_stop;
  __shared__ float sh_kth;
  const int scene = blockIdx.x;
  const int n = n_ptr[scene];
  const float4* boxes = reinterpret_cast<const float4*>(sorted_boxes) + (size_t)scene * total_slots;
  const float* scores = sorted_scores + (size_t)scene * total_slots;
  M += (size_t)scene * total_slots * words;
  o = scene_outputs(o, scene);
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  const int nw = (n + 63) >> 6;
  for (int w = tid; w < WIDE_MAX_WORDS; w += 1024) removed[w] = 0;
  if (tid == 0) {
    sh_total = 0;
    sh_stop = 0;
    sh_kth = -1.0f;
  }
  __syncthreads();
  // Every wave owns four rows of a chunk (grp, grp + 16, grp + 32, grp + 48), a lane the words c + 1 + lane + 64 q behind the
  // diagonal.  The rows of chunk c + 1 are requested before chunk c is resolved: nothing on the serial path waits for HBM.
  constexpr int Q = WIDE_MAX_WORDS / 64;
  u64 nxt[4][Q];
  u64 nd = 0;
  float nsc = 0.f;
  auto fetch = [&](int c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = c * 64 + grp + 16 * r;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int w = c + 1 + lane + 64 * q;
        nxt[r][q] = (row < n && w < nw) ? M[(size_t)row * words + w] : 0ull;
      }
    }
    if (grp == 0) {
      const int row = c * 64 + lane;
      nd = row < n ? M[(size_t)row * words + c] : 0ull;
      nsc = row < n ? scores[row] : 0.f;
    }
  };
  fetch(0);
  for (int c = 0; c < nw; ++c) {
    u64 cur[4][Q];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < Q; ++q) cur[r][q] = nxt[r][q];
    const u64 diag = nd;
    const float sc = nsc;
    if (c + 1 < nw) fetch(c + 1);
    const int lim = min(64, n - c * 64);
    if (grp == 0) {
      u64 rem = removed[c];
      if (lim < 64) rem |= ~0ull << lim;
      u64 kept = 0;
      for (int i = 0; i < lim; ++i) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag & 0xFFFFFFFFull), i);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), i);
        if (!((rem >> i) & 1ull)) {
          kept |= 1ull << i;
          rem |= ((u64)hi << 32) | lo;
        }
      }
      // the cut: the first max_keep kept boxes, then every kept box whose score equals the max_keep-th's (centernet.py:733-741)
      const int total0 = sh_total;
      const bool mine = (kept >> lane) & 1ull;
      const int r = total0 + __popcll(kept & ((1ull << lane) - 1ull));
      float kth = sh_kth;
      const u64 at_k = __ballot(mine && r == max_keep - 1);
      if (at_k) kth = __shfl(sc, __ffsll((long long)at_k) - 1, 64);
      const bool take = mine && (r < max_keep || sc >= kth);
      const u64 taken = __ballot(take);
      const bool over = __ballot(mine && !take) != 0;
      if (take && r < o.cap) {
        reinterpret_cast<float4*>(o.out_boxes)[r] = boxes[c * 64 + lane];
        o.out_scores[r] = sc;
      }
      if (lane == 0) {
        const int total = total0 + __popcll(taken);
        sh_total = total;
        sh_kth = kth;
        sh_kept = taken;
        // every later score is <= this chunk's last: past the cut only equal scores can still be taken
        int stop = over ? 1 : 0;
        if (!stop && total >= max_keep) {
          const int nx = (c + 1) * 64;
          if (nx >= n || scores[nx] < kth) stop = 1;
        }
        sh_stop = stop;
      }
    }
    __syncthreads();
    if (sh_stop) break;
    const u64 kept = sh_kept;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      u64 acc = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if ((kept >> (grp + 16 * r)) & 1ull) acc |= cur[r][q];
      const int w = c + 1 + lane + 64 * q;
      if (acc && w < nw) atomicOr(&removed[w], acc);
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) *o.out_count = sh_total < o.cap ? sh_total : o.cap;
}

// ------------------------------------------------------------------------------------------------------
// fast_rcnn_inference in one launch: threshold + sort + per-class NMS + top-k (+ unique rows)
// ------------------------------------------------------------------------------------------------------
// All detections of one proposal row carry the row's class-agnostic box, so per-class NMS never needs an IoU between two
// candidates: it needs the IoU between two ROWS.  The kernel builds the R x R bit matrix M (bit r' of row r: IoU(box r, box r') >
// thr) once -- one 64-bit word per thread -- and greedy NMS becomes bit tests: a candidate (row r, class c) is suppressed iff bit
// r of supp[c] is set, where supp[c] is the OR of M[r'] over the kept (r', c).  Candidates are sorted in batches: the first batch
// holds the >= 1024 best scores (histogram of the score bits, as in cn_level_topk_kernel), which is nearly always enough to keep
// `topk`; only if it is not, the rest is sorted and the walk continues (same state, same order: exactly the greedy NMS over the
// fully sorted list).  The walk itself is done by ONE wave (no workgroup barrier per chunk of 64 candidates).
#define DET_MAX_R 512
#define DET_WORDS (DET_MAX_R / 64)
#define DET_MAX_C 24

__global__ __launch_bounds__(1024) void det_select_kernel(const float* boxes, const float* scores, const int* count, int R_cap, int C1,
                                                           float img_w, float img_h, float thr, float nms_thresh, int topk, ScanOut o) {
  EOD_CHAIN_PRIO();
  constexpr int EMAX = 8;
  {
    const int scene = blockIdx.x;      // one workgroup per scene of a batch
    boxes += (size_t)scene * R_cap * 4;
    scores += (size_t)scene * R_cap * C1;
    if (count) count += scene;
    o = scene_outputs(o, scene);
  }
  __shared__ u64 buf[1024 * EMAX];                 // compaction target / sort exchange / sorted keys of the batch
  __shared__ u64 Mx[DET_MAX_R * DET_WORDS];        // IoU bit matrix of the rows
  __shared__ float cbx[DET_MAX_R * 4];             // clipped boxes
  __shared__ u64 supp[DET_MAX_C * DET_WORDS];      // per class: rows suppressed by the kept detections of that class
  __shared__ int hist[4096];
  __shared__ u64 kept_key[NMS_KEPT_MAX];
  __shared__ int flag[NMS_KEPT_MAX];
  __shared__ int wsum[16];
  __shared__ u64 keptbits[1024 * 8 / 64];          // bit q: entry q of the sorted batch survives its class's NMS
  __shared__ int wordpre[128];
  __shared__ unsigned char row_ok[DET_MAX_R];
  __shared__ int sh_cnt, sh_cut, sh_n2, sh_total, sh_done;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int R = R_cap;
  if (count) {
    const int c = *count;
    R = c < R ? c : R;
  }
  const int C = C1 - 1;
  const int slots = R_cap * C;
  const int W = (R + 63) >> 6;
#ifdef EOD_STAMPS
  const int sb = topk <= 100 ? 8 : 24;
#endif
  EOD_STAMP(sb + 0);
  for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
  for (int i = tid; i < DET_MAX_C * DET_WORDS; i += 1024) supp[i] = 0;
  if (tid == 0) {
    sh_cnt = 0;
    sh_cut = 0;
    sh_n2 = 0;
    sh_total = 0;
    sh_done = 0;
  }
  // a row takes part only if its box and ALL its scores are finite (d2 fast_rcnn_inference drops the others)
  for (int r = tid; r < R_cap; r += 1024) {
    bool fin = r < R;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (fin) {
      b0 = boxes[r * 4 + 0]; b1 = boxes[r * 4 + 1]; b2 = boxes[r * 4 + 2]; b3 = boxes[r * 4 + 3];
      fin = isfinite(b0) && isfinite(b1) && isfinite(b2) && isfinite(b3);
      for (int q = 0; q < C1; ++q) fin = fin & (bool)isfinite(scores[r * C1 + q]);
    }
    row_ok[r] = fin ? 1 : 0;
    cbx[r * 4 + 0] = fminf(fmaxf(b0, 0.f), img_w);
    cbx[r * 4 + 1] = fminf(fmaxf(b1, 0.f), img_h);
    cbx[r * 4 + 2] = fminf(fmaxf(b2, 0.f), img_w);
    cbx[r * 4 + 3] = fminf(fmaxf(b3, 0.f), img_h);
  }
  // the thread's scores: eight independent loads, one latency (slot i = e * 1024 + tid = (row, class))
  float sc[EMAX];
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = e * 1024 + tid;
    const int r = i / C, c = i - r * C;
    sc[e] = (i < slots && r < R) ? scores[r * C1 + c] : -1.0f;
  }
  __syncthreads();
  EOD_STAMP(sb + 1);
  // IoU bit matrix: one word (64 partner rows) per thread and step
  for (int idx = tid; idx < R * W; idx += 1024) {
    const int r = idx / W, w = idx - r * W;
    const float x1 = cbx[r * 4 + 0], y1 = cbx[r * 4 + 1], x2 = cbx[r * 4 + 2], y2 = cbx[r * 4 + 3];
    const float area = (x2 - x1) * (y2 - y1);
    u64 bits = 0;
    for (int b = 0; b < 64; ++b) {
      const int r2 = w * 64 + b;
      if (r2 < R && r2 != r &&
          iou_over(x1, y1, x2, y2, area, cbx[r2 * 4 + 0], cbx[r2 * 4 + 1], cbx[r2 * 4 + 2], cbx[r2 * 4 + 3], nms_thresh))
        bits |= 1ull << b;
    }
    Mx[r * DET_WORDS + w] = bits;
  }
  EOD_STAMP(sb + 2);
  // candidates: score > thr on a finite row; histogram of their score bits
  u64 key[EMAX];
  int local = 0;
#pragma unroll
  for (int e = 0; e < EMAX; ++e) {
    const int i = e * 1024 + tid;
    u64 k = 0;
    if (i < slots && row_ok[i / C] && sc[e] > thr) {
      k = make_key(sc[e], (unsigned)i);
      atomicAdd(&hist[score_bin(__float_as_uint(sc[e]))], 1);
      ++local;
    }
    key[e] = k;
  }
  if (local) atomicAdd(&sh_cnt, local);
  __syncthreads();
  const int n = sh_cnt;
  // the lowest bin whose suffix holds min(n, 1024) candidates
  {
    const int want = n < 1024 ? n : 1024;
    const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    const int mine = h0 + h1 + h2 + h3;
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_down(inc, off, 64);
      if (lane + off < 64) inc += t;
    }
    if (lane == 0) wsum[wave] = inc;
    __syncthreads();
    int after = 0;
    for (int w = wave + 1; w < 16; ++w) after += wsum[w];
    const int above = after + inc - mine;
    const int s3 = above + h3, s2 = s3 + h2, s1 = s2 + h1, s0 = s1 + h0;
    if (want > 0) {
      if (above < want && s3 >= want) sh_cut = 4 * tid + 3;
      else if (s3 < want && s2 >= want) sh_cut = 4 * tid + 2;
      else if (s2 < want && s1 >= want) sh_cut = 4 * tid + 1;
      else if (s1 < want && s0 >= want) sh_cut = 4 * tid;
    }
  }
  __syncthreads();
  const int cut = sh_cut;
  EOD_STAMP(sb + 3);
  for (int batch = 0; batch < 2; ++batch) {
    // batch 0: the candidates of the bins >= cut; batch 1 (only if batch 0 did not fill the list): all the others
#pragma unroll
    for (int e = 0; e < EMAX; ++e) {
      bool in = false;
      if (key[e]) {
        const bool hi = score_bin((unsigned)(key[e] >> 32)) >= cut;
        in = batch == 0 ? hi : !hi;
      }
      const u64 bal = __ballot(in);
      int base = 0;
      if (lane == 0 && bal) base = atomicAdd(&sh_n2, __popcll(bal));
      base = __shfl(base, 0, 64);
      if (in) buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = key[e];
    }
    __syncthreads();
    const int n2 = sh_n2;
    if (n2 <= 1024) sort_in_place<1>(buf, n2);
    else if (n2 <= 2048) sort_in_place<2>(buf, n2);
    else if (n2 <= 4096) sort_in_place<4>(buf, n2);
    else sort_in_place<8>(buf, n2);
    EOD_STAMP(sb + 4 + 2 * batch);
    // Greedy NMS over the sorted batch, per class and in parallel: suppression only acts inside a class, so the greedy walk over
    // the whole list is the same as C independent walks over each class's entries in list order.  Wave w takes classes w, w + 16:
    // per 64-entry chunk one ballot finds its class's entries (a few per chunk), each is a bit test against the class's suppressed-
    // row set and, if kept, an OR of its row of the IoU matrix into that set (LDS, in program order within the wave).  (First
    // version: ONE wave walked all entries of all classes, 128 dependent shuffle steps per chunk: 45 us to keep 100, 131 us to keep
    // 300.)  The kept entries are then ranked in list order and the first (topk - total) of them taken.
    for (int i = tid; i < 1024 * EMAX / 64; i += 1024) keptbits[i] = 0;
    __syncthreads();
    for (int cls = wave; cls < C; cls += 16) {
      // the class's suppressed-row set lives in registers for the walk: lane w holds word w
      u64 myw = lane < DET_WORDS ? supp[cls * DET_WORDS + lane] : 0ull;
      for (int c0 = 0; c0 < n2; c0 += 64) {
        const int q = c0 + lane;
        int r = 0, mycl = -1;
        if (q < n2) {
          const int slot = (int)key_index(buf[q]);
          r = slot / C;
          mycl = slot - r * C;
        }
        u64 m = __ballot(mycl == cls);
        u64 keptmask = 0;
        while (m) {
          const int j = (int)__ffsll((long long)m) - 1;
          m &= m - 1;
          // j and rj are wave-uniform (they come from the ballot): v_readlane into scalar registers, no LDS-crossbar shuffle
          const int rj = __builtin_amdgcn_readlane(r, j);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myw & 0xFFFFFFFFull), rj >> 6);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myw >> 32), rj >> 6);
          const u64 word = ((u64)hi << 32) | lo;
          if (!((word >> (rj & 63)) & 1ull)) {               // wave-uniform
            keptmask |= 1ull << j;
            if (lane < W) myw |= Mx[rj * DET_WORDS + lane];
          }
        }
        if (lane == 0 && keptmask) atomicOr(&keptbits[c0 >> 6], keptmask);
      }
      if (lane < DET_WORDS) supp[cls * DET_WORDS + lane] = myw;
    }
    __syncthreads();
    // rank of every kept entry in list order: popcounts of the words before it + of the bits below it
    {
      const int nwords = (n2 + 63) >> 6;                   // <= 128
      if (tid < 128) {
        const int pc = tid < nwords ? __popcll(keptbits[tid]) : 0;
        int inc = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_up(inc, off, 64);
          if (lane >= off) inc += t;
        }
        wordpre[tid] = inc - pc;                             // exclusive inside the wave
        if (lane == 63) wsum[wave] = inc;                    // waves 0 and 1
      }
      __syncthreads();
      const int total0 = sh_total;
      const int kept_all = wsum[0] + (nwords > 64 ? wsum[1] : 0);
#pragma unroll
      for (int e = 0; e < EMAX; ++e) {
        const int q = e * 1024 + tid;
        if (q < n2) {
          const u64 wbits = keptbits[q >> 6];
          if ((wbits >> (q & 63)) & 1ull) {
            const int rank = wordpre[q >> 6] + ((q >> 6) >= 64 ? wsum[0] : 0) + __popcll(wbits & ((1ull << (q & 63)) - 1ull));
            const int pos = total0 + rank;
            if (pos < topk && pos < NMS_KEPT_MAX) kept_key[pos] = buf[q];
          }
        }
      }
      __syncthreads();
      if (tid == 0) {
        int total = total0 + kept_all;
        const bool full = total >= topk;
        if (full) total = topk;
        sh_total = total;
        // done when the list is full or no candidate is left for a second batch
        sh_done = (full || batch == 1 || n2 >= n) ? 1 : 0;
        sh_n2 = 0;
      }
    }
    __syncthreads();
    EOD_STAMP(sb + 5 + 2 * batch);
    if (sh_done) break;
  }
  __syncthreads();
  int total = sh_total;
  if (total > o.cap) total = o.cap;
  if (total > NMS_KEPT_MAX) total = NMS_KEPT_MAX;
  if (tid == 0 && o.out_count) *o.out_count = total;
  int my_row = -1;
  if (tid < total) {
    const u64 k = kept_key[tid];
    const int slot = (int)key_index(k);
    const int r = slot / C, cl = slot - r * C;
    my_row = r;
    if (o.out_boxes) {
      o.out_boxes[tid * 4 + 0] = cbx[r * 4 + 0];
      o.out_boxes[tid * 4 + 1] = cbx[r * 4 + 1];
      o.out_boxes[tid * 4 + 2] = cbx[r * 4 + 2];
      o.out_boxes[tid * 4 + 3] = cbx[r * 4 + 3];
    }
    if (o.out_scores) o.out_scores[tid] = key_score(k);
    if (o.out_labels) o.out_labels[tid] = cl;
    if (o.out_rows) o.out_rows[tid] = r;
  }
  if (o.rep_of) {
    // entries of one source row carry the same box: the first of them represents the group (the mask head is class agnostic)
    if (tid < NMS_KEPT_MAX) flag[tid] = 0x7FFFFFFF;
    __syncthreads();
    if (tid < total && my_row >= 0 && my_row < NMS_KEPT_MAX) atomicMin(&flag[my_row], tid);
    __syncthreads();
    int is_rep = 0;
    if (tid < total) {
      const int rep = (my_row >= 0 && my_row < NMS_KEPT_MAX) ? flag[my_row] : tid;
      o.rep_of[tid] = rep;
      is_rep = rep == tid;
    }
    u64 rb = 0;
    if (tid < NMS_KEPT_MAX) {
      rb = __ballot(is_rep != 0);
      if (lane == 0) wsum[wave] = __popcll(rb);
    }
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NMS_KEPT_MAX / 64; ++w) {
      const int cw = wsum[w];
      if (w < wave) before += cw;
      all += cw;
    }
    if (tid < NMS_KEPT_MAX && is_rep) o.rep_list[before + __popcll(rb & ((1ull << lane) - 1ull))] = tid;
    if (tid == 0) *o.rep_count = all;
    __syncthreads();
  }
  if (o.uniq_rows) {
    // torch.unique of the kept rows: flags over the row ids (< NMS_KEPT_MAX), ballot compaction, ascending
    if (tid < NMS_KEPT_MAX) flag[tid] = 0;
    __syncthreads();
    if (tid < total && my_row >= 0 && my_row < NMS_KEPT_MAX) flag[my_row] = 1;
    __syncthreads();
    int f = 0;
    u64 fb = 0;
    if (tid < NMS_KEPT_MAX) {
      f = flag[tid];
      fb = __ballot(f != 0);
      if (lane == 0) wsum[wave] = __popcll(fb);
    }
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NMS_KEPT_MAX / 64; ++w) {
      const int cw = wsum[w];
      if (w < wave) before += cw;
      all += cw;
    }
    if (tid < NMS_KEPT_MAX && f) {
      const int pos = before + __popcll(fb & ((1ull << lane) - 1ull));
      if (pos < o.uniq_cap) o.uniq_rows[pos] = tid;
    }
    if (tid == 0 && o.uniq_count) *o.uniq_count = all < o.uniq_cap ? all : o.uniq_cap;
  }
  EOD_STAMP(sb + 8);
#ifdef EOD_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    eod_stamps[sb + 9] = (unsigned long long)sh_cnt;       // candidates above the threshold
    eod_stamps[sb + 10] = (unsigned long long)sh_total;    // kept
  }
#endif
}

// The scene-local index lists of a batch (kept proposal rows, detection-group representatives) as ONE list of global indices
// b * id_stride + list[b][k], scene by scene: the mask head then runs once over the concatenation with a single count.
__global__ __launch_bounds__(256) void concat_lists_kernel(const int* __restrict__ lists, const int* __restrict__ counts, int cap_in,
                                                            int id_stride, int batch, int* __restrict__ out, int* __restrict__ out_count) {
  EOD_CHAIN_PRIO();
  int off = 0;
  for (int b = 0; b < batch; ++b) {
    int c = counts[b];
    c = c < 0 ? 0 : (c > cap_in ? cap_in : c);
    for (int k = threadIdx.x; k < c; k += blockDim.x) out[off + k] = b * id_stride + lists[(size_t)b * cap_in + k];
    off += c;
  }
  if (threadIdx.x == 0) *out_count = off;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct SelWs {
  float* sorted_boxes;
  float* sorted_scores;
  int* sorted_labels;
  int* sorted_rows;
  u64* cand_keys;
  int* cand_cnt;
  size_t bytes;
};

SelWs carve(void* base, int cap_sort, int keep_cap, int cand_slots) {
  SelWs w{};
  size_t off = 0;
  char* b = static_cast<char*>(base);
  (void)keep_cap;
  auto take = [&](size_t bytes) {
    char* p = b ? b + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  w.sorted_boxes = reinterpret_cast<float*>(take((size_t)cap_sort * 4 * sizeof(float)));
  w.sorted_scores = reinterpret_cast<float*>(take((size_t)cap_sort * sizeof(float)));
  w.sorted_labels = reinterpret_cast<int*>(take((size_t)cap_sort * sizeof(int)));
  w.sorted_rows = reinterpret_cast<int*>(take((size_t)cap_sort * sizeof(int)));
  w.cand_keys = reinterpret_cast<u64*>(take((size_t)(cand_slots > 0 ? cand_slots : 1) * sizeof(u64)));
  w.cand_cnt = reinterpret_cast<int*>(take(8 * EOD_MAX_BATCH * sizeof(int)));
  w.bytes = off;
  return w;
}

}  // namespace

// Wide lists (levels * pre_nms_topk > 8192, the training thresholds): the packed candidate slots are bounded by the pyramid's
// positions; behind the narrow layout sit the candidate count per scene and the suppression bit matrix.
struct WideWs {
  int slots;       // bound of the packed candidate slots of one scene
  int words;       // 64-bit words per matrix row
  int* n;          // [batch]
  u64* matrix;     // [batch][slots][words]
  size_t bytes;
};

static WideWs carve_wide(void* base, size_t narrow_bytes, int total_positions, int levels, int pre_nms_topk, int nb) {
  WideWs w{};
  w.slots = std::min(total_positions, levels * pre_nms_topk);
  w.words = (w.slots + 63) / 64;
  char* b = static_cast<char*>(base);
  size_t off = narrow_bytes;
  w.n = reinterpret_cast<int*>(b ? b + off : nullptr);
  off += align_up(EOD_MAX_BATCH * sizeof(int));
  w.matrix = reinterpret_cast<u64*>(b ? b + off : nullptr);
  off += align_up((size_t)nb * w.slots * w.words * sizeof(u64));
  w.bytes = off;
  return w;
}

extern "C" size_t eod_proposals_workspace_bytes(int total_positions, int levels, int pre_nms_topk, int batch) {
  const int slots = levels * pre_nms_topk, nb = batch > 1 ? batch : 1;
  if (slots > 8192) {
    const int ws = std::max(std::min(total_positions, slots), 1);
    const size_t narrow = carve(nullptr, ws * nb, ws, ws * nb).bytes;
    return carve_wide(nullptr, narrow, total_positions, levels, pre_nms_topk, nb).bytes;
  }
  return carve(nullptr, slots * nb, slots, slots * nb).bytes;
}

// centernet.py:603-745 with lists beyond one workgroup's LDS (see the kernels above)
static int centernet_proposals_wide(const EodProposalDesc* d, hipStream_t s) {
  const int total = d->level_off[d->levels];
  int packed = 0, max_level = 0;
  for (int l = 0; l < d->levels; ++l) {
    const int n = d->level_off[l + 1] - d->level_off[l];
    if (n <= 0 || n > EOD_SORT_MAX || d->level_w[l] <= 0 || n % d->level_w[l] != 0) return EOD_ERR_CAPACITY;
    packed += std::min(n, d->pre_nms_topk);
    max_level = std::max(max_level, n);
  }
  if (packed > WIDE_MAX_SLOTS || d->cap < d->post_nms_topk || d->cap > WIDE_MAX_CAP) return EOD_ERR_CAPACITY;
  const int nb = d->batch > 1 ? d->batch : 1;
  if (nb > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  const int ws = std::min(total, d->levels * d->pre_nms_topk);          // >= packed
  const SelWs w = carve(d->workspace, ws * nb, ws, ws * nb);
  const WideWs ww = carve_wide(d->workspace, w.bytes, total, d->levels, d->pre_nms_topk, nb);
  if (d->workspace_bytes < ww.bytes) return EOD_ERR_CAPACITY;
  CnArgs a{};
  a.batch = nb;
  a.head = d->head_out; a.head_stride = d->head_stride; a.levels = d->levels;
  for (int l = 0; l <= d->levels; ++l) a.level_off[l] = d->level_off[l];
  for (int l = 0; l < d->levels; ++l) {
    a.level_w[l] = d->level_w[l];
    a.level_stride[l] = d->level_stride[l];
    a.level_scale[l] = d->level_scale[l];
  }
  a.score_thresh = d->score_thresh; a.topk = d->pre_nms_topk; a.cand_keys = w.cand_keys; a.cand_cnt = w.cand_cnt;
  a.pk_off[0] = 0;
  for (int l = 0; l < d->levels; ++l) a.pk_off[l + 1] = a.pk_off[l] + std::min(d->level_off[l + 1] - d->level_off[l], d->pre_nms_topk);
  if (max_level <= 8192)
    hipLaunchKernelGGL(cn_level_topk_kernel<8>, dim3(d->levels, nb), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(cn_level_topk_kernel<16>, dim3(d->levels, nb), dim3(1024), 0, s, a);
  // the kernels below address the scenes' lists with the PACKED slot count as their stride
  const int words = (packed + 63) / 64;
  hipLaunchKernelGGL(cn_rank_decode_kernel, dim3((packed + 255) / 256, nb), dim3(256), 0, s, a, w.sorted_boxes, w.sorted_scores, ww.n);
  hipLaunchKernelGGL(nms_matrix_kernel, dim3((words + 3) / 4, words, nb), dim3(256), 0, s, w.sorted_boxes, ww.n, packed, words,
                     d->nms_thresh, ww.matrix);
  ScanOut o{d->out_boxes, d->out_scores, nullptr, nullptr, d->out_count, d->cap, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(nb), dim3(1024), 0, s, w.sorted_boxes, w.sorted_scores, ww.n, packed, words, ww.matrix,
                     d->post_nms_topk, o);
  return eod_launch_status();
}

extern "C" int eod_centernet_proposals(const EodProposalDesc* d, eod_stream_t stream) {
  if (!d || !d->head_out || !d->out_boxes || !d->out_scores || !d->out_count || !d->workspace) return EOD_ERR_NULL;
  if (d->levels < 1 || d->levels > 5 || d->pre_nms_topk < 1 || d->head_stride < 5) return EOD_ERR_BAD_DIMS;
  const int slots = d->levels * d->pre_nms_topk;
  if (slots > 8192) return centernet_proposals_wide(d, (hipStream_t)stream);
  for (int l = 0; l < d->levels; ++l) {
    const int n = d->level_off[l + 1] - d->level_off[l];
    if (n <= 0 || n > EOD_SORT_MAX || d->level_w[l] <= 0 || n % d->level_w[l] != 0) return EOD_ERR_CAPACITY;
  }
  if (d->cap < d->post_nms_topk || d->cap > NMS_KEPT_MAX) return EOD_ERR_CAPACITY;
  const int nb = d->batch > 1 ? d->batch : 1;
  if (nb > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  const SelWs w = carve(d->workspace, slots * nb, slots, slots * nb);
  if (d->workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  CnArgs a{};
  a.batch = nb;
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
    hipLaunchKernelGGL(cn_level_topk_kernel<8>, dim3(d->levels, nb), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL(cn_level_topk_kernel<16>, dim3(d->levels, nb), dim3(1024), 0, s, a);
  ScanOut o{d->out_boxes, d->out_scores, nullptr, nullptr, d->out_count, d->cap, nullptr, nullptr, 0, nullptr, nullptr, nullptr};
  if (a.pk_off[d->levels] <= 4096)
    hipLaunchKernelGGL(cn_merge_nms_kernel<4>, dim3(nb), dim3(1024), 0, s, a, w.sorted_boxes, w.sorted_scores, d->nms_thresh,
                       d->post_nms_topk, o);
  else
    hipLaunchKernelGGL(cn_merge_nms_kernel<8>, dim3(nb), dim3(1024), 0, s, a, w.sorted_boxes, w.sorted_scores, d->nms_thresh,
                       d->post_nms_topk, o);
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
  if (d->R_cap <= 0 || d->R_cap > DET_MAX_R || d->C1 < 2 || d->C1 - 1 > DET_MAX_C || d->topk <= 0 || d->topk > NMS_KEPT_MAX)
    return EOD_ERR_BAD_DIMS;
  const int slots = d->R_cap * (d->C1 - 1);
  if (slots > 8192) return EOD_ERR_CAPACITY;
  const SelWs w = carve(d->workspace, slots, slots, 0);
  if (d->workspace_bytes < w.bytes) return EOD_ERR_CAPACITY;
  hipStream_t s = (hipStream_t)stream;
  if (d->out_unique_rows && (!d->out_unique_count || d->unique_cap <= 0 || d->R_cap > NMS_KEPT_MAX)) return EOD_ERR_BAD_DIMS;
  if (d->out_rep_of && (!d->out_rep_list || !d->out_rep_count || d->R_cap > NMS_KEPT_MAX)) return EOD_ERR_BAD_DIMS;
  ScanOut o{d->out_boxes, d->out_scores, d->out_classes, d->out_rows, d->out_count, d->topk, d->out_unique_rows, d->out_unique_count,
            d->unique_cap, d->out_rep_of, d->out_rep_list, d->out_rep_count};
  if (d->batch > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(det_select_kernel, dim3(d->batch > 1 ? d->batch : 1), dim3(1024), 0, s, d->boxes, d->scores, d->count, d->R_cap, d->C1, d->img_w, d->img_h,
                     d->score_thresh, d->nms_thresh, d->topk, o);
  return eod_launch_status();
}

extern "C" int eod_concat_lists(const int32_t* lists, const int32_t* counts, int cap_in, int id_stride, int batch, int32_t* out,
                                int32_t* out_count, eod_stream_t stream) {
  if (!lists || !counts || !out || !out_count) return EOD_ERR_NULL;
  if (cap_in <= 0 || id_stride <= 0 || batch < 1 || batch > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(concat_lists_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lists, counts, cap_in, id_stride, batch, out, out_count);
  return eod_launch_status();
}
