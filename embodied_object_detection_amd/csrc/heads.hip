// Small per-ROI kernels of the cascade heads and the post-processing:
//   zero-shot classifier tail, Box2BoxTransform, cascade score fusion, memory-side CLIP scoring,
//   detector_postprocess compaction and paste_masks_in_image.
// All rows are bounded by a device-side count; everything is latency-trivial next to the dense kernels.
#include "eod_common.h"
#include "../../include/eod_hip.h"

namespace {

__device__ __forceinline__ int dyn_rows(const int* count, int cap) {
  if (!count) return cap;
  const int c = *count;
  return c < cap ? c : cap;
}

// B ROI lists back to back (EOD batch convention): row r of R_cap * batch holds work iff (r mod R_cap) < count[r / R_cap]
__device__ __forceinline__ bool row_has_work(const int* count, int R_cap, int row) {
  const int s = row / R_cap;
  const int c = count ? count[s] : R_cap;
  return row - s * R_cap < (c < R_cap ? c : R_cap);
}

// one wave per row, 4 rows per workgroup.  feat [R,D] (D = 512 -> 8 per lane).  The class matrix zs [D,C1] is staged once per
// workgroup in LDS, transposed to [c][k], so that every class costs two conflict-free ds_read_b128 per lane instead of eight
// strided global loads behind a dependent chain (80 / 37 us -> a few us on the cascade's critical path).  Same arithmetic and
// summation order as before: per lane 8 products in channel order, then the wave butterfly.
constexpr int ZS_MAX_C = 24;

// zs [512][C1] (global, 16-byte aligned) -> zt [C1][512] (LDS) by the 256 threads of a workgroup.  All of a thread's 16-byte loads are
// issued before the first LDS write: the first version's loop (load one float, write it, 42 times) was a chain of 42 dependent L2
// round trips -- 20 of the kernel's 29 us.
__device__ __forceinline__ void stage_class_matrix(const float* __restrict__ zs, float* zt, int n, int C1) {
  constexpr int PER = ZS_MAX_C * 512 / 4 / 256;           // 12 float4 per thread cover the largest class matrix
  const int n4 = n >> 2;                                   // n = 512 C1 is a multiple of 4
  f32x4 buf[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i4 = threadIdx.x + 256 * j;
    if (i4 < n4) buf[j] = *reinterpret_cast<const f32x4*>(zs + 4 * i4);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i4 = threadIdx.x + 256 * j;
    if (i4 < n4) {
      int k = (4 * i4) / C1, c = 4 * i4 - k * C1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        zt[c * 512 + k] = buf[j][e];
        if (++c == C1) {
          c = 0;
          ++k;
        }
      }
    }
  }
}
// The rest of a cascade stage's predictor in the same launch (eod_cascade_stage_tail): bbox_pred.2 (Linear 1024 -> 4 on the ReLU'd
// bbox_pred.0 output, detic_fast_rcnn.py:109-116) and Box2BoxTransform.apply_deltas onto the stage's boxes (detic_roi_heads.py:121-122,
// 314).  One wave per row like the classifier: lane l owns inputs 16 l .. 16 l + 15 of the row, four butterfly sums.
struct BoxTail {
  const float* hb;        // [R, K] bbox_pred.0 output; null = no tail
  const float* w2;        // [4][ld] bbox_pred.2 weight rows
  const float* b2;        // [4]
  int K, ld;
  const float* boxes_in;  // [R,4]
  float* boxes_out;       // [R,4]
  float* deltas_out;      // [R,4] or null
  float wx, wy, ww, wh;
  int clip;
  float img_w, img_h;
};

__device__ __forceinline__ void box_tail_row(const BoxTail& t, int row, int lane) {
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = lane * 16; k0 < t.K; k0 += 64 * 16) {
    f32x4 h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = *reinterpret_cast<const f32x4*>(t.hb + (size_t)row * t.K + k0 + 4 * q);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(t.w2 + (size_t)o * t.ld + k0 + 4 * q);
        acc[o] += h[q].x * w.x; acc[o] += h[q].y * w.y; acc[o] += h[q].z * w.z; acc[o] += h[q].w * w.w;
      }
    }
  }
  float d[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) d[o] = wave_reduce_sum(acc[o]) + t.b2[o];
  if (lane == 0) {
    if (t.deltas_out) {
      t.deltas_out[row * 4 + 0] = d[0]; t.deltas_out[row * 4 + 1] = d[1]; t.deltas_out[row * 4 + 2] = d[2]; t.deltas_out[row * 4 + 3] = d[3];
    }
    const float x1 = t.boxes_in[row * 4 + 0], y1 = t.boxes_in[row * 4 + 1], x2 = t.boxes_in[row * 4 + 2], y2 = t.boxes_in[row * 4 + 3];
    float ox1, oy1, ox2, oy2;
    eod_apply_deltas_one(d, x1, y1, x2, y2, t.wx, t.wy, t.ww, t.wh, t.clip, t.img_w, t.img_h, ox1, oy1, ox2, oy2);
    t.boxes_out[row * 4 + 0] = ox1; t.boxes_out[row * 4 + 1] = oy1; t.boxes_out[row * 4 + 2] = ox2; t.boxes_out[row * 4 + 3] = oy2;
  }
}

__global__ __launch_bounds__(256) void zs_classify_kernel(const float* __restrict__ feat, const float* __restrict__ zs,
                                                           float* __restrict__ prob_acc, int accumulate, float* __restrict__ featn_out,
                                                           const int* __restrict__ count, int R_cap, int D, int C1, float temp,
                                                           const float* __restrict__ zs_mem, const float* __restrict__ prop_scores,
                                                           float* __restrict__ mem_scores, float final_inv_stages, int batch, BoxTail tail) {
  EOD_CHAIN_PRIO();
  __shared__ __attribute__((aligned(16))) float zt[ZS_MAX_C * 512];
  // R_cap % 4 == 0 whenever the rows are a batch of lists: a workgroup's four rows belong to one list
  if (!row_has_work(count, R_cap, (int)(blockIdx.x * 4))) return;    // whole workgroup beyond its list's count
  stage_class_matrix(zs, zt, D * C1, C1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool active = row < R_cap * batch && row_has_work(count, R_cap, row);
  if (active && tail.hb) box_tail_row(tail, row, lane);
  float x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = 0.f;
  if (active) {
    float ss = 0.f;
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(feat + (size_t)row * D + lane * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(feat + (size_t)row * D + lane * 8 + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) ss += x[q] * x[q];
    ss = wave_reduce_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = temp * (x[q] / denom);
    if (featn_out) {
      *reinterpret_cast<f32x4*>(featn_out + (size_t)row * D + lane * 8) = f32x4{x[0], x[1], x[2], x[3]};
      *reinterpret_cast<f32x4*>(featn_out + (size_t)row * D + lane * 8 + 4) = f32x4{x[4], x[5], x[6], x[7]};
    }
    const float ps = (final_inv_stages > 0.f && prop_scores) ? prop_scores[row] : 0.f;
    // lane c owns class c: the row's accumulated probabilities are read with ONE coalesced load before the loop and written with one
    // store after it (lane 0 reading and writing them one by one inside the loop was a chain of C1 dependent global round trips: 29 us
    // for a launch of 5 MFLOP).  The butterfly leaves the identical sum in every lane, so lane c's logit is lane 0's.
    float* o = prob_acc + (size_t)row * C1 + lane;
    const float old = (accumulate && lane < C1) ? *o : 0.f;
    float logit = 0.f;
    for (int c = 0; c < C1; ++c) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(zt + c * 512 + lane * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(zt + c * 512 + lane * 8 + 4);
      float s = 0.f;
      s += x[0] * w0.x; s += x[1] * w0.y; s += x[2] * w0.z; s += x[3] * w0.w;
      s += x[4] * w1.x; s += x[5] * w1.y; s += x[6] * w1.z; s += x[7] * w1.w;
      s = wave_reduce_sum(s);
      if (lane == c) logit = s;
    }
    if (lane < C1) {
      const float p = eod_sigmoid_precise(logit);
      float v = accumulate ? (old + p) : p;
      if (final_inv_stages > 0.f) v = sqrtf(v * final_inv_stages * ps);         // cascade score fusion (detic_roi_heads.py:164-173)
      *o = v;
    }
  }
  if (!zs_mem) return;
  // memory-side CLIP re-score (custom_rcnn.py:838-861) with the meta-architecture's own class matrix: same staging, same
  // per-lane channel order and butterfly as eod_memory_scores on feat_norm_out
  __syncthreads();
  stage_class_matrix(zs_mem, zt, D * C1, C1);
  __syncthreads();
  if (!active) return;
  const float p = prop_scores[row];
  float logit = 0.f;
  for (int c = 0; c < C1; ++c) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(zt + c * 512 + lane * 8);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(zt + c * 512 + lane * 8 + 4);
    float s = 0.f;
    s += x[0] * w0.x; s += x[1] * w0.y; s += x[2] * w0.z; s += x[3] * w0.w;
    s += x[4] * w1.x; s += x[5] * w1.y; s += x[6] * w1.z; s += x[7] * w1.w;
    s = wave_reduce_sum(s);
    if (lane == c) logit = s;
  }
  if (lane < C1) mem_scores[(size_t)row * C1 + lane] = (p < 1.0f) ? sqrtf(eod_sigmoid_precise(logit) * p) : 0.0f;
}

__global__ void apply_deltas_kernel(const float* __restrict__ deltas, int ld, const float* __restrict__ boxes, float* __restrict__ out,
                                    const int* __restrict__ count, int R_cap, float wx, float wy, float ww, float wh, int clip,
                                    float img_w, float img_h, int batch) {
  EOD_CHAIN_PRIO();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R_cap * batch || !row_has_work(count, R_cap, r)) return;
  const float x1 = boxes[r * 4 + 0], y1 = boxes[r * 4 + 1], x2 = boxes[r * 4 + 2], y2 = boxes[r * 4 + 3];
  float ox1, oy1, ox2, oy2;
  eod_apply_deltas_one(deltas + (size_t)r * ld, x1, y1, x2, y2, wx, wy, ww, wh, clip, img_w, img_h, ox1, oy1, ox2, oy2);
  out[r * 4 + 0] = ox1;
  out[r * 4 + 1] = oy1;
  out[r * 4 + 2] = ox2;
  out[r * 4 + 3] = oy2;
}

__global__ void cascade_scores_kernel(float* __restrict__ prob_acc, const float* __restrict__ prop_scores, const int* __restrict__ count,
                                      int R_cap, int C1, float inv_stages) {
  EOD_CHAIN_PRIO();
  const int R = dyn_rows(count, R_cap);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * C1) return;
  const int r = i / C1;
  prob_acc[i] = sqrtf(prob_acc[i] * inv_stages * prop_scores[r]);
}

// one wave per row: scores[r][c] = sqrt(sigmoid(featn[r] . zs[:,c]) * ps[r]); rows with ps >= 1 are excluded (score 0)
__global__ __launch_bounds__(256) void memory_scores_kernel(const float* __restrict__ featn, const float* __restrict__ zs,
                                                             const float* __restrict__ ps, float* __restrict__ scores,
                                                             const int* __restrict__ count, int R_cap, int D, int C1) {
  EOD_CHAIN_PRIO();
  const int R = dyn_rows(count, R_cap);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int per = D / 64;
  float x[8];
  for (int q = 0; q < per; ++q) x[q] = featn[(size_t)row * D + lane * per + q];
  const float p = ps[row];
  for (int c = 0; c < C1; ++c) {
    float s = 0.f;
    for (int q = 0; q < per; ++q) s += x[q] * zs[(size_t)(lane * per + q) * C1 + c];
    s = wave_reduce_sum(s);
    if (lane == 0) scores[(size_t)row * C1 + c] = (p < 1.0f) ? sqrtf(eod_sigmoid_precise(s) * p) : 0.0f;
  }
}

// detector_postprocess: scale, clip, drop empty boxes (single block)
__global__ __launch_bounds__(512) void postprocess_kernel(const float* boxes, const float* scores, const int* classes, const int* count,
                                                           int cap, float sx, float sy, float out_w, float out_h, float* ob, float* os,
                                                           int* oc, int* osrc, int* ocount, const int* remap) {
  __shared__ int sh_keep[512];
  __shared__ int sh_pos[512];
  {
    const int b = blockIdx.x;         // scene of a batch: every buffer is `batch` single-scene buffers back to back
    boxes += (size_t)b * cap * 4; scores += (size_t)b * cap; classes += (size_t)b * cap;
    if (count) count += b;
    ob += (size_t)b * cap * 4; os += (size_t)b * cap; oc += (size_t)b * cap; osrc += (size_t)b * cap; ocount += b;
    if (remap) remap += (size_t)b * cap;
  }
  const int D = dyn_rows(count, cap);
  const int t = threadIdx.x;
  float b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  int keep = 0;
  if (t < D) {
    b0 = fminf(fmaxf(boxes[t * 4 + 0] * sx, 0.f), out_w);
    b1 = fminf(fmaxf(boxes[t * 4 + 1] * sy, 0.f), out_h);
    b2 = fminf(fmaxf(boxes[t * 4 + 2] * sx, 0.f), out_w);
    b3 = fminf(fmaxf(boxes[t * 4 + 3] * sy, 0.f), out_h);
    keep = ((b2 - b0) > 0.f && (b3 - b1) > 0.f) ? 1 : 0;
  }
  sh_keep[t] = keep;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 512; ++i) {
      sh_pos[i] = run;
      run += sh_keep[i];
    }
    *ocount = run;
  }
  __syncthreads();
  if (keep) {
    const int q = sh_pos[t];
    ob[q * 4 + 0] = b0;
    ob[q * 4 + 1] = b1;
    ob[q * 4 + 2] = b2;
    ob[q * 4 + 3] = b3;
    os[q] = scores[t];
    oc[q] = classes[t];
    osrc[q] = remap ? remap[t] : t;
  }
}

// grid (pixel tiles, K).  Exact op order of F.grid_sample(bilinear, zeros, align_corners=False) on the
// normalised grid detectron2 builds in _do_paste_mask.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* prob, const float* boxes, const int* rows, const int* count,
                                                           int K_cap, int H, int W, float thr, uint8_t* out, int prob_units) {
  {
    const int b = blockIdx.z;         // scene of a batch
    prob += (size_t)b * prob_units * 784; boxes += (size_t)b * K_cap * 4;
    if (rows) rows += (size_t)b * K_cap;
    if (count) count += b;
    out += (size_t)b * K_cap * H * W;
  }
  const int K = dyn_rows(count, K_cap);
  const int k = blockIdx.y;
  if (k >= K) return;
  __shared__ float m[28 * 28];
  const int src = rows ? rows[k] : k;
  for (int i = threadIdx.x; i < 784; i += blockDim.x) m[i] = prob[(size_t)src * 784 + i];
  __syncthreads();
  const float x0 = boxes[k * 4 + 0], y0 = boxes[k * 4 + 1], x1 = boxes[k * 4 + 2], y1 = boxes[k * 4 + 3];
  const int total = H * W;
  // each thread produces 16 consecutive pixels of one row and writes them with ONE 16-byte store (W % 16 == 0)
  const int groups = total >> 4;
  // a sample further than one mask pixel outside the box is exactly 0: whole 16-pixel runs out there skip the sampling
  const float mx = fabsf(x1 - x0) * (1.0f / 14.0f), my = fabsf(y1 - y0) * (1.0f / 14.0f);
  const bool degenerate = !(x1 > x0) || !(y1 > y0);
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += gridDim.x * blockDim.x) {
    const int p0 = g << 4;
    const int y = p0 / W, xb = p0 - y * W;
    unsigned wv[4] = {0u, 0u, 0u, 0u};
    const float fy0 = (float)y + 0.5f;
    const bool row_out = !degenerate && (fy0 < y0 - my || fy0 > y1 + my || (float)xb + 16.0f < x0 - mx || (float)xb > x1 + mx);
    if (!row_out) {
      const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.0f - 1.0f;
      const float iy = ((gy + 1.0f) * 28.0f - 1.0f) / 2.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int x = xb + j;
        const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.0f - 1.0f;
        const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
        float v = 0.f;
        if (ix > -1.0f && ix < 28.0f && iy > -1.0f && iy < 28.0f) {  // also false for NaN
          const float fx = floorf(ix), fy = floorf(iy);
          const int xw = (int)fx, yn = (int)fy;
          const int xe = xw + 1, ys = yn + 1;
          const float nw = ((float)xe - ix) * ((float)ys - iy);
          const float ne = (ix - (float)xw) * ((float)ys - iy);
          const float sw = ((float)xe - ix) * (iy - (float)yn);
          const float se = (ix - (float)xw) * (iy - (float)yn);
          const bool xwv = (unsigned)xw < 28u, xev = (unsigned)xe < 28u, ynv = (unsigned)yn < 28u, ysv = (unsigned)ys < 28u;
          if (xwv && ynv) v += m[yn * 28 + xw] * nw;
          if (xev && ynv) v += m[yn * 28 + xe] * ne;
          if (xwv && ysv) v += m[ys * 28 + xw] * sw;
          if (xev && ysv) v += m[ys * 28 + xe] * se;
        }
        if (v >= thr) wv[j >> 2] |= 1u << ((j & 3) * 8);
      }
    }
    uint4 pk;
    pk.x = wv[0]; pk.y = wv[1]; pk.z = wv[2]; pk.w = wv[3];
    *reinterpret_cast<uint4*>(out + (size_t)k * total + p0) = pk;
  }
}

}  // namespace

extern "C" int eod_zs_classify(const float* feat, const float* zs, float* prob_acc, int accumulate, float* feat_norm_out,
                               const int32_t* count, int R_cap, int D, int C1, float temp, const float* zs_mem, const float* prop_scores,
                               float* mem_scores_out, float final_inv_stages, int batch, eod_stream_t stream) {
  if (!feat || !zs || !prob_acc) return EOD_ERR_NULL;
  if (zs_mem && (!prop_scores || !mem_scores_out)) return EOD_ERR_NULL;
  if (final_inv_stages > 0.f && !prop_scores) return EOD_ERR_NULL;
  // the class matrix is staged in 48 KB of LDS: vocabularies of more than ZS_MAX_C - 1 classes (RESET_CLS_TESTS) are refused here
  // with a capacity error that ops.zs_classify words out
  if (D != 512 || C1 < 2 || R_cap <= 0) return EOD_ERR_BAD_DIMS;
  if (C1 > ZS_MAX_C) return EOD_ERR_CAPACITY;
  if (!eod_aligned16(feat) || (feat_norm_out && !eod_aligned16(feat_norm_out)) || !eod_aligned16(zs) || (zs_mem && !eod_aligned16(zs_mem)))
    return EOD_ERR_ALIGN;
  if (batch > 1 && (batch > EOD_MAX_BATCH || R_cap % 4 != 0)) return EOD_ERR_BAD_DIMS;
  const int nb = batch > 1 ? batch : 1;
  hipLaunchKernelGGL(zs_classify_kernel, dim3((R_cap * nb + 3) / 4), dim3(256), 0, (hipStream_t)stream, feat, zs, prob_acc, accumulate,
                     feat_norm_out, count, R_cap, D, C1, temp, zs_mem, prop_scores, mem_scores_out, final_inv_stages, nb, BoxTail{});
  return eod_launch_status();
}

extern "C" int eod_cascade_stage_tail(const EodStageTailDesc* d, eod_stream_t stream) {
  if (!d || !d->feat || !d->zs || !d->prob_acc || !d->hb || !d->w2 || !d->b2 || !d->boxes_in || !d->boxes_out) return EOD_ERR_NULL;
  if (d->zs_mem && (!d->prop_scores || !d->mem_scores_out)) return EOD_ERR_NULL;
  if (d->final_inv_stages > 0.f && !d->prop_scores) return EOD_ERR_NULL;
  if (d->D != 512 || d->C1 < 2 || d->R_cap <= 0 || d->hb_dim <= 0 || d->hb_dim % 16 != 0 || d->w2_ld < d->hb_dim || d->w2_ld % 4 != 0)
    return EOD_ERR_BAD_DIMS;
  if (d->C1 > ZS_MAX_C) return EOD_ERR_CAPACITY;
  if (!eod_aligned16(d->feat) || (d->feat_norm_out && !eod_aligned16(d->feat_norm_out)) || !eod_aligned16(d->zs) ||
      (d->zs_mem && !eod_aligned16(d->zs_mem)) || !eod_aligned16(d->hb) || !eod_aligned16(d->w2))
    return EOD_ERR_ALIGN;
  if (d->batch > 1 && (d->batch > EOD_MAX_BATCH || d->R_cap % 4 != 0)) return EOD_ERR_BAD_DIMS;
  const int nb = d->batch > 1 ? d->batch : 1;
  BoxTail t{d->hb, d->w2, d->b2, d->hb_dim, d->w2_ld, d->boxes_in, d->boxes_out, d->deltas_out, d->wx, d->wy, d->ww, d->wh,
            d->clip, d->img_w, d->img_h};
  hipLaunchKernelGGL(zs_classify_kernel, dim3((d->R_cap * nb + 3) / 4), dim3(256), 0, (hipStream_t)stream, d->feat, d->zs, d->prob_acc,
                     d->accumulate, d->feat_norm_out, d->count, d->R_cap, d->D, d->C1, d->temp, d->zs_mem, d->prop_scores,
                     d->mem_scores_out, d->final_inv_stages, nb, t);
  return eod_launch_status();
}

extern "C" int eod_apply_deltas(const float* deltas, int ld, const float* boxes, float* out, const int32_t* count, int R_cap, float wx,
                                float wy, float ww, float wh, int clip, float img_w, float img_h, int batch, eod_stream_t stream) {
  if (!deltas || !boxes || !out) return EOD_ERR_NULL;
  if (ld < 4 || R_cap <= 0 || batch > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  const int nb = batch > 1 ? batch : 1;
  hipLaunchKernelGGL(apply_deltas_kernel, dim3((R_cap * nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, deltas, ld, boxes, out, count,
                     R_cap, wx, wy, ww, wh, clip, img_w, img_h, nb);
  return eod_launch_status();
}

extern "C" int eod_cascade_scores(float* prob_acc, const float* prop_scores, const int32_t* count, int R_cap, int C1, float inv_stages,
                                  eod_stream_t stream) {
  if (!prob_acc || !prop_scores) return EOD_ERR_NULL;
  if (R_cap <= 0 || C1 <= 0) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(cascade_scores_kernel, dim3((R_cap * C1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob_acc, prop_scores,
                     count, R_cap, C1, inv_stages);
  return eod_launch_status();
}

extern "C" int eod_memory_scores(const float* featn, const float* zs, const float* prop_scores, float* scores, const int32_t* count,
                                 int R_cap, int D, int C1, eod_stream_t stream) {
  if (!featn || !zs || !prop_scores || !scores) return EOD_ERR_NULL;
  if (D != 512 || C1 < 2 || R_cap <= 0) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(memory_scores_kernel, dim3((R_cap + 3) / 4), dim3(256), 0, (hipStream_t)stream, featn, zs, prop_scores, scores,
                     count, R_cap, D, C1);
  return eod_launch_status();
}

extern "C" int eod_detector_postprocess(const float* boxes, const float* scores, const int32_t* classes, const int32_t* count, int cap,
                                        float sx, float sy, float out_w, float out_h, float* out_boxes, float* out_scores,
                                        int32_t* out_classes, int32_t* out_src, int32_t* out_count, const int32_t* remap,
                                        int batch, eod_stream_t stream) {
  if (!boxes || !scores || !classes || !out_boxes || !out_scores || !out_classes || !out_src || !out_count) return EOD_ERR_NULL;
  if (cap <= 0 || cap > 512) return EOD_ERR_CAPACITY;
  if (batch > EOD_MAX_BATCH) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(postprocess_kernel, dim3(batch > 1 ? batch : 1), dim3(512), 0, (hipStream_t)stream, boxes, scores, classes, count, cap, sx, sy, out_w,
                     out_h, out_boxes, out_scores, out_classes, out_src, out_count, remap);
  return eod_launch_status();
}

extern "C" int eod_paste_masks(const float* prob, const float* boxes, const int32_t* rows, const int32_t* count, int K_cap, int H, int W,
                               float threshold, uint8_t* out, int batch, int prob_units, eod_stream_t stream) {
  if (!prob || !boxes || !out) return EOD_ERR_NULL;
  if (K_cap <= 0 || H <= 0 || W <= 0 || (W & 15) || batch > EOD_MAX_BATCH || (batch > 1 && prob_units <= 0)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(out)) return EOD_ERR_ALIGN;
  int tiles = (H * W / 16 + 256 * 2 - 1) / (256 * 2);
  hipLaunchKernelGGL(paste_masks_kernel, dim3(tiles, K_cap, batch > 1 ? batch : 1), dim3(256), 0, (hipStream_t)stream, prob, boxes, rows,
                     count, K_cap, H, W, threshold, out, batch > 1 ? prob_units : 0);
  return eod_launch_status();
}
