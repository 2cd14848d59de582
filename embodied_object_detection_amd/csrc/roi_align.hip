// ROIAlignV2 (aligned=True, sampling_ratio=0) over the three FPN levels p3..p5, NHWC fp32.
// Replaces detectron2 ROIPooler + torchvision roi_align as called by the cascade box heads
// (Detic/detic/modeling/roi_heads/detic_roi_heads.py:332) and the mask pooler (:265); semantics per
// SURVEY.md Appendix A7.
//
// One wave per output bin: the 4 bilinear taps of every sample point are wave-uniform scalars, each lane
// owns 4 consecutive channels (C = 256 -> one 1 KiB coalesced read per tap), samples are accumulated in the
// upstream order (iy outer, ix inner) and divided by the sample count.  HBM/L2-bound gather.
#include <cstdlib>
#include "eod_common.h"
#include "../../include/eod_hip.h"

namespace {

struct RoiArgs {
  const float* feat[3];
  int h[3], w[3];
  float scale[3];
  int C;
  const float* boxes;
  const int* box_rows;  // optional gather: ROI r pools box box_rows[r]
  const int* count;
  int R_cap;
  int S;
  float* out;
  int batch, boxes_per_image;   // batch > 1: feat[l] holds `batch` images; box j belongs to image j / boxes_per_image
  // optional refine-on-load (EodBoxRefine): the ROI's box is apply_deltas(boxes[br], deltas[br]); the wave of bin 0 also stores it
  const float* deltas;
  int ld, clip;
  float wx, wy, ww, wh, img_w, img_h;
  float* boxes_out;
  FastDiv div_bins, div_s;   // wave id -> (roi, bin) -> (ph, pw) without integer division sequences
};

__global__ __launch_bounds__(256) void roi_align_kernel(RoiArgs p) {
  EOD_CHAIN_PRIO();
  int R = p.R_cap;
  const bool segmented = p.batch > 1 && !p.box_rows;      // one ROI list per image, a count per list
  if (p.count && !segmented) {
    const int c = *p.count;
    R = c < R ? c : R;
  }
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int bins = p.S * p.S;
  const int total = R * bins;                       // < 2^31: checked on the host
  for (int wid = blockIdx.x * wpb + (threadIdx.x >> 6); wid < total; wid += gridDim.x * wpb) {
    const int r = (int)fdiv((unsigned)wid, p.div_bins);
    const int b = wid - r * bins;
    const int ph = (int)fdiv((unsigned)b, p.div_s), pw = b - ph * p.S;
    const int br = p.box_rows ? p.box_rows[r] : r;
    int img = 0;
    if (p.batch > 1) {
      img = br / p.boxes_per_image;
      if (segmented && p.count && br - img * p.boxes_per_image >= p.count[img]) continue;
    }
    float bx1 = p.boxes[br * 4 + 0], by1 = p.boxes[br * 4 + 1], bx2 = p.boxes[br * 4 + 2], by2 = p.boxes[br * 4 + 3];
    if (p.deltas) {
      float ox1, oy1, ox2, oy2;
      eod_apply_deltas_one(p.deltas + (size_t)br * p.ld, bx1, by1, bx2, by2, p.wx, p.wy, p.ww, p.wh, p.clip, p.img_w, p.img_h, ox1, oy1, ox2,
                           oy2);
      bx1 = ox1; by1 = oy1; bx2 = ox2; by2 = oy2;
      if (b == 0 && lane == 0) {
        p.boxes_out[br * 4 + 0] = bx1; p.boxes_out[br * 4 + 1] = by1; p.boxes_out[br * 4 + 2] = bx2; p.boxes_out[br * 4 + 3] = by2;
      }
    }
    // assign_boxes_to_levels: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [3,5]
    const float area = (bx2 - bx1) * (by2 - by1);
    float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    lv = fminf(fmaxf(lv, 3.0f), 5.0f);
    const int l = (int)lv - 3;
    const int H = p.h[l], W = p.w[l];
    const float* feat = p.feat[l] + (size_t)img * H * W * p.C;
    const float sc = p.scale[l];
    const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f, x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
    const float roi_w = x2 - x1, roi_h = y2 - y1;
    const float bin_h = roi_h / (float)p.S, bin_w = roi_w / (float)p.S;
    const float ghf = ceilf(roi_h / (float)p.S), gwf = ceilf(roi_w / (float)p.S);
    // non-finite or absurd boxes (exploded deltas) produce an all-zero bin instead of an unbounded loop
    const bool sane = (ghf == ghf) && (gwf == gwf) && ghf < 1.0e6f && gwf < 1.0e6f;
    const int gh = sane ? (int)ghf : 0;
    const int gw = sane ? (int)gwf : 0;
    const float cnt = fmaxf((float)gh * (float)gw, 1.0f);
    // samples outside [-1,H]x[-1,W] contribute 0: only walk the index range that can fall inside (+-1 margin)
    int iy_lo = 0, iy_hi = gh - 1, ix_lo = 0, ix_hi = gw - 1;
    if (gh > 0) {
      const float ys = y1 + (float)ph * bin_h, st = bin_h / (float)gh;
      iy_lo = max(0, (int)floorf((-1.0f - ys) / st - 0.5f) - 1);
      iy_hi = min(gh - 1, (int)ceilf(((float)H - ys) / st - 0.5f) + 1);
    }
    if (gw > 0) {
      const float xs = x1 + (float)pw * bin_w, st = bin_w / (float)gw;
      ix_lo = max(0, (int)floorf((-1.0f - xs) / st - 0.5f) - 1);
      ix_hi = min(gw - 1, (int)ceilf(((float)W - xs) / st - 0.5f) + 1);
    }
    for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
          float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
          float yy = y;
          if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
          if (yy <= 0.f) yy = 0.f;
          if (x <= 0.f) x = 0.f;
          int y_low = (int)yy, x_low = (int)x;
          int y_high, x_high;
          if (y_low >= H - 1) {
            y_high = y_low = H - 1;
            yy = (float)y_low;
          } else {
            y_high = y_low + 1;
          }
          if (x_low >= W - 1) {
            x_high = x_low = W - 1;
            x = (float)x_low;
          } else {
            x_high = x_low + 1;
          }
          const float ly = yy - (float)y_low, lx = x - (float)x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(feat + ((size_t)y_low * W + x_low) * p.C + c0);
          const f32x4 v2 = *reinterpret_cast<const f32x4*>(feat + ((size_t)y_low * W + x_high) * p.C + c0);
          const f32x4 v3 = *reinterpret_cast<const f32x4*>(feat + ((size_t)y_high * W + x_low) * p.C + c0);
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(feat + ((size_t)y_high * W + x_high) * p.C + c0);
          acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      acc.x /= cnt;
      acc.y /= cnt;
      acc.z /= cnt;
      acc.w /= cnt;
      *reinterpret_cast<f32x4*>(p.out + ((size_t)r * bins + b) * p.C + c0) = acc;
    }
  }
}

// Backward of the pooling above (training slices; torchvision's roi_align backward as detectron2's ROIPooler reaches it): the
// gradient of an output bin, divided by the bin's sample count, is spread over the four bilinear taps of every sample point with
// hardware fp32 atomics into the gradient of the ROI's pyramid level (dfeat is accumulated into: the three cascade stages and the
// mask pooler add into the same buffers).  Same wave / lane mapping and the same sample walk as the forward; single image.
__global__ __launch_bounds__(256) void roi_align_backward_kernel(RoiArgs p, float* d3, float* d4, float* d5, const float* __restrict__ g) {
  int R = p.R_cap;
  if (p.count) {
    const int c = *p.count;
    R = c < R ? c : R;
  }
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int bins = p.S * p.S;
  const int total = R * bins;
  for (int wid = blockIdx.x * wpb + (threadIdx.x >> 6); wid < total; wid += gridDim.x * wpb) {
    const int r = (int)fdiv((unsigned)wid, p.div_bins);
    const int b = wid - r * bins;
    const int ph = (int)fdiv((unsigned)b, p.div_s), pw = b - ph * p.S;
    const float bx1 = p.boxes[r * 4 + 0], by1 = p.boxes[r * 4 + 1], bx2 = p.boxes[r * 4 + 2], by2 = p.boxes[r * 4 + 3];
    const float area = (bx2 - bx1) * (by2 - by1);
    float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    lv = fminf(fmaxf(lv, 3.0f), 5.0f);
    const int l = (int)lv - 3;
    const int H = p.h[l], W = p.w[l];
    float* dfeat = l == 0 ? d3 : (l == 1 ? d4 : d5);
    const float sc = p.scale[l];
    const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f, x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
    const float roi_w = x2 - x1, roi_h = y2 - y1;
    const float bin_h = roi_h / (float)p.S, bin_w = roi_w / (float)p.S;
    const float ghf = ceilf(roi_h / (float)p.S), gwf = ceilf(roi_w / (float)p.S);
    const bool sane = (ghf == ghf) && (gwf == gwf) && ghf < 1.0e6f && gwf < 1.0e6f;
    const int gh = sane ? (int)ghf : 0;
    const int gw = sane ? (int)gwf : 0;
    const float cnt = fmaxf((float)gh * (float)gw, 1.0f);
    int iy_lo = 0, iy_hi = gh - 1, ix_lo = 0, ix_hi = gw - 1;
    if (gh > 0) {
      const float ys = y1 + (float)ph * bin_h, st = bin_h / (float)gh;
      iy_lo = max(0, (int)floorf((-1.0f - ys) / st - 0.5f) - 1);
      iy_hi = min(gh - 1, (int)ceilf(((float)H - ys) / st - 0.5f) + 1);
    }
    if (gw > 0) {
      const float xs = x1 + (float)pw * bin_w, st = bin_w / (float)gw;
      ix_lo = max(0, (int)floorf((-1.0f - xs) / st - 0.5f) - 1);
      ix_hi = min(gw - 1, (int)ceilf(((float)W - xs) / st - 0.5f) + 1);
    }
    for (int c0 = lane * 4; c0 < p.C; c0 += 256) {
      f32x4 gv = *reinterpret_cast<const f32x4*>(g + ((size_t)r * bins + b) * p.C + c0);
      gv.x /= cnt; gv.y /= cnt; gv.z /= cnt; gv.w /= cnt;
      for (int iy = iy_lo; iy <= iy_hi; ++iy) {
        const float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
        for (int ix = ix_lo; ix <= ix_hi; ++ix) {
          float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
          float yy = y;
          if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
          if (yy <= 0.f) yy = 0.f;
          if (x <= 0.f) x = 0.f;
          int y_low = (int)yy, x_low = (int)x;
          int y_high, x_high;
          if (y_low >= H - 1) {
            y_high = y_low = H - 1;
            yy = (float)y_low;
          } else {
            y_high = y_low + 1;
          }
          if (x_low >= W - 1) {
            x_high = x_low = W - 1;
            x = (float)x_low;
          } else {
            x_high = x_low + 1;
          }
          const float ly = yy - (float)y_low, lx = x - (float)x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float wt[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
          float* dst[4] = {dfeat + ((size_t)y_low * W + x_low) * p.C + c0, dfeat + ((size_t)y_low * W + x_high) * p.C + c0,
                           dfeat + ((size_t)y_high * W + x_low) * p.C + c0, dfeat + ((size_t)y_high * W + x_high) * p.C + c0};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(dst[k] + 0, wt[k] * gv.x);
            unsafeAtomicAdd(dst[k] + 1, wt[k] * gv.y);
            unsafeAtomicAdd(dst[k] + 2, wt[k] * gv.z);
            unsafeAtomicAdd(dst[k] + 3, wt[k] * gv.w);
          }
        }
      }
    }
  }
}

// The same gradient in "footprint" form: one wave per (ROI, feature row of the ROI's footprint).  The bilinear weights of a sample
// are separable, so the gradient an ROI leaves in cell (cy, cx) is  sum_{ph, pw} Ay[cy][ph] Ax[cx][pw] g[r, ph, pw, :] / count  with
// Ay[cy][ph] = the summed y-weights of bin row ph's valid samples on feature row cy (Ax alike).  Lane i computes Ay[.][i] / Ax[.][i],
// the non-zero bins are walked through a ballot mask, and a cell takes ONE atomic add per channel and ROI instead of one per sample
// tap: 49 bins x gh gw samples x 4 taps become the (rows x columns) of the footprint -- about 10x fewer atomics on the cascade's
// 7x7 poolers (the sample-tap form spent 2.6 ms per launch on 472 ROIs, 30 % of a training iteration).  S <= 64.
__device__ __forceinline__ float roi_axis_weight(float lo, float bin, int g, int bin_idx, int extent, int cell) {
  // summed weight of the g samples of bin `bin_idx` along one axis on feature index `cell` (the forward's sample walk and clamps)
  float a = 0.f;
  for (int i = 0; i < g; ++i) {
    float v = lo + (float)bin_idx * bin + ((float)i + 0.5f) * bin / (float)g;
    if (v < -1.0f || v > (float)extent) continue;
    if (v <= 0.f) v = 0.f;
    int v_low = (int)v, v_high;
    if (v_low >= extent - 1) {
      v_high = v_low = extent - 1;
      v = (float)v_low;
    } else {
      v_high = v_low + 1;
    }
    const float l = v - (float)v_low;
    if (v_low == cell) a += 1.f - l;
    if (v_high == cell && v_high != v_low) a += l;
  }
  return a;
}

__global__ __launch_bounds__(256) void roi_align_backward_rows_kernel(RoiArgs p, float* d3, float* d4, float* d5, const float* __restrict__ g,
                                                                       int rows_max) {
  int R = p.R_cap;
  if (p.count) {
    const int c = *p.count;
    R = c < R ? c : R;
  }
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  const int S = p.S;
  const long total = (long)R * rows_max;
  for (long wid = (long)blockIdx.x * wpb + (threadIdx.x >> 6); wid < total; wid += (long)gridDim.x * wpb) {
    const int r = (int)(wid / rows_max);
    const int j = (int)(wid - (long)r * rows_max);
    const float bx1 = p.boxes[r * 4 + 0], by1 = p.boxes[r * 4 + 1], bx2 = p.boxes[r * 4 + 2], by2 = p.boxes[r * 4 + 3];
    const float area = (bx2 - bx1) * (by2 - by1);
    float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    lv = fminf(fmaxf(lv, 3.0f), 5.0f);
    const int l = (int)lv - 3;
    const int H = p.h[l], W = p.w[l];
    float* dfeat = l == 0 ? d3 : (l == 1 ? d4 : d5);
    const float sc = p.scale[l];
    const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f, x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
    const float roi_w = x2 - x1, roi_h = y2 - y1;
    const float bin_h = roi_h / (float)S, bin_w = roi_w / (float)S;
    const float ghf = ceilf(roi_h / (float)S), gwf = ceilf(roi_w / (float)S);
    const bool sane = (ghf == ghf) && (gwf == gwf) && ghf < 1.0e6f && gwf < 1.0e6f && ghf >= 1.0f && gwf >= 1.0f;
    if (!sane) continue;
    const int gh = (int)ghf, gw = (int)gwf;
    const float inv_cnt = 1.0f / ((float)gh * (float)gw);
    // every sample lies in (y1, y2): its taps in rows floor(y1) .. floor(y2) + 1, clamped to the level
    const int cy_lo = min(max((int)floorf(fmaxf(y1, -2.0f)), 0), H - 1), cy_hi = min(max((int)floorf(fminf(y2, (float)H + 1.0f)) + 1, 0), H - 1);
    const int cy = cy_lo + j;
    if (cy > cy_hi) continue;
    const float ay = lane < S ? roi_axis_weight(y1, bin_h, gh, lane, H, cy) : 0.f;
    const unsigned long long my = __ballot(ay != 0.f);
    if (!my) continue;
    const int cx_lo = min(max((int)floorf(fmaxf(x1, -2.0f)), 0), W - 1), cx_hi = min(max((int)floorf(fminf(x2, (float)W + 1.0f)) + 1, 0), W - 1);
    for (int cx = cx_lo; cx <= cx_hi; ++cx) {
      const float ax = lane < S ? roi_axis_weight(x1, bin_w, gw, lane, W, cx) : 0.f;
      const unsigned long long mx = __ballot(ax != 0.f);
      if (!mx) continue;
      for (int cb = 0; cb < p.C; cb += 256) {                      // wave-uniform trip count: the shuffles below see every lane
        const int c0 = cb + lane * 4;
        const bool on = c0 < p.C;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (unsigned long long m1 = my; m1; m1 &= m1 - 1) {
          const int ph = __builtin_ctzll(m1);
          const float wy = __shfl(ay, ph);
          for (unsigned long long m2 = mx; m2; m2 &= m2 - 1) {
            const int pw = __builtin_ctzll(m2);
            const float wgt = wy * __shfl(ax, pw);
            if (on) {
              const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (((size_t)r * S + ph) * S + pw) * p.C + c0);
              acc.x += wgt * gv.x; acc.y += wgt * gv.y; acc.z += wgt * gv.z; acc.w += wgt * gv.w;
            }
          }
        }
        if (on) {
          float* dst = dfeat + ((size_t)cy * W + cx) * p.C + c0;
          unsafeAtomicAdd(dst + 0, acc.x * inv_cnt);
          unsafeAtomicAdd(dst + 1, acc.y * inv_cnt);
          unsafeAtomicAdd(dst + 2, acc.z * inv_cnt);
          unsafeAtomicAdd(dst + 3, acc.w * inv_cnt);
        }
      }
    }
  }
}

}  // namespace

extern "C" int eod_roi_align(const float* p3, const float* p4, const float* p5, int h3, int w3, int C, const float* boxes,
                             const int32_t* box_rows, const int32_t* count, int R_cap, int out_size, float* out, int batch,
                             int boxes_per_image, const EodBoxRefine* refine, eod_stream_t stream) {
  if (!p3 || !p4 || !p5 || !boxes || !out) return EOD_ERR_NULL;
  if (h3 <= 0 || w3 <= 0 || (h3 & 3) || (w3 & 3) || C % 4 != 0 || R_cap <= 0 || out_size <= 0) return EOD_ERR_BAD_DIMS;
  if ((long)R_cap * out_size * out_size >= (1L << 30)) return EOD_ERR_BAD_DIMS;
  if (batch > 1 && (batch > EOD_MAX_BATCH || boxes_per_image <= 0 || (!box_rows && R_cap != batch * boxes_per_image))) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(p3) || !eod_aligned16(p4) || !eod_aligned16(p5) || !eod_aligned16(out)) return EOD_ERR_ALIGN;
  RoiArgs a{};
  a.feat[0] = p3; a.feat[1] = p4; a.feat[2] = p5;
  a.h[0] = h3; a.w[0] = w3; a.h[1] = h3 / 2; a.w[1] = w3 / 2; a.h[2] = h3 / 4; a.w[2] = w3 / 4;
  a.scale[0] = 1.0f / 8; a.scale[1] = 1.0f / 16; a.scale[2] = 1.0f / 32;
  a.C = C; a.boxes = boxes; a.box_rows = box_rows; a.count = count; a.R_cap = R_cap; a.S = out_size; a.out = out;
  a.batch = batch > 1 ? batch : 1; a.boxes_per_image = boxes_per_image;
  if (refine) {
    if (!refine->deltas || !refine->boxes_out || refine->ld < 4 || box_rows) return EOD_ERR_BAD_DIMS;      // list form: boxes are final
    a.deltas = refine->deltas; a.ld = refine->ld; a.clip = refine->clip; a.boxes_out = refine->boxes_out;
    a.wx = refine->wx; a.wy = refine->wy; a.ww = refine->ww; a.wh = refine->wh; a.img_w = refine->img_w; a.img_h = refine->img_h;
  }
  a.div_bins = eod_make_fastdiv((unsigned)(out_size * out_size));
  a.div_s = eod_make_fastdiv((unsigned)out_size);
  const long waves = (long)R_cap * out_size * out_size;
  long blocks = (waves + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(roi_align_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

// Gather form (round 4): one wave per CELL of a level's gradient.  The footprint kernel above still issues one atomic per (ROI,
// footprint cell, channel) -- 512 sampled rows with ~200-cell footprints are ~3 x 10^7 fp32 atomics per launch (0.56 ms; 1.7 ms of a
// training iteration), and their order is not fixed.  Here a cell's wave finds the ROIs whose footprint holds the cell (every lane
// tests R / 64 boxes: level, row range, column range), walks them in ROW ORDER, accumulates weight x dY in registers (lane = 4
// channels) and adds the sum to the cell once: no atomics, and the result does not depend on the schedule.  C <= 256, S <= 64.
static __global__ __launch_bounds__(256) void roi_align_backward_gather_kernel(RoiArgs p, float* d3, float* d4, float* d5, const float* __restrict__ g) {
  int R = p.R_cap;
  if (p.count) {
    const int c = *p.count;
    R = c < R ? c : R;
  }
  const int lane = threadIdx.x & 63;
  const int S = p.S;
  const int n0 = p.h[0] * p.w[0], n1 = p.h[1] * p.w[1], n2 = p.h[2] * p.w[2];
  const int cell_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (cell_id >= n0 + n1 + n2) return;                               // wave-uniform
  const int l = cell_id < n0 ? 0 : (cell_id < n0 + n1 ? 1 : 2);
  const int local = cell_id - (l == 0 ? 0 : (l == 1 ? n0 : n0 + n1));
  const int H = p.h[l], W = p.w[l];
  const int cy = local / W, cx = local - cy * W;
  const float sc = p.scale[l];
  float* dfeat = l == 0 ? d3 : (l == 1 ? d4 : d5);
  const int c0 = lane * 4;
  const bool on = c0 < p.C;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < R; base += 64) {
    // does ROI base + lane reach this cell?  (its level; the rows / columns its samples' taps can touch)
    bool hit = false;
    const int rr = base + lane;
    if (rr < R) {
      const float bx1 = p.boxes[rr * 4 + 0], by1 = p.boxes[rr * 4 + 1], bx2 = p.boxes[rr * 4 + 2], by2 = p.boxes[rr * 4 + 3];
      const float area = (bx2 - bx1) * (by2 - by1);
      float lv = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
      lv = fminf(fmaxf(lv, 3.0f), 5.0f);
      if ((int)lv - 3 == l) {
        const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f, x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
        const int cy_lo = min(max((int)floorf(fmaxf(y1, -2.0f)), 0), H - 1), cy_hi = min(max((int)floorf(fminf(y2, (float)H + 1.0f)) + 1, 0), H - 1);
        const int cx_lo = min(max((int)floorf(fmaxf(x1, -2.0f)), 0), W - 1), cx_hi = min(max((int)floorf(fminf(x2, (float)W + 1.0f)) + 1, 0), W - 1);
        hit = cy >= cy_lo && cy <= cy_hi && cx >= cx_lo && cx <= cx_hi;
      }
    }
    for (unsigned long long m = __ballot(hit); m; m &= m - 1) {
      const int r = base + __builtin_ctzll(m);                       // wave-uniform
      const float bx1 = p.boxes[r * 4 + 0], by1 = p.boxes[r * 4 + 1], bx2 = p.boxes[r * 4 + 2], by2 = p.boxes[r * 4 + 3];
      const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f, x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
      const float roi_w = x2 - x1, roi_h = y2 - y1;
      const float bin_h = roi_h / (float)S, bin_w = roi_w / (float)S;
      const float ghf = ceilf(roi_h / (float)S), gwf = ceilf(roi_w / (float)S);
      const bool sane = (ghf == ghf) && (gwf == gwf) && ghf < 1.0e6f && gwf < 1.0e6f && ghf >= 1.0f && gwf >= 1.0f;
      if (!sane) continue;
      const int gh = (int)ghf, gw = (int)gwf;
      const float inv_cnt = 1.0f / ((float)gh * (float)gw);
      const float ay = lane < S ? roi_axis_weight(y1, bin_h, gh, lane, H, cy) : 0.f;
      const unsigned long long my = __ballot(ay != 0.f);
      if (!my) continue;
      const float ax = lane < S ? roi_axis_weight(x1, bin_w, gw, lane, W, cx) : 0.f;
      const unsigned long long mx = __ballot(ax != 0.f);
      if (!mx) continue;
      f32x4 part = {0.f, 0.f, 0.f, 0.f};
      for (unsigned long long m1 = my; m1; m1 &= m1 - 1) {
        const int ph = __builtin_ctzll(m1);
        const float wy = __shfl(ay, ph);
        for (unsigned long long m2 = mx; m2; m2 &= m2 - 1) {
          const int pw = __builtin_ctzll(m2);
          const float wgt = wy * __shfl(ax, pw);
          if (on) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + (((size_t)r * S + ph) * S + pw) * p.C + c0);
            part.x += wgt * gv.x; part.y += wgt * gv.y; part.z += wgt * gv.z; part.w += wgt * gv.w;
          }
        }
      }
      acc.x += part.x * inv_cnt; acc.y += part.y * inv_cnt; acc.z += part.z * inv_cnt; acc.w += part.w * inv_cnt;
    }
  }
  if (on && (acc.x != 0.f || acc.y != 0.f || acc.z != 0.f || acc.w != 0.f)) {
    f32x4* dst = reinterpret_cast<f32x4*>(dfeat + (size_t)local * p.C + c0);
    f32x4 v = *dst;
    v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
    *dst = v;
  }
}

extern "C" int eod_roi_align_backward(float* dp3, float* dp4, float* dp5, int h3, int w3, int C, const float* boxes, const int32_t* count,
                                      int R_cap, int out_size, const float* g, eod_stream_t stream) {
  if (!dp3 || !dp4 || !dp5 || !boxes || !g) return EOD_ERR_NULL;
  if (h3 <= 0 || w3 <= 0 || (h3 & 3) || (w3 & 3) || C % 4 != 0 || R_cap <= 0 || out_size <= 0) return EOD_ERR_BAD_DIMS;
  if ((long)R_cap * out_size * out_size >= (1L << 30)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(g)) return EOD_ERR_ALIGN;
  RoiArgs a{};
  a.h[0] = h3; a.w[0] = w3; a.h[1] = h3 / 2; a.w[1] = w3 / 2; a.h[2] = h3 / 4; a.w[2] = w3 / 4;
  a.scale[0] = 1.0f / 8; a.scale[1] = 1.0f / 16; a.scale[2] = 1.0f / 32;
  a.C = C; a.boxes = boxes; a.count = count; a.R_cap = R_cap; a.S = out_size;
  a.batch = 1;
  a.div_bins = eod_make_fastdiv((unsigned)(out_size * out_size));
  a.div_s = eod_make_fastdiv((unsigned)out_size);
  static const bool by_samples = getenv("EOD_ROI_BWD_SAMPLES") != nullptr;     // the sample-tap form, for A/B runs (tools/)
  static const bool by_rows = getenv("EOD_ROI_BWD_ROWS") != nullptr;           // the footprint form with atomics (round 3), for A/B runs
  if (out_size <= 64 && C <= 256 && !by_samples && !by_rows && eod_aligned16(dp3) && eod_aligned16(dp4) && eod_aligned16(dp5)) {
    const long cells = (long)h3 * w3 + (long)(h3 / 2) * (w3 / 2) + (long)(h3 / 4) * (w3 / 4);
    hipLaunchKernelGGL(roi_align_backward_gather_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, dp3, dp4,
                       dp5, g);
    return eod_launch_status();
  }
  if (out_size <= 64 && !by_samples) {
    const int rows_max = h3 + 1;                                             // the tallest footprint: a level-3 ROI over the whole height
    const long waves = (long)R_cap * rows_max;
    long blocks = (waves + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(roi_align_backward_rows_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a, dp3, dp4, dp5, g, rows_max);
    return eod_launch_status();
  }
  const long waves = (long)R_cap * out_size * out_size;
  long blocks = (waves + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(roi_align_backward_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a, dp3, dp4, dp5, g);
  return eod_launch_status();
}
