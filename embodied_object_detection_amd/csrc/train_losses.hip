// Training losses of the proposal generator with their gradients (training slices): CenterNet.losses for the
// recurrent configuration -- ONLY_PROPOSAL + WITH_AGN_HM, not_norm_reg -- i.e. the class-agnostic heatmap focal loss
// (binary_heatmap_focal_loss, centernet/modeling/layers/heatmap_focal_loss.py:52-84) and the GIoU regression loss (IOULoss 'giou',
// layers/iou_loss.py:10-64) as centernet/modeling/dense_heads/centernet.py:241-318 combines them, evaluated on the head's raw
// output rows [P, stride] (col 0: agnostic logit, cols 1..4: bbox_pred before the level's Scale and the ReLU,
// centernet_head.py:141-161).  Two launches: a dense pass over the P positions (negative term, regression term, their gradients;
// per-workgroup partial sums in double) and one workgroup over the positive locations that also adds the partials in a fixed order.
// Elementwise + reductions: HBM-bound, P x (stride + 5) floats read, P x stride written.
#include "eod_common.h"
#include "../../include/eod_hip.h"

namespace {

struct CnLossArgs {
  const float* head;
  int stride;
  const float* heat;    // [P] agnostic heatmap target (flattened_hms.max(dim=1))
  const float* reg_t;   // [P,4] (l, t, r, b) targets, rows with max < 0: no regression target
  const int* pos;       // [n_pos]
  int n_pos, P, levels;
  int lv_off[9];
  float lv_scale[8];
  float alpha, beta, gamma, clampv, ignore_fp;
  float c_pos, c_neg, c_reg;   // pos_weight / num_pos_avg, neg_weight / num_pos_avg, reg_weight / reg_norm
  // counts on the device (no host round trip between the target assignment and the losses): counts_local[0] = length of `pos`
  // (clamped to n_pos = its capacity), counts_total = {positives, regression rows} summed over the ranks; the three constants
  // above are then weight / max(count / world, 1), evaluated by every thread as the host evaluates them
  const int* counts_local;
  const int* counts_total;
  float w_pos, w_neg, w_reg, world;
  float* d_head;        // [P, stride]
  double* partial;      // [blocks][2]
  float* losses;        // [3]: loc, agn_pos, agn_neg
};

__device__ __forceinline__ float half_on_tie(float a, float b, bool take_less) {
  // d min(a, b) / da (take_less) or d max(a, b) / da: 1 where a wins, 1/2 on a tie (torch.minimum / maximum backward), else 0
  if (a == b) return 0.5f;
  return (take_less ? a < b : a > b) ? 1.f : 0.f;
}

__device__ __forceinline__ void cn_loss_constants(CnLossArgs& a) {
  if (!a.counts_total) return;
  const float num_pos_avg = fmaxf((float)((double)a.counts_total[0] / (double)a.world), 1.f);
  const float reg_norm = fmaxf((float)((double)a.counts_total[1] / (double)a.world), 1.f);
  a.c_pos = a.w_pos / num_pos_avg;
  a.c_neg = a.w_neg / num_pos_avg;
  a.c_reg = a.w_reg / reg_norm;
  const int n = a.counts_local[0];
  a.n_pos = n < a.n_pos ? (n < 0 ? 0 : n) : a.n_pos;
}

__global__ __launch_bounds__(256) void centernet_loss_dense_kernel(CnLossArgs a) {
  cn_loss_constants(a);
  __shared__ double red[2][4];
  double neg_sum = 0.0, loc_sum = 0.0;
  const float an = a.alpha >= 0.f ? 1.f - a.alpha : 1.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.P; i += gridDim.x * blockDim.x) {
    const float* h = a.head + (size_t)i * a.stride;
    float* d = a.d_head + (size_t)i * a.stride;
    // ---- negative term of the agnostic focal loss
    const float s = eod_sigmoid_precise(h[0]);
    const float p = fminf(fmaxf(s, a.clampv), 1.f - a.clampv);
    const bool inside = s >= a.clampv && s <= 1.f - a.clampv;
    const float nw = powf(1.f - a.heat[i], a.beta);
    const bool keep = !(a.ignore_fp > 0.f) || p < a.ignore_fp;
    float gz = 0.f;
    if (keep) {
      const float l1p = logf(1.f - p), pg = powf(p, a.gamma);
      neg_sum += (double)(l1p * pg * nw);
      const float dp = (-pg / (1.f - p) + a.gamma * powf(p, a.gamma - 1.f) * l1p) * nw;
      if (inside) gz = -an * a.c_neg * dp * s * (1.f - s);
    }
    d[0] = gz;
    for (int c = 5; c < a.stride; ++c) d[c] = 0.f;
    // ---- GIoU regression term
    const float tl = a.reg_t[i * 4 + 0], tt = a.reg_t[i * 4 + 1], tr = a.reg_t[i * 4 + 2], tb = a.reg_t[i * 4 + 3];
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    if (fmaxf(fmaxf(tl, tt), fmaxf(tr, tb)) >= 0.f) {
      // the level's Scale by a chain of selects over the (ascending) level offsets: indexing the kernel arguments with a per-lane
      // level makes the compiler keep a copy of the argument struct in scratch memory (216 bytes per lane)
      float sc = a.lv_scale[0];
#pragma unroll
      for (int q = 1; q < 8; ++q)
        if (q < a.levels && i >= a.lv_off[q]) sc = a.lv_scale[q];
      const float r0 = h[1] * sc, r1 = h[2] * sc, r2 = h[3] * sc, r3 = h[4] * sc;
      const float pl = fmaxf(r0, 0.f), pt = fmaxf(r1, 0.f), pr = fmaxf(r2, 0.f), pb = fmaxf(r3, 0.f);
      const float t_area = (tl + tr) * (tt + tb), p_area = (pl + pr) * (pt + pb);
      const float wi = fminf(pl, tl) + fminf(pr, tr), hi = fminf(pb, tb) + fminf(pt, tt);
      const float gw = fmaxf(pl, tl) + fmaxf(pr, tr), gh = fmaxf(pb, tb) + fmaxf(pt, tt);
      const float ac = gw * gh, ai = wi * hi, au = t_area + p_area - ai;
      const float iou = (ai + 1.f) / (au + 1.f);
      const float giou = iou - (ac - au) / ac;
      loc_sum += (double)(1.f - giou);
      // d giou / d x for x in (pl, pt, pr, pb): through p_area, ai (wi or hi) and ac (gw or gh)
      const float dpa[4] = {pt + pb, pl + pr, pt + pb, pl + pr};
      const float dai[4] = {half_on_tie(pl, tl, true) * hi, half_on_tie(pt, tt, true) * wi, half_on_tie(pr, tr, true) * hi,
                            half_on_tie(pb, tb, true) * wi};
      const float dac[4] = {half_on_tie(pl, tl, false) * gh, half_on_tie(pt, tt, false) * gw, half_on_tie(pr, tr, false) * gh,
                            half_on_tie(pb, tb, false) * gw};
      const float raw[4] = {r0, r1, r2, r3};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dau = dpa[k] - dai[k];
        const float diou = (dai[k] * (au + 1.f) - (ai + 1.f) * dau) / ((au + 1.f) * (au + 1.f));
        const float dfrac = (dau * ac - au * dac[k]) / (ac * ac);       // d (au / ac)
        const float dg = diou + dfrac;
        g4[k] = raw[k] > 0.f ? -a.c_reg * dg * sc : 0.f;
      }
    }
    d[1] = g4[0]; d[2] = g4[1]; d[3] = g4[2]; d[4] = g4[3];
  }
  // workgroup sums, waves added in order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) {
    neg_sum += __shfl_xor(neg_sum, off, 64);
    loc_sum += __shfl_xor(loc_sum, off, 64);
  }
  if (lane == 0) { red[0][wv] = neg_sum; red[1][wv] = loc_sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partial[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    a.partial[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ __launch_bounds__(256) void centernet_loss_pos_kernel(CnLossArgs a, int dense_blocks) {
  cn_loss_constants(a);
  __shared__ double red[4];
  double pos_sum = 0.0;
  const float ap = a.alpha >= 0.f ? a.alpha : 1.f;
  for (int j = threadIdx.x; j < a.n_pos; j += blockDim.x) {
    const int i = a.pos[j];
    if (i < 0 || i >= a.P) continue;
    const float s = eod_sigmoid_precise(a.head[(size_t)i * a.stride]);
    const float p = fminf(fmaxf(s, a.clampv), 1.f - a.clampv);
    const bool inside = s >= a.clampv && s <= 1.f - a.clampv;
    const float lp = logf(p), q = powf(1.f - p, a.gamma);
    pos_sum += (double)(lp * q);
    if (inside) {
      const float dp = q / p - a.gamma * lp * powf(1.f - p, a.gamma - 1.f);
      unsafeAtomicAdd(a.d_head + (size_t)i * a.stride, -ap * a.c_pos * dp * s * (1.f - s));     // a location may be listed twice
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) pos_sum += __shfl_xor(pos_sum, off, 64);
  if (lane == 0) red[wv] = pos_sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double neg = 0.0, loc = 0.0;
    for (int b = 0; b < dense_blocks; ++b) { neg += a.partial[b * 2]; loc += a.partial[b * 2 + 1]; }
    const double pos = red[0] + red[1] + red[2] + red[3];
    const float an = a.alpha >= 0.f ? 1.f - a.alpha : 1.f;
    a.losses[0] = (float)((double)a.c_reg * loc);
    a.losses[1] = (float)(-(double)(ap * a.c_pos) * pos);
    a.losses[2] = (float)(-(double)(an * a.c_neg) * neg);
  }
}

constexpr int kDenseBlocksMax = 512;

// DeticFastRCNNOutputLayers.losses for USE_SIGMOID_CE + class-agnostic regression (detic_fast_rcnn.py:157-197): one workgroup per
// ROI row -- sigmoid cross entropy over the C foreground columns against the one-hot of gt_classes (the background column takes no
// part, :205-207), optional per-class weight (:213-225), and for foreground rows the smooth-L1 distance between the predicted deltas
// and Box2BoxTransform.get_deltas(proposal, gt) (:282-291); both normalised by the number of rows B (:232, :303).  Row sums in
// double, added in row order by the second launch.
struct BoxLossArgs {
  const float* scores; int ld;     // [B, ld], columns 0..C
  const float* deltas;             // [B,4]
  const float* prop; const float* gtb;   // [B,4]
  const int* gt;                   // [B]
  const float* cw;                 // [C] or null
  int B, C;
  float wx, wy, ww, wh, beta;
  float* d_scores; float* d_deltas;
  double* partial;                 // [B][2]
  float* losses;                   // [2]: loss_cls, loss_box_reg
};

__global__ __launch_bounds__(256) void fast_rcnn_loss_rows_kernel(BoxLossArgs a) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const int g = a.gt[b];
  const float inv_b = 1.f / (float)a.B;
  const float* z = a.scores + (size_t)b * a.ld;
  float* dz = a.d_scores + (size_t)b * a.ld;
  double sum = 0.0;
  for (int c = threadIdx.x; c < a.ld; c += blockDim.x) {
    if (c >= a.C) { dz[c] = 0.f; continue; }
    const float w = a.cw ? a.cw[c] : 1.f;
    const float y = c == g ? 1.f : 0.f;
    const float v = z[c];
    // binary_cross_entropy_with_logits: (1 - y) v + log1p(exp(-|v|)) + max(-v, 0)
    sum += (double)(w * ((1.f - y) * v + log1pf(expf(-fabsf(v))) + fmaxf(-v, 0.f)));
    dz[c] = w * (eod_sigmoid_precise(v) - y) * inv_b;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) red[wv] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partial[b * 2] = red[0] + red[1] + red[2] + red[3];
    double reg = 0.0;
    float gd[4] = {0.f, 0.f, 0.f, 0.f};
    if (g >= 0 && g < a.C) {
      const float* s = a.prop + b * 4;
      const float* t = a.gtb + b * 4;
      const float sw = s[2] - s[0], sh = s[3] - s[1], sx = s[0] + 0.5f * sw, sy = s[1] + 0.5f * sh;
      const float tw = t[2] - t[0], th = t[3] - t[1], tx = t[0] + 0.5f * tw, ty = t[1] + 0.5f * th;
      const float tgt[4] = {a.wx * (tx - sx) / sw, a.wy * (ty - sy) / sh, a.ww * logf(tw / sw), a.wh * logf(th / sh)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float e = a.deltas[b * 4 + k] - tgt[k], ae = fabsf(e);
        const float sgn = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
        if (a.beta < 1e-5f) {
          reg += (double)ae;
          gd[k] = sgn * inv_b;
        } else if (ae < a.beta) {
          reg += (double)(0.5f * e * e / a.beta);
          gd[k] = e / a.beta * inv_b;
        } else {
          reg += (double)(ae - 0.5f * a.beta);
          gd[k] = sgn * inv_b;
        }
      }
    }
    a.partial[b * 2 + 1] = reg;
#pragma unroll
    for (int k = 0; k < 4; ++k) a.d_deltas[b * 4 + k] = gd[k];
  }
}

__global__ __launch_bounds__(64) void fast_rcnn_loss_sum_kernel(BoxLossArgs a) {
  if (threadIdx.x != 0) return;
  double cls = 0.0, reg = 0.0;
  for (int b = 0; b < a.B; ++b) { cls += a.partial[b * 2]; reg += a.partial[b * 2 + 1]; }
  a.losses[0] = (float)(cls / (double)a.B);
  a.losses[1] = (float)(reg / (double)a.B);
}

// CenterNet target assignment for ONLY_PROPOSAL, one image (centernet.py:342-479).  Every comparison below decides a target, so the
// arithmetic follows the reference's fp32 operations one by one (no fused multiply-add: `fp contract(off)`).
struct CnTargetArgs {
  const float* boxes;   // [N,4]
  int N, P, levels;
  int lv_off[9], lv_w[8], lv_stride[8];
  float soi_lo[8], soi_hi[8];
  float radius_scale;   // (float)(delta^2 * 2)
  float min_radius2;
  float* heat;          // [P]
  float* reg;           // [P,4]
  int* pos;             // [N * levels]
  int* counts;          // [2]: positives, regression rows
};

// positive locations (`_get_label_inds`, :441-479): box-major, level-minor list of the cells that hold a box's centre on the levels
// whose size range contains the box (`assign_fpn_level`, :482-498).  One wave; also resets the regression-row counter.
__global__ __launch_bounds__(64) void centernet_pos_inds_kernel(CnTargetArgs a) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x;
  int base = 0;
  const int total = a.N * a.levels;
  for (int i0 = 0; i0 < total; i0 += 64) {
    const int i = i0 + lane;
    bool cared = false;
    int idx = 0;
    if (i < total) {
      const int n = i / a.levels, l = i - n * a.levels;
      const float x1 = a.boxes[n * 4], y1 = a.boxes[n * 4 + 1], x2 = a.boxes[n * 4 + 2], y2 = a.boxes[n * 4 + 3];
      const float w = x2 - x1, h = y2 - y1;
      const float crit = sqrtf(w * w + h * h) / 2.f;
      cared = crit >= a.soi_lo[l] && crit <= a.soi_hi[l];
      const float s = (float)a.lv_stride[l];
      const int cxi = (int)(((x1 + x2) / 2.f) / s), cyi = (int)(((y1 + y2) / 2.f) / s);
      idx = a.lv_off[l] + cyi * a.lv_w[l] + cxi;
    }
    const unsigned long long bal = __ballot(cared);
    if (cared) a.pos[base + __popcll(bal & ((1ull << lane) - 1ull))] = idx;
    base += __popcll(bal);
  }
  if (lane == 0) { a.counts[0] = base; a.counts[1] = 0; }
}

// per grid position (`_get_ground_truth`, :342-438): the regression target of the nearest (radius-weighted) object among those whose
// 3x3 centre region contains the position on this level, and the class-agnostic Gaussian heatmap over all objects.
__global__ __launch_bounds__(256) void centernet_targets_kernel(CnTargetArgs a) {
#pragma clang fp contract(off)
  __shared__ float sb[256][8];   // x1, y1, x2, y2, cx, cy, radius2
  const float INF = 100000000.f;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = m < a.P;
  int l = 0;
  if (live)
    while (l + 1 < a.levels && m >= a.lv_off[l + 1]) ++l;
  const int si = a.lv_stride[l];
  const float s = (float)si;
  const int rel = live ? m - a.lv_off[l] : 0;
  const int iy = rel / a.lv_w[l], ix = rel - iy * a.lv_w[l];
  const float gx = (float)(ix * si + si / 2), gy = (float)(iy * si + si / 2);     // compute_grids (:321-339)
  float best = INF, hmin = INF * 10.f;
  float r0 = -INF, r1 = -INF, r2 = -INF, r3 = -INF;
  for (int n0 = 0; n0 < a.N; n0 += 256) {
    __syncthreads();
    const int nb = n0 + threadIdx.x;
    if (nb < a.N) {
      const float x1 = a.boxes[nb * 4], y1 = a.boxes[nb * 4 + 1], x2 = a.boxes[nb * 4 + 2], y2 = a.boxes[nb * 4 + 3];
      sb[threadIdx.x][0] = x1; sb[threadIdx.x][1] = y1; sb[threadIdx.x][2] = x2; sb[threadIdx.x][3] = y2;
      sb[threadIdx.x][4] = (x1 + x2) / 2.f;
      sb[threadIdx.x][5] = (y1 + y2) / 2.f;
      const float area = (x2 - x1) * (y2 - y1);
      sb[threadIdx.x][6] = fmaxf(area * a.radius_scale, a.min_radius2);
    }
    __syncthreads();
    const int cnt = min(256, a.N - n0);
    if (!live) continue;
    for (int j = 0; j < cnt; ++j) {
      const float x1 = sb[j][0], y1 = sb[j][1], x2 = sb[j][2], y2 = sb[j][3], cx = sb[j][4], cy = sb[j][5];
      const float lt = gx - x1, tt = gy - y1, rt = x2 - gx, bt = y2 - gy;
      const float cdx = (float)(int)(cx / s) * s + s / 2.f, cdy = (float)(int)(cy / s) * s + s / 2.f;
      const float ex = gx - cdx, ey = gy - cdy;
      const bool is_peak = (ex * ex + ey * ey) == 0.f;
      const bool in_box = fminf(fminf(lt, tt), fminf(rt, bt)) > 0.f;
      const bool c3 = fabsf(ex) <= s && fabsf(ey) <= s && in_box;
      const float sw = lt + rt, sh = tt + bt;
      const float crit = sqrtf(sw * sw + sh * sh) / 2.f;
      const bool mask = c3 && crit >= a.soi_lo[l] && crit <= a.soi_hi[l];
      const float dx = gx - cx, dy = gy - cy;
      const float dist2 = is_peak ? 0.f : dx * dx + dy * dy;
      const float wd = dist2 / sb[j][6];
      hmin = fminf(hmin, wd);
      if (mask && wd < best) {          // strict: the first of equal distances wins, as torch.min(dim) returns it
        best = wd;
        r0 = lt; r1 = tt; r2 = rt; r3 = bt;
      }
    }
  }
  bool has = false;
  if (live) {
    has = best < INF;
    if (!has) { r0 = r1 = r2 = r3 = -INF; }
    f32x4 o = {r0 / s, r1 / s, r2 / s, r3 / s};
    *reinterpret_cast<f32x4*>(a.reg + (size_t)m * 4) = o;
    float hm = a.N > 0 ? expf(-hmin) : 0.f;
    if (hm < 1e-4f) hm = 0.f;
    a.heat[m] = hm;
  }
  const unsigned long long bal = __ballot(has);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(a.counts + 1, __popcll(bal));
}

}  // namespace

extern "C" int eod_centernet_targets(const EodCenterNetTargetDesc* d, eod_stream_t stream) {
  if (!d || !d->agn_heatmap || !d->reg_targets || !d->pos_inds || !d->counts) return EOD_ERR_NULL;
  if (d->n_boxes > 0 && !d->gt_boxes) return EOD_ERR_NULL;
  if (d->n_boxes < 0 || d->n_boxes > 4096 || d->levels < 1 || d->levels > 8 || d->level_off[0] != 0) return EOD_ERR_BAD_DIMS;
  for (int l = 0; l < d->levels; ++l)
    if (d->level_w[l] <= 0 || d->level_stride[l] <= 0 || d->level_off[l + 1] <= d->level_off[l] ||
        (d->level_off[l + 1] - d->level_off[l]) % d->level_w[l] != 0)
      return EOD_ERR_BAD_DIMS;
  if (!(d->hm_min_overlap > 0.0 && d->hm_min_overlap < 1.0) || !(d->min_radius >= 0.0)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(d->reg_targets)) return EOD_ERR_ALIGN;
  CnTargetArgs a{};
  a.boxes = d->gt_boxes; a.N = d->n_boxes; a.levels = d->levels; a.P = d->level_off[d->levels];
  for (int l = 0; l <= d->levels; ++l) a.lv_off[l] = d->level_off[l];
  for (int l = 0; l < d->levels; ++l) {
    a.lv_w[l] = d->level_w[l]; a.lv_stride[l] = d->level_stride[l];
    a.soi_lo[l] = d->soi_lo[l]; a.soi_hi[l] = d->soi_hi[l];
  }
  const double delta = (1.0 - d->hm_min_overlap) / (1.0 + d->hm_min_overlap);          // centernet.py:117
  a.radius_scale = (float)(delta * delta * 2.0);                                       // :414
  a.min_radius2 = (float)(d->min_radius * d->min_radius);
  a.heat = d->agn_heatmap; a.reg = d->reg_targets; a.pos = d->pos_inds; a.counts = d->counts;
  hipLaunchKernelGGL(centernet_pos_inds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(centernet_targets_kernel, dim3((a.P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

extern "C" size_t eod_fast_rcnn_loss_workspace_bytes(int B) { return B > 0 ? (size_t)B * 2 * sizeof(double) : 0; }

extern "C" int eod_fast_rcnn_loss(const float* scores, int ld, const float* deltas, const float* proposal_boxes, const float* gt_boxes,
                                  const int32_t* gt_classes, const float* class_weight, int B, int num_classes, float wx, float wy,
                                  float ww, float wh, float smooth_l1_beta, float* d_scores, float* d_deltas, float* losses,
                                  void* workspace, size_t workspace_bytes, eod_stream_t stream) {
  if (!scores || !deltas || !proposal_boxes || !gt_boxes || !gt_classes || !d_scores || !d_deltas || !losses || !workspace) return EOD_ERR_NULL;
  if (B <= 0 || num_classes <= 0 || ld < num_classes + 1 || !(smooth_l1_beta >= 0.f)) return EOD_ERR_BAD_DIMS;
  if (workspace_bytes < eod_fast_rcnn_loss_workspace_bytes(B)) return EOD_ERR_CAPACITY;
  BoxLossArgs a{};
  a.scores = scores; a.ld = ld; a.deltas = deltas; a.prop = proposal_boxes; a.gtb = gt_boxes; a.gt = gt_classes; a.cw = class_weight;
  a.B = B; a.C = num_classes; a.wx = wx; a.wy = wy; a.ww = ww; a.wh = wh; a.beta = smooth_l1_beta;
  a.d_scores = d_scores; a.d_deltas = d_deltas; a.partial = static_cast<double*>(workspace); a.losses = losses;
  hipLaunchKernelGGL(fast_rcnn_loss_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(fast_rcnn_loss_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

extern "C" size_t eod_centernet_loss_workspace_bytes(void) { return (size_t)kDenseBlocksMax * 2 * sizeof(double); }

extern "C" int eod_centernet_loss(const EodCenterNetLossDesc* d, eod_stream_t stream) {
  if (!d || !d->head_out || !d->agn_heatmap || !d->reg_targets || !d->d_head_out || !d->losses || !d->workspace) return EOD_ERR_NULL;
  if (d->n_pos > 0 && !d->pos_inds) return EOD_ERR_NULL;
  if (d->P <= 0 || d->head_stride < 5 || d->levels < 1 || d->levels > 8 || d->n_pos < 0) return EOD_ERR_BAD_DIMS;
  if (d->level_off[0] != 0 || d->level_off[d->levels] != d->P) return EOD_ERR_BAD_DIMS;
  for (int l = 0; l < d->levels; ++l)
    if (d->level_off[l + 1] < d->level_off[l]) return EOD_ERR_BAD_DIMS;
  const bool dev_counts = d->counts_total != nullptr;
  if (dev_counts && (!d->counts_local || !(d->world_size >= 1.f))) return EOD_ERR_BAD_DIMS;
  if (!dev_counts && (!(d->num_pos_avg >= 1.f) || !(d->reg_norm >= 1.f))) return EOD_ERR_BAD_DIMS;
  if (!(d->sigmoid_clamp > 0.f && d->sigmoid_clamp < 0.5f)) return EOD_ERR_BAD_DIMS;
  if (d->workspace_bytes < eod_centernet_loss_workspace_bytes()) return EOD_ERR_CAPACITY;
  CnLossArgs a{};
  a.head = d->head_out; a.stride = d->head_stride; a.heat = d->agn_heatmap; a.reg_t = d->reg_targets;
  a.pos = d->pos_inds; a.n_pos = d->n_pos; a.P = d->P; a.levels = d->levels;
  for (int l = 0; l <= d->levels; ++l) a.lv_off[l] = d->level_off[l];
  for (int l = 0; l < d->levels; ++l) a.lv_scale[l] = d->level_scale[l];
  a.alpha = d->hm_focal_alpha; a.beta = d->hm_focal_beta; a.gamma = d->loss_gamma; a.clampv = d->sigmoid_clamp;
  a.ignore_fp = d->ignore_high_fp;
  if (dev_counts) {
    a.counts_local = d->counts_local; a.counts_total = d->counts_total; a.world = d->world_size;
    a.w_pos = d->pos_weight; a.w_neg = d->neg_weight; a.w_reg = d->reg_weight;
  } else {
    a.c_pos = d->pos_weight / d->num_pos_avg; a.c_neg = d->neg_weight / d->num_pos_avg; a.c_reg = d->reg_weight / d->reg_norm;
  }
  a.d_head = d->d_head_out; a.partial = static_cast<double*>(d->workspace); a.losses = d->losses;
  int blocks = (d->P + 255) / 256;
  if (blocks > kDenseBlocksMax) blocks = kDenseBlocksMax;
  hipLaunchKernelGGL(centernet_loss_dense_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(centernet_loss_pos_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, blocks);
  return eod_launch_status();
}

// ==================================================================================================================================
// The ROI heads' half of the training forward (custom_rcnn.py:642-650 -> detic_roi_heads.py:226-240,88-147): what stands between the
// proposals and `eod_fast_rcnn_loss`.
//  * match_label_kernel: detectron2's `pairwise_iou` + `Matcher([thr], [0, 1], allow_low_quality_matches=False)` + the labelling of
//    `ROIHeads._sample_proposals` / `CascadeROIHeads._match_and_label_boxes` (called at detic_roi_heads.py:232,115): every proposal
//    takes the FIRST ground-truth box of maximal IoU (torch.max(dim=0)), is foreground iff that IoU >= the stage's threshold, and
//    carries the matched box and its class (background = num_classes).  One thread per proposal, the ground-truth boxes staged
//    through LDS 256 at a time; no fused multiply-adds, so that every IoU is the reference's fp32 value and the >= decides alike.
//  * sample_proposals_kernel: `subsample_labels` (called through `label_and_sample_proposals`, detic_roi_heads.py:232) as a
//    selection by random KEYS: of the foreground rows the min(int(batch * fraction), #fg) with the smallest (key, row), of the
//    background rows the min(batch - that, #bg) smallest; foreground rows first, each class in ascending row order.  With
//    independent uniform keys this is the uniform random subset the reference draws with torch.randperm (its RNG stream itself is
//    not reproducible across devices); one workgroup, keys and kinds in LDS, rank by counting.
//  * zs_logits_kernel: DeticFastRCNNOutputLayers.forward's `scores` (detic_fast_rcnn.py:437-466 -> zero_shot_classifier.py:71-111,
//    NORM_WEIGHT, USE_BIAS 0): logits = temp * normalize(feat) . zs_weight, the value `zs_classify_kernel` (heads.hip) feeds its
//    sigmoid, with that kernel's lane ownership and summation order -- but written out, for any number of classes.
// Integer / index work + a [B,512] x [512,C1] product: launch-latency bound at these sizes (R <= 8192, B <= 512).
// ==================================================================================================================================
namespace {

struct MatchArgs {
  const float* boxes; int R;
  const float* gtb; const int* gtc; int G;
  float thr; int C;
  int* matched; float* iou; int* cls; float* out_gtb;
  // eod_match_label_proposals: `boxes` is a capacity-sized proposal list with its count on the device; row i of the R = cap + G
  // rows is proposal i (i < count), ground-truth box i - count (add_ground_truth_to_proposals, when `append`), or no row at all
  // (class -1: ignored by the sampling); `all_boxes` [R,4] receives the rows' boxes
  const int* count; int cap; int append; float* all_boxes;
};

__global__ __launch_bounds__(256) void match_label_kernel(MatchArgs a) {
#pragma clang fp contract(off)
  __shared__ float g[256 * 5];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float bx0 = 0.f, by0 = 0.f, bx1 = 0.f, by1 = 0.f, barea = 0.f;
  bool valid = i < a.R;
  if (i < a.R) {
    const float* src = a.boxes + (size_t)i * 4;
    if (a.count) {
      int n = *a.count;
      n = n < 0 ? 0 : (n > a.cap ? a.cap : n);
      if (i >= n) {
        valid = a.append && i - n < a.G;
        src = a.gtb + (size_t)(valid ? i - n : 0) * 4;
      }
    }
    if (valid) {
      bx0 = src[0]; by0 = src[1]; bx1 = src[2]; by1 = src[3];
      barea = (bx1 - bx0) * (by1 - by0);                                 // Boxes.area
    }
    if (a.all_boxes) {
      a.all_boxes[i * 4] = bx0; a.all_boxes[i * 4 + 1] = by0; a.all_boxes[i * 4 + 2] = bx1; a.all_boxes[i * 4 + 3] = by1;
    }
  }
  float best = -1.f;
  int bi = 0;
  for (int base = 0; base < a.G; base += 256) {
    const int n = a.G - base < 256 ? a.G - base : 256;
    __syncthreads();
    if ((int)threadIdx.x < n) {
      const float* p = a.gtb + (size_t)(base + threadIdx.x) * 4;
      const float x0 = p[0], y0 = p[1], x1 = p[2], y1 = p[3];
      float* q = g + threadIdx.x * 5;
      q[0] = x0; q[1] = y0; q[2] = x1; q[3] = y1; q[4] = (x1 - x0) * (y1 - y0);
    }
    __syncthreads();
    if (i < a.R) {
      for (int j = 0; j < n; ++j) {
        const float* q = g + j * 5;
        float w = fminf(q[2], bx1) - fmaxf(q[0], bx0);                     // pairwise_intersection: min of the maxima - max of the minima
        float h = fminf(q[3], by1) - fmaxf(q[1], by0);
        w = fmaxf(w, 0.f); h = fmaxf(h, 0.f);                              // clamp_(min=0)
        const float inter = w * h;
        float v = 0.f;
        if (inter > 0.f) {                                                 // torch.where(inter > 0, inter / (area1 + area2 - inter), 0)
          const float s = q[4] + barea;
          v = inter / (s - inter);
        }
        if (v > best) { best = v; bi = base + j; }                         // first maximum
      }
    }
  }
  if (i >= a.R) return;
  if (!valid) {                                                            // beyond the list: no row (ignored by the sampling)
    a.matched[i] = 0; a.iou[i] = 0.f; a.cls[i] = -1;
    a.out_gtb[i * 4] = 0.f; a.out_gtb[i * 4 + 1] = 0.f; a.out_gtb[i * 4 + 2] = 0.f; a.out_gtb[i * 4 + 3] = 0.f;
    return;
  }
  if (a.G == 0) {                                                          // Matcher on an empty matrix: match 0, label 0
    a.matched[i] = 0; a.iou[i] = 0.f; a.cls[i] = a.C;
    a.out_gtb[i * 4] = 0.f; a.out_gtb[i * 4 + 1] = 0.f; a.out_gtb[i * 4 + 2] = 0.f; a.out_gtb[i * 4 + 3] = 0.f;
    return;
  }
  a.matched[i] = bi;
  a.iou[i] = best;
  a.cls[i] = best >= a.thr ? a.gtc[bi] : a.C;
  const float* p = a.gtb + (size_t)bi * 4;
  a.out_gtb[i * 4] = p[0]; a.out_gtb[i * 4 + 1] = p[1]; a.out_gtb[i * 4 + 2] = p[2]; a.out_gtb[i * 4 + 3] = p[3];
}

#define SAMPLE_MAX_R 8192
struct SampleArgs {
  const int* cls; const float* keys;
  int R, C, batch, max_pos;
  int* sampled; int* counts;
};

// exclusive scan of one int per thread over the 1024 threads of the workgroup (wave shuffles + one LDS pass); *total = the sum
__device__ __forceinline__ int sample_block_scan(int v, int* wsum, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();                                   // wsum may still be read from the previous scan
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int c = wsum[w];
    if (w < wave) before += c;
    all += c;
  }
  *total = all;
  return before + inc - v;
}

// Ranks by counting are O(R^2) in one workgroup (0.55 ms at R = 2072: the training step's 4 000 / 2 000 proposal lists).  Uniform
// keys in [0, 1) bucket well: rows are dealt into 1024 value buckets per kind (counting sort in LDS), a row's rank among its kind =
// rows in smaller buckets + rows of its own bucket with a smaller (key, row) -- two or three candidates on average, the whole bucket
// when many keys are equal (still exact: the comparison is the full (key, row) order).  The sampled rows' slots are a block scan.
__global__ __launch_bounds__(1024) void sample_proposals_kernel(SampleArgs a) {
  constexpr int NB = 1024;
  __shared__ float key[SAMPLE_MAX_R];
  __shared__ unsigned char kind[SAMPLE_MAX_R];      // 0: ignored (-1), 1: foreground, 2: background; bit 2: sampled
  __shared__ unsigned short list[SAMPLE_MAX_R];     // rows grouped by (kind, bucket)
  __shared__ int start[2][NB + 1];                  // first list slot of a (kind, bucket); then the running fill position
  __shared__ int wsum[16];
  __shared__ int tot[2];
  const int tid = threadIdx.x;
  if (tid < 2) tot[tid] = 0;
  for (int b = tid; b < 2 * (NB + 1); b += 1024) (&start[0][0])[b] = 0;
  __syncthreads();
  auto bucket_of = [](float k) {
    int b = (int)(k * (float)NB);
    return b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
  };
  for (int i = tid; i < a.R; i += 1024) {
    const int c = a.cls[i];
    const unsigned char k = c == a.C ? 2 : (c != -1 ? 1 : 0);            // subsample_labels: (labels != -1) & (labels != bg) / labels == bg
    const float kv = a.keys[i];
    key[i] = kv;
    kind[i] = k;
    if (k) {
      atomicAdd(&tot[k - 1], 1);
      atomicAdd(&start[k - 1][bucket_of(kv) + 1], 1);                    // counts, shifted by one for the exclusive prefix
    }
  }
  __syncthreads();
  // exclusive prefix over the buckets of each kind (thread t owns bucket t); the background lists follow the foreground ones
  {
    int total = 0;
    const int c0 = start[0][tid + 1];
    const int e0 = sample_block_scan(c0, wsum, &total);
    const int n_fg = total;
    const int c1 = start[1][tid + 1];
    const int e1 = sample_block_scan(c1, wsum, &total);
    __syncthreads();
    start[0][tid] = e0;
    start[1][tid] = n_fg + e1;
    if (tid == 0) { start[0][NB] = n_fg; start[1][NB] = n_fg + total; }
  }
  __syncthreads();
  // fill the lists: `fillp` = a second copy of the bucket starts, advanced atomically (the order inside a bucket does not matter)
  __shared__ int fillp[2][NB];
  fillp[0][tid] = start[0][tid];
  fillp[1][tid] = start[1][tid];
  __syncthreads();
  for (int i = tid; i < a.R; i += 1024) {
    const unsigned char k = kind[i];
    if (k) list[atomicAdd(&fillp[k - 1][bucket_of(key[i])], 1)] = (unsigned short)i;
  }
  __syncthreads();
  const int n_pos = tot[0] < a.max_pos ? tot[0] : a.max_pos;
  const int n_neg = tot[1] < a.batch - n_pos ? tot[1] : a.batch - n_pos;
  const int base1 = start[0][NB];                                         // first background slot
  unsigned keep = 0;                                                      // bit q: row tid + 1024 q is sampled
  for (int i = tid, q = 0; i < a.R; i += 1024, ++q) {
    const unsigned char k = kind[i];
    if (!k) continue;
    const float ki = key[i];
    const int b = bucket_of(ki);
    const int lo = start[k - 1][b], hi = start[k - 1][b + 1];
    int rank = lo - (k == 2 ? base1 : 0);                                 // rows of this kind in smaller buckets
    for (int p = lo; p < hi; ++p) {
      const int j = list[p];
      const float kj = key[j];
      rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
    }
    if (rank < (k == 1 ? n_pos : n_neg)) keep |= 1u << q;
  }
  __syncthreads();
  for (int i = tid, q = 0; i < a.R; i += 1024, ++q)
    if (keep >> q & 1u) kind[i] |= 4;
  __syncthreads();
  // slots: foreground rows first, each kind in ascending row order = exclusive scans over the rows.  Thread t scans a CONTIGUOUS
  // run of rows so that the order of the scan is the row order
  const int per = (a.R + 1023) / 1024;
  const int r0 = tid * per, r1 = min(r0 + per, a.R);
  int cf = 0, cb = 0;
  for (int i = r0; i < r1; ++i) {
    cf += kind[i] == 5 ? 1 : 0;
    cb += kind[i] == 6 ? 1 : 0;
  }
  int total = 0;
  int sf = sample_block_scan(cf, wsum, &total);
  int sb = sample_block_scan(cb, wsum, &total) + n_pos;
  for (int i = r0; i < r1; ++i) {
    if (kind[i] == 5) a.sampled[sf++] = i;
    else if (kind[i] == 6) a.sampled[sb++] = i;
  }
  if (tid == 0) { a.counts[0] = n_pos; a.counts[1] = n_pos + n_neg; }
}

// one wave per row; lane owns channels [8 lane, 8 lane + 8) as in zs_classify_kernel (D = 512)
__global__ __launch_bounds__(256) void zs_logits_kernel(const float* __restrict__ feat, const float* __restrict__ zs, int B, int C1,
                                                         float temp, float* __restrict__ logits, int ld, float* __restrict__ featn_out) {
  const int D = 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float x[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = feat[(size_t)row * D + lane * 8 + q];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) ss += x[q] * x[q];
  ss = wave_reduce_sum(ss);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);                            // F.normalize eps
#pragma unroll
  for (int q = 0; q < 8; ++q) x[q] = temp * (x[q] / denom);
  if (featn_out) {
#pragma unroll
    for (int q = 0; q < 8; ++q) featn_out[(size_t)row * D + lane * 8 + q] = x[q];
  }
  for (int c0 = 0; c0 < C1; c0 += 64) {
    const int n = C1 - c0 < 64 ? C1 - c0 : 64;
    float mine = 0.f;
    for (int c = 0; c < n; ++c) {
      const float* w = zs + (size_t)(lane * 8) * C1 + c0 + c;            // zs_weight [D, C1]
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += x[q] * w[(size_t)q * C1];
      s = wave_reduce_sum(s);
      if (lane == c) mine = s;
    }
    if (lane < n) logits[(size_t)row * ld + c0 + lane] = mine;
  }
}

// backward of zs_logits_kernel with respect to the feature (the class matrix is a buffer, zero_shot_classifier.py:54): with
// n = max(|f|, eps), u = f / n and g = temp * d_logits . zs^T:  d f = (g - u (u . g)) / n  (|f| > eps; below it the clamp's gradient is 0
// and d f = g / eps, as F.normalize's autograd gives).  Same row / lane ownership as the forward.
__global__ __launch_bounds__(256) void zs_logits_backward_kernel(const float* __restrict__ feat, const float* __restrict__ zs,
                                                                  const float* __restrict__ d_logits, int ld, int B, int C1, float temp,
                                                                  float* __restrict__ d_feat) {
  const int D = 512;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float u[8], g[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { u[q] = feat[(size_t)row * D + lane * 8 + q]; g[q] = 0.f; }
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) ss += u[q] * u[q];
  ss = wave_reduce_sum(ss);
  const float norm = sqrtf(ss);
  const float denom = fmaxf(norm, 1e-12f);
#pragma unroll
  for (int q = 0; q < 8; ++q) u[q] = u[q] / denom;
  for (int c = 0; c < C1; ++c) {
    const float dl = d_logits[(size_t)row * ld + c];
    const float* w = zs + (size_t)(lane * 8) * C1 + c;
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] += dl * w[(size_t)q * C1];
  }
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { g[q] *= temp; dot += g[q] * u[q]; }
  dot = wave_reduce_sum(dot);
  const bool clamped = !(norm > 1e-12f);
#pragma unroll
  for (int q = 0; q < 8; ++q) d_feat[(size_t)row * D + lane * 8 + q] = (clamped ? g[q] : g[q] - u[q] * dot) / denom;
}

}  // namespace

extern "C" int eod_match_label(const float* boxes, int R, const float* gt_boxes, const int32_t* gt_classes, int G, float iou_thresh,
                               int num_classes, int32_t* matched_idx, float* matched_iou, int32_t* out_classes, float* out_gt_boxes,
                               eod_stream_t stream) {
  if (!boxes || !matched_idx || !matched_iou || !out_classes || !out_gt_boxes) return EOD_ERR_NULL;
  if (G > 0 && (!gt_boxes || !gt_classes)) return EOD_ERR_NULL;
  if (R <= 0 || G < 0 || num_classes <= 0 || !(iou_thresh >= 0.f && iou_thresh <= 1.f)) return EOD_ERR_BAD_DIMS;
  MatchArgs a{boxes, R, gt_boxes, gt_classes, G, iou_thresh, num_classes, matched_idx, matched_iou, out_classes, out_gt_boxes,
              nullptr, 0, 0, nullptr};
  hipLaunchKernelGGL(match_label_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

extern "C" int eod_match_label_proposals(const float* prop_boxes, const int32_t* prop_count, int cap, const float* gt_boxes,
                                         const int32_t* gt_classes, int G, int append_gt, float iou_thresh, int num_classes,
                                         float* all_boxes, int32_t* matched_idx, float* matched_iou, int32_t* out_classes,
                                         float* out_gt_boxes, eod_stream_t stream) {
  if (!prop_boxes || !prop_count || !all_boxes || !matched_idx || !matched_iou || !out_classes || !out_gt_boxes) return EOD_ERR_NULL;
  if (G > 0 && (!gt_boxes || !gt_classes)) return EOD_ERR_NULL;
  if (cap <= 0 || G < 0 || num_classes <= 0 || !(iou_thresh >= 0.f && iou_thresh <= 1.f)) return EOD_ERR_BAD_DIMS;
  const int R = cap + (append_gt ? G : 0);
  MatchArgs a{prop_boxes, R, gt_boxes, gt_classes, G, iou_thresh, num_classes, matched_idx, matched_iou, out_classes, out_gt_boxes,
              prop_count, cap, append_gt ? 1 : 0, all_boxes};
  hipLaunchKernelGGL(match_label_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

extern "C" int eod_sample_proposals(const int32_t* classes, const float* keys, int R, int num_classes, int batch_size_per_image,
                                    float positive_fraction, int32_t* sampled_idx, int32_t* counts, eod_stream_t stream) {
  if (!classes || !keys || !sampled_idx || !counts) return EOD_ERR_NULL;
  if (R <= 0 || num_classes <= 0 || batch_size_per_image <= 0 || !(positive_fraction >= 0.f && positive_fraction <= 1.f))
    return EOD_ERR_BAD_DIMS;
  if (R > SAMPLE_MAX_R) return EOD_ERR_CAPACITY;
  SampleArgs a{classes, keys, R, num_classes, batch_size_per_image, (int)((double)batch_size_per_image * (double)positive_fraction),
               sampled_idx, counts};
  hipLaunchKernelGGL(sample_proposals_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return eod_launch_status();
}

extern "C" int eod_zs_logits(const float* feat, const float* zs_weight, int B, int D, int C1, float temp, float* logits, int ld,
                             float* featn_out, eod_stream_t stream) {
  if (!feat || !zs_weight || !logits) return EOD_ERR_NULL;
  if (B <= 0 || D != 512 || C1 < 1 || ld < C1) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(zs_logits_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, feat, zs_weight, B, C1, temp, logits, ld,
                     featn_out);
  return eod_launch_status();
}

extern "C" int eod_zs_logits_backward(const float* feat, const float* zs_weight, const float* d_logits, int ld, int B, int D, int C1,
                                      float temp, float* d_feat, eod_stream_t stream) {
  if (!feat || !zs_weight || !d_logits || !d_feat) return EOD_ERR_NULL;
  if (B <= 0 || D != 512 || C1 < 1 || ld < C1) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(zs_logits_backward_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, feat, zs_weight, d_logits, ld, B, C1,
                     temp, d_feat);
  return eod_launch_status();
}
