// Memory-bound helper kernels of the dense path (NHWC fp32): image normalisation, stem max-pool,
// GroupNorm(32)+ReLU over the concatenated FPN levels, mask predictor + sigmoid, fills.
// All are pure streaming kernels: 16-byte accesses per lane, one pass over the data.
#include "eod_common.h"
#include "../../include/eod_hip.h"

namespace {

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W,
                                                          int Hp, int Wp, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int total = Hp * Wp;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    const int y = p / Wp, x = p - y * Wp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y < H && x < W) {
      const size_t o = (size_t)y * W + x;
      const size_t plane = (size_t)H * W;
      v.x = ((float)img[o] - m0) / s0;
      v.y = ((float)img[plane + o] - m1) / s1;
      v.z = ((float)img[2 * plane + o] - m2) / s2;
    }
    *reinterpret_cast<f32x4*>(out + (size_t)p * 4) = v;
  }
}

__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                                       int OH, int OW) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * OH * OW * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    size_t t = i / c4;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int n = (int)(t / OH);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + iy) * W + ix) * C + cc * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

struct LevelOff {
  int off[EOD_MAX_LEVELS + 1];   // 5 levels x up to EOD_MAX_BATCH scenes in lock-step: every (level, scene) image has its own statistics
  int levels;
};

// GroupNorm statistics in two launches.  (1) one block per 32-row chunk of one level: thread = channel (coalesced
// 1 KiB rows), double accumulation of sum / sum of squares, 8-lane shuffle reduce to the group -> partial[chunk][group].
// (2) one thread per (level, group) adds its level's chunk partials in chunk order (deterministic) and emits mean, rstd.
#define GN_ROWS 32
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, LevelOff lo, int C, int groups,
                                                          double* __restrict__ partial) {
  EOD_CHAIN_PRIO();
  // chunk -> level
  int level = 0, first_chunk = 0;
  for (;;) {
    const int rows = lo.off[level + 1] - lo.off[level];
    const int nch = (rows + GN_ROWS - 1) / GN_ROWS;
    if ((int)blockIdx.x < first_chunk + nch || level + 1 >= lo.levels) break;
    first_chunk += nch;
    ++level;
  }
  const int r0 = lo.off[level] + ((int)blockIdx.x - first_chunk) * GN_ROWS;
  const int r1 = min(r0 + GN_ROWS, lo.off[level + 1]);
  const int cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s = 0.0, q = 0.0;
    for (int r = r0; r < r1; ++r) {
      const double v = (double)x[(size_t)r * C + c];
      s += v;
      q += v * v;
    }
    // reduce over the cpg (power of two, <= 64) consecutive channels of the group
    for (int off = 1; off < cpg; off <<= 1) {
      s += __shfl_xor(s, off, 64);
      q += __shfl_xor(q, off, 64);
    }
    if ((c % cpg) == 0) {
      const int g = c / cpg;
      partial[((size_t)blockIdx.x * groups + g) * 2 + 0] = s;
      partial[((size_t)blockIdx.x * groups + g) * 2 + 1] = q;
    }
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partial, LevelOff lo, int C, int groups, float eps,
                                                           float* __restrict__ stats) {
  EOD_CHAIN_PRIO();
  // one wave per (level, group): lanes stride over the chunks, fixed-shape shuffle tree -> deterministic
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= lo.levels * groups) return;
  const int level = i / groups, g = i - level * groups;
  int first_chunk = 0;
  for (int l = 0; l < level; ++l) first_chunk += (lo.off[l + 1] - lo.off[l] + GN_ROWS - 1) / GN_ROWS;
  const int rows = lo.off[level + 1] - lo.off[level];
  const int nch = (rows + GN_ROWS - 1) / GN_ROWS;
  double s = 0.0, q = 0.0;
  for (int c = lane; c < nch; c += 64) {
    s += partial[((size_t)(first_chunk + c) * groups + g) * 2 + 0];
    q += partial[((size_t)(first_chunk + c) * groups + g) * 2 + 1];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if (lane != 0) return;
  const double n = (double)rows * (C / groups);
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[i * 2 + 0] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void gn_apply_relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             LevelOff lo, int C, int groups, const float* __restrict__ stats) {
  EOD_CHAIN_PRIO();
  const int c4 = C >> 2;
  const int cpg = C / groups;
  const size_t total = (size_t)lo.off[lo.levels] * c4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4) * 4;
    const int row = (int)(i / c4);
    int level = 0;
    while (level + 1 < lo.levels && row >= lo.off[level + 1]) ++level;
    const int g = cc / cpg;
    const float mean = stats[(level * groups + g) * 2 + 0];
    const float rstd = stats[(level * groups + g) * 2 + 1];
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cc);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + cc);
    f32x4 o;
    o.x = fmaxf((v.x - mean) * rstd * ga.x + be.x, 0.f);
    o.y = fmaxf((v.y - mean) * rstd * ga.y + be.y, 0.f);
    o.z = fmaxf((v.z - mean) * rstd * ga.z + be.z, 0.f);
    o.w = fmaxf((v.w - mean) * rstd * ga.w + be.w, 0.f);
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}

// Finalize + apply in ONE launch (one launch less on the frame's dependent chain): a workgroup owns one 32-row chunk of one level,
// first reduces that level's chunk partials to (mean, rstd) for every group -- one wave per group, lanes striding over the chunks,
// the same fixed-shape shuffle tree and the same double arithmetic as gn_finalize_kernel, hence the same bits -- into LDS, then
// normalises its rows.  The redundant reductions are 32 groups x <= 200 partial pairs per workgroup out of L2.
// mean / rstd of every group of one level image from the chunks' partial sums (double): eight lanes per group take every eighth
// chunk in chunk order and are added by a three-step butterfly -- a fixed order, all groups at once (the one-wave-per-group form
// walked 8 groups one after the other: ~20 us of prologue in every workgroup of the apply passes).
__device__ __forceinline__ void gn_level_stats(const double* __restrict__ fwd_partial, int first_chunk, int nch, int rows, int C,
                                               int groups, float eps, float (*s_stats)[2]) {
  const int sub = threadIdx.x & 7;
  for (int g = threadIdx.x >> 3; g < groups; g += (int)(blockDim.x >> 3)) {
    double s = 0.0, q = 0.0;
    for (int c = sub; c < nch; c += 8) {
      s += fwd_partial[((size_t)(first_chunk + c) * groups + g) * 2 + 0];
      q += fwd_partial[((size_t)(first_chunk + c) * groups + g) * 2 + 1];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      q += __shfl_xor(q, off, 64);
    }
    if (sub == 0) {
      const double n = (double)rows * (C / groups);
      const double mean = s / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      s_stats[g][0] = (float)mean;
      s_stats[g][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void gn_finalize_apply_relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      LevelOff lo, int C, int groups, float eps,
                                                                      const double* __restrict__ partial) {
  EOD_CHAIN_PRIO();
  __shared__ float s_stats[64][2];
  int level = 0, first_chunk = 0;
  for (;;) {
    const int rows_l = lo.off[level + 1] - lo.off[level];
    const int nch_l = (rows_l + GN_ROWS - 1) / GN_ROWS;
    if ((int)blockIdx.x < first_chunk + nch_l || level + 1 >= lo.levels) break;
    first_chunk += nch_l;
    ++level;
  }
  const int rows = lo.off[level + 1] - lo.off[level];
  const int nch = (rows + GN_ROWS - 1) / GN_ROWS;
  gn_level_stats(partial, first_chunk, nch, rows, C, groups, eps, s_stats);
  const int r0 = lo.off[level] + ((int)blockIdx.x - first_chunk) * GN_ROWS;
  const int r1 = min(r0 + GN_ROWS, lo.off[level + 1]);
  const int c4 = C >> 2;
  const int cpg = C / groups;
  const int total = (r1 - r0) * c4;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int cc = (i % c4) * 4;
    const int row = r0 + i / c4;
    const int g = cc / cpg;
    const float mean = s_stats[g][0], rstd = s_stats[g][1];
    const size_t o = (size_t)row * C + cc;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + cc);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + cc);
    f32x4 r;
    r.x = fmaxf((v.x - mean) * rstd * ga.x + be.x, 0.f);
    r.y = fmaxf((v.y - mean) * rstd * ga.y + be.y, 0.f);
    r.z = fmaxf((v.z - mean) * rstd * ga.z + be.z, 0.f);
    r.w = fmaxf((v.w - mean) * rstd * ga.w + be.w, 0.f);
    *reinterpret_cast<f32x4*>(y + o) = r;
  }
}

// one wave per row: dot(x[row,:], w) + b -> sigmoid
__global__ __launch_bounds__(256) void mask_predictor_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                              float* __restrict__ prob, int rows, int C,
                                                              const int* __restrict__ unit_count, int unit_rows,
                                                              const int* __restrict__ out_units) {
  int R = rows;
  if (unit_count) {
    const int lim = *unit_count * unit_rows;
    R = lim < R ? lim : R;
  }
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += gridDim.x * wpb) {
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * C + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      s += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
    s = wave_reduce_sum(s);
    if (lane == 0) {
      int orow = row;
      if (out_units) {   // scatter: unit u of the compact input list is unit out_units[u] of the output
        const int u = row / unit_rows;
        orow = out_units[u] * unit_rows + (row - u * unit_rows);
      }
      prob[orow] = eod_sigmoid_precise(s + bias);
    }
  }
}

__global__ void fill_f32_kernel(float* p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_i32_kernel(int* p, int v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

inline int grid_for(size_t work, int per_block = 256, int cap = 2048) {
  size_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int eod_abi_version(void) { return 1; }

extern "C" int eod_preprocess_image(const uint8_t* img, float* out, int H, int W, int Hp, int Wp, const float* mean3,
                                    const float* std3, eod_stream_t stream) {
  if (!img || !out || !mean3 || !std3) return EOD_ERR_NULL;
  if (H <= 0 || W <= 0 || Hp < H || Wp < W) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(out)) return EOD_ERR_ALIGN;
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((size_t)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, img, out, H, W, Hp, Wp,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return eod_launch_status();
}

extern "C" int eod_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, eod_stream_t stream) {
  if (!x || !y) return EOD_ERR_NULL;
  if (C % 4 != 0 || OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1 || N <= 0) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(x) || !eod_aligned16(y)) return EOD_ERR_ALIGN;
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((size_t)N * OH * OW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C,
                     OH, OW);
  return eod_launch_status();
}

// ---- GroupNorm(32) + ReLU backward (training slices): the tower layers downstream of the memory fusion --------
// y = relu(xhat * gamma + beta), xhat = (x - mean) * rstd per (level image, group).  With dyr = dy * [y > 0], n = rows_l * C / groups:
//   dbeta_c = sum dyr, dgamma_c = sum dyr * xhat (over all rows of all levels: the layer's parameters are shared by the levels)
//   dx = rstd * (dyr * gamma_c - s1_g / n - xhat * s2_g / n),  s1_g = sum_{c in g} gamma_c a_c, s2_g = sum_{c in g} gamma_c b_c,
//   a_c / b_c = the level's sums of dyr / dyr * xhat for channel c.
// Three launches: per 32-row chunk the per-channel sums (double, row order); one workgroup adds them per level in chunk order and
// over the levels in level order (deterministic); the apply pass.  mean / rstd are recomputed from the forward's own partial sums.
__device__ __forceinline__ void gn_chunk_of_block(const LevelOff& lo, int& level, int& first_chunk) {
  level = 0;
  first_chunk = 0;
  for (;;) {
    const int rows_l = lo.off[level + 1] - lo.off[level];
    const int nch_l = (rows_l + GN_ROWS - 1) / GN_ROWS;
    if ((int)blockIdx.x < first_chunk + nch_l || level + 1 >= lo.levels) break;
    first_chunk += nch_l;
    ++level;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy, LevelOff lo, int C, int groups, float eps,
                                                              const double* __restrict__ fwd_partial, double* __restrict__ part) {
  __shared__ float s_stats[64][2];
  int level, first_chunk;
  gn_chunk_of_block(lo, level, first_chunk);
  const int rows = lo.off[level + 1] - lo.off[level];
  gn_level_stats(fwd_partial, first_chunk, (rows + GN_ROWS - 1) / GN_ROWS, rows, C, groups, eps, s_stats);
  const int r0 = lo.off[level] + ((int)blockIdx.x - first_chunk) * GN_ROWS;
  const int r1 = min(r0 + GN_ROWS, lo.off[level + 1]);
  const int cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float mean = s_stats[c / cpg][0], rstd = s_stats[c / cpg][1];
    double a = 0.0, b = 0.0;
    for (int r = r0; r < r1; ++r) {
      const size_t o = (size_t)r * C + c;
      const float dyr = y[o] > 0.f ? dy[o] : 0.f;
      const float xh = (x[o] - mean) * rstd;
      a += (double)dyr;
      b += (double)dyr * (double)xh;
    }
    part[((size_t)blockIdx.x * C + c) * 2 + 0] = a;
    part[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

// per level the chunk sums -> level totals [levels][C][2]; over the levels -> dgamma, dbeta.  A workgroup owns 32 channels; its eight
// sub-lanes per channel take every eighth chunk of a level (in chunk order) and are added in sub-lane order through LDS: a fixed
// order of summation, and 8 x C / 32 times the one-workgroup form's parallelism (66 -> ~10 us for the tower's 267 chunks).
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const double* __restrict__ part, LevelOff lo, int C, double* __restrict__ level_tot,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s_part[8][32][2];
  const int cl = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double ga = 0.0, gb = 0.0;
  int chunk0 = 0;
  for (int l = 0; l < lo.levels; ++l) {
    const int nch = (lo.off[l + 1] - lo.off[l] + GN_ROWS - 1) / GN_ROWS;
    double a = 0.0, b = 0.0;
    if (c < C) {
      for (int k = sub; k < nch; k += 8) {
        a += part[((size_t)(chunk0 + k) * C + c) * 2 + 0];
        b += part[((size_t)(chunk0 + k) * C + c) * 2 + 1];
      }
    }
    s_part[sub][cl][0] = a;
    s_part[sub][cl][1] = b;
    __syncthreads();
    if (sub == 0 && c < C) {
      a = 0.0; b = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a += s_part[j][cl][0];
        b += s_part[j][cl][1];
      }
      level_tot[((size_t)l * C + c) * 2 + 0] = a;
      level_tot[((size_t)l * C + c) * 2 + 1] = b;
      ga += a;
      gb += b;
    }
    __syncthreads();
    chunk0 += nch;
  }
  if (sub == 0 && c < C) {
    dbeta[c] = (float)ga;
    dgamma[c] = (float)gb;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, LevelOff lo, int C, int groups, float eps,
                                                            const double* __restrict__ fwd_partial, const double* __restrict__ level_tot,
                                                            float* __restrict__ dx) {
  __shared__ float s_stats[64][2];
  __shared__ double s_sum[64][2];
  int level, first_chunk;
  gn_chunk_of_block(lo, level, first_chunk);
  const int rows = lo.off[level + 1] - lo.off[level];
  gn_level_stats(fwd_partial, first_chunk, (rows + GN_ROWS - 1) / GN_ROWS, rows, C, groups, eps, s_stats);
  const int cpg = C / groups;
  // s1_g, s2_g of this level: gamma-weighted sums of the level totals over the group's channels, in channel order
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      const int c = g * cpg + j;
      s1 += (double)gamma[c] * level_tot[((size_t)level * C + c) * 2 + 0];
      s2 += (double)gamma[c] * level_tot[((size_t)level * C + c) * 2 + 1];
    }
    s_sum[g][0] = s1;
    s_sum[g][1] = s2;
  }
  __syncthreads();
  const int r0 = lo.off[level] + ((int)blockIdx.x - first_chunk) * GN_ROWS;
  const int r1 = min(r0 + GN_ROWS, lo.off[level + 1]);
  const double inv_n = 1.0 / ((double)rows * cpg);
  const int total = (r1 - r0) * C;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int c = i % C;
    const size_t o = (size_t)(r0 + i / C) * C + c;
    const int g = c / cpg;
    const float mean = s_stats[g][0], rstd = s_stats[g][1];
    const float dyr = y[o] > 0.f ? dy[o] : 0.f;
    const float xh = (x[o] - mean) * rstd;
    const float m1 = (float)(s_sum[g][0] * inv_n), m2 = (float)(s_sum[g][1] * inv_n);
    dx[o] = rstd * (dyr * gamma[c] - m1 - xh * m2);
  }
}

extern "C" size_t eod_groupnorm_backward_workspace_bytes(const int32_t* level_off_host, int levels, int C) {
  if (!level_off_host || levels < 1 || levels > EOD_MAX_LEVELS || C <= 0) return 0;
  size_t chunks = 0;
  for (int i = 0; i < levels; ++i) chunks += (size_t)(level_off_host[i + 1] - level_off_host[i] + GN_ROWS - 1) / GN_ROWS;
  return (chunks + (size_t)levels) * C * 2 * sizeof(double);
}

extern "C" int eod_groupnorm_relu_backward(const float* x, const float* y, const float* dy, const float* gamma, const int32_t* level_off_host,
                                           int levels, int C, int groups, float eps, const float* fwd_stats, void* workspace, float* dx,
                                           float* dgamma, float* dbeta, eod_stream_t stream) {
  if (!x || !y || !dy || !gamma || !level_off_host || !fwd_stats || !workspace || !dx || !dgamma || !dbeta) return EOD_ERR_NULL;
  if (levels < 1 || levels > EOD_MAX_LEVELS || groups < 1 || groups > 64 || C % groups != 0) return EOD_ERR_BAD_DIMS;
  LevelOff lo{};
  lo.levels = levels;
  for (int i = 0; i <= levels; ++i) lo.off[i] = level_off_host[i];
  int chunks = 0;
  for (int i = 0; i < levels; ++i) {
    if (lo.off[i + 1] <= lo.off[i]) return EOD_ERR_BAD_DIMS;
    chunks += (lo.off[i + 1] - lo.off[i] + GN_ROWS - 1) / GN_ROWS;
  }
  // the forward's workspace: [2 * levels * groups floats | pad to 8 B | chunks * groups * 2 doubles]
  const double* fwd_partial = reinterpret_cast<const double*>(reinterpret_cast<const char*>(fwd_stats) +
                                                              ((size_t)2 * levels * groups * sizeof(float) + 7) / 8 * 8);
  double* part = static_cast<double*>(workspace);
  double* level_tot = part + (size_t)chunks * C * 2;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(chunks), dim3(256), 0, s, x, y, dy, lo, C, groups, eps, fwd_partial, part);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3((C + 31) / 32), dim3(256), 0, s, part, lo, C, level_tot, dgamma, dbeta);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks), dim3(256), 0, s, x, y, dy, gamma, lo, C, groups, eps, fwd_partial, level_tot, dx);
  return eod_launch_status();
}

extern "C" size_t eod_groupnorm_workspace_bytes(const int32_t* level_off_host, int levels, int groups) {
  if (!level_off_host || levels < 1 || levels > EOD_MAX_LEVELS) return 0;
  size_t chunks = 0;
  for (int i = 0; i < levels; ++i) chunks += (size_t)(level_off_host[i + 1] - level_off_host[i] + GN_ROWS - 1) / GN_ROWS;
  return ((size_t)2 * levels * groups * sizeof(float) + 7) / 8 * 8 + chunks * groups * 2 * sizeof(double);
}

extern "C" size_t eod_groupnorm_partial_offset(int levels, int groups) {
  return ((size_t)2 * levels * groups * sizeof(float) + 7) / 8 * 8;
}

extern "C" int eod_groupnorm_relu(const float* x, float* y, const float* gamma, const float* beta, const int32_t* level_off_host,
                                  int levels, int C, int groups, float eps, float* stats, int partial_ready, eod_stream_t stream) {
  if (!x || !y || !gamma || !beta || !level_off_host || !stats) return EOD_ERR_NULL;
  if (levels < 1 || levels > EOD_MAX_LEVELS || C % groups != 0 || C % 4 != 0 || (C / groups) % 4 != 0) return EOD_ERR_BAD_DIMS;
  LevelOff lo{};
  lo.levels = levels;
  for (int i = 0; i <= levels; ++i) lo.off[i] = level_off_host[i];
  for (int i = 0; i < levels; ++i)
    if (lo.off[i + 1] <= lo.off[i]) return EOD_ERR_BAD_DIMS;
  int chunks = 0;
  for (int i = 0; i < levels; ++i) chunks += (lo.off[i + 1] - lo.off[i] + GN_ROWS - 1) / GN_ROWS;
  // stats layout: [2*levels*groups floats | pad to 8 B | chunks*groups*2 doubles]
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(stats) + ((size_t)2 * levels * groups * sizeof(float) + 7) / 8 * 8);
  const int cpg = C / groups;
  if (cpg > 64 || (cpg & (cpg - 1)) != 0 || 256 % cpg != 0) return EOD_ERR_BAD_DIMS;
  if (!partial_ready)
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, x, lo, C, groups, partial);
  if (groups <= 64) {
    hipLaunchKernelGGL(gn_finalize_apply_relu_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, x, y, gamma, beta, lo, C, groups,
                       eps, partial);
  } else {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((levels * groups + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, lo, C, groups,
                       eps, stats);
    hipLaunchKernelGGL(gn_apply_relu_kernel, dim3(grid_for((size_t)lo.off[levels] * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                       gamma, beta, lo, C, groups, stats);
  }
  return eod_launch_status();
}

extern "C" int eod_mask_predictor_sigmoid(const float* x, const float* w, float bias, float* prob, int rows, int C,
                                          const int32_t* unit_count, int unit_rows, const int32_t* out_units, eod_stream_t stream) {
  if (!x || !w || !prob) return EOD_ERR_NULL;
  if (rows <= 0 || C % 4 != 0 || (out_units && unit_rows <= 0)) return EOD_ERR_BAD_DIMS;
  hipLaunchKernelGGL(mask_predictor_kernel, dim3(grid_for((size_t)rows, 4, 8192)), dim3(256), 0, (hipStream_t)stream, x, w, bias, prob,
                     rows, C, unit_count, unit_rows, out_units);
  return eod_launch_status();
}

extern "C" int eod_fill_f32(float* p, float v, size_t n, eod_stream_t stream) {
  if (!p) return EOD_ERR_NULL;
  if (n == 0) return EOD_OK;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  return eod_launch_status();
}
extern "C" int eod_fill_i32(int32_t* p, int32_t v, size_t n, eod_stream_t stream) {
  if (!p) return EOD_ERR_NULL;
  if (n == 0) return EOD_OK;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  return eod_launch_status();
}
