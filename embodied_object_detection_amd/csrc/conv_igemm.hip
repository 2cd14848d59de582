// eod_conv2d: argument checks, tile / split-K planning, arithmetic-mode switch and the C ABI of the implicit-GEMM convolution.
//
//   y[m][n] = act( (sum_k A[m][k] * Wt[n][k] + bias[n]) * out_scale + res[m][n] )
//
// A[m][k] is the im2col view of the NHWC input (m = (img, oy, ox), k = (ky, kx, c), c fastest), Wt is [Cout][Kpad].  The kernels
// live in conv_fp32.hip (fp32 MFMA, default) and conv_bf16x3.hip (opt-in split-bf16 arithmetic); this file picks one, sizes the grid and, for small problems, splits K into fp32 slabs that
// conv_splitk_reduce_kernel sums in a fixed order before the fused epilogue (deterministic, no atomics).
#include "conv_common.h"
#include "../../include/eod_hip.h"
#include <atomic>
#include <cstdlib>

namespace {

using namespace eodconv;

// Sums the K slabs in slab order and applies the fused epilogue.  VEC = 4: one float4 of four consecutive output channels per
// thread and slab (Cout % 4 == 0); VEC = 1 for the odd-width heads.  32-bit indices (check_desc bounds rows x Cout below 2^31).
template <int VEC>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs p) {
  EOD_CHAIN_PRIO();
  const int M = conv_row_limit(p, p.M);
  const unsigned per_row = (unsigned)p.Cout / VEC;
  const unsigned total = (unsigned)M * per_row;
  const size_t slab = (size_t)p.M * p.Cout;
  for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const unsigned m = fdiv(idx, p.div_row);
    if (!conv_row_active(p, (int)m)) continue;          // segmented counts: the tile that would have written this row's slabs left early
    const unsigned n = (idx - m * per_row) * VEC;
    const float* src = p.partial + (size_t)m * p.Cout + n;
    if (VEC == 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      // four slabs at a time: independent loads in flight together, added in slab order
      for (int zb = 0; zb < p.splitk; zb += 4) {
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (zb + j < p.splitk) t[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(zb + j) * slab);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (zb + j < p.splitk) v += t[j];
      }
      epilogue_store<true>(p, v.x, (int)m, (int)n);
      epilogue_store<true>(p, v.y, (int)m, (int)n + 1);
      epilogue_store<true>(p, v.z, (int)m, (int)n + 2);
      epilogue_store<true>(p, v.w, (int)m, (int)n + 3);
    } else {
      float v = 0.f;
      for (int z = 0; z < p.splitk; ++z) v += src[z * slab];
      epilogue_store<true>(p, v, (int)m, (int)n);
    }
  }
}

// The slab reduce of a pyramid-mode layer that is followed by GroupNorm (CenterNet tower, centernet_head.py:76-79): the same sums
// in slab order + epilogue, one workgroup per 32-row chunk of one level (eod_groupnorm_relu's chunks), and the chunk's GroupNorm
// partial sums on the way -> partial[chunk][group][2] -- so that the statistics launch is not needed.  Memory access as in the
// plain reduce: a thread owns one float4 column (4 consecutive channels) and every (256 / (C/4))-th row of the chunk; sum and sum
// of squares are accumulated in double per thread (rows in ascending order, the four channels in order), combined over the
// threads of a channel group with shuffles and over the row subsets through LDS in subset order: deterministic.  (The first
// version walked one channel per thread over 32 rows x slabs: 44 us against 12 + 11 for the two separate launches.)
#define EOD_GN_ROWS 32
__global__ __launch_bounds__(256) void conv_splitk_reduce_gn_kernel(ConvArgs p) {
  EOD_CHAIN_PRIO();
  __shared__ double red[4][64][2];           // [row subset][group][sum, sum of squares]; groups <= 64, subsets <= 4
  int level = 0, first_chunk = 0;
  for (;;) {
    const int rows = p.lv_off[level + 1] - p.lv_off[level];
    const int nch = (rows + EOD_GN_ROWS - 1) / EOD_GN_ROWS;
    if ((int)blockIdx.x < first_chunk + nch || level + 1 >= p.nlv) break;
    first_chunk += nch;
    ++level;
  }
  const int r0 = p.lv_off[level] + ((int)blockIdx.x - first_chunk) * EOD_GN_ROWS;
  const int r1 = min(r0 + EOD_GN_ROWS, p.lv_off[level + 1]);
  const int C = p.Cout;
  const int c4 = C >> 2;                     // float4 columns per row: 64 for the tower (checked on the host: 256 % c4 == 0)
  const int nsub = 256 / c4;                 // row subsets: 4
  const int col = threadIdx.x % c4, sub = threadIdx.x / c4;
  const int cpg4 = (C / p.gn_groups) >> 2;   // float4 columns per channel group: 2
  const size_t slab = (size_t)p.M * C;
  double s = 0.0, q = 0.0;
  for (int r = r0 + sub; r < r1; r += nsub) {
    const float* src = p.partial + (size_t)r * C + col * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int zb = 0; zb < p.splitk; zb += 4) {
      f32x4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (zb + j < p.splitk) t[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(zb + j) * slab);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (zb + j < p.splitk) v += t[j];
    }
    const double o0 = (double)epilogue_store<true>(p, v.x, r, col * 4 + 0);
    const double o1 = (double)epilogue_store<true>(p, v.y, r, col * 4 + 1);
    const double o2 = (double)epilogue_store<true>(p, v.z, r, col * 4 + 2);
    const double o3 = (double)epilogue_store<true>(p, v.w, r, col * 4 + 3);
    s += o0; q += o0 * o0;
    s += o1; q += o1 * o1;
    s += o2; q += o2 * o2;
    s += o3; q += o3 * o3;
  }
  for (int off = 1; off < cpg4; off <<= 1) {      // the columns of one group are adjacent lanes (c4 is a multiple of cpg4, 64 of c4)
    s += __shfl_xor(s, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if ((col % cpg4) == 0) {
    red[sub][col / cpg4][0] = s;
    red[sub][col / cpg4][1] = q;
  }
  __syncthreads();
  if ((int)threadIdx.x < p.gn_groups) {
    double ts = 0.0, tq = 0.0;
    for (int u = 0; u < nsub; ++u) {
      ts += red[u][threadIdx.x][0];
      tq += red[u][threadIdx.x][1];
    }
    p.gn_partial[((size_t)blockIdx.x * p.gn_groups + threadIdx.x) * 2 + 0] = ts;
    p.gn_partial[((size_t)blockIdx.x * p.gn_groups + threadIdx.x) * 2 + 1] = tq;
  }
}

inline int big_tile_env() {
  static const int v = [] {
    const char* e = getenv("EOD_CONV_BIG_TILE");
    return e ? atoi(e) : 0;
  }();
  return v;
}

inline int default_bk() {
  static const int v = [] {
    const char* e = getenv("EOD_CONV_BK");
    // measured end to end on the final schedule (profiles/r01_bench_640_default.log era): BK=32 140.8 frames/s / 119.6 TFLOP/s on the
    // mask GEMM vs BK=64 139.4 / 117.4 (smaller LDS tile -> one more workgroup per CU; the halved barrier count of BK=64 stopped
    // paying once the K-loop bookkeeping became incremental)
    return (e && atoi(e) == 64) ? 64 : 32;
  }();
  return v;
}

// Process-wide arithmetic mode of eod_conv2d (eod_set_conv_math / EOD_CONV_MATH): 0 fp32 MFMA, 1 bf16x3 split.
std::atomic<int>& math_mode() {
  static std::atomic<int> v([] {
    const char* e = getenv("EOD_CONV_MATH");
    return (e && e[0] == 'b') ? EOD_MATH_BF16X3 : EOD_MATH_FP32;
  }());
  return v;
}

struct Plan {
  int tile;  // 1=128x128 2=128x64 3=64x64
  int bk;    // K chunk staged per barrier pair: 32 or 64
  int glds;  // 2: bf16x3 split kernel (BK = 32); 0: fp32 MFMA kernel
  int bm, bn, tiles_m, tiles_n, splitk, cps, nchunks;
  int wavek;  // 0, or the number of waves (4 / 8) of the 32x32-tile kernel that splits K over the waves of a workgroup
};

inline int midsplit_env() {          // experiment knob: EOD_CONV_MIDSPLIT=0 turns the 256..1024-tile split-K rule off
  static const int v = [] {
    const char* e = getenv("EOD_CONV_MIDSPLIT");
    return e ? atoi(e) : 1;
  }();
  return v;
}

inline int wavek_env() {
  static const int v = [] {
    const char* e = getenv("EOD_CONV_WAVEK");
    return e ? atoi(e) : 1;
  }();
  return v;
}

Plan make_plan(const EodConvDesc* d, int M, int nchunks32) {
  // 256x128: bf16x3 8-wave kernel only; 64x256: fused mask-head tail (out_mode 2) only
  static const int cfg[5][2] = {{128, 128}, {128, 64}, {64, 64}, {256, 128}, {64, 256}};
  Plan pl{};
  // Measured on MI355X (tools/conv_bench.py, profiles/r01_conv_bench.log): the 64x64 tile (7 waves/SIMD, finest
  // tile quantisation over 256 CUs) is the fastest or ties on every shape of this path, including the
  // 50k x 256 x 2304 mask-head GEMM (107 vs 92 TFLOP/s for 128x128).  The larger tiles stay selectable.
  int pick = 2;
  const int ft = d->force_tile % 10, fbk = d->force_tile / 10;   // force_tile = tile + 10 (BK 32) / + 20 (BK 64)
  if (ft >= 1 && ft <= 3) pick = ft - 1;
  else if (ft == 4 && fbk == 5 && !d->tap4 && !d->in_relu) pick = 3;
  else if (M >= 32768 && big_tile_env() >= 1 && big_tile_env() <= 3) pick = big_tile_env() - 1;   // experiment knob
  const bool bk64_ok = !d->tap4 && d->Cin % 64 == 0 && d->Kpad % 64 == 0;
  pl.glds = (fbk == 5 && !d->tap4 && !d->in_relu) ? 2 : 0;
  if (d->force_tile == 0 && math_mode().load(std::memory_order_relaxed) == EOD_MATH_BF16X3 && !d->tap4 && !d->in_relu) {
    pl.glds = 2;
    const long t128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    const long t256 = (long)((M + 255) / 256) * ((d->Cout + 127) / 128);
    // measured (profiles/r01_bf16x3_accuracy_speed.log): 256x128 (8 waves, one workgroup per CU) once it fills the chip,
    // 128x128 once it fills two workgroups per CU, else the finest tile
    pick = t256 >= 256 ? 3 : (t128 >= 512 ? 0 : 2);
  }
  if (d->out_mode == 2) {   // one workgroup owns all 256 channels of a deconv quadrant; fp32 kernel in every arithmetic mode
    pick = 4;
    pl.glds = 0;
  }
  if (d->gate) {            // the gated epilogue lives in the 64x64 fp32 tile, the wave-K kernel and the slab reduces
    pick = 2;
    pl.glds = 0;
  }
  pl.bk = (pl.glds || d->out_mode == 2) ? 32 : ((fbk == 2 && bk64_ok) ? 64 : (fbk == 1 ? 32 : (bk64_ok && default_bk() == 64 ? 64 : 32)));
  const int nchunks = d->Kpad / pl.bk;
  (void)nchunks32;
  pl.nchunks = nchunks;
  pl.tile = pick + 1;
  pl.bm = cfg[pick][0];
  pl.bn = cfg[pick][1];
  pl.tiles_m = (M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (d->Cout + pl.bn - 1) / pl.bn;
  // split-K is decided on `plan_rows` when given (a batch planned like one image: identical K walk, bitwise equal results)
  const int Mp = (d->plan_rows > 0 && d->plan_rows < M) ? d->plan_rows : M;
  const long tiles = (long)((Mp + pl.bm - 1) / pl.bm) * pl.tiles_n;
  int splitk = 1;
  if (d->out_mode == 2) {
    splitk = 1;
  } else if (d->force_splitk > 0) {
    splitk = d->force_splitk;
  } else if (pl.glds == 0 && !d->tap4 && nchunks >= 8 &&
             ((d->force_tile == 6 || d->force_tile == 7) ||
              (tiles < 256 && nchunks <= 192 && pl.bk == 32 && d->force_tile == 0 && wavek_env() &&
               ((Mp <= 512 && nchunks / (nchunks >= 64 ? 8 : 4) >= 8) || nchunks >= 128)))) {
    // Few rows and a deep K (ResNet layer4 at batch 1, FPN lateral / output 5, P6/P7, the box heads' 1024-wide FC layers): K is
    // split over the waves of a workgroup that owns a 32x32 tile (conv_wavek_kernel) -- no slabs, no reduce launch.  Eight waves
    // once a wave's share would exceed 16 chunks.  Measured per layer against the slab form (rocprofv3, 640x640 trunk,
    // profiles/r03_*): it wins or ties where a wave gets >= 8 chunks and the row count is small (layer4 conv2 25 against 38 us,
    // the FC layers equal with one launch less); with many rows (layer2/3: 32x32 tiles re-read the weights per tile) or a
    // shallow K per wave (layer4 conv3) the slabs are faster (31 against 42 us, 21 against 34 us) and keep the layer.
    pl.wavek = d->force_tile == 6 ? 4 : (d->force_tile == 7 ? 8 : (nchunks >= 64 ? 8 : 4));     // force_tile 6 / 7: benchmarks, tests
    pl.bk = 32;
    pl.nchunks = d->Kpad / 32;
    pl.bm = pl.bn = 32;
    pl.tiles_m = (M + 31) / 32;
    pl.tiles_n = (d->Cout + 31) / 32;
  } else if (tiles < 256 && nchunks >= 8) {
    // Deep-K, few-row GEMMs (the box head's fc1: 320 x 12544 -> 1024, 80 tiles, 392 chunks) are a serial chain of ~1.5 us chunk
    // round trips per workgroup: 7 workgroups per CU instead of 2 shorten the chain 3.5x (105 -> ~35 us on the cascade's critical
    // path) for 4x the slab traffic of a 10 us reduce.
    const int target = (nchunks >= 128 && Mp <= 512) ? 1792 : 512;
    int want = (int)((target + tiles - 1) / tiles);
    int maxs = nchunks / 4;
    splitk = want < maxs ? want : maxs;
    if (splitk < 1) splitk = 1;
  } else if (tiles >= 256 && tiles <= 1024 && d->m_count == nullptr && nchunks >= 18 && midsplit_env()) {
    // A few hundred tiles on 256 CUs: every workgroup is resident at once and a CU works through its workgroups' MFMAs one
    // after the other, so the launch takes ceil(tiles / 256) tile times -- the CenterNet tower's 536 tiles take 3 where 2.09
    // would do.  Split-K makes the units finer: time ~ ceil(tiles * s / 256) / s (+ ~0.12 of a tile for the slab reduce).
    // Measured (tools/conv_bench.py tower, reduce included): 8 556 rows x 256 x 2304: s = 1 122-135 us, 2 109-118, 3 106-113,
    // 4 110.  Static row counts only: a device-side count (the mask passes) is not known here, and their two variants
    // (lazy / all proposals) must walk K in the same order to stay bitwise equal.
    double best = 1e30;
    for (int sct = 1; sct <= 4; ++sct) {
      if (nchunks / sct < 9) break;
      const double est = (double)((tiles * sct + 255) / 256) / sct + (sct > 1 ? 0.12 : 0.0);
      if (est < best - 1e-9) {
        best = est;
        splitk = sct;
      }
    }
  }
  if (splitk > nchunks) splitk = nchunks;
  pl.cps = (nchunks + splitk - 1) / splitk;
  pl.splitk = (nchunks + pl.cps - 1) / pl.cps;
  return pl;
}

int total_rows(const EodConvDesc* d) { return d->levels > 0 ? d->level_off[d->levels] : d->N * d->OH * d->OW; }

int check_desc(const EodConvDesc* d) {
  if (!d || !d->x || !d->w || !d->y) return EOD_ERR_NULL;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return EOD_ERR_BAD_DIMS;
  if (d->levels <= 0 && (d->H <= 0 || d->W <= 0 || d->OH <= 0 || d->OW <= 0)) return EOD_ERR_BAD_DIMS;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->pad < 0) return EOD_ERR_BAD_DIMS;
  if (d->Kpad % 32 != 0 || d->Kpad < d->KH * d->KW * d->Cin) return EOD_ERR_BAD_DIMS;
  if (d->tap4) {
    if (d->Cin != 4) return EOD_ERR_BAD_DIMS;
  } else {
    if (d->Cin % 32 != 0 || d->Kpad != d->KH * d->KW * d->Cin) return EOD_ERR_BAD_DIMS;
  }
  if (d->levels > 0) {
    // pyramid mode: stride-1 'same' conv over up to 5 level images stored back to back
    if (d->levels > EOD_MAX_LEVELS || d->N != 1 || d->stride != 1 || d->KH != d->KW || d->pad != d->KH / 2 || d->tap4 || d->out_mode != 0 ||
        d->res_mode == 2 || d->level_off[0] != 0)
      return EOD_ERR_BAD_DIMS;
    for (int l = 0; l < d->levels; ++l)
      if (d->level_h[l] <= 0 || d->level_w[l] <= 0 || d->level_off[l + 1] - d->level_off[l] != d->level_h[l] * d->level_w[l])
        return EOD_ERR_BAD_DIMS;
  } else {
    if (d->OH != (d->H + 2 * d->pad - d->KH) / d->stride + 1) return EOD_ERR_BAD_DIMS;
    if (d->OW != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return EOD_ERR_BAD_DIMS;
  }
  // operand buffers are addressed with 32-bit byte offsets (buffer descriptors): < 2 GiB each
  if (d->levels <= 0 && (long)d->N * d->H * d->W * d->Cin >= (1L << 29)) return EOD_ERR_BAD_DIMS;
  if ((long)d->Cout * d->Kpad >= (1L << 29)) return EOD_ERR_BAD_DIMS;
  if (d->KH * d->KW > 64) return EOD_ERR_BAD_DIMS;
  if ((long)total_rows(d) * (d->Cout > d->Cin ? d->Cout : d->Cin) >= (1L << 31)) return EOD_ERR_BAD_DIMS;
  if (d->out_mode == 1 && (d->Cout % 4 != 0)) return EOD_ERR_BAD_DIMS;
  if (d->out_mode == 2) {
    if (d->Cout != 1024 || d->KH != 1 || d->KW != 1 || d->stride != 1 || d->tap4 || d->levels > 0 || d->res_mode != 0 || d->in_relu ||
        d->force_splitk > 1)
      return EOD_ERR_BAD_DIMS;
    if (!d->fuse_w) return EOD_ERR_NULL;
  }
  if (d->out_mode < 0 || d->out_mode > 2) return EOD_ERR_BAD_DIMS;
  if (d->res_mode != 0 && !d->res) return EOD_ERR_NULL;
  if (d->res_mode == 2 && ((d->OH & 1) || (d->OW & 1))) return EOD_ERR_BAD_DIMS;
  if (d->m_count && d->m_unit <= 0) return EOD_ERR_BAD_DIMS;
  if (d->m_segments > 1) {
    // B unit lists back to back: whole images per list, a count per list
    if (!d->m_count || d->levels > 0 || d->out_mode == 2 || d->m_segments > EOD_MAX_BATCH || d->N % d->m_segments != 0)
      return EOD_ERR_BAD_DIMS;
    if (((long)(d->N / d->m_segments) * d->OH * d->OW) % d->m_unit != 0) return EOD_ERR_BAD_DIMS;
  }
  if (d->gate && (d->out_mode != 0 || d->split_n != 0 || d->gn_partial)) return EOD_ERR_BAD_DIMS;
  if (d->gate && d->force_tile != 0 && d->force_tile % 10 != 3 && d->force_tile != 6 && d->force_tile != 7) return EOD_ERR_BAD_DIMS;
  if (d->gate && d->force_tile / 10 == 5) return EOD_ERR_BAD_DIMS;          // no gated epilogue in the bf16x3 kernels
  if (d->lds_reserve < 0 || d->lds_reserve > 48 * 1024) return EOD_ERR_BAD_DIMS;
  if (d->split_n != 0) {
    if (!d->y2) return EOD_ERR_NULL;
    if (d->split_n < 0 || d->split_n >= d->Cout || d->out_mode != 0 || d->res_mode != 0 || d->levels > 0 || d->gn_partial)
      return EOD_ERR_BAD_DIMS;
  }
  if (d->gn_partial) {
    // statistics for a following GroupNorm: pyramid mode, static row count, plain epilogue, channel groups of a power of two <= 64
    const int cpg = d->gn_groups > 0 ? d->Cout / d->gn_groups : 0;
    if (d->levels <= 0 || d->m_count || d->out_mode != 0 || d->gn_groups <= 0 || d->gn_groups > 64 || d->Cout % d->gn_groups != 0 ||
        cpg > 64 || (cpg & (cpg - 1)) != 0 || cpg < 4 || d->Cout % 4 != 0 || d->Cout > 1024 || 256 % (d->Cout / 4) != 0 ||
        64 % (cpg / 4) != 0)
      return EOD_ERR_BAD_DIMS;
  }
  if (!eod_aligned16(d->x) || !eod_aligned16(d->w) || !eod_aligned16(d->w_split)) return EOD_ERR_ALIGN;
  return EOD_OK;
}

}  // namespace

extern "C" int eod_set_conv_math(int mode) {
  if (mode != EOD_MATH_FP32 && mode != EOD_MATH_BF16X3) return EOD_ERR_BAD_DIMS;
  return math_mode().exchange(mode);
}

extern "C" int eod_get_conv_math(void) { return math_mode().load(); }

extern "C" size_t eod_conv_split_weights_bytes(int Cout, int Kpad) {
  return (Cout > 0 && Kpad > 0 && Kpad % 32 == 0) ? (size_t)Cout * Kpad * 6 : 0;
}

extern "C" int eod_conv_split_weights_bf16x3(const float* w, int Cout, int Kpad, void* out, eod_stream_t stream) {
  if (!w || !out) return EOD_ERR_NULL;
  if (Cout <= 0 || Kpad <= 0 || Kpad % 32 != 0 || (long)Cout * Kpad * 6 >= (1L << 32)) return EOD_ERR_BAD_DIMS;
  if (!eod_aligned16(w) || !eod_aligned16(out)) return EOD_ERR_ALIGN;
  launch_split_weights(w, out, Cout, Kpad, static_cast<hipStream_t>(stream));
  return eod_launch_status();
}

extern "C" int eod_conv2d_gn_fused(const EodConvDesc* d) {
  EodConvDesc t = *d;
  t.gn_partial = nullptr;
  if (check_desc(&t) != EOD_OK) return 0;
  return make_plan(&t, total_rows(&t), t.Kpad / 32).splitk > 1 ? 1 : 0;
}

extern "C" size_t eod_conv2d_workspace_bytes(const EodConvDesc* d) {
  if (check_desc(d) != EOD_OK) return 0;
  const int M = total_rows(d);
  const int nchunks = d->Kpad / 32;
  const Plan pl = make_plan(d, M, nchunks);
  if (pl.splitk <= 1) return 0;
  return (size_t)pl.splitk * M * d->Cout * sizeof(float);
}

extern "C" int eod_conv2d(const EodConvDesc* d, eod_stream_t stream) {
  const int st = check_desc(d);
  if (st != EOD_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvArgs a{};
  a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->res; a.y = d->y;
  a.partial = d->workspace;
  a.m_count = d->m_count; a.m_unit = d->m_unit;
  a.m_segs = d->m_segments > 1 ? d->m_segments : 1;
  a.seg_rows = a.m_segs > 1 ? (d->N / d->m_segments) * d->OH * d->OW : 0;
  a.fuse_w = d->fuse_w; a.out_units = d->out_units; a.fuse_b = d->fuse_b;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.Kpad = d->Kpad;
  a.M = total_rows(d);
  a.div_ow = eod_make_fastdiv((unsigned)(d->OW > 0 ? d->OW : 1));
  a.div_oh = eod_make_fastdiv((unsigned)(d->OH > 0 ? d->OH : 1));
  a.div_row = eod_make_fastdiv((unsigned)(d->Cout % 4 == 0 ? d->Cout / 4 : d->Cout));   // split-K reduce: work items per output row
  a.div_cd = eod_make_fastdiv((unsigned)((d->Cout >> 2) > 0 ? (d->Cout >> 2) : 1));
  {
    const size_t xe = d->levels > 0 ? (size_t)d->level_off[d->levels] * d->Cin : (size_t)d->N * d->H * d->W * d->Cin;
    a.x_bytes = (unsigned)(xe * sizeof(float));
    a.w_bytes = (unsigned)((size_t)d->Cout * d->Kpad * sizeof(float));
    a.w3 = d->w_split;
    a.w3_bytes = (unsigned)((size_t)d->Cout * d->Kpad * 6);
  }
  a.nlv = d->levels > 0 ? d->levels : 0;
  for (int l = 0; l < a.nlv; ++l) {
    a.lv_off[l] = d->level_off[l];
    a.lv_h[l] = d->level_h[l];
    a.lv_w[l] = d->level_w[l];
  }
  if (a.nlv) a.lv_off[a.nlv] = d->level_off[a.nlv];
  a.relu = d->relu; a.res_mode = d->res_mode; a.in_relu = d->in_relu; a.out_mode = d->out_mode;
  a.gate = d->gate;
  a.out_scale = d->out_scale;
  a.gn_partial = d->gn_partial;
  a.gn_groups = d->gn_groups;
  a.y2 = d->y2;
  a.split_n = d->split_n;
  const Plan pl = make_plan(d, a.M, d->Kpad / 32);
  if (d->gn_partial && pl.splitk <= 1) return EOD_ERR_BAD_DIMS;      // the statistics ride on the slab reduce (see eod_conv2d_gn_fused)
  a.nchunks = pl.nchunks;
  a.splitk = pl.splitk; a.cps = pl.cps; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n;
  if (pl.splitk > 1) {
    const size_t need = (size_t)pl.splitk * a.M * a.Cout * sizeof(float);
    if (!d->workspace || d->workspace_bytes < need) return EOD_ERR_CAPACITY;
  }
  dim3 grid(pl.tiles_m * pl.tiles_n, pl.splitk);
  if (pl.wavek) launch_conv_wavek(a, pl.wavek, grid, s);
  else if (pl.glds == 2) launch_conv_bf16x3(a, pl.tile, grid, s);
  else launch_conv_fp32(a, pl.tile, pl.bk, d->tap4 != 0, grid, s, d->lds_reserve, d->prefetch2);
  if (pl.splitk > 1) {
    const size_t total = (size_t)a.M * a.Cout;
    const bool vec = a.Cout % 4 == 0;
    int blocks = (int)((total / (vec ? 4 : 1) + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (a.gn_partial) {
      int chunks = 0;
      for (int l = 0; l < a.nlv; ++l) chunks += (a.lv_off[l + 1] - a.lv_off[l] + EOD_GN_ROWS - 1) / EOD_GN_ROWS;
      hipLaunchKernelGGL(conv_splitk_reduce_gn_kernel, dim3(chunks), dim3(256), 0, s, a);
    } else if (vec) hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, a);
  }
  return eod_launch_status();
}
