// Shared helpers for the gfx950 kernels of the embodied-detector hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EOD_OK 0
#define EOD_ERR_BAD_DIMS (-1)
#define EOD_ERR_ALIGN (-2)
#define EOD_ERR_LAUNCH (-3)
#define EOD_ERR_NULL (-4)
#define EOD_ERR_CAPACITY (-5)

#define EOD_WAVE 64
// First statement of every small kernel of the latency-bound chains (proposal decoding, cascade glue, selection, memory write):
// their waves share SIMDs with resident GEMM waves of the concurrently running mask passes; with a raised wave priority the
// instruction arbiter serves them first (the GEMM waves keep the default 0).
#define EOD_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int eod_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? EOD_OK : EOD_ERR_LAUNCH;
}

static inline bool eod_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float eod_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

// exact-ish sigmoid used where scores are compared against thresholds / sorted: expf, IEEE divide
__device__ __forceinline__ float eod_sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }

// Box2BoxTransform.apply_deltas on one box (detic_roi_heads.py:121-122,314), shared by apply_deltas_kernel and by roi_align_kernel's
// refine-on-load form: the same instructions in both, hence the same bits.
__device__ __forceinline__ void eod_apply_deltas_one(const float* __restrict__ d, float x1, float y1, float x2, float y2, float wx, float wy,
                                                     float ww, float wh, int clip, float img_w, float img_h, float& ox1, float& oy1,
                                                     float& ox2, float& oy2) {
  const float w = x2 - x1, h = y2 - y1;
  const float cx = x1 + 0.5f * w, cy = y1 + 0.5f * h;
  const float clampv = 4.135166556742356f;  // log(1000/16)
  const float dx = d[0] / wx;
  const float dy = d[1] / wy;
  const float dw = fminf(d[2] / ww, clampv);
  const float dh = fminf(d[3] / wh, clampv);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  ox1 = pcx - 0.5f * pw; oy1 = pcy - 0.5f * ph; ox2 = pcx + 0.5f * pw; oy2 = pcy + 0.5f * ph;
  if (clip) {
    ox1 = fminf(fmaxf(ox1, 0.f), img_w);
    oy1 = fminf(fmaxf(oy1, 0.f), img_h);
    ox2 = fminf(fmaxf(ox2, 0.f), img_w);
    oy2 = fminf(fmaxf(oy2, 0.f), img_h);
  }
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups that share an XCD (id % 8 equal under
// round-robin dispatch) get a contiguous range of tile ids, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

// Division by a launch-invariant integer with one mul_hi + shifts (Granlund-Montgomery, exact for every 32-bit n).
struct FastDiv {
  unsigned mp, sh1, sh2, d;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.mp, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

static inline FastDiv eod_make_fastdiv(unsigned d) {
  FastDiv f{};
  if (d == 0) d = 1;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.mp = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 0 ? l - 1 : 0;
  f.d = d;
  return f;
}
