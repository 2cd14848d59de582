// How much does a resident dense kernel slow a latency-bound chain of tiny kernels on another (high-priority) stream, and does it
// depend on HOW the dense kernel occupies the chip?  Dense = an MFMA spin loop (no memory traffic), launched either as many short
// workgroups (the shape of an implicit GEMM: thousands of ~50 us workgroups queued behind full CUs) or as a few long-running
// ones (a persistent kernel that leaves workgroup slots free).  Chain = 60 dependent launches of a one-workgroup kernel that
// chases 24 dependent global loads.
// hipcc --offload-arch=gfx950 -O3 -o chain_contention chain_contention.hip && ./chain_contention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ long long g_stamp[256];      // [0] dense start (first workgroup), [1] dense end (any workgroup, max), [2 + i] start of chained launch i


__global__ __launch_bounds__(256) void dense(float* out, int iters) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0 && blockIdx.x == 0) g_stamp[0] = wall_clock64();
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.37f;
  if (iters < 0) iters = (-iters * (48 + (int)((blockIdx.x * 2654435761u) >> 27))) >> 6;      // 0.75 .. 1.23 x
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 12345.678f) { out[0] = s; smem[threadIdx.x] = s; }
  if (threadIdx.x == 0) atomicMax((unsigned long long*)&g_stamp[1], (unsigned long long)wall_clock64());
}

template <bool PRIO>
__global__ __launch_bounds__(64) void tiny(const int* __restrict__ next, int* __restrict__ out, int hops, int stamp) {
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  if (threadIdx.x == 0 && stamp >= 0) g_stamp[2 + stamp] = wall_clock64();
  int p = threadIdx.x;
  for (int h = 0; h < hops; ++h) p = next[p];
  out[threadIdx.x] = p;
}

static float run(hipStream_t hi, hipStream_t lo, float* dout, int dense_grid, int dense_iters, size_t dense_lds, const int* next, int* tout,
                 bool prio, int n_dense_launches) {
  hipEvent_t e0, e1, d0, d1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventCreate(&d0);
  hipEventCreate(&d1);
  long long zero[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), zero, sizeof(zero));
  hipEventRecord(d0, lo);
  for (int i = 0; i < n_dense_launches; ++i) hipLaunchKernelGGL(dense, dim3(dense_grid), dim3(256), dense_lds, lo, dout, dense_iters);
  // let the dense kernel get going
  hipEventRecord(d1, lo);
  hipLaunchKernelGGL(tiny<false>, dim3(1), dim3(64), 0, hi, next, tout, 200, -1);
  hipEventRecord(e0, hi);
  for (int i = 0; i < 60; ++i) {
    if (prio) hipLaunchKernelGGL(tiny<true>, dim3(1), dim3(64), 0, hi, next, tout, 24, i);
    else hipLaunchKernelGGL(tiny<false>, dim3(1), dim3(64), 0, hi, next, tout, 24, i);
  }
  hipEventRecord(e1, hi);
  hipDeviceSynchronize();
  float ms, dms, lead;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventElapsedTime(&dms, d0, d1);
  hipEventElapsedTime(&lead, d0, e1);
  if (n_dense_launches && getenv("VERBOSE")) {
    long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    printf("    dense first workgroup .. last end: 0 .. %.2f ms (last launch only); chained launches started at (ms):", (st[1] - st[0]) * 1e-5);
    for (int i = 0; i < 60; i += 6) printf(" %.2f", (st[2 + i] - st[0]) * 1e-5);
    printf("\n");
  }
  if (n_dense_launches && getenv("VERBOSE")) printf("    [dense %.2f ms; chain ended %.2f ms after the dense start]\n", dms, lead);
  return ms * 1e3f / 60.f;
}

int main() {
  int least, greatest;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  printf("stream priority range: least %d, greatest %d\n", least, greatest);
  float* dout;
  hipMalloc(&dout, 4);
  const int n = 1 << 20;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 40503 + 12345) % n);
  int *next, *tout;
  hipMalloc(&next, n * 4);
  hipMalloc(&tout, 256);
  hipMemcpy(next, h.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(dense), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  // one workgroup's MFMA loop: iters x 16 MFMAs x 64 cycles; 2 workgroups per CU share the SIMDs
  const int it50us = (int)(50e-6 * 2.3e9 / (16 * 64));            // ~50 us when alone on its SIMDs
  struct Case { const char* name; int grid, iters; size_t lds; int launches; };
  const Case cases[] = {
      {"no dense kernel                                       ", 0, 0, 0, 0},
      {"GEMM-shaped:   8 x 4000 workgroups, uneven (37..63 us)  ", 4000, -it50us, 64 * 1024, 8},
      {"GEMM-shaped:  32 x 1000 workgroups, uneven              ", 1000, -it50us, 64 * 1024, 32},
      {"GEMM-shaped:  64 x  500 workgroups, uneven              ", 500, -it50us, 64 * 1024, 64},
      {"GEMM-shaped: 128 x  250 workgroups, uneven              ", 250, -it50us, 64 * 1024, 128},
      {"persistent: 256 workgroups (1 per CU), 64 KiB LDS       ", 256, it50us * 64, 64 * 1024, 1},
  };
  // Are the two streams' hardware queues served by the same command-processor pipe?  k dummy streams are created (and used once)
  // between the chain's stream and the dense stream, which shifts the dense stream's queue by k.
  const int env_k = getenv("DUMMY_STREAMS") ? atoi(getenv("DUMMY_STREAMS")) : -1;
  for (int k = (env_k >= 0 ? env_k : 0); k <= (env_k >= 0 ? env_k : 6); ++k) {
    for (int pc = 0; pc < 2; ++pc) {
      hipStream_t hi, lo, dummy[8];
      const int ph = pc == 0 ? greatest : 0, pl = pc == 0 ? least : 0;
      hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, ph);
      hipLaunchKernelGGL(tiny<false>, dim3(1), dim3(64), 0, hi, next, tout, 4, -1);
      for (int d = 0; d < k; ++d) {
        hipStreamCreateWithPriority(&dummy[d], hipStreamNonBlocking, pl);
        hipLaunchKernelGGL(tiny<false>, dim3(1), dim3(64), 0, dummy[d], next, tout, 4, -1);
      }
      hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, pl);
      hipDeviceSynchronize();
      printf("=== %d dummy streams; chain stream priority %d, dense stream priority %d\n", k, ph, pl);
      for (const Case& c : cases) {
        float best = 1e9f;
        for (int r = 0; r < 2; ++r) {
          const float us = run(hi, lo, dout, c.grid ? c.grid : 1, c.grid ? c.iters : 1, c.lds, next, tout, true, c.launches);
          best = us < best ? us : best;
        }
        printf("%s : %6.1f us per chained launch\n", c.name, best);
      }
      hipStreamDestroy(hi);
      hipStreamDestroy(lo);
      for (int d = 0; d < k; ++d) hipStreamDestroy(dummy[d]);
    }
  }
  return 0;
}
