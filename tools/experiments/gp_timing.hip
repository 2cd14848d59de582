// Phase timing of gather_pool_kernel with s_memtime stamps (diagnostics; built and run by hand on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/gp_timing.hip -o /tmp/gp_timing && /tmp/gp_timing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_stamps;      // [blocks][4 waves][6]
#define GP_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if ((threadIdx.x & 63) == 0) g_stamps[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 6 + (k)] = __builtin_readcyclecounter(); \
  } while (0)
#include "../../embodied_object_detection_amd/csrc/memory_read.hip"

int main() {
  const int H = 640, W = 640, N = 40000;
  std::vector<int> proj(H * W);
  const char* names[3] = {"constant", "columns", "blocks8"};
  unsigned short* mem;
  int* dproj;
  unsigned short* pooled;
  unsigned long long* stamps;
  const int blocks = (H / 32) * (W / 32);
  hipMalloc(&mem, (size_t)N * 512 * 2);
  hipMemset(mem, 0, (size_t)N * 512 * 2);
  hipMalloc(&dproj, proj.size() * 4);
  hipMalloc(&pooled, eod_memory_pooled_halves(H, W) * 2);
  hipMalloc(&stamps, (size_t)blocks * 4 * 6 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  for (int pat = 0; pat < 3; ++pat) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        proj[y * W + x] = pat == 0 ? 7 : (pat == 1 ? (x * 5) % N : ((y / 8) * 200 + x / 8) % N);
    hipMemcpy(dproj, proj.data(), proj.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) eod_memory_gather_pool(mem, dproj, H, W, 512, N, pooled, nullptr, 0, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * 4 * 6);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double d[5] = {0, 0, 0, 0, 0};
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < blocks * 4; ++w) {
      for (int k = 0; k < 5; ++k) d[k] += (double)(h[w * 6 + k + 1] - h[w * 6 + k]);
      if (h[w * 6] < t0) t0 = h[w * 6];
      if (h[w * 6 + 5] > t1) t1 = h[w * 6 + 5];
    }
    printf("%-9s per wave (cycles of the 100 MHz counter x?): index load %.0f | dedup rounds %.0f | dma wait %.0f | pooling %.0f | tail %.0f | "
           "first-to-last stamp %llu\n", names[pat], d[0] / (blocks * 4), d[1] / (blocks * 4), d[2] / (blocks * 4), d[3] / (blocks * 4),
           d[4] / (blocks * 4), t1 - t0);
  }
  return 0;
}
