#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n * 4, 0x00020000);
  unsigned voff = (threadIdx.x & 1) ? 0xFFFFFFFFu : threadIdx.x * 16;   // odd lanes out of range
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.0f;
  __syncthreads();
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 v = *reinterpret_cast<f32x4*>(lds + 256 + threadIdx.x * 4);
  y[threadIdx.x * 4 + 0] = v.x; y[threadIdx.x * 4 + 1] = v.y; y[threadIdx.x * 4 + 2] = v.z; y[threadIdx.x * 4 + 3] = v.w;
}
int main() {
  float *x, *y; hipMalloc(&x, 1024 * 4); hipMalloc(&y, 256 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 1024);
  float o[256]; hipMemcpy(o, y, sizeof(o), hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("%g ", o[i]); printf("\n");
  return 0;
}
