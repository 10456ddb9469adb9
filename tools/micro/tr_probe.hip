#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)i;
  __syncthreads();
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) o[threadIdx.x * 4 + j] = (float)v[j];
}
int main() {
  float* o; hipMalloc(&o, 256 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); 
  float h[256]; hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4.0f", h[l * 4 + j]); printf("\n"); }
  return 0;
}
