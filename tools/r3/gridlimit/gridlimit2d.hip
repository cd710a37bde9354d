// 2-D launch whose total exceeds 2^32 work-items while every dimension stays below it
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* cnt, unsigned long long* maxb) {
  if (threadIdx.x == 0) {
    atomicAdd(cnt, 1ULL);
    atomicMax(maxb, (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x);
  }
}
int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  for (unsigned by : {1u, 16u, 28u, 100u}) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(1u << 20, by), dim3(256), 0, 0, d, d + 1);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("grid=(%u,%u) work-items=%llu launch=%s sync=%s counted=%llu expected=%llu max_block=%llu\n", 1u << 20, by,
           (unsigned long long)by * (1ull << 28), hipGetErrorName(e1), hipGetErrorName(e2), h[0], (unsigned long long)by << 20, h[1]);
  }
  return 0;
}
