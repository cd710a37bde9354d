// Does a 1-D launch with more than 2^32 work-items run on this ROCm?  (AQL grid_size_x is a 32-bit work-item count.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* cnt, unsigned long long* maxb) {
  if (threadIdx.x == 0) {
    atomicAdd(cnt, 1ULL);
    atomicMax(maxb, (unsigned long long)blockIdx.x);
  }
}
int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  for (unsigned long long blocks : {1000000ULL, 16000000ULL, 17000000ULL, 28000000ULL, 100000000ULL}) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, 0, d, d + 1);
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("blocks=%llu work-items=%llu launch=%s sync=%s counted=%llu max_block=%llu\n", blocks, blocks * 256, hipGetErrorName(e1),
           hipGetErrorName(e2), h[0], h[1]);
  }
  return 0;
}
