// how many 64-thread workgroups with a given LDS size are resident on a CU (occupancy API), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float* out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = lds[63 - threadIdx.x];
}
int main() {
  for (int bytes : {13312, 13653, 14336, 15360, 16384, 17408, 17920, 18432, 20480, 40960}) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes);
    printf("LDS %6d B per 64-thread workgroup -> %d workgroups per CU\n", bytes, n);
  }
  return 0;
}
