// micro-benchmark: issue rate of fp64 vector instructions on gfx950 at 1, 2 and 4 waves per SIMD.
// Question it answers: does the fp64 fused kernel lose throughput by running ONE wave per SIMD,
// the way the fp32 kernel did?  build: hipcc --offload-arch=gfx950 -O2 -o f64 f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
  const double m = 0.999, c = 0.001;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) {  // independent v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      } else if (MODE == 1) {  // dependent chain
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 2) {  // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                     "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
      } else if (MODE == 3) {  // v_add_f64
        asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 4) {  // 4 v_fma_f64 + 4 v_mov_b32 interleaved (fp64 moves/selects are pairs of 32-bit ops)
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_mov_b32 %4, %5\n v_fma_f64 %1, %1, %8, %9\n v_mov_b32 %5, %6\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_mov_b32 %6, %7\n v_fma_f64 %3, %3, %8, %9\n v_mov_b32 %7, %4\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(m), "v"(c));
      } else if (MODE == 5) {  // v_rcp_f64
        asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                     "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 6) {  // v_fma_f32 for reference
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(0.999f), "v"(0.001f));
      } else if (MODE == 7) {  // two dependent chains interleaved
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
}
template <int MODE>
void run(const char* name, int blocks) {
  double* out;
  hipMalloc(&out, blocks * 64 * 8);
  const int iters = 2000;
  k<MODE><<<blocks, 64>>>(out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(out, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks / 1024.0;
  printf("%-14s %4.0f wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms,
         ms * 1e6 / (iters * 64.0) / waves_per_simd);
  hipFree(out);
}
int main() {
  for (int blocks : {1024, 2048, 4096}) {
    run<0>("v_fma_f64", blocks); run<1>("fma_f64 dep", blocks); run<7>("fma_f64 2dep", blocks); run<2>("v_mul_f64", blocks);
    run<3>("v_add_f64", blocks); run<4>("fma64+mov32", blocks); run<5>("v_rcp_f64", blocks); run<6>("v_fma_f32", blocks);
  }
  return 0;
}
