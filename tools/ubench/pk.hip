// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs transcendental, 1 wave and 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float m = 0.999f, c = 0.001f;
  const float2v m2 = {m, m}, c2 = {c, c};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    } else if (MODE == 3) {  // dependent chain of fma
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      }
    } else if (MODE == 4) {  // v_pk_mul_f32
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2));
      }
    } else if (MODE == 5) {  // v_mov_b32
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                     "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  k<MODE><<<blocks, 64>>>(out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double per = (double)h / (iters * 64.0);
  printf("%-12s blocks %5d: %.2f counter-ticks per instruction per wave, kernel %.3f ms -> %.2f ns*SIMD/instr\n", name, blocks, per, ms,
         ms * 1e6 / (iters * 64.0) / (blocks / 1024.0 > 1 ? blocks / 1024.0 : 1));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {1024, 2048, 4096}) {
    run<0>("v_fma", blocks); run<1>("v_pk_fma", blocks); run<4>("v_pk_mul", blocks); run<2>("v_exp", blocks); run<3>("fma_dep", blocks); run<5>("v_mov", blocks);
  }
  return 0;
}
