// micro-benchmark: what a vector instruction costs in SHADER CYCLES and in WALL time on gfx950, so that the clock the
// part holds under a vector-ALU-dense load falls out (the guide prices a wave64 v_fma_f32 at 2
// cycles with >= 2 waves per SIMD; tools/ubench/pk.hip measured 1.15 ns and nobody had measured the clock).
//
// Per mode and occupancy: every wave stamps s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its
// loop (MI355X_MICROARCH.md, DVFS give-back item 6: clock = d(memtime) / d(memrealtime) x 100 MHz), the host times the
// launch with HIP events.  The launches run back to back for `--seconds` per mode before the measured one so that the
// power management has settled.  Output: one JSON object per line.
// build: hipcc --offload-arch=gfx950 -O2 -o clock clock.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Stamp {
  unsigned long long cyc, real, real_begin;  // shader cycles and 100 MHz ticks of the loop, absolute 100 MHz time of its start
};

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, Stamp* st, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
  const float m = 0.999f, c = 0.001f;
  const double md = 0.999, cd = 0.001;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) {  // independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
      } else if (MODE == 1) {  // dependent v_fma_f32 chain
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(a0) : "v"(m), "v"(c));
      } else if (MODE == 2) {  // v_exp_f32 (quarter-rate class)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 3) {  // independent v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(md), "v"(cd));
      } else if (MODE == 4) {  // dependent v_fma_f64 chain
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(d0) : "v"(md), "v"(cd));
      } else if (MODE == 5) {  // v_mov_b32 / v_cndmask class (32-bit moves: half of an fp64 select)
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                     "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 6) {  // v_rcp_f64 (fp64 transcendental class)
        asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                     "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
      } else if (MODE == 7) {  // v_sin_f32
        asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n"
                     "v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 8) {  // v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                     "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(md));
      } else if (MODE == 9) {  // v_add_f64
        asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd));
      } else if (MODE == 11) {  // packed fp32 fma: two fp32 results per instruction, on register pairs
        asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                     "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd));
      } else if (MODE == 12) {  // packed and plain fp32 fma alternating (what a partly packed kernel would issue)
        asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_fma_f32 %4, %4, %9, %10\n v_pk_fma_f32 %1, %1, %8, %8\n v_fma_f32 %5, %5, %9, %10\n"
                     "v_pk_fma_f32 %2, %2, %8, %8\n v_fma_f32 %6, %6, %9, %10\n v_pk_fma_f32 %3, %3, %8, %8\n v_fma_f32 %7, %7, %9, %10\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(cd), "v"(m), "v"(c));
      } else if (MODE == 13) {  // packed fp32 fma with one operand broadcast from the low half (op_sel_hi)
        asm volatile("v_pk_fma_f32 %0, %8, %0, %8 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %8, %1, %8 op_sel_hi:[0,1,1]\n"
                     "v_pk_fma_f32 %2, %8, %2, %8 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %8, %3, %8 op_sel_hi:[0,1,1]\n"
                     "v_pk_fma_f32 %4, %8, %4, %8 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %5, %8, %5, %8 op_sel_hi:[0,1,1]\n"
                     "v_pk_fma_f32 %6, %8, %6, %8 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %7, %8, %7, %8 op_sel_hi:[0,1,1]\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(cd));
      } else if (MODE == 10) {  // v_mul_f32 with a DPP operand (the group traffic of the fused kernel)
        asm volatile("v_mul_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "v_mul_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] =
      a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0, r0};
}

static double g_seconds = 2.0;

template <int MODE>
void run(const char* name, int blocks) {
  float* out;
  Stamp* st;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipMalloc(&st, (size_t)blocks * sizeof(Stamp));
  const int iters = 4000;  // 256 k instructions per wave: a launch of a few hundred microseconds to milliseconds
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // settle: back-to-back launches for g_seconds
  float ms = 0.f;
  double spent = 0.0;
  int launches = 0;
  while (spent < g_seconds * 1e3) {
    hipEventRecord(e0);
    for (int r = 0; r < 16; ++r) k<MODE><<<blocks, 64>>>(out, st, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    spent += ms;
    launches += 16;
  }
  hipEventRecord(e0);
  k<MODE><<<blocks, 64>>>(out, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks);
  hipMemcpy(h.data(), st, (size_t)blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> ghz(blocks), cyc(blocks);
  for (int b = 0; b < blocks; ++b) {
    ghz[b] = h[b].real ? (double)h[b].cyc / (double)h[b].real * 0.1 : 0.0;
    cyc[b] = (double)h[b].cyc;
  }
  // how many waves were resident together: waves whose loop covers the mid-point of the launch
  unsigned long long rb_min = ~0ull, re_max = 0;
  for (int b = 0; b < blocks; ++b) {
    rb_min = std::min(rb_min, h[b].real_begin);
    re_max = std::max(re_max, h[b].real_begin + h[b].real);
  }
  const unsigned long long mid = rb_min + (re_max - rb_min) / 2;
  int resident = 0;
  for (int b = 0; b < blocks; ++b)
    if (h[b].real_begin <= mid && mid <= h[b].real_begin + h[b].real) ++resident;
  const double span_us = (double)(re_max - rb_min) * 0.01;
  std::sort(ghz.begin(), ghz.end());
  std::sort(cyc.begin(), cyc.end());
  const double n_inst = (double)iters * 64.0;
  const double waves_per_simd = blocks / 1024.0;
  // cycles of SIMD time per wave-instruction: a wave's cycles / its instructions / the waves sharing its SIMD
  const double cyc_per_inst_wave = cyc[blocks / 2] / n_inst;
  const double resident_per_simd = resident / 1024.0;
  // a wave's interval between instructions / the waves that really shared its SIMD (not the number launched)
  const double cyc_per_inst_simd = cyc_per_inst_wave / (resident_per_simd > 1 ? resident_per_simd : 1);
  const double ns_per_inst_simd = ms * 1e6 / n_inst / (waves_per_simd > 1 ? waves_per_simd : 1);
  printf("{\"mode\": \"%s\", \"waves_per_simd\": %.0f, \"settle_launches\": %d, \"kernel_ms\": %.4f, "
         "\"resident_waves_per_simd_at_midpoint\": %.2f, \"first_start_to_last_end_us\": %.1f, "
         "\"cycles_per_instr_wave\": %.3f, \"cycles_per_instr_simd\": %.3f, \"ns_per_instr_simd\": %.4f, "
         "\"clock_GHz_median\": %.4f, \"clock_GHz_min\": %.4f, \"clock_GHz_max\": %.4f, "
         "\"clock_GHz_from_wall\": %.4f}\n",
         name, waves_per_simd, launches, ms, resident_per_simd, span_us, cyc_per_inst_wave, cyc_per_inst_simd, ns_per_inst_simd, ghz[blocks / 2],
         ghz[0], ghz[blocks - 1], cyc_per_inst_simd / ns_per_inst_simd);
  fflush(stdout);
  hipFree(out);
  hipFree(st);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  bool packed_only = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--seconds") && i + 1 < argc) g_seconds = atof(argv[i + 1]);
    if (!strcmp(argv[i], "--packed")) packed_only = true;  // only the packed-fp32 modes (and plain fma beside them)
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"clockRate_kHz\": %d, \"CUs\": %d}\n", prop.name, prop.gcnArchName,
         prop.clockRate, prop.multiProcessorCount);
  if (packed_only) {
    for (int blocks : {1024, 2048, 3072, 4096}) {
      run<0>("v_fma_f32", blocks);
      run<11>("v_pk_fma_f32", blocks);
      run<13>("v_pk_fma_f32 op_sel_hi broadcast", blocks);
      run<12>("v_pk_fma_f32 + v_fma_f32 alternating", blocks);
    }
    return 0;
  }
  for (int blocks : {1024, 2048, 3072, 4096, 8192}) {
    run<0>("v_fma_f32", blocks);
    run<1>("v_fma_f32 dependent", blocks);
    run<2>("v_exp_f32", blocks);
    run<7>("v_sin_f32", blocks);
    run<5>("v_mov_b32", blocks);
    run<10>("v_mul_f32_dpp", blocks);
    run<3>("v_fma_f64", blocks);
    run<4>("v_fma_f64 dependent", blocks);
    run<8>("v_mul_f64", blocks);
    run<9>("v_add_f64", blocks);
    run<6>("v_rcp_f64", blocks);
  }
  return 0;
}
