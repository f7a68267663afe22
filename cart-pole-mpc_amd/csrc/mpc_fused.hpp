// mpc_fused.hpp -- all SQP iterations of a batch in ONE kernel, a problem spread over L = S-1 lanes.
//
// The split pipeline (mpc_kernels.hpp) streams the sensitivities Gamma/Phi and the QP factors W/T through
// HBM between its kernels (measured 6.6 KB per problem and iteration, 40 % of the HBM peak, profiles/).
// Here lane s of a group of L consecutive lanes owns shooting interval s of one problem for the whole
// solve: its node x_s, Phi_s and defect live in registers; its SP controls, the step, Gamma_s and the
// per-control QP factors {1/d_k, (U^-1 g)_k} in lane-private LDS (80 scalars per lane = 20 KB per wave, so 8
// waves per CU = 2 per SIMD, which is also what the ~185 VGPRs allow); nothing of it reaches memory.
// Per iteration:
//   linearise      every lane integrates its own interval (RK4 + forward sensitivities), in parallel
//   QP sweeps      T = U D U^T, W = U^-1 R^T, gw = U^-1 g (down the horizon), then U^-T and the state recovery
//                  (up) are linear recurrences in k.  Every lane solves its own block in block-local
//                  coordinates (carry-in = 0) plus the scalar chain that says how a carry-in propagates; a
//                  chain of L-1 steps over the lanes (DPP row shifts) fixes the true block-boundary values
//                  {Psi_s, w_in, gw_in} / {dx_s, du_in}; each lane then corrects its block locally.  The
//                  pivots d_k depend on (lambda, weights) only: a Moebius recurrence whose block-boundary
//                  values come from powers of one 2x2 matrix, so no lane walks the whole horizon.
//   NX x NX LDL^T  every lane of the group solves the same small system (bitwise identical inputs: group
//                  sums are commutative butterflies over quad permutes)
//   line search    every lane rolls out its own interval of the trial point; the merit is a group sum
// 64/L problems share a wave, so the lock-step line search pays the maximum trial count over 16 (L = 4)
// problems instead of 64.  prepare_kernel / finalize_kernel of mpc_kernels.hpp run before / after,
// unchanged: the workspace fields zx, zu (iterate = warm start), sc and ist are this kernel's only global
// traffic.
//
// Same algorithm as the split pipeline (same QP, same line search, same decisions); the sweeps associate
// their sums differently, so results agree to rounding (fp64: 1e-9), not bitwise.
#pragma once
#include "mpc_kernels.hpp"

namespace cpmpc {

// ---- 2x2 matrices for the pivot recurrence (compile-time exponent, by squaring) -----------------------------
template <typename R>
struct Mat2 {
  R a, b, c, d;  // [[a, b], [c, d]]
};
template <typename R>
__device__ __forceinline__ Mat2<R> mat2_mul(const Mat2<R>& x, const Mat2<R>& y) {
  return Mat2<R>{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
template <typename R, int E>
__device__ __forceinline__ Mat2<R> mat2_pow(const Mat2<R>& t) {
  if constexpr (E <= 0) {
    return Mat2<R>{R(1), R(0), R(0), R(1)};
  } else if constexpr (E == 1) {
    return t;
  } else if constexpr (E % 2 == 0) {
    const Mat2<R> h = mat2_pow<R, E / 2>(t);
    return mat2_mul<R>(h, h);
  } else {
    return mat2_mul<R>(mat2_pow<R, E - 1>(t), t);
  }
}

template <typename R>
__device__ __forceinline__ Mat2<R> mat2_pow_rt(const Mat2<R>& t, int e) {  // run-time exponent, e >= 0
  Mat2<R> acc{R(1), R(0), R(0), R(1)}, base = t;
  while (e > 0) {
    if (e & 1) acc = mat2_mul<R>(acc, base);
    base = mat2_mul<R>(base, base);
    e >>= 1;
  }
  return acc;
}

// ---- lane traffic inside a group ------------------------------------------------------------------------
// Groups are L consecutive lanes with L | 16, so they never straddle a 16-lane DPP row: neighbour moves are
// row shifts and, for L = 4 (one quad) and L = 2, sums and broadcasts are quad permutes.  DPP moves are VALU
// operand modifiers: no trip through the LDS crossbar as for ds_bpermute (__shfl), which matters in the
// boundary chains where every move is on the critical path.
#ifndef CPMPC_DPP_NO_OLD
#define CPMPC_DPP_NO_OLD 1
#endif
template <int CTRL>
__device__ __forceinline__ int dpp32(int v) {
#if CPMPC_DPP_NO_OLD
  // every lane is written (all rows and banks enabled; a lane whose source falls off its row reads 0), so there is no
  // previous value to preserve and no `v_mov_b32 dst, 0` in front of the move
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
#else
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
#endif
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, dpp32<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)dpp32<CTRL>((int)(unsigned)(b & 0xffffffffll));
  const unsigned hi = (unsigned)dpp32<CTRL>((int)(unsigned)((unsigned long long)b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kDppQuadXor1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);  // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);  // quad_perm:[2,3,0,1]
constexpr int kDppRowShl1 = 0x101;                                // lane i reads lane i+1 (inside its row)
constexpr int kDppRowShr1 = 0x111;                                // lane i reads lane i-1 (inside its row)

// sum over the group, bitwise identical in all its lanes: a commutative butterfly when L is a power of two, else
// (groups of 5, 10, ... lanes) every lane adds the group's values in the same fixed order
template <typename R, int L>
__device__ __forceinline__ R group_sum(R v) {
  if constexpr (L == 4) {
    v += dpp<kDppQuadXor1>(v);
    v += dpp<kDppQuadXor2>(v);
  } else if constexpr (L == 2) {
    v += dpp<kDppQuadXor1>(v);
  } else if constexpr ((L & (L - 1)) == 0) {
#pragma unroll
    for (int off = 1; off < L; off <<= 1) v += __shfl_xor(v, off);
  } else {
    const int gb = (int)threadIdx.x - (int)threadIdx.x % L;
    R acc = __shfl(v, gb);
#pragma unroll
    for (int j = 1; j < L; ++j) acc += __shfl(v, gb + j);
    v = acc;
  }
  return v;
}
// maximum over the group, identical in all its lanes (a NaN operand is dropped, like nan_max)
template <typename R, int L>
__device__ __forceinline__ R group_max(R v) {
  if constexpr (L == 4) {
    v = nan_max(v, dpp<kDppQuadXor1>(v));
    v = nan_max(v, dpp<kDppQuadXor2>(v));
  } else if constexpr (L == 2) {
    v = nan_max(v, dpp<kDppQuadXor1>(v));
  } else if constexpr ((L & (L - 1)) == 0) {
#pragma unroll
    for (int off = 1; off < L; off <<= 1) v = nan_max(v, __shfl_xor(v, off));
  } else {
    const int gb = (int)threadIdx.x - (int)threadIdx.x % L;
    R acc = __shfl(v, gb);
#pragma unroll
    for (int j = 1; j < L; ++j) acc = nan_max(acc, __shfl(v, gb + j));
    v = acc;
  }
  return v;
}
// value of my right / left neighbour lane; callers discard it at the group edge.  Groups whose size divides 16
// never straddle a DPP row; the others go through the wave shuffle.
template <typename R, int L>
__device__ __forceinline__ R lane_right(R v) {
  if constexpr (16 % L == 0) return dpp<kDppRowShl1>(v);
  else return __shfl(v, (int)threadIdx.x + 1);
}
template <typename R, int L>
__device__ __forceinline__ R lane_left(R v) {
  if constexpr (16 % L == 0) return dpp<kDppRowShr1>(v);
  else return __shfl(v, (int)threadIdx.x - 1);
}
// value held by the first / last lane of my group
template <typename R, int L>
__device__ __forceinline__ R group_first(R v, int gbase) {
  if constexpr (L == 4) return dpp<0x00>(v);                              // quad_perm:[0,0,0,0]
  else if constexpr (L == 2) return dpp<(0 | (0 << 2) | (2 << 4) | (2 << 6))>(v);  // quad_perm:[0,0,2,2]
  else return __shfl(v, gbase);
}
template <typename R, int L>
__device__ __forceinline__ R group_last(R v, int gbase) {
  if constexpr (L == 4) return dpp<0xff>(v);                              // quad_perm:[3,3,3,3]
  else if constexpr (L == 2) return dpp<(1 | (1 << 2) | (3 << 4) | (3 << 6))>(v);  // quad_perm:[1,1,3,3]
  else return __shfl(v, gbase + L - 1);
}

// Waves per SIMD the register allocator must leave room for.  The sweeps are latency chains (LDS round trips,
// lane shuffles), so a second resident wave is worth a few spilled dwords in fp32 (measured +15%); fp64 values
// take two registers each and stay at one wave.
#ifndef CPMPC_FUSED_WAVES_F32
#define CPMPC_FUSED_WAVES_F32 2
#endif
// 1 (default): the fp32 fused kernel also refines the terminal multipliers once through the factored operator (the
// fp64 one always does); 0 builds the A/B variant without it (tools: build_variant("norefine32", ...))
#ifndef CPMPC_FUSED_REFINE_F32
#define CPMPC_FUSED_REFINE_F32 1
#endif
// refinement passes in a float kernel whose terminal system is carried in double (wide.hpp; measured in round 4)
#ifndef CPMPC_FUSED_REFINE_WIDE
#define CPMPC_FUSED_REFINE_WIDE 0
#endif
// refinement passes of the terminal multipliers for horizons of more than four intervals (fp64)
#ifndef CPMPC_FUSED_REFINE_LONG
#define CPMPC_FUSED_REFINE_LONG 2
#endif
// the Gamma update loop of the linearisation reads one column ahead (default: fp64 only; -DCPMPC_FUSED_GAMMA_AHEAD_ALL=0/1
// forces it off / on for both dtypes)
#ifdef CPMPC_FUSED_GAMMA_AHEAD_ALL
#define CPMPC_FUSED_GAMMA_AHEAD(R) (CPMPC_FUSED_GAMMA_AHEAD_ALL != 0)
#else
#define CPMPC_FUSED_GAMMA_AHEAD(R) (sizeof(R) == 8)
#endif
// unroll factor of the block-local sweep passes (LDS reads of several controls in flight)
// The exit test's rounding floor (cpmpc_solver_opts.exit_defect_floor) in the DOUBLE kernels: left out by default.  In
// double the floor is 3e-14 -- it could change an exit decision only where |D| is within mu x 3e-14 of the tolerance -- and
// carrying the few instructions costs the fp64 fused kernel 1.6 % (4.845 -> 4.925 ms at B = 262 144, same-session A/B: the
// kernel sits at its register limit and the extra live values reshuffle its spills).  1 compiles it in.
#ifndef CPMPC_EXIT_FLOOR_F64
#define CPMPC_EXIT_FLOOR_F64 0
#endif
// 1: the float 4-state kernels carry the whole terminal part of the QP in double (mpc_fused_body.inc: kWideQP); measured in
// round 5, see HISTORY.md
#ifndef CPMPC_FUSED_WIDE_QP_F32
#define CPMPC_FUSED_WIDE_QP_F32 0
#endif
#ifndef CPMPC_SWEEP_UNROLL
#define CPMPC_SWEEP_UNROLL 5
#endif
#ifndef CPMPC_FUSED_EXTRA_ATTR
#define CPMPC_FUSED_EXTRA_ATTR  // experiments: e.g. -DCPMPC_FUSED_EXTRA_ATTR='__attribute__((amdgpu_num_vgpr(168)))'
#endif
#define CPMPC_FUSED_BOUNDS __launch_bounds__(64, ((sizeof(R) == 4 && M::NX <= 4) ? CPMPC_FUSED_WAVES_F32 : 1))

// Debug build only (-DCPMPC_FUSED_TIMING): shader-clock cycles per phase, summed over waves, read back by
// cpmpc_debug_phase_cycles().  Not part of the product library.
#ifdef CPMPC_FUSED_TIMING
static __device__ unsigned long long g_fused_phase_cycles[8];  // one copy per translation unit
#define CPMPC_TICK_INIT() unsigned long long tick_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tick_last = __builtin_readcyclecounter()
#define CPMPC_TICK(IDX)                                              \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long tick_now = __builtin_readcyclecounter(); \
    tick_acc[IDX] += tick_now - tick_last;                           \
    tick_last = tick_now;                                            \
  } while (0)
#define CPMPC_TICK_FLUSH()                                                                       \
  do {                                                                                           \
    if (threadIdx.x == 0)                                                                        \
      for (int ti = 0; ti < 8; ++ti) atomicAdd(&g_fused_phase_cycles[ti], tick_acc[ti]);         \
  } while (0)
#else
#define CPMPC_TICK_INIT() do { } while (0)
#define CPMPC_TICK(IDX) do { } while (0)
#define CPMPC_TICK_FLUSH() do { } while (0)
#endif

// Debug build only (-DCPMPC_FUSED_CLOCK): every wave stamps the shader clock (s_memtime) and the constant 100 MHz
// counter (s_memrealtime) once on entry and once on exit; the sums over waves give the clock the part holds while this
// kernel runs (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps go to a buffer of their own that nothing else
// reads; read back by cpmpc_debug_kernel_clock().  Not part of the product library.
#ifdef CPMPC_FUSED_CLOCK
static __device__ unsigned long long g_fused_clock[4];  // sum of shader cycles, sum of 100 MHz ticks, waves, max cycles
#define CPMPC_CLOCK_BEGIN()                                                  \
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime();           \
  const unsigned long long clk_r0 = __builtin_amdgcn_s_memrealtime()
#define CPMPC_CLOCK_END()                                                                          \
  do {                                                                                             \
    const unsigned long long clk_c1 = __builtin_amdgcn_s_memtime();                                \
    const unsigned long long clk_r1 = __builtin_amdgcn_s_memrealtime();                            \
    if (threadIdx.x == 0) {                                                                        \
      atomicAdd(&g_fused_clock[0], clk_c1 - clk_c0);                                               \
      atomicAdd(&g_fused_clock[1], clk_r1 - clk_r0);                                               \
      atomicAdd(&g_fused_clock[2], 1ull);                                                          \
      atomicMax(&g_fused_clock[3], clk_c1 - clk_c0);                                               \
    }                                                                                              \
  } while (0)
#else
#define CPMPC_CLOCK_BEGIN() do { } while (0)
#define CPMPC_CLOCK_END() do { } while (0)
#endif

// Lean LDS layout, an experiment kept as a build option (round 3, measured negative, default off).  Motivation
// (tools/ubench/clock.hip): ONE wave issues a vector instruction every ~5.1 cycles at best, so the two waves per SIMD the
// fp32 kernel runs keep a 2-cycle issue port 39 % busy and a third would saturate it.  Three waves need <= 168 registers
// (costs this kernel 24 spilled values, none inside a loop) and less LDS: with (U^-1 g)_k sharing the slot of du_k (the
// previous step is dead when sweep 1 writes it; sweep 1b turns it into v_k in place) a wave needs 17.5 KB instead of 20
// and nine fit a CU instead of eight (-DCPMPC_FUSED_LEAN_LDS=1): measured 108.4 M re-plans/s against 116.3 M; with the
// 1/d_k of a lane's controls in registers as well (15 KB, ten waves per CU, -DCPMPC_FUSED_ID_REGS=1) the fully unrolled
// sweeps spill 109-151 values: 99.0 M.  A third wave on one or two of a CU's four SIMDs does not pay for the registers
// it takes from the other two.
#ifndef CPMPC_FUSED_LEAN_LDS
#define CPMPC_FUSED_LEAN_LDS 0
#endif
template <typename R, typename M, int SP>
__host__ __device__ constexpr bool fused_lean() {
  return CPMPC_FUSED_LEAN_LDS && sizeof(R) == 4 && M::NX <= 4 && SP <= 10;
}
// Slim layout for the double kernel of the 6-state model (round 5): with (U^-1 g)_k in the slot of du_k (as in the lean
// layout above) and the 1/d_k of a lane's controls in registers, a wave needs u + du + Gamma = (8 + 8 + 48) x SP x 64 bytes
// = 40 KB at SP = 10 instead of 60 KB: four waves per CU -- one per SIMD, all the 512-register kernel can use -- instead of
// two.  Not for the REFINE instantiation (its second solve needs gw and du side by side).
// (Round 5 also tried the float 6-state kernel in this layout with a 256-register budget, two waves per SIMD: 533 spilled
// dwords, 33 against 51 M re-plans/s; the option was removed in round 6, HISTORY.md has the numbers.)
#ifndef CPMPC_FUSED_SLIM_F64_NX6
#define CPMPC_FUSED_SLIM_F64_NX6 1
#endif
template <typename R, typename M, int SP, bool REFINE>
__host__ __device__ constexpr bool fused_slim() {
  return CPMPC_FUSED_SLIM_F64_NX6 && sizeof(R) == 8 && M::NX > 4 && !REFINE && SP <= 10;
}
#undef CPMPC_FUSED_BOUNDS
#define CPMPC_FUSED_BOUNDS_S __launch_bounds__(64, (fused_lean<R, M, SP>() ? 3 : ((sizeof(R) == 4 && M::NX <= 4) ? CPMPC_FUSED_WAVES_F32 : 1)))
#define CPMPC_FUSED_BOUNDS __launch_bounds__(64, ((sizeof(R) == 4 && M::NX <= 4) ? CPMPC_FUSED_WAVES_F32 : 1))

// ---- Gamma in LDS ---------------------------------------------------------------------------------------------------
// A column of Gamma_s is an NX-vector: 16 bytes (float, NX = 4), 32 (double NX = 4; float NX = 6, padded) or 64 (double,
// NX = 6).  Stored as one element per lane ([i * 64 + lane], rounds 1-3) the wider ones put consecutive lanes 32 / 64
// bytes apart: by the bank rule of ds_read_b128 ((address / 4) mod 64 within its 16-lane groups, MI355X_MICROARCH.md
// LDS section) lanes l and l + 8 of a group then share banks: a 2-way (4-way) conflict on every read and write of Gamma.
// CPMPC_FUSED_G_PLANES = 1 stores the vector as 16-byte pieces in separate planes, [(i * pieces + p) * 64 + lane]:
// lane-consecutive 16-byte slots, the conflict-free pattern.  Same bytes of LDS; float / NX = 4 is one piece either way.
// Measured (round 4, fp64, B = 262 144, same session; profiles/r04_f64_{planes,noplanes}_pmc_summary.json):
// SQ_LDS_BANK_CONFLICT 7 800 -> 0 cycles per wave (39 % of the LDS-array cycles gone, exactly the predicted 2-way), and the
// kernel is 0.5 % SLOWER (4.832 vs 4.805 ms; 50.7 vs 50.95 M re-plans/s), SQ_WAIT_ANY unchanged (11.7 % vs 11.3 % of
// wave-cycles), SQ_WAIT_INST_LDS 0.1 % in both: the LDS array works 3 % of the wave's cycles either way, so its conflicts
// were never what the lone wave waits for, and the second address per access costs more than they did.  Default off.
#ifndef CPMPC_FUSED_G_PLANES
#define CPMPC_FUSED_G_PLANES 0
#endif
// Round 5: a column occupies only the pieces it fills -- 48 bytes instead of the padded 64 for double / NX = 6, 24 instead of
// 32 for float / NX = 6 (CPMPC_FUSED_G_UNPADDED = 0 restores the padded elements): Gamma is 30 KB instead of 40 KB per wave
// for the double 6-state kernel, three waves per CU instead of two (and four with the slim layout below).
#ifndef CPMPC_FUSED_G_UNPADDED
#define CPMPC_FUSED_G_UNPADDED 1
#endif
// A piece is 16 bytes where the column is a whole number of them (float / NX = 4: one; double / NX = 4: two; double /
// NX = 6: three) and 8 bytes otherwise (float / NX = 6: 24 bytes = three pieces of 8 instead of the padded 32: 15 KB of Gamma
// per wave instead of 20, which is what lets the float 6-state kernel's LDS fit eight waves per CU, round 5).
template <int BYTES>
struct alignas(BYTES) GPieceT {
  unsigned w[BYTES / 4];
};
template <typename R, int NX>
__host__ __device__ constexpr int fused_g_piece_bytes() {
  return (CPMPC_FUSED_G_UNPADDED && (NX * sizeof(R)) % 16 != 0) ? 8 : 16;
}
template <typename R, int NX>
using GPieceOf = GPieceT<fused_g_piece_bytes<R, NX>()>;
template <typename R, int NX>
__host__ __device__ constexpr int fused_g_pieces() {
  constexpr int PB = fused_g_piece_bytes<R, NX>();
  return CPMPC_FUSED_G_UNPADDED ? (int)((NX * sizeof(R) + PB - 1) / PB) : (int)(sizeof(XV<R, NX>) / 16);
}
template <typename R, int NX>
__host__ __device__ constexpr size_t fused_g_bytes() {  // LDS bytes of one column
  return (size_t)fused_g_pieces<R, NX>() * fused_g_piece_bytes<R, NX>();
}
template <typename R, int NX>
__device__ __forceinline__ XV<R, NX> fused_g_ld(const GPieceOf<R, NX>* g, int i, int lane) {
  constexpr int K = fused_g_pieces<R, NX>();
  XV<R, NX> v;
  GPieceOf<R, NX> pc[K];
#pragma unroll
  for (int p = 0; p < K; ++p) pc[p] = CPMPC_FUSED_G_PLANES ? g[(i * K + p) * 64 + lane] : g[(i * 64 + lane) * K + p];
  __builtin_memcpy(&v, pc, sizeof pc <= sizeof v ? sizeof pc : sizeof v);  // (the padding of v, if any, is never read)
  return v;
}
template <typename R, int NX>
__device__ __forceinline__ void fused_g_st(GPieceOf<R, NX>* g, int i, int lane, const XV<R, NX> v) {
  constexpr int K = fused_g_pieces<R, NX>();
  GPieceOf<R, NX> pc[K];
  __builtin_memcpy(pc, &v, sizeof pc <= sizeof v ? sizeof pc : sizeof v);
#pragma unroll
  for (int p = 0; p < K; ++p) {
    if (CPMPC_FUSED_G_PLANES) g[(i * K + p) * 64 + lane] = pc[p];
    else g[(i * 64 + lane) * K + p] = pc[p];
  }
}

// SHARED: the batch shares one parameter set -> the model constants stay wave-uniform (scalar registers)
// REFINE (double kernels only; CPMPC_CREATE_REFINE_QP): one step of iterative refinement of the whole QP solution with
// residuals from the original data (the block after sweep 2 in mpc_fused_body.inc)
template <typename R, typename M, int SP, int L, bool SHARED, bool REFINE>
__global__ CPMPC_FUSED_BOUNDS_S CPMPC_FUSED_EXTRA_ATTR void fused_sqp_kernel(const SolverArgs<R, M> a, const int max_iters) {
  constexpr int NX = M::NX;
  constexpr int PPW = 64 / L;  // problems per wave
  constexpr bool kSlim = fused_slim<R, M, SP, REFINE>();
  constexpr bool kLean = fused_lean<R, M, SP>() || kSlim;
  __shared__ R lds_u[SP * 64];
  __shared__ R lds_du[SP * 64];
  __shared__ GPieceOf<R, NX> lds_G[SP * 64 * fused_g_pieces<R, NX>()];  // column i of my Gamma_s (fused_g_ld / fused_g_st)
  __shared__ R lds_gw_own[kLean ? 1 : SP * 64];  // (U^-1 g)_k of my controls
#ifndef CPMPC_FUSED_ID_REGS
#define CPMPC_FUSED_ID_REGS 0  // 1: 1/d_k in registers (15 KB of LDS, but the full unroll it needs spills 151 values: slower)
#endif
  constexpr bool kIdRegs = (fused_lean<R, M, SP>() && CPMPC_FUSED_ID_REGS) || kSlim;
  __shared__ R lds_id[kIdRegs ? 1 : SP * 64];      // 1/d_k of my controls
  R* const lds_gw = kLean ? lds_du : lds_gw_own;
  R id_reg[kIdRegs ? SP : 1];
  constexpr int kSweepUnroll = kIdRegs ? SP : CPMPC_SWEEP_UNROLL;  // register-held 1/d_k need static indices: full unroll
#define CPMPC_ID_SET(I, V)                    \
  do {                                        \
    if constexpr (kIdRegs) id_reg[I] = (V);   \
    else lds_id[(I) * 64 + lane] = (V);       \
  } while (0)
#define CPMPC_ID_GET(I) (kIdRegs ? id_reg[kIdRegs ? (I) : 0] : lds_id[kIdRegs ? 0 : (I) * 64 + lane])
#define CPMPC_SWEEP_PRAGMA _Pragma("unroll kSweepUnroll")
// with the sweeps fully unrolled the scheduler hoists all ten controls' LDS reads to the top and spills; a scheduling
// barrier per control keeps the live ranges those of the rolled loop
#define CPMPC_SWEEP_FENCE()                                     \
  do {                                                          \
    if constexpr (kIdRegs) __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#define CPMPC_FUSED_MAT2_POW(T, E) mat2_pow<R, (E)>(T)
#define CPMPC_FUSED_BODY_DYN 0
#include "mpc_fused_body.inc"
#undef CPMPC_FUSED_BODY_DYN
#undef CPMPC_FUSED_MAT2_POW
#undef CPMPC_SWEEP_PRAGMA
#undef CPMPC_SWEEP_FENCE
#undef CPMPC_ID_SET
#undef CPMPC_ID_GET
}

// The same kernel for a state spacing without a compiled specialisation: SP is a.SP at run time, the LDS arrays are
// carved from dynamic shared memory (fused_dyn_lds_bytes), the matrix powers use a run-time exponent.
template <typename R, typename M>
__host__ __device__ constexpr size_t fused_dyn_lds_bytes(int sp) {
  return (size_t)sp * 64 * (4 * sizeof(R) + fused_g_bytes<R, M::NX>());
}
template <typename R, typename M, int L, bool SHARED, bool REFINE>
__global__ CPMPC_FUSED_BOUNDS void fused_sqp_dyn_kernel(const SolverArgs<R, M> a, const int max_iters) {
  constexpr int NX = M::NX;
  constexpr int PPW = 64 / L;
  const int SP = a.SP;
  extern __shared__ __align__(32) unsigned char fused_dyn_lds[];
  GPieceOf<R, NX>* lds_G = reinterpret_cast<GPieceOf<R, NX>*>(fused_dyn_lds);   // widest elements first: stays aligned
  R* lds_u = reinterpret_cast<R*>(lds_G + (size_t)SP * 64 * fused_g_pieces<R, NX>());
  R* lds_du = lds_u + (size_t)SP * 64;
  R* lds_gw = lds_du + (size_t)SP * 64;
  R* lds_id = lds_gw + (size_t)SP * 64;
#define CPMPC_ID_SET(I, V) lds_id[(I) * 64 + lane] = (V)
#define CPMPC_ID_GET(I) (lds_id[(I) * 64 + lane])
#define CPMPC_SWEEP_PRAGMA _Pragma("unroll 5")
#define CPMPC_SWEEP_FENCE() do { } while (0)
#define CPMPC_FUSED_MAT2_POW(T, E) mat2_pow_rt<R>(T, (E))
#define CPMPC_FUSED_BODY_DYN 1
#include "mpc_fused_body.inc"
#undef CPMPC_FUSED_BODY_DYN
#undef CPMPC_FUSED_MAT2_POW
#undef CPMPC_SWEEP_PRAGMA
#undef CPMPC_SWEEP_FENCE
#undef CPMPC_ID_SET
#undef CPMPC_ID_GET
}

}  // namespace cpmpc
