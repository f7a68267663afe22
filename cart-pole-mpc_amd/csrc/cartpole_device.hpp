// cartpole_device.hpp -- gfx950 device functions: cart-pole dynamics with analytic Jacobians and
// the RK4 step with/without sensitivities.  One problem per lane, everything in registers.
//
// Behaviour follows (reference paths relative to /root/reference):
//   gen::single_pendulum_dynamics        optimization/single_pendulum_dynamics.hpp:13-186
//     (written from the Lagrangian spec symbolic/dynamics_single.py:58-143)
//   runge_kutta_4th_order<D>             optimization/integration.hpp:13-49
//   runge_kutta_4th_order_no_jacobians   optimization/integration.hpp:52-62
//   mod_pi                               optimization/integration.hpp:65-73
#pragma once
#include <hip/hip_runtime.h>

namespace cpmpc {

// ------------------------------------------------------------------------------------------------
// scalar math per dtype
// ------------------------------------------------------------------------------------------------
template <typename R>
struct Math;

#ifndef CPMPC_F32_LIBM
#define CPMPC_F32_LIBM 0  // 1: ocml sincosf/tanhf/sqrtf and IEEE division in the fp32 kernels
#endif
#ifndef CPMPC_F32_NATIVE_TRIG
// 1 (default): v_sin_f32 / v_cos_f32.  Measured on MI355X: +13 % re-plans/s over the
// polynomial below, and the fp32-vs-fp64 control error and the closed-loop balancing accuracy are the same
// with either (fp32 rounding of the rest of the pipeline dominates).  0 selects the ~1-ulp polynomial.
#define CPMPC_F32_NATIVE_TRIG 1
#endif

// fp32: the reference is fp64-only, so the fp32 kernels are free to use bounded-range routines.
// Angles reaching sincos are at most a few turns (theta is wrapped to (-pi, pi] at every shooting node
// and RK4 stages move it by < 1 rad), where a 3-term Cody-Waite reduction by pi/2 and degree-7/8
// minimax polynomials are accurate to ~1 ulp; tanh uses the hardware exp2; sqrt and reciprocal use the
// 1-ulp hardware instructions.
template <>
struct Math<float> {
#if CPMPC_F32_LIBM
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
  static __device__ __forceinline__ float tanh(float x) { return ::tanhf(x); }
  static __device__ __forceinline__ float tanh_scaled(float x, float scale, float) { return ::tanhf(x * scale); }
  static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
  static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
#else
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) {
#if CPMPC_F32_NATIVE_TRIG
    s = __sinf(x);  // v_sin_f32 / v_cos_f32 on x/(2 pi); arguments here are at most a few turns
    c = __cosf(x);
    return;
#endif
    const float kf = ::rintf(x * 0.63661977236758134f);  // nearest multiple of pi/2
    float r = ::fmaf(-kf, 1.5703125f, x);                // pi/2 split in three (Cody-Waite)
    r = ::fmaf(-kf, 4.837512969970703125e-4f, r);
    r = ::fmaf(-kf, 7.54978995489188e-8f, r);
    const int q = (int)kf;
    const float r2 = r * r;
    float sp = ::fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = ::fmaf(sp, r2, -1.6666654611e-1f);
    sp = ::fmaf(sp * r2, r, r);
    float cp = ::fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = ::fmaf(cp, r2, 4.166664568298827e-2f);
    cp = ::fmaf(cp * r2, r2, ::fmaf(r2, -0.5f, 1.0f));
    const bool swap = q & 1;
    const float ss = swap ? cp : sp;
    const float cc = swap ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
  }
  static __device__ __forceinline__ float tanh(float x) {
    const float t = __expf(-2.0f * ::fabsf(x));  // in (0, 1]
    const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return ::copysignf(r, x);
  }
  // tanh(x * scale) with k2 = -2 log2(e) scale precomputed: one multiply feeds v_exp_f32 directly
  static __device__ __forceinline__ float tanh_scaled(float x, float /*scale*/, float k2) {
    const float t = __builtin_amdgcn_exp2f(::fabsf(x) * k2);  // exp(-2 |x| scale), in (0, 1]
    const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return ::copysignf(r, x);
  }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
  static constexpr bool kIncrementalTrig = false;  // v_sin_f32 / v_cos_f32 are cheaper than a rotation
  static constexpr bool kMergedReciprocals = false;
  static __device__ __forceinline__ float div(float a, float b) { return a / b; }
  // |v| and 1/|v| (0 when |v| = 0: the hardware sqrt flushes a denormal argument to zero, see cartpole_accel)
  static __device__ __forceinline__ void sqrt_inv(float x, float& s, float& r) {
    s = sqrt(x);
    r = (0.0f < s) ? rcp(s) : 0.0f;
  }
  static __device__ __forceinline__ float sqrt_only(float x) { return sqrt(x); }
  static __device__ __forceinline__ float fabs(float x) { return ::fabsf(x); }
  static __device__ __forceinline__ float trunc(float x) { return ::truncf(x); }
  static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static __device__ __forceinline__ bool finite(float x) { return ::isfinite(x); }
};

// fp64: the parity dtype.  The inlined library sincos / tanh cost 155 / 165 instructions per call (an RK4 step
// without Jacobians 1 470, against ~45 per stage of actual dynamics), so fp64 gets its own routines, accurate to
// about one ulp on the ranges this path produces (-DCPMPC_F64_LIBM=1 restores the library calls for A/B):
//   sincos  Cody-Waite reduction by pi/2 in three 33-bit pieces (exact products for |x| < 2^20 * pi/2; pole angles
//           are wrapped to (-pi, pi] at every node) + the classic degree-13 / degree-14 minimax kernels on
//           [-pi/4, pi/4]; beyond that range a finite argument gives (0, 1) and a non-finite one NaN
//   tanh    -t / (t + 2) with t = expm1(-2|x|): n = rint(y / ln 2), r = y - n ln 2 (hi/lo), degree-13 Taylor
//           kernel on |r| <= ln2/2, t = 2^n p + (2^n - 1); exact odd symmetry, full relative accuracy at small |x|
#ifndef CPMPC_F64_LIBM
#define CPMPC_F64_LIBM 0
#endif
// Horner step p*z + C with the coefficient C read straight from a scalar register pair (VOP3 v_fma_f64 takes one
// SGPR source).  Left to itself hipcc 7.2 keeps the ~24 fp64 polynomial coefficients of sincos/tanh in VGPRs across
// the RK4 loops and, the kernel being at its register limit, re-assembles each one into an aligned VGPR pair
// (v_mov_b64 + v_mov_b32) in front of a two-address v_fmac_f64: about 30 of the ~160 vector instructions of a stage
// were such copies.  Materialising a coefficient in SGPRs is scalar-unit work that issues beside the vector stream.
#ifndef CPMPC_F64_SGPR_COEF
#define CPMPC_F64_SGPR_COEF 1
#endif
__device__ __forceinline__ double horner(double p, double z, double c) {
#if CPMPC_F64_SGPR_COEF
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
  return r;
#else
  return ::fma(p, z, c);
#endif
}

#ifndef CPMPC_F64_IEEE_DIV
#define CPMPC_F64_IEEE_DIV 0  // 1: compiler-expanded IEEE division and sqrt in the fp64 kernels (A/B of the Newton routines)
#endif
// The fp64 polynomial coefficients, read from constant memory instead of being immediates: as immediates every use costs
// two s_mov_b32 in front of the v_fma_f64 that takes the pair (the kernel has no scalar registers to keep ~35 pairs
// resident), and at ONE wave per SIMD -- where the fp64 fused kernel runs -- a scalar instruction takes an issue slot of
// the wave like a vector one does (measured: 163 k instructions per wave x 4.75 cycles = the wave's residence time).
// A scalar load brings up to eight coefficients in one instruction.  -DCPMPC_F64_COEF_TABLE=0 restores the immediates.
#ifndef CPMPC_F64_COEF_TABLE
#define CPMPC_F64_COEF_TABLE 1
#endif
#if CPMPC_F64_COEF_TABLE
__constant__ double kCoef64[23] = {  // [0..7] sincos, [8..18] expm1 (tanh), [19..22] the rotation's own (it shares 12, 15, 16)
    2.75573137070700676789e-06,
    -1.98412698298579493134e-04,
    8.33333333332248946124e-03,
    -1.66666666666666324348e-01,
    -2.75573143513906633035e-07,
    2.48015872894767294178e-05,
    -1.38888888888741095749e-03,
    4.16666666666666019037e-02,
    2.08767569878680989792e-09,
    2.50521083854417187751e-08,
    2.75573192239858906526e-07,
    2.75573192239858906526e-06,
    2.48015873015873015873e-05,
    1.98412698412698412698e-04,
    1.38888888888888888889e-03,
    8.33333333333333333333e-03,
    4.16666666666666666667e-02,
    1.66666666666666666667e-01,
    0.5,
    -1.98412698412698412698e-04,
    -1.66666666666666666667e-01,
    -1.38888888888888888889e-03,
    -0.5,
};
// (Reading the table through a pointer made opaque at the top of each routine, so that the loads stay inside the RK4
// loops instead of being hoisted, spilled and read back with v_readlane, was tried: 27 scalar loads and 17 waits per step
// replace 18 v_readlane and 22 s_mov -- no gain.)
#define CPMPC_C64(I, LITERAL) (kCoef64[I])
#else
#define CPMPC_C64(I, LITERAL) (LITERAL)
#endif
template <>
struct Math<double> {
#if CPMPC_F64_LIBM
  static __device__ __forceinline__ void sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
  static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
#else
  static __device__ __forceinline__ void sincos(double x, double& s, double& c) {
    const double kf = ::rint(x * 6.36619772367581382433e-01);  // nearest multiple of pi/2
    double r = ::fma(-kf, 1.57079632673412561417e+00, x);      // pi/2 = P1 + P2 + P3 (+ 8.5e-32)
    r = ::fma(-kf, 6.07710050630396597660e-11, r);
    r = ::fma(-kf, 2.02226624871116645580e-21, r);
    const double z = r * r;
    // sin(r) = r + r z S(z)
    double sp = ::fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    sp = horner(sp, z, CPMPC_C64(0, 2.75573137070700676789e-06));
    sp = horner(sp, z, CPMPC_C64(1, -1.98412698298579493134e-04));
    sp = horner(sp, z, CPMPC_C64(2, 8.33333333332248946124e-03));
    sp = horner(sp, z, CPMPC_C64(3, -1.66666666666666324348e-01));
    const double sr = ::fma(r * z, sp, r);
    // cos(r) = w + ((1 - w) - z/2 + z z C(z)),  w = 1 - z/2
    double cp = ::fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    cp = horner(cp, z, CPMPC_C64(4, -2.75573143513906633035e-07));
    cp = horner(cp, z, CPMPC_C64(5, 2.48015872894767294178e-05));
    cp = horner(cp, z, CPMPC_C64(6, -1.38888888888741095749e-03));
    cp = horner(cp, z, CPMPC_C64(7, 4.16666666666666019037e-02));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * z * cp);
    // quadrant.  |x| beyond the exact-reduction range is not an angle a live problem holds, but a diverging line
    // search trial can: there the C library (and so the reference) still returns SOME value in [-1, 1], the merit of
    // the trial stays finite-and-huge and the backtracking takes its "far too long" branch; a NaN here would take the
    // other branch (measured on a double-pendulum problem: step halved instead of cut to a tenth, a different
    // accepted step after five trials).  So: huge finite x -> (0, 1); inf / NaN -> NaN, as sin and cos do.
    const bool ok = ::fabs(kf) < 1048576.0;
    const int q = ok ? (int)kf : 0;
    const double ss = (q & 1) ? cr : sr;
    const double cc = (q & 1) ? sr : cr;
    const double zero_or_nan = x - x;  // 0 for a finite x, NaN otherwise
    s = ok ? ((q & 2) ? -ss : ss) : zero_or_nan;
    c = ok ? (((q + 1) & 2) ? -cc : cc) : 1.0 + zero_or_nan;
  }
  static __device__ __forceinline__ double tanh(double x) {
    double y = -2.0 * ::fabs(x);
    y = (y < -80.0) ? -80.0 : y;  // 1 - tanh(40) < 2^-53: the clamp does not change the rounded result (NaN stays NaN)
    const double nf = ::rint(y * 1.44269504088896338700e+00);
    double r = ::fma(-nf, 6.93147180369123816490e-01, y);  // ln 2 = hi + lo
    r = ::fma(-nf, 1.90821492927058770002e-10, r);
    // expm1(r) = r + r^2 (1/2! + r (1/3! + ... + r / 13!)),  |r| <= ln2 / 2
    double p = 1.6059043836821613e-10;                       // 1/13!
    p = horner(p, r, CPMPC_C64(8, 2.08767569878680989792e-09));            // 1/12!
    p = horner(p, r, CPMPC_C64(9, 2.50521083854417187751e-08));            // 1/11!
    p = horner(p, r, CPMPC_C64(10, 2.75573192239858906526e-07));            // 1/10!
    p = horner(p, r, CPMPC_C64(11, 2.75573192239858906526e-06));            // 1/9!
    p = horner(p, r, CPMPC_C64(12, 2.48015873015873015873e-05));            // 1/8!
    p = horner(p, r, CPMPC_C64(13, 1.98412698412698412698e-04));            // 1/7!
    p = horner(p, r, CPMPC_C64(14, 1.38888888888888888889e-03));            // 1/6!
    p = horner(p, r, CPMPC_C64(15, 8.33333333333333333333e-03));            // 1/5!
    p = horner(p, r, CPMPC_C64(16, 4.16666666666666666667e-02));            // 1/4!
    p = horner(p, r, CPMPC_C64(17, 1.66666666666666666667e-01));            // 1/3!
    p = horner(p, r, CPMPC_C64(18, 0.5));                                   // 1/2!
    p = ::fma(r * r, p, r);
    const double two_n = ::ldexp(1.0, (int)nf);              // n in [-116, 0]
    const double t = ::fma(two_n, p, two_n - 1.0);           // expm1(y) in (-1, 0]
    return ::copysign(div(-t, t + 2.0), x);
  }
  // tanh(x) = sign(x) num / den with num = -expm1(-2|x|) >= 0, den = expm1(-2|x|) + 2 in (1, 2]: the quotient is left to
  // the caller, who has another reciprocal to take at the same place and takes ONE of the product (cartpole_accel_sc)
  // e2 = exp(-2|x|) comes along for the slope: sech^2 x = 4 e2 / den^2 has full relative accuracy where 1 - tanh^2
  // cancels (a saturated friction term with a tiny v_mu multiplies that slope by up to 1e6).
  static __device__ __forceinline__ void tanh_parts(double x, double& num, double& den, double& e2) {
    double y = -2.0 * ::fabs(x);
    y = ::fmax(y, -80.0);  // one v_max_f64 (a NaN argument becomes -80 here; the caller's other terms keep the NaN)
    const double nf = ::rint(y * 1.44269504088896338700e+00);
    double r = ::fma(-nf, 6.93147180369123816490e-01, y);
    r = ::fma(-nf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = horner(p, r, CPMPC_C64(8, 2.08767569878680989792e-09));
    p = horner(p, r, CPMPC_C64(9, 2.50521083854417187751e-08));
    p = horner(p, r, CPMPC_C64(10, 2.75573192239858906526e-07));
    p = horner(p, r, CPMPC_C64(11, 2.75573192239858906526e-06));
    p = horner(p, r, CPMPC_C64(12, 2.48015873015873015873e-05));
    p = horner(p, r, CPMPC_C64(13, 1.98412698412698412698e-04));
    p = horner(p, r, CPMPC_C64(14, 1.38888888888888888889e-03));
    p = horner(p, r, CPMPC_C64(15, 8.33333333333333333333e-03));
    p = horner(p, r, CPMPC_C64(16, 4.16666666666666666667e-02));
    p = horner(p, r, CPMPC_C64(17, 1.66666666666666666667e-01));
    p = horner(p, r, CPMPC_C64(18, 0.5));
    p = ::fma(r * r, p, r);
    const double two_n = ::ldexp(1.0, (int)nf);
    const double t = ::fma(two_n, p, two_n - 1.0);
    num = -t;
    den = t + 2.0;
    e2 = ::fma(two_n, p, two_n);
  }
  // sin and cos of th0 + d from (s0, c0) = sincos(th0) by one rotation, for |d| <= 1: the Taylor kernels of sin and
  // cos - 1 are evaluated at d/4 (|d/4| <= 1/4: exact to 3e-18 through d^11 / d^12) and doubled twice
  // (sin 2a = 2 sin a (1 + (cos a - 1)), cos 2a - 1 = -2 sin^2 a: no cancellation), and the result is formed as
  // s0 + (small), c0 + (small), so it carries the rounding of the base pair plus an ulp or two.  The stages of one RK4
  // step evaluate the pole angle at th, th + (h/2) w1, th + (h/2) w2, th + h w3: |d| <= 1 covers poles up to 100 rad/s
  // at the reference's step of 10 ms (measured on the benchmark distribution: 12 % of the steps exceed 25 rad/s, none
  // 90) -- 26 vector instructions against ~45 for the argument reduction, both minimax kernels and the quadrant selects
  // of a full sincos.
  static __device__ __forceinline__ void sincos_delta(double s0, double c0, double d, double& s, double& c) {
    const double q = 0.25 * d;
    const double z = q * q;
    double sp = ::fma(z, -2.50521083854417187751e-08, 2.75573192239858906526e-06);  // -1/11!, 1/9!
    sp = horner(sp, z, CPMPC_C64(19, -1.98412698412698412698e-04));                                // -1/7!
    sp = horner(sp, z, CPMPC_C64(15, 8.33333333333333333333e-03));                                 // 1/5!
    sp = horner(sp, z, CPMPC_C64(20, -1.66666666666666666667e-01));                                // -1/3!
    double sd = ::fma(q * z, sp, q);                                                // sin(d/4)
    double cp = ::fma(z, 2.08767569878680989792e-09, -2.75573192239858906526e-07);  // 1/12!, -1/10!
    cp = horner(cp, z, CPMPC_C64(12, 2.48015873015873015873e-05));                                 // 1/8!
    cp = horner(cp, z, CPMPC_C64(21, -1.38888888888888888889e-03));                                // -1/6!
    cp = horner(cp, z, CPMPC_C64(16, 4.16666666666666666667e-02));                                 // 1/4!
    cp = horner(cp, z, CPMPC_C64(22, -0.5));                                                       // -1/2!
    double cm1 = z * cp;                                                            // cos(d/4) - 1
#pragma unroll
    for (int dbl = 0; dbl < 2; ++dbl) {  // a -> 2a
      const double t = sd + sd;
      const double s2 = ::fma(t, cm1, t);
      cm1 = -t * sd;
      sd = s2;
    }
    s = s0 + ::fma(c0, sd, s0 * cm1);
    c = c0 + ::fma(-s0, sd, c0 * cm1);
  }
#endif
#ifndef CPMPC_F64_TRIG_ROTATE
#define CPMPC_F64_TRIG_ROTATE 1  // 0: a full sincos at every RK4 stage (A/B of the rotation above)
#endif
#ifndef CPMPC_F64_MERGED_RCP
#define CPMPC_F64_MERGED_RCP 1   // 0: tanh's quotient and 1/den by two separate Newton chains (A/B)
#endif
  static constexpr bool kIncrementalTrig = CPMPC_F64_TRIG_ROTATE && !CPMPC_F64_LIBM;
  static constexpr bool kMergedReciprocals = CPMPC_F64_MERGED_RCP && !CPMPC_F64_LIBM && !CPMPC_F64_IEEE_DIV;
  static __device__ __forceinline__ double tanh_scaled(double x, double scale, double) { return tanh(x * scale); }
#if CPMPC_F64_LIBM || CPMPC_F64_IEEE_DIV
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ void sqrt_inv(double x, double& s, double& r) {
    s = ::sqrt(x);
    r = (0.0 < s) ? 1.0 / s : 0.0;
  }
  static __device__ __forceinline__ double sqrt_only(double x) { return ::sqrt(x); }
#else
  // The compiler's IEEE division is 11 instructions around v_rcp_f64 (two v_div_scale, four refinement fma, the
  // quotient and its residual, v_div_fmas, v_div_fixup) and its sqrt 15 around v_rsq_f64 (range scaling either side);
  // an RK4 stage holds three reciprocals and a square root.  The operands here are ordinary magnitudes (no scaling
  // needed), so: hardware seed + two Newton steps (each squares the error: seed >= 2^-23 -> 2^-46 -> below an ulp),
  // and for a quotient one residual correction.  Results are within an ulp of the correctly rounded ones
  // (test_fp64_math_routines_over_wide_ranges); x = 0 gives NaN instead of inf (no caller divides by zero on a
  // problem that is still being solved: pivots and |v| are tested first).
  static __device__ __forceinline__ double rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = ::fma(-x, y, 1.0);
    y = ::fma(y, e, y);
    e = ::fma(-x, y, 1.0);
    return ::fma(y, e, y);
  }
  // Quotients sit in the cold places (penalty update, step-length interpolation, the 4x4 pivots), where the operands are
  // not guaranteed to be ordinary: v_div_fixup_f64 gives the IEEE result for a zero, infinite, denormal or NaN operand
  // (b = +-inf -> 0, b = 0 -> +-inf, a denormal b whose seed overflowed -> the correctly signed infinity instead of
  // inf - inf = NaN), one instruction.  rcp() above stays bare: its callers (the dynamics' den > 0, pivots tested
  // positive first) guarantee an ordinary operand.
  static __device__ __forceinline__ double div(double a, double b) {
    const double y = rcp(b);
    const double q = a * y;
    return __builtin_amdgcn_div_fixup(::fma(::fma(-b, q, a), y, q), b, a);
  }
  // sqrt(x) and 1/sqrt(x) together (coupled Goldschmidt iteration from v_rsq_f64): the drag terms need both.
  // x = 0 -> (0, 0): "no drag at rest" (single_pendulum_dynamics.hpp:75-84 guards on 0 < |v|^2).
  static __device__ __forceinline__ void sqrt_inv(double x, double& s, double& r) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = ::fma(-h, g, 0.5);
    g = ::fma(g, e, g);
    h = ::fma(h, e, h);
    e = ::fma(-h, g, 0.5);
    g = ::fma(g, e, g);
    h = ::fma(h, e, h);
    g = ::fma(::fma(-g, g, x), h, g);
    const bool zero = (x == 0.0);
    s = zero ? 0.0 : g;
    r = zero ? 0.0 : h + h;
  }
  static __device__ __forceinline__ double sqrt(double x) {
    double s, r;
    sqrt_inv(x, s, r);
    return s;
  }
  // sqrt alone, for the Jacobian-free dynamics (|v| of the drag term; 1/|v| is only needed by the partials): one
  // coupled step from the seed (error 2^-23 -> 2^-45) and the residual correction (-> below an ulp), and the zero
  // argument handled by a floor of 1e-200 (sqrt 1e-100: the drag products it enters vanish with the velocities) instead
  // of a compare and two selects.  Ten instructions against fifteen.
  static __device__ __forceinline__ double sqrt_only(double x) {
    const double xm = ::fmax(x, 1.0e-200);
    const double y = __builtin_amdgcn_rsq(xm);
    double g = xm * y;
    const double h = 0.5 * y;
    const double e = ::fma(-h, g, 0.5);
    g = ::fma(g, e, g);
    const double h1 = ::fma(h, e, h);
    return ::fma(::fma(-g, g, xm), h1, g);
  }
#endif
  static __device__ __forceinline__ double fabs(double x) { return ::fabs(x); }
  static __device__ __forceinline__ double trunc(double x) { return ::trunc(x); }
  static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static __device__ __forceinline__ bool finite(double x) { return ::isfinite(x); }
};

// Map an angle to (-pi, pi]; same cut as integration.hpp:65-73 (mod_pi(+-pi) = +pi).
// fmod(a, 2pi) is evaluated as a - trunc(a/2pi)*2pi with one fused multiply-add: the exact remainder
// is representable, so the fma returns it exactly whenever the quotient is right, and a quotient off
// by one (a within an ulp of a multiple of 2pi) is repaired by the two range fix-ups below.
template <typename R>
__device__ __forceinline__ R mod_pi(R angle) {
  constexpr R pi = static_cast<R>(3.14159265358979323846);
  constexpr R two_pi = 2 * pi;
  constexpr R inv_two_pi = static_cast<R>(0.15915494309189533577);
  const R n = Math<R>::trunc(angle * inv_two_pi);
  angle = Math<R>::fma(-n, two_pi, angle);
  angle += (angle < R(0)) ? two_pi : R(0);
  angle -= (angle > pi) ? two_pi : R(0);
  return angle;
}

// ------------------------------------------------------------------------------------------------
// model constants: SingleCartPoleParams (structs.hpp:8-41) plus the sub-expressions that depend on
// the parameters only, hoisted out of the per-stage evaluation.
// ------------------------------------------------------------------------------------------------
template <typename R>
struct CartPoleConsts {
  R m_1, L, g, xs, ks;
  R mt;         // m_1 + m_b
  R inv_v_mu;   // 1 / max(v_mu_b, 1e-6)
  R fr;         // -(mt * mu_b) * g      friction scale
  R inv_L;      // 1 / L
  R kap;        // mt / (m_1 L^2)
  R m1L;        // m_1 * L
  R gm1L;       // g * m_1 * L
  R half_cd;    // c_d / 2
  R half_cd_L;  // c_d L / 2
  R tanh_k2;    // -2 log2(e) / v_mu   (fp32 tanh: exp2 argument scale)
  R fr_vmu;     // fr / v_mu           (slope scale of the friction term)
  R two_m1L;    // 2 m_1 L
};

template <typename R, typename P>
__host__ __device__ inline CartPoleConsts<R> make_consts(const P* p) {
  // p = {m_b, m_1, l_1, g, mu_b, v_mu_b, c_d_1, x_s, k_s}; evaluated in P, rounded once to R.
  const P m_b = p[0], m_1 = p[1], L = p[2], g = p[3], mu = p[4], v_mu_in = p[5], cd = p[6];
  const P mt = m_1 + m_b;
  const P v_mu = (P(1.0e-6) < v_mu_in) ? v_mu_in : P(1.0e-6);
  CartPoleConsts<R> k;
  k.m_1 = R(m_1);
  k.L = R(L);
  k.g = R(g);
  k.xs = R(p[7]);
  k.ks = R(p[8]);
  k.mt = R(mt);
  k.inv_v_mu = R(P(1) / v_mu);
  k.fr = R((mt * mu) * -g);
  k.inv_L = R(P(1) / L);
  k.kap = R(mt / (m_1 * L * L));
  k.m1L = R(m_1 * L);
  k.gm1L = R(g * m_1 * L);
  k.half_cd = R(P(0.5) * cd);
  k.half_cd_L = R(P(0.5) * cd * L);
  k.tanh_k2 = R(P(-2.8853900817779268) / v_mu);
  k.fr_vmu = R(((mt * mu) * -g) / v_mu);
  k.two_m1L = R(P(2) * m_1 * L);
  return k;
}

// External forces {f_base.x, f_mass.x, f_mass.y}; f_base.y never enters the equations.
template <typename R>
struct ExtForce {
  R fbx, fmx, fmy;
};

// ------------------------------------------------------------------------------------------------
// accelerations (b_x'', th_1'') and, optionally, their partials.
//   Ja[r][c] : d(acc_r)/d(b_x, th, b_x', th')      Jua[r] : d(acc_r)/du
// The full 4x4 stage Jacobian is [[0 0 1 0],[0 0 0 1],[Ja]] (single_pendulum_dynamics.hpp:159-166);
// only the two non-trivial rows are ever materialised.
// ------------------------------------------------------------------------------------------------
template <typename R, bool WITH_J, bool HAS_EXT>
__device__ __forceinline__ void cartpole_accel_sc(const CartPoleConsts<R>& k, const R bx, const R s, const R c,
                                                  const R v, const R w, const R u,
                                                  const ExtForce<R>& fe, R& a_x, R& a_th,
                                                  R (&Ja)[2][4], R (&Jua)[2]) {
  // (s, c) = sin, cos of the pole angle; the angle itself enters the equations of motion through them only

  // bumper springs (strict comparisons, as the generated branches)
  const R e_r = bx - k.xs;
  const R e_l = -(bx + k.xs);
  const bool on_r = R(0) < e_r;
  const bool on_l = R(0) < e_l;
  const R F_s = k.ks * ((on_l ? e_l : R(0)) - (on_r ? e_r : R(0)));

  // smoothed Coulomb friction, and 1 / (m_t - m_1 s^2).  Where a reciprocal is a Newton chain (fp64) the friction
  // tanh = num / dt and 1 / den share ONE reciprocal, of the product: 1/den = r dt, tanh = num (r den)
  const R den = k.mt - k.m_1 * s * s;
  R tv, inv_den, sech2 = R(0);
  if constexpr (Math<R>::kMergedReciprocals) {
    R num, dt, e2;
    Math<R>::tanh_parts(v * k.inv_v_mu, num, dt, e2);
    const R r = Math<R>::rcp(den * dt);
    inv_den = r * dt;
    const R idt = r * den;
    const R q = num * idt;
    if (WITH_J) sech2 = R(4) * e2 * (idt * idt);
    tv = __builtin_copysign(q, v);  // one v_bfi_b32 (q >= 0; a NaN speed gives a NaN quotient either way)
  } else {
    tv = Math<R>::tanh_scaled(v, k.inv_v_mu, k.tanh_k2);
    inv_den = Math<R>::rcp(den);
  }
  const R F_f = tv * k.fr;

  // air drag on the pole mass
  const R Lw = k.L * w;
  const R vx = v - Lw * s;
  const R vy = Lw * c;
  const R n2 = vx * vx + vy * vy;
  R n, inv_n = R(0);  // |v| and 1/|v| (0 at rest); inv_n is only used by the partials
  if (WITH_J) {
    Math<R>::sqrt_inv(n2, n, inv_n);
  } else {
    n = Math<R>::sqrt_only(n2);
  }
  const R e = Lw - s * v;  // = c*vy - s*vx
  // the generated code guards these with |v|^2 > 0; at |v| = 0 the products are 0 anyway
  const R Dx = k.half_cd * n * vx;
  const R Dth = k.half_cd_L * n * e;

  R F_b = u + F_f + F_s - Dx + k.m1L * w * w * c;
  R F_th = -k.gm1L * c - Dth;
  if (HAS_EXT) {
    F_b += fe.fbx + fe.fmx;
    F_th += k.L * (fe.fmy * c - fe.fmx * s);
  }

  const R sl = s * k.inv_L;
  const R N_x = F_b + sl * F_th;
  const R N_th = sl * F_b + k.kap * F_th;
  a_x = N_x * inv_den;
  a_th = N_th * inv_den;

  if (WITH_J) {
    // partials of the drag terms with respect to (th, v, w); all vanish at |v| = 0 (the guard of the
    // generated code), which a zero 1/|v| reproduces without a branch.  (The guard is on |v| itself, not |v|^2:
    // the fp32 hardware sqrt flushes a denormal |v|^2 to |v| = 0, and 1/0 would turn the vanishing products
    // below into NaN; in fp64 the two tests are equivalent.)
    // d|v|/d(th, v, w), with  vx + L w s = v  and  c vy - s vx = e  folded in
    const R dn0 = -(vy * v) * inv_n;
    const R dn1 = vx * inv_n;
    const R dn2 = (k.L * e) * inv_n;
    const R dDx0 = k.half_cd * (dn0 * vx - n * vy);
    const R dDx1 = k.half_cd * (dn1 * vx + n);
    const R dDx2 = k.half_cd * (dn2 * vx - n * (k.L * s));
    const R dDt0 = k.half_cd_L * (dn0 * e - n * (c * v));
    const R dDt1 = k.half_cd_L * (dn1 * e - n * s);
    const R dDt2 = k.half_cd_L * (dn2 * e + n * k.L);
    const R dFf_dv = (Math<R>::kMergedReciprocals ? sech2 : (R(1) - tv * tv)) * k.fr_vmu;
    const R dFs_dbx = k.ks * ((on_l ? R(-1) : R(0)) - (on_r ? R(1) : R(0)));

    const R dFb0 = -dDx0 - k.m1L * w * w * s;
    const R dFb1 = dFf_dv - dDx1;
    const R dFb2 = -dDx2 + k.two_m1L * w * c;
    R dFt0 = k.gm1L * s - dDt0;
    if (HAS_EXT) dFt0 += k.L * (-fe.fmx * c - fe.fmy * s);
    const R dFt1 = -dDt1;
    const R dFt2 = -dDt2;

    const R dden = R(-2) * k.m_1 * s * c;
    const R cl = c * k.inv_L;
    const R dNx0 = dFb0 + cl * F_th + sl * dFt0;
    const R dNx1 = dFb1 + sl * dFt1;
    const R dNx2 = dFb2 + sl * dFt2;
    const R dNt0 = cl * F_b + sl * dFb0 + k.kap * dFt0;
    const R dNt1 = sl * dFb1 + k.kap * dFt1;
    const R dNt2 = sl * dFb2 + k.kap * dFt2;

    Ja[0][0] = dFs_dbx * inv_den;
    Ja[0][1] = (dNx0 - a_x * dden) * inv_den;
    Ja[0][2] = dNx1 * inv_den;
    Ja[0][3] = dNx2 * inv_den;
    Ja[1][0] = sl * dFs_dbx * inv_den;
    Ja[1][1] = (dNt0 - a_th * dden) * inv_den;
    Ja[1][2] = dNt1 * inv_den;
    Ja[1][3] = dNt2 * inv_den;
    Jua[0] = inv_den;       // single_pendulum_dynamics.hpp:179-184
    Jua[1] = sl * inv_den;
  }
}

template <typename R, bool WITH_J, bool HAS_EXT>
__device__ __forceinline__ void cartpole_accel(const CartPoleConsts<R>& k, const R bx, const R th,
                                               const R v, const R w, const R u,
                                               const ExtForce<R>& fe, R& a_x, R& a_th,
                                               R (&Ja)[2][4], R (&Jua)[2]) {
  R s, c;
  Math<R>::sincos(th, s, c);
  cartpole_accel_sc<R, WITH_J, HAS_EXT>(k, bx, s, c, v, w, u, fe, a_x, a_th, Ja, Jua);
}

// The pole angle's sine and cosine across the stages of an RK4 step and the steps of a rollout: the first step's stage 1
// evaluates them in full and keeps the base; everything after is a rotation away (Math<R>::sincos_delta) unless a lane
// moves further than the kernels cover (|d| > 1 rad within one step: a pole at more than 100 rad/s, or a wrapped angle),
// which takes the full path for that lane.
template <typename R>
struct TrigBase {
  R th0, s0, c0;
  bool valid = false;  // (th0, s0, c0) hold the previous step's stage-1 pair: this step's stage 1 may rotate from it
  __device__ __forceinline__ void invalidate() { valid = false; }  // the next stage 1 evaluates in full (re-anchors the chain)
};
// The far branch of a rotation (|d| > 1 rad, or a NaN) is taken per LANE, never per wave: a problem's arithmetic must
// not depend on what its neighbours in the wave do (batch-position independence: staged, sharded and stand-alone solves
// are bitwise equal).
template <typename R>
__device__ __forceinline__ void sincos_from_base(const TrigBase<R>& tb, const R th, R& s, R& c) {
  const R d = th - tb.th0;
  Math<R>::sincos_delta(tb.s0, tb.c0, d, s, c);
#ifndef CPMPC_F64_TRIG_NO_FALLBACK  // (defined only to count the main path's instructions in a listing)
  if (__builtin_expect(!(Math<R>::fabs(d) <= R(1)), 0)) Math<R>::sincos(th, s, c);
#endif
}
// 0 (default): stage 1 of every RK4 step evaluates sine and cosine in full.  1: stage 1 rotates from the previous step's
// pair when there is one (round 3: the branch on the loop-carried flag and the second code path cost more than the 19
// instructions saved, 48.9 against 49.9 M re-plans/s).  2 (round 5): the same, and the fused kernel's two rollouts evaluate
// the base in full BEFORE their loops (Model::chain_begin), so that inside them the flag is a constant and stage 1 is
// always a rotation -- no branch, one code path.
#ifndef CPMPC_F64_TRIG_CHAIN
#define CPMPC_F64_TRIG_CHAIN 0
#endif
template <typename R, int STAGE, int CHAIN = CPMPC_F64_TRIG_CHAIN>
__device__ __forceinline__ void stage_sincos(TrigBase<R>& tb, const R th, R& s, R& c) {
  if constexpr (Math<R>::kIncrementalTrig) {
    if constexpr (STAGE == 1) {
      // consecutive steps of one rollout: this step's angle is within h |w| of the previous step's, so stage 1 rotates
      // from that pair too (one more rotation per step: the chain's error grows like the square root of its length,
      // ~1.5 ulp over the ten steps of an interval); a wrapped angle jumps by 2 pi and takes the far branch
      if (CHAIN && tb.valid) {
        sincos_from_base<R>(tb, th, s, c);
      } else {
        Math<R>::sincos(th, s, c);
      }
      tb.th0 = th;
      tb.s0 = s;
      tb.c0 = c;
      tb.valid = true;
    } else {
      sincos_from_base<R>(tb, th, s, c);
    }
  } else {
    Math<R>::sincos(th, s, c);
  }
}

// ------------------------------------------------------------------------------------------------
// RK4 without sensitivities (integration.hpp:52-62).  x updated in place.
// ------------------------------------------------------------------------------------------------
template <typename R, bool HAS_EXT>
__device__ __forceinline__ void rk4_step(const CartPoleConsts<R>& k, const R h, R (&x)[4],
                                         const R u, const ExtForce<R>& fe) {
  R Ja[2][4], Jua[2];
  const R hh = h / R(2);
  R a1x, a1t, a2x, a2t, a3x, a3t, a4x, a4t;
  // k1 = f(x)
  cartpole_accel<R, false, HAS_EXT>(k, x[0], x[1], x[2], x[3], u, fe, a1x, a1t, Ja, Jua);
  const R k1_0 = x[2], k1_1 = x[3];
  // k2 = f(x + k1 h/2)
  const R k2_0 = x[2] + a1x * hh, k2_1 = x[3] + a1t * hh;
  cartpole_accel<R, false, HAS_EXT>(k, x[0] + k1_0 * hh, x[1] + k1_1 * hh, k2_0, k2_1, u, fe, a2x,
                                    a2t, Ja, Jua);
  // k3 = f(x + k2 h/2)
  const R k3_0 = x[2] + a2x * hh, k3_1 = x[3] + a2t * hh;
  cartpole_accel<R, false, HAS_EXT>(k, x[0] + k2_0 * hh, x[1] + k2_1 * hh, k3_0, k3_1, u, fe, a3x,
                                    a3t, Ja, Jua);
  // k4 = f(x + k3 h)
  const R k4_0 = x[2] + a3x * h, k4_1 = x[3] + a3t * h;
  cartpole_accel<R, false, HAS_EXT>(k, x[0] + k3_0 * h, x[1] + k3_1 * h, k4_0, k4_1, u, fe, a4x,
                                    a4t, Ja, Jua);
  const R h6 = h / R(6);
  x[0] += h6 * (k1_0 + k2_0 * R(2) + k3_0 * R(2) + k4_0);
  x[1] += h6 * (k1_1 + k2_1 * R(2) + k3_1 * R(2) + k4_1);
  x[2] += h6 * (a1x + a2x * R(2) + a3x * R(2) + a4x);
  x[3] += h6 * (a1t + a2t * R(2) + a3t * R(2) + a4t);
}

// ------------------------------------------------------------------------------------------------
// RK4 with sensitivities A = dx+/dx (4x4), Bv = dx+/du (4)   (integration.hpp:13-49).
//
// Stage chain rule with K_j = [[0 I],[Ja_j]] the stage Jacobian at the stage argument:
//   D_1 = K_1,  D_{j+1} = K_{j+1} (I + a_j D_j),  a = {h/2, h/2, h}
//   A = I + h/6 (D_1 + 2 D_2 + 2 D_3 + D_4)
// The top two rows of K X are rows 2,3 of X and the bottom two are Ja X, so each stage product
// costs 2x4x4 multiply-adds instead of 4x4x4.  Same for the control column d_j.
// ------------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ void stage_chain(const R (&Ja)[2][4], const R (&Jua)[2], const R a,
                                            const R (&D)[4][4], const R (&d)[4], R (&Dn)[4][4],
                                            R (&dn)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    Dn[0][c] = a * D[2][c] + (c == 2 ? R(1) : R(0));
    Dn[1][c] = a * D[3][c] + (c == 3 ? R(1) : R(0));
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const R acc = Ja[r][0] * D[0][c] + Ja[r][1] * D[1][c] + Ja[r][2] * D[2][c] + Ja[r][3] * D[3][c];
      Dn[2 + r][c] = Ja[r][c] + a * acc;
    }
  }
  dn[0] = a * d[2];
  dn[1] = a * d[3];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const R acc = Ja[r][0] * d[0] + Ja[r][1] * d[1] + Ja[r][2] * d[2] + Ja[r][3] * d[3];
    dn[2 + r] = a * acc + Jua[r];
  }
}

template <typename R, bool HAS_EXT>
__device__ __forceinline__ void rk4_step_jac(const CartPoleConsts<R>& k, const R h, R (&x)[4],
                                             const R u, const ExtForce<R>& fe, R (&A)[4][4],
                                             R (&Bv)[4]) {
  const R hh = h / R(2);
  R Ja[2][4], Jua[2];
  R D[4][4], d[4], Dn[4][4], dn[4];
  R As[4][4], bs[4];  // running sums D_1 + 2 D_2 + 2 D_3 + D_4

  // stage 1
  R a1x, a1t;
  cartpole_accel<R, true, HAS_EXT>(k, x[0], x[1], x[2], x[3], u, fe, a1x, a1t, Ja, Jua);
  const R k1_0 = x[2], k1_1 = x[3];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    D[0][c] = (c == 2 ? R(1) : R(0));
    D[1][c] = (c == 3 ? R(1) : R(0));
    D[2][c] = Ja[0][c];
    D[3][c] = Ja[1][c];
  }
  d[0] = R(0);
  d[1] = R(0);
  d[2] = Jua[0];
  d[3] = Jua[1];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) As[r][c] = D[r][c];
    bs[r] = d[r];
  }

  // stage 2
  R a2x, a2t;
  const R k2_0 = x[2] + a1x * hh, k2_1 = x[3] + a1t * hh;
  cartpole_accel<R, true, HAS_EXT>(k, x[0] + k1_0 * hh, x[1] + k1_1 * hh, k2_0, k2_1, u, fe, a2x,
                                   a2t, Ja, Jua);
  stage_chain<R>(Ja, Jua, hh, D, d, Dn, dn);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      As[r][c] += Dn[r][c] * R(2);
      D[r][c] = Dn[r][c];
    }
    bs[r] += dn[r] * R(2);
    d[r] = dn[r];
  }

  // stage 3
  R a3x, a3t;
  const R k3_0 = x[2] + a2x * hh, k3_1 = x[3] + a2t * hh;
  cartpole_accel<R, true, HAS_EXT>(k, x[0] + k2_0 * hh, x[1] + k2_1 * hh, k3_0, k3_1, u, fe, a3x,
                                   a3t, Ja, Jua);
  stage_chain<R>(Ja, Jua, hh, D, d, Dn, dn);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      As[r][c] += Dn[r][c] * R(2);
      D[r][c] = Dn[r][c];
    }
    bs[r] += dn[r] * R(2);
    d[r] = dn[r];
  }

  // stage 4
  R a4x, a4t;
  const R k4_0 = x[2] + a3x * h, k4_1 = x[3] + a3t * h;
  cartpole_accel<R, true, HAS_EXT>(k, x[0] + k3_0 * h, x[1] + k3_1 * h, k4_0, k4_1, u, fe, a4x, a4t,
                                   Ja, Jua);
  stage_chain<R>(Ja, Jua, h, D, d, Dn, dn);

  const R h6 = h / R(6);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) A[r][c] = (r == c ? R(1) : R(0)) + h6 * (As[r][c] + Dn[r][c]);
    Bv[r] = h6 * (bs[r] + dn[r]);
  }
  x[0] += h6 * (k1_0 + k2_0 * R(2) + k3_0 * R(2) + k4_0);
  x[1] += h6 * (k1_1 + k2_1 * R(2) + k3_1 * R(2) + k4_1);
  x[2] += h6 * (a1x + a2x * R(2) + a3x * R(2) + a4x);
  x[3] += h6 * (a1t + a2t * R(2) + a3t * R(2) + a4t);
}

}  // namespace cpmpc
