// wide.hpp -- the type the terminal Schur complement of the condensed QP is carried in.
//
// S = W^T D^-1 W is a normal-equations matrix: forming it squares the conditioning of the terminal sensitivities, and
// state elimination through the horizon of an unstable plant makes those sensitivities large (e^{6 t} for the default
// pole).  In a float kernel every quantity that feeds the NX x NX system -- the per-control block sums, their
// combination across shooting intervals, the sums over the group, the LDL^T factors and the solve for the multipliers --
// is therefore carried in double.  The products that enter the sums are exact in double (24 + 24 bits), so the
// accumulated S is the Gram matrix of the ROUNDED columns to double precision: positive semi-definite by construction,
// which is what keeps the pivots of the terminal system positive during swing-up (round 3's fp32 kernels reported
// QP_INDEFINITE on up to 0.3 % of controllers per tick there; optimization_test.cc:44-46 asserts it never happens).
// Measured on a CPU model of the elimination (N = 40, 300 cold-start problems, error of the QP step against a long-double
// dense KKT solve): all in float 2.7 worst / 1.0 p99 / 3e-3 median; S in double 4e-3 / 3e-3 / 1e-4 -- better than a
// dense float KKT solve with pivoting (2e-2 / 2e-2 / 1e-3).
// A double kernel keeps its own type.  (Round 4 tried double-double there for horizons beyond 1 s: it removes the
// rounding of S but not that of W q and of the forward state recovery, which amplify by the same e^{6 t}; the result
// was no better than plain double with its refinement pass, so it is not in the code.  Such horizons are solved with a
// warning, or refused with CPMPC_CREATE_STRICT_HORIZON: include/cpmpc.h.)
// Round 5: CPMPC_CREATE_WIDE_QP extends the double part of a float kernel from the NX x NX system to the whole terminal
// part of the QP (mpc_fused_body.inc: type Q).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

// 1 (default): the float kernels carry the terminal system in double; 0 builds the A/B variant that keeps it in float
// (the state of round 3)
#ifndef CPMPC_WIDE_F32
#define CPMPC_WIDE_F32 1
#endif

namespace cpmpc {

template <typename R>
struct WideOf {
  using type = R;
};
template <>
struct WideOf<float> {
  using type = std::conditional_t<CPMPC_WIDE_F32 != 0, double, float>;
};

template <typename W>
struct Wide {
  template <typename R>
  static __device__ __forceinline__ W of(R a) { return (W)a; }
  // product of two narrow values: exact when W = double and R = float; the rounded product when W = R
  template <typename R>
  static __device__ __forceinline__ W prod(R a, R b) { return (W)a * (W)b; }
};

}  // namespace cpmpc
