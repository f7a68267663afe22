// gen_terms_check.cc -- the HIP output of tools/gen_dynamics.py (round 6: M q'' = F terms with folded constants, sines / cosines
// as inputs, structure masks) compiled FOR THE HOST and held to the generator's C output for the CPU check, entry by entry.
// No GPU: the generated headers are scalar-templated C++ once the HIP decorations are defined away; the small LDL^T solves of
// the model policies (csrc/models.hpp, device code) are restated here in a dozen lines.
// Build (tests/test_generated_dynamics.py does it): g++ -O1 -std=c++17 -I<repo> tests/host/gen_terms_check.cc
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#define __device__
#define __host__
#define __forceinline__ inline
namespace cpmpc {
template <typename R>
struct Math;
template <>
struct Math<double> {  // the host's libm where the kernels have their own routines (cartpole_device.hpp)
  static constexpr bool kMergedReciprocals = false;  // (the kernels' double type has the split tanh; here sech^2 = 1 - tanh^2)
  static void tanh_parts(double, double&, double&, double&) {}
  static double rcp(double x) { return 1.0 / x; }
  static double tanh_scaled(double x, double scale, double) { return std::tanh(x * scale); }
  static void sqrt_inv(double x, double& s, double& r) {
    s = std::sqrt(x);
    r = (0.0 < x) ? 1.0 / s : 0.0;
  }
  static double sqrt_only(double x) { return std::sqrt(x); }
};
}  // namespace cpmpc
#include "cart-pole-mpc_amd/csrc/double_pendulum_gen.hpp"
#include "cart-pole-mpc_amd/csrc/single_pendulum_gen.hpp"
namespace ref {  // the generator's C output (what the CPU check runs)
#include "oracle/double_pendulum_gen.inc"
#include "oracle/single_pendulum_gen.inc"
}  // namespace ref

static int failures = 0;
static void expect(bool ok, const char* what, int i, double got, double want) {
  if (!ok && failures++ < 20) std::printf("FAIL %s case %d: got %.17g want %.17g (rel %.2e)\n", what, i, got, want, std::fabs(got - want) / std::fmax(1e-300, std::fabs(want)));
}
static bool close(double a, double b, double scale) { return std::fabs(a - b) <= 1e-12 * std::fmax(1.0, scale); }

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  using namespace cpmpc;
  // ---- cart + double pendulum: every entry of M, F, dF/dx, dM/dth against the straight CSE'd C code; masked entries are 0 there
  int n_double = 0;
  for (int i = 0; i < 2000; ++i) {
    double p[6] = {1.0 + 0.5 * U(rng), 0.1 + 0.05 * U(rng), 0.1 + 0.05 * U(rng), 0.25 + 0.1 * U(rng), 0.2 + 0.1 * U(rng), 9.81};
    double x[6] = {U(rng), 3.2 * U(rng), 3.2 * U(rng), 2 * U(rng), 6 * U(rng), 6 * U(rng)};
    const double u = 50 * U(rng);
    double M0[9], F0[3], dF0[18], dA0[9], dB0[9];
    ref::double_pendulum_terms(p, x, u, M0, F0, dF0, dA0, dB0);
    const DoublePendulumGenConsts<double> K = double_pendulum_gen_consts<double, double>(p);
    double M[9], F[3], dF[18], dA[9], dB[9];
    for (double& v : dF) v = 12345.0;  // masked entries must stay untouched
    for (double& v : dA) v = 12345.0;
    for (double& v : dB) v = 12345.0;
    double_pendulum_terms_sc<double>(K, std::sin(x[1]), std::cos(x[1]), std::sin(x[2]), std::cos(x[2]), x, u, M, F, dF, dA, dB);
    double sc = 0;
    for (double v : dF0) sc = std::fmax(sc, std::fabs(v));
    for (int k = 0; k < 9; ++k) expect(close(M[k], M0[k], std::fabs(M0[k])), "double M", i, M[k], M0[k]);
    for (int k = 0; k < 3; ++k) expect(close(F[k], F0[k], std::fabs(F0[k])), "double F", i, F[k], F0[k]);
    for (int k = 0; k < 18; ++k) {
      if (DoublePendulumGenSparsity::dFdx[k]) expect(close(dF[k], dF0[k], sc), "double dFdx", i, dF[k], dF0[k]);
      else expect(dF0[k] == 0.0 && dF[k] == 12345.0, "double dFdx masked entry", i, dF[k], dF0[k]);
    }
    for (int k = 0; k < 9; ++k) {
      if (DoublePendulumGenSparsity::dM1[k]) expect(close(dA[k], dA0[k], 1.0), "double dM1", i, dA[k], dA0[k]);
      else expect(dA0[k] == 0.0 && dA[k] == 12345.0, "double dM1 masked entry", i, dA[k], dA0[k]);
      if (DoublePendulumGenSparsity::dM2[k]) expect(close(dB[k], dB0[k], 1.0), "double dM2", i, dB[k], dB0[k]);
      else expect(dB0[k] == 0.0 && dB[k] == 12345.0, "double dM2 masked entry", i, dB[k], dB0[k]);
    }
    // a column of da/dx flagged as vanishing has dF/dx_c = 0 for every row and (for an angle) dM/dx_c = 0
    for (int c = 0; c < 6; ++c)
      if ((DoublePendulumGenSparsity::ja_zero_cols >> c) & 1u)
        for (int r = 0; r < 3; ++r) expect(dF0[r * 6 + c] == 0.0 && c != 1 && c != 2, "double ja_zero_cols", i, dF0[r * 6 + c], 0.0);
    ++n_double;
  }
  // ---- cart + single pole: accelerations and their partials through the M / F terms and a 2 x 2 LDL^T (as
  // SingleModelGenerated does) against the closed-form C code, with and without external forces, every branch
  int n_single = 0;
  const double sets[4][9] = {{1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0}, {1.0, 0.1, 0.25, 9.81, 0.03, 0.1, 0.13, 0.8, 100.0},
                             {1.3, 0.2, 0.4, 9.81, 0.0, 0.1, 0.0, 0.8, 10.0}, {1.0, 0.1, 0.25, 9.81, 0.2, 1e-9, 5.0, 0.75, 100.0}};
  for (int ps = 0; ps < (getenv("GEN_CHECK_SETS") ? atoi(getenv("GEN_CHECK_SETS")) : 4); ++ps) {
    const double* p = sets[ps];
    const ref::SinglePendulumGenConsts K0 = ref::single_pendulum_gen_consts(p);
    const SinglePendulumMFConsts<double> K = single_pendulum_mf_consts<double, double>(p);
    const double inv_m00 = 1.0 / (p[0] + p[1]);
    for (int i = 0; i < 1500; ++i) {
      double x[4] = {1.5 * U(rng), 4 * U(rng), 2 * U(rng), 6 * U(rng)};
      if (i % 10 == 0) x[2] = x[3] = 0.0;                           // at rest: the |v|^2 > 0 guard
      if (i % 10 == 1) x[0] = p[7] * ((i % 20 == 1) ? 1 : -1);      // exactly on a bumper edge: spring off (strict 0 < arg)
      if (i % 10 == 2) x[2] = 3e-6 * U(rng);
      const double u = 50 * U(rng);
      const bool ext = (i % 2) == 0;
      const double fb = ext ? 3 * U(rng) : 0.0, fmx = ext ? 3 * U(rng) : 0.0, fmy = ext ? 3 * U(rng) : 0.0;
      double a0[2], Ja0[2][4], Jua0[2];
      if (ext) ref::single_pendulum_gen_accel_ext(&K0, x[0], x[1], x[2], x[3], u, fb, fmx, fmy, a0, Ja0, Jua0, 1);
      else ref::single_pendulum_gen_accel_noext(&K0, x[0], x[1], x[2], x[3], u, 0, 0, 0, a0, Ja0, Jua0, 1);
      const double s = std::sin(x[1]), c = std::cos(x[1]);
      double tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy;
      single_pendulum_mf_helpers<double, true>(K, s, c, x[0], x[2], x[3], tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy);
      double M[4], F[2], dF[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dM[4] = {0, 0, 0, 0};
      if (ext) single_pendulum_mf_terms_ext<double, true>(K, s, c, tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy, x[2], x[3], u, fb, fmx, fmy, M, F, dF, dM);
      else single_pendulum_mf_terms_noext<double, true>(K, s, c, tv, sech2, n, inv_n, sr, sl, on_r, on_l, vx, vy, x[2], x[3], u, 0.0, 0.0, 0.0, M, F, dF, dM);
      const double L = M[2] * inv_m00, id1 = 1.0 / (M[3] - L * M[2]);
      auto solve = [&](double b0, double b1, double& y0, double& y1) {
        y1 = (b1 - L * b0) * id1;
        y0 = b0 * inv_m00 - L * y1;
      };
      double a[2], Ja[2][4], Jua[2];
      solve(F[0], F[1], a[0], a[1]);
      for (int cc = 0; cc < 4; ++cc) {
        double r0 = SinglePendulumMFSparsity::dFdx[cc] ? dF[cc] : 0.0, r1 = SinglePendulumMFSparsity::dFdx[4 + cc] ? dF[4 + cc] : 0.0;
        if (cc == 1) {
          r0 -= (SinglePendulumMFSparsity::dM1[0] ? dM[0] : 0.0) * a[0] + (SinglePendulumMFSparsity::dM1[1] ? dM[1] : 0.0) * a[1];
          r1 -= (SinglePendulumMFSparsity::dM1[2] ? dM[2] : 0.0) * a[0] + (SinglePendulumMFSparsity::dM1[3] ? dM[3] : 0.0) * a[1];
        }
        solve(r0, r1, Ja[0][cc], Ja[1][cc]);
      }
      solve(1.0, 0.0, Jua[0], Jua[1]);
      // scale of the partials: the largest entry, and the friction slope's scale mu (m_b + m_1) g / max(v_mu, 1e-6) / m_00 -- both
      // sides form 1 - tanh^2 on the host, whose rounding that slope multiplies (1e6 for the fourth parameter set)
      double sj = p[4] * (p[0] + p[1]) * p[3] / std::fmax(p[5], 1e-6) * inv_m00;
      for (int r = 0; r < 2; ++r)
        for (int cc = 0; cc < 4; ++cc) sj = std::fmax(sj, std::fabs(Ja0[r][cc]));
      for (int r = 0; r < 2; ++r) {
        expect(close(a[r], a0[r], std::fabs(a0[r])), "single a", i, a[r], a0[r]);
        expect(close(Jua[r], Jua0[r], 1.0), "single Jua", i, Jua[r], Jua0[r]);
        for (int cc = 0; cc < 4; ++cc) expect(close(Ja[r][cc], Ja0[r][cc], sj), "single Ja", i, Ja[r][cc], Ja0[r][cc]);
        expect((Ja[r][0] == 0.0) == (Ja0[r][0] == 0.0), "single spring column structure", i, Ja[r][0], Ja0[r][0]);
      }
      ++n_single;
    }
  }
  if (failures) {
    std::printf("%d failures\n", failures);
    return 1;
  }
  std::printf("OK generated terms: %d double-pendulum cases, %d single-pendulum cases\n", n_double, n_single);
  return 0;
}
