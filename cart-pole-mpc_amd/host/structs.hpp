// structs.hpp -- plain-data types of the cart-pole MPC API, with the reference's names and field
// order so that callers of pendulum::Optimization / pendulum::Simulator compile unchanged.
// Counterpart of optimization/structs.hpp:8-70 in the reference (Eigen-free: ToVector returns
// std::array, the only Eigen use in the reference's public structs).
#pragma once
#include <array>

namespace pendulum {

// Parameters of the single-pole system (structs.hpp:8-41).
struct SingleCartPoleParams {
  double m_b;     // mass of the base (kg)
  double m_1;     // mass on the pole (kg)
  double l_1;     // length of the pole (m)
  double g;       // gravity (m/s^2)
  double mu_b;    // friction at the base
  double v_mu_b;  // cutoff velocity of the smoothed Coulomb model (m/s)
  double c_d_1;   // drag coefficient on the pole mass
  double x_s;     // position of the bumper springs (m)
  double k_s;     // spring constant of the bumpers (N/m)

  SingleCartPoleParams() noexcept = default;
  constexpr SingleCartPoleParams(double m_b, double m_1, double l_1, double g, double mu_b, double v_mu_b,
                                 double c_d_1, double x_s, double k_s) noexcept
      : m_b(m_b), m_1(m_1), l_1(l_1), g(g), mu_b(mu_b), v_mu_b(v_mu_b), c_d_1(c_d_1), x_s(x_s), k_s(k_s) {}

  std::array<double, 9> ToArray() const noexcept { return {m_b, m_1, l_1, g, mu_b, v_mu_b, c_d_1, x_s, k_s}; }
};

// State of the single cart-pole system (structs.hpp:44-64).
struct SingleCartPoleState {
  double b_x;       // base position
  double th_1;      // pole angle, measured from +x: upright = +pi/2
  double b_x_dot;
  double th_1_dot;

  SingleCartPoleState() noexcept = default;
  constexpr SingleCartPoleState(double b_x, double th_1, double b_x_dot, double th_1_dot) noexcept
      : b_x(b_x), th_1(th_1), b_x_dot(b_x_dot), th_1_dot(th_1_dot) {}
  explicit SingleCartPoleState(const std::array<double, 4>& x) noexcept
      : SingleCartPoleState(x[0], x[1], x[2], x[3]) {}

  std::array<double, 4> ToVector() const noexcept { return {b_x, th_1, b_x_dot, th_1_dot}; }
};

// structs.hpp:67-70
struct Vector2 {
  double x;
  double y;
};

}  // namespace pendulum
