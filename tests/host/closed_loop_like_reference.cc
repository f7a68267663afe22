// closed_loop_like_reference.cc -- TEST INFRASTRUCTURE: the "a caller of the reference compiles unchanged" proof.
//
// The body of main() below IS the reference's own closed-loop test, optimization/optimization_test.cc:12-66
// (gareth-cross/cart-pole-mpc, MIT licence, (c) Gareth Cross): same variable names, same order of statements, same
// tolerances, with the gtest macros replaced by CHECK_NEAR and the two #include lines pointing at this repo's
// drop-in headers (cart-pole-mpc_amd/host/optimization.hpp, simulator.hpp).  It is kept that close on purpose: it
// shows that code written against pendulum::Optimization / pendulum::Simulator builds and passes against this
// repo's classes, which run on the GPU through libcpmpc.so.  Nothing in the product uses this file.
// Built by cart-pole-mpc_amd/build.py into lib/host_smoke; run by tests/test_pypendulum.py.  Exit code 0 = all hold.
#include <cmath>
#include <cstdio>
#include <vector>

#include "optimization.hpp"
#include "simulator.hpp"

using namespace pendulum;

#define CHECK_NEAR(a, b, tol)                                                                  \
  do {                                                                                         \
    if (!(std::fabs((a) - (b)) <= (tol))) {                                                    \
      std::printf("FAIL %s:%d: |%s - %s| = %g > %g (t=%zu)\n", __FILE__, __LINE__, #a, #b,     \
                  std::fabs((a) - (b)), (double)(tol), t);                                     \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

int main(int argc, char** argv) {
  constexpr std::size_t num_steps = 200;
  OptimizationParams optimization_params{};
  optimization_params.control_dt = 0.01;
  optimization_params.window_length = 40;
  optimization_params.state_spacing = 5;
  optimization_params.max_iterations = 10;
  constexpr SingleCartPoleParams dynamics_params{1.0, 0.1, 0.25, 9.81, 0.03, 0.1, 0.13, 0.8, 100.0};
  constexpr SingleCartPoleState x0{0.0, -M_PI / 2, 0.0, 0.0};

  std::vector<SingleCartPoleState> states{};
  states.push_back(x0);
  Simulator sim{};
  sim.SetState(x0);
  Optimization optimization{optimization_params};
  std::size_t t = 0;
  for (; t < num_steps; ++t) {
    const OptimizationOutputs outputs = optimization.Step(sim.GetState(), dynamics_params, 0.0);
    if (outputs.solver_outputs.termination_state == NLSTerminationState::QP_INDEFINITE ||
        outputs.solver_outputs.termination_state == NLSTerminationState::MAX_LAMBDA) {
      std::printf("FAIL: termination %d at t=%zu\n", (int)outputs.solver_outputs.termination_state, t);
      return 1;
    }
    const SingleCartPoleState& terminal_state = outputs.predicted_states.back();
    if (t > 20) {
      CHECK_NEAR(0.0, terminal_state.b_x_dot, 1.0e-4);
      CHECK_NEAR(0.0, terminal_state.th_1_dot, 1.0e-4);
      CHECK_NEAR(M_PI / 2, terminal_state.th_1, 1.0e-4);
    }
    states.push_back(sim.GetState());
    sim.Step(dynamics_params, optimization_params.control_dt, outputs.u.front(), {0, 0}, {0, 0});
    if (argc > 1 && t % 20 == 0) std::printf("t=%zu u0=%.6f %s", t, outputs.u.front(), outputs.solver_outputs.ToString().c_str());
  }
  const SingleCartPoleState& terminal_state = states.back();
  CHECK_NEAR(0.0, terminal_state.b_x_dot, 1.0e-4);
  CHECK_NEAR(0.0, terminal_state.th_1_dot, 1.0e-3);
  CHECK_NEAR(M_PI / 2, terminal_state.th_1, 1.0e-4);
  std::printf("OK closed loop: final state b_x=%.3e th_1-pi/2=%.3e b_x_dot=%.3e th_1_dot=%.3e\n", terminal_state.b_x,
              terminal_state.th_1 - M_PI / 2, terminal_state.b_x_dot, terminal_state.th_1_dot);
  return 0;
}
