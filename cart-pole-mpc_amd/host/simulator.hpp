// simulator.hpp -- pendulum::Simulator with the reference's signatures
// (optimization/simulator.hpp:10-29); the integration runs in libcpmpc.so's sim kernel (fp64).
#pragma once
#include <array>

#include "structs.hpp"

namespace pendulum {

// Encapsulates the system state and integrates it forward in time.
class Simulator {
 public:
  // Step the simulator forward by `dt` with control input `u` (simulator.cc:11-23).
  // Throws std::invalid_argument for dt < 0 or non-finite u (simulator.cc:13-14).
  void Step(const SingleCartPoleParams& params, double dt, double u, const Vector2& f_base,
            const Vector2& f_mass);

  SingleCartPoleState GetState() const noexcept {
    return SingleCartPoleState{state_[0], state_[1], state_[2], state_[3]};
  }

  void SetState(const SingleCartPoleState& state) noexcept { state_ = state.ToVector(); }

 private:
  std::array<double, 4> state_{0.0, -3.14159265358979323846 / 2, 0.0, 0.0};  // simulator.hpp:28
};

}  // namespace pendulum
