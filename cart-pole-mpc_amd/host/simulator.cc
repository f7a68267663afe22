// simulator.cc -- pendulum::Simulator over the C-ABI (Simulator::Step, optimization/simulator.cc:11-36).
#include "simulator.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

#include "../../include/cpmpc.h"

namespace pendulum {

void Simulator::Step(const SingleCartPoleParams& params, double dt, const double u, const Vector2& f_base,
                     const Vector2& f_mass) {
  if (!(dt >= 0.0)) throw std::invalid_argument("Simulator::Step: dt must be >= 0 (simulator.cc:13)");
  if (!std::isfinite(u)) throw std::invalid_argument("Simulator::Step: u is not finite (simulator.cc:14)");
  const auto dyn = params.ToArray();
  const double fext[4] = {f_base.x, f_base.y, f_mass.x, f_mass.y};
  const int rc = cpmpc_sim_step_batch_host(1, dyn.data(), dt, &u, fext, state_.data());
  if (rc == CPMPC_ERR_INVALID_ARG) throw std::invalid_argument(std::string("cpmpc: ") + cpmpc_last_error());
  if (rc != CPMPC_OK) throw std::runtime_error(std::string("cpmpc: ") + cpmpc_last_error());
}

}  // namespace pendulum
