// optimization.cc -- pendulum::Optimization over the C-ABI (include/cpmpc.h).
// Mirrors the call sequence of the reference's Optimization::Step (optimization/optimization.cc:39-97);
// the numerical work happens in libcpmpc.so's HIP kernels.
#include "optimization.hpp"

#include <sstream>
#include <stdexcept>

#include "../../include/cpmpc.h"

namespace pendulum {

static const char* TerminationName(NLSTerminationState s) {
  switch (s) {
    case NLSTerminationState::NONE: return "NONE";
    case NLSTerminationState::MAX_ITERATIONS: return "MAX_ITERATIONS";
    case NLSTerminationState::SATISFIED_ABSOLUTE_TOL: return "SATISFIED_ABSOLUTE_TOL";
    case NLSTerminationState::SATISFIED_RELATIVE_TOL: return "SATISFIED_RELATIVE_TOL";
    case NLSTerminationState::SATISFIED_FIRST_ORDER_TOL: return "SATISFIED_FIRST_ORDER_TOL";
    case NLSTerminationState::QP_INDEFINITE: return "QP_INDEFINITE";
    case NLSTerminationState::USER_CALLBACK: return "USER_CALLBACK";
    case NLSTerminationState::MAX_LAMBDA: return "MAX_LAMBDA";
    case NLSTerminationState::NON_FINITE: return "NON_FINITE";
  }
  return "?";
}

std::string NLSSolverOutputs::ToString() const {
  std::ostringstream os;
  os << "Iterations: " << iterations << ", termination = " << TerminationName(termination_state)
     << "\n  final cost (1/2 |r|^2) = " << final_cost << ", final |c|_1 = " << final_equality_l1 << "\n";
  if (horizon_beyond_parity)
    os << "  horizon beyond " << cpmpc_max_parity_horizon()
       << " s (cpmpc_max_parity_horizon): far from the optimum a cold start may differ from a full-space solve by more than 1e-5\n";
  return os.str();
}

cpmpc_params ToCParams(const OptimizationParams& p);  // also used by sharded_optimization.cc
static cpmpc_params ToC(const OptimizationParams& p) { return ToCParams(p); }
cpmpc_params ToCParams(const OptimizationParams& p) {
  cpmpc_params c;
  c.control_dt = p.control_dt;
  c.window_length = p.window_length;
  c.state_spacing = p.state_spacing;
  c.max_iterations = p.max_iterations;
  c.relative_exit_tol = p.relative_exit_tol;
  c.absolute_first_derivative_tol = p.absolute_first_derivative_tol;
  c.equality_penalty_initial = p.equality_penalty_initial;
  c.u_guess_sinusoid_amplitude = p.u_guess_sinusoid_amplitude;
  c.u_cost_weight = p.u_cost_weight;
  c.u_derivative_cost_weight = p.u_derivative_cost_weight;
  c.b_x_final_cost_weight = p.b_x_final_cost_weight;
  c.th_final_cost_weight = p.th_final_cost_weight;
  c.b_x_dot_final_cost_weight = p.b_x_dot_final_cost_weight;
  c.th_dot_final_cost_weight = p.th_dot_final_cost_weight;
  return c;
}

[[noreturn]] static void Throw(int rc) {
  const std::string text = std::string("cpmpc: ") + cpmpc_last_error();
  if (rc == CPMPC_ERR_INVALID_ARG) throw std::invalid_argument(text);
  throw std::runtime_error(text);
}

Optimization::Optimization(const OptimizationParams& params, std::size_t max_batch, int device, bool strict_horizon)
    : params_(params), max_batch_(max_batch) {
  const cpmpc_params c = ToC(params);
  cpmpc_create_info info{};
  info.struct_size = sizeof info;
  info.flags = strict_horizon ? CPMPC_CREATE_STRICT_HORIZON : 0u;
  info.dtype = CPMPC_F64;  // fp64, like the reference.
  info.model = CPMPC_MODEL_SINGLE;
  info.device = device;
  info.max_batch = static_cast<std::int64_t>(max_batch);
  info.params = &c;
  const int rc = cpmpc_create_ex(&info, &solver_);
  if (rc != CPMPC_OK) Throw(rc);
}

Optimization::~Optimization() { cpmpc_destroy(solver_); }

bool Optimization::HorizonBeyondParity() const noexcept { return cpmpc_horizon_beyond_parity(solver_) == 1; }

void Optimization::Reset() {
  previous_solution_.resize(0);
  cpmpc_reset(solver_);
}

void Optimization::SetPreviousSolution(const std::vector<double>& guess) {
  if (static_cast<int>(guess.size()) != cpmpc_dim(solver_)) {
    // the reference would fail later inside the solver on a wrongly sized guess; fail here instead
    throw std::invalid_argument("SetPreviousSolution: guess has the wrong dimension");
  }
  previous_solution_ = guess;
  const int rc = cpmpc_set_previous_solution_host(solver_, 1, guess.data());
  if (rc != CPMPC_OK) Throw(rc);
}

OptimizationOutputs Optimization::Step(const SingleCartPoleState& current_state,
                                       const SingleCartPoleParams& dynamics_params,
                                       const double b_x_set_point) {
  const std::size_t N = params_.window_length;
  const auto x0 = current_state.ToVector();
  const auto dyn = dynamics_params.ToArray();
  std::vector<double> u(N), pred(4 * N);
  std::int32_t status = 0, iterations = 0;
  double cost = 0.0, eq = 0.0;
  // one round trip: the solution z (previous_solution_ of the next Step, optimization.cc:84-85) comes back with u
  std::vector<double> z(static_cast<std::size_t>(cpmpc_dim(solver_)));
  const cpmpc_step_host_outputs ho = {u.data(), pred.data(), &status, &iterations, &cost, &eq, z.data()};
  const int rc = cpmpc_step_batch_host_ex(solver_, 1, x0.data(), dyn.data(), b_x_set_point, &ho);
  if (rc != CPMPC_OK) Throw(rc);

  OptimizationOutputs out;
  out.initial_state = current_state;
  out.previous_solution = previous_solution_;  // the solution the guess was shifted from (optimization.cc:84)
  out.solver_outputs.termination_state = static_cast<NLSTerminationState>(status);
  out.solver_outputs.iterations = iterations;
  out.solver_outputs.final_cost = cost;
  out.solver_outputs.final_equality_l1 = eq;
  out.solver_outputs.horizon_beyond_parity = HorizonBeyondParity();
  out.u = std::move(u);
  out.predicted_states.reserve(N);
  for (std::size_t k = 0; k < N; ++k)  // [N][4][1]
    out.predicted_states.emplace_back(pred[4 * k + 0], pred[4 * k + 1], pred[4 * k + 2], pred[4 * k + 3]);

  previous_solution_ = std::move(z);  // optimization.cc:85
  return out;
}

void Optimization::StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                                 double b_x_set_point, double* u, double* predicted_states, std::int32_t* status,
                                 std::int32_t* iterations, double* final_cost, double* final_equality_l1) {
  if (states_soa == nullptr || B == 0) throw std::invalid_argument("StepBatch: states_soa must be [4][B], B >= 1");
  if (B > max_batch_) throw std::invalid_argument("StepBatch: batch exceeds the capacity given at construction");
  const auto dyn = dynamics_params.ToArray();
  const int rc = cpmpc_step_batch_host(solver_, static_cast<std::int64_t>(B), states_soa, dyn.data(), b_x_set_point, u,
                                       predicted_states, status, iterations, final_cost, final_equality_l1);
  if (rc != CPMPC_OK) Throw(rc);
}

void Optimization::StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                                 double b_x_set_point, const PerProblemInputs& pp, double* u, double* predicted_states,
                                 std::int32_t* status, std::int32_t* iterations, double* final_cost,
                                 double* final_equality_l1, double* solution) {
  if (states_soa == nullptr || B == 0) throw std::invalid_argument("StepBatch: states_soa must be [4][B], B >= 1");
  if (B > max_batch_) throw std::invalid_argument("StepBatch: batch exceeds the capacity given at construction");
  const auto dyn = dynamics_params.ToArray();
  const cpmpc_step_host_inputs in = {states_soa, pp.dynamics_params ? nullptr : dyn.data(), pp.dynamics_params,
                                     b_x_set_point, pp.set_points, pp.terminal_weights};
  const cpmpc_step_host_outputs ho = {u, predicted_states, status, iterations, final_cost, final_equality_l1, solution};
  const int rc = cpmpc_step_batch_host_in(solver_, static_cast<std::int64_t>(B), &in, &ho);
  if (rc != CPMPC_OK) Throw(rc);
}

void Optimization::SetPreviousSolutionBatch(const std::vector<double>& z_soa, std::size_t B) {
  if (B == 0 || z_soa.size() != Dim() * B) throw std::invalid_argument("SetPreviousSolutionBatch: z_soa must be [dim][B]");
  const int rc = cpmpc_set_previous_solution_host(solver_, static_cast<std::int64_t>(B), z_soa.data());
  if (rc != CPMPC_OK) Throw(rc);
}

std::vector<double> Optimization::GetSolutionBatch(std::size_t B) {
  std::vector<double> z(Dim() * B);
  const int rc = cpmpc_get_solution_host(solver_, static_cast<std::int64_t>(B), z.data());
  if (rc != CPMPC_OK) Throw(rc);
  return z;
}

void Optimization::SetHostChunk(std::size_t problems) {
  const int rc = cpmpc_set_host_chunk(solver_, static_cast<std::int64_t>(problems));
  if (rc != CPMPC_OK) Throw(rc);
}

std::size_t Optimization::Dim() const { return static_cast<std::size_t>(cpmpc_dim(solver_)); }

BatchOptimizationOutputs Optimization::StepBatch(const std::vector<double>& states_soa,
                                                 const SingleCartPoleParams& dynamics_params,
                                                 const double b_x_set_point) {
  if (states_soa.empty() || states_soa.size() % 4 != 0)
    throw std::invalid_argument("StepBatch: states_soa must be [4][B]");
  const std::size_t B = states_soa.size() / 4;
  if (B > max_batch_) throw std::invalid_argument("StepBatch: batch exceeds the capacity given at construction");
  const std::size_t N = params_.window_length;
  const auto dyn = dynamics_params.ToArray();
  BatchOptimizationOutputs out;
  out.batch = B;
  out.u.resize(N * B);
  out.predicted_states.resize(4 * N * B);
  out.status.resize(B);
  out.iterations.resize(B);
  out.final_cost.resize(B);
  out.final_equality_l1.resize(B);
  const int rc = cpmpc_step_batch_host(solver_, static_cast<std::int64_t>(B), states_soa.data(), dyn.data(),
                                       b_x_set_point, out.u.data(), out.predicted_states.data(),
                                       out.status.data(), out.iterations.data(), out.final_cost.data(),
                                       out.final_equality_l1.data());
  if (rc != CPMPC_OK) Throw(rc);
  return out;
}

}  // namespace pendulum
