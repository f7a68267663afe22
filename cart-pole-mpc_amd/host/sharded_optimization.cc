// sharded_optimization.cc -- pendulum::ShardedOptimization over cpmpc_sharded_* (include/cpmpc.h).
#include "sharded_optimization.hpp"

#include <stdexcept>
#include <string>

#include "../../include/cpmpc.h"

namespace pendulum {

cpmpc_params ToCParams(const OptimizationParams& p);  // optimization.cc

[[noreturn]] static void ThrowSharded(int rc) {
  const std::string text = std::string("cpmpc: ") + cpmpc_last_error();
  if (rc == CPMPC_ERR_INVALID_ARG) throw std::invalid_argument(text);
  throw std::runtime_error(text);
}

ShardedOptimization::ShardedOptimization(const OptimizationParams& params, std::size_t max_batch,
                                         const std::vector<int>& devices, bool strict_horizon)
    : params_(params), max_batch_(max_batch) {
  const cpmpc_params c = ToCParams(params);
  cpmpc_create_info info{};
  info.struct_size = sizeof info;
  info.flags = strict_horizon ? CPMPC_CREATE_STRICT_HORIZON : 0u;
  info.dtype = CPMPC_F64;
  info.model = CPMPC_MODEL_SINGLE;
  info.max_batch = static_cast<std::int64_t>(max_batch);
  info.params = &c;
  const int rc = cpmpc_sharded_create_ex(&info, devices.empty() ? nullptr : devices.data(),
                                         static_cast<int>(devices.size()), &sharded_);
  if (rc != CPMPC_OK) ThrowSharded(rc);
}

void ShardedOptimization::StepBatchInto(const double* states_soa, std::size_t B,
                                        const SingleCartPoleParams& dynamics_params, double b_x_set_point,
                                        const PerProblemInputs& pp, double* u, double* predicted_states,
                                        std::int32_t* status, std::int32_t* iterations, double* final_cost,
                                        double* final_equality_l1, double* solution) {
  if (states_soa == nullptr || B == 0) throw std::invalid_argument("StepBatch: states_soa must be [4][B], B >= 1");
  if (B > max_batch_) throw std::invalid_argument("StepBatch: batch exceeds the capacity given at construction");
  const auto dyn = dynamics_params.ToArray();
  const cpmpc_step_host_inputs in = {states_soa, pp.dynamics_params ? nullptr : dyn.data(), pp.dynamics_params,
                                     b_x_set_point, pp.set_points, pp.terminal_weights};
  const cpmpc_step_host_outputs ho = {u, predicted_states, status, iterations, final_cost, final_equality_l1, solution};
  const int rc = cpmpc_sharded_step_batch_host_in(sharded_, static_cast<std::int64_t>(B), &in, &ho);
  if (rc != CPMPC_OK) ThrowSharded(rc);
}

void ShardedOptimization::SetPreviousSolution(const std::vector<double>& z_soa, std::size_t B) {
  if (B == 0 || z_soa.size() != Dim() * B) throw std::invalid_argument("SetPreviousSolution: z_soa must be [dim][B]");
  const int rc = cpmpc_sharded_set_previous_solution_host(sharded_, static_cast<std::int64_t>(B), z_soa.data());
  if (rc != CPMPC_OK) ThrowSharded(rc);
}

std::vector<double> ShardedOptimization::GetSolution(std::size_t B) {
  std::vector<double> z(Dim() * B);
  const int rc = cpmpc_sharded_get_solution_host(sharded_, static_cast<std::int64_t>(B), z.data());
  if (rc != CPMPC_OK) ThrowSharded(rc);
  return z;
}

std::size_t ShardedOptimization::PreviousSolutionBatch() const noexcept {
  return static_cast<std::size_t>(cpmpc_sharded_previous_solution_batch(sharded_));
}
void ShardedOptimization::SetHostChunk(std::size_t problems) {
  for (int i = 0; i < cpmpc_sharded_num_shards(sharded_); ++i) {
    const int rc = cpmpc_set_host_chunk(cpmpc_sharded_handle(sharded_, i), static_cast<std::int64_t>(problems));
    if (rc != CPMPC_OK) ThrowSharded(rc);
  }
}
std::size_t ShardedOptimization::Dim() const noexcept {
  return static_cast<std::size_t>(cpmpc_dim(cpmpc_sharded_handle(sharded_, 0)));
}

ShardedOptimization::~ShardedOptimization() { cpmpc_sharded_destroy(sharded_); }

void ShardedOptimization::Reset() { cpmpc_sharded_reset(sharded_); }

bool ShardedOptimization::HorizonBeyondParity() const noexcept { return cpmpc_sharded_horizon_beyond_parity(sharded_) == 1; }
std::size_t ShardedOptimization::NumShards() const noexcept {
  return static_cast<std::size_t>(cpmpc_sharded_num_shards(sharded_));
}
int ShardedOptimization::DeviceOfShard(std::size_t shard) const noexcept {
  return cpmpc_sharded_device(sharded_, static_cast<int>(shard));
}
std::pair<std::size_t, std::size_t> ShardedOptimization::ShardRange(std::size_t shard, std::size_t B) const {
  std::int64_t lo = 0, hi = 0;
  const int rc = cpmpc_sharded_range(sharded_, static_cast<int>(shard), static_cast<std::int64_t>(B), &lo, &hi);
  if (rc != CPMPC_OK) ThrowSharded(rc);
  return {static_cast<std::size_t>(lo), static_cast<std::size_t>(hi)};
}

void ShardedOptimization::StepBatchInto(const double* states_soa, std::size_t B,
                                        const SingleCartPoleParams& dynamics_params, double b_x_set_point, double* u,
                                        double* predicted_states, std::int32_t* status, std::int32_t* iterations,
                                        double* final_cost, double* final_equality_l1) {
  if (states_soa == nullptr || B == 0) throw std::invalid_argument("StepBatch: states_soa must be [4][B], B >= 1");
  if (B > max_batch_) throw std::invalid_argument("StepBatch: batch exceeds the capacity given at construction");
  const auto dyn = dynamics_params.ToArray();
  const cpmpc_step_host_outputs ho = {u, predicted_states, status, iterations, final_cost, final_equality_l1, nullptr};
  const int rc = cpmpc_sharded_step_batch_host(sharded_, static_cast<std::int64_t>(B), states_soa, dyn.data(),
                                               b_x_set_point, &ho);
  if (rc != CPMPC_OK) ThrowSharded(rc);
}

BatchOptimizationOutputs ShardedOptimization::StepBatch(const std::vector<double>& states_soa,
                                                        const SingleCartPoleParams& dynamics_params,
                                                        const double b_x_set_point) {
  if (states_soa.empty() || states_soa.size() % 4 != 0)
    throw std::invalid_argument("StepBatch: states_soa must be [4][B]");
  const std::size_t B = states_soa.size() / 4;
  const std::size_t N = params_.window_length;
  BatchOptimizationOutputs out;
  out.batch = B;
  out.u.resize(N * B);
  out.predicted_states.resize(4 * N * B);
  out.status.resize(B);
  out.iterations.resize(B);
  out.final_cost.resize(B);
  out.final_equality_l1.resize(B);
  StepBatchInto(states_soa.data(), B, dynamics_params, b_x_set_point, out.u.data(), out.predicted_states.data(),
                out.status.data(), out.iterations.data(), out.final_cost.data(), out.final_equality_l1.data());
  return out;
}

}  // namespace pendulum
