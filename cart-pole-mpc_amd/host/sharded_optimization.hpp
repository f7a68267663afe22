// sharded_optimization.hpp -- pendulum::ShardedOptimization: B controllers in lock-step over SEVERAL GPUs from one
// C++ process, behind the same StepBatch call as pendulum::Optimization (optimization.hpp).
//
// The reference has no parallelism of any kind (its Optimization is one controller on one CPU thread,
// optimization/optimization.hpp:73-108); independent controllers shard embarrassingly, so this class is the multi-GPU
// form of "one Optimization per controller": one solver handle and one stream per shard (include/cpmpc.h,
// cpmpc_sharded_*), the batch split contiguously, all shards running concurrently, the outputs assembled in global
// problem order.  The only traffic between devices is the scatter of the measured states and the gather of the
// results.
#pragma once
#include <cstddef>
#include <vector>

#include "optimization.hpp"

struct cpmpc_sharded;

namespace pendulum {

class ShardedOptimization {
 public:
  // devices: HIP device of each shard (a device may appear more than once); empty = every visible gfx950 device.
  // max_batch is the TOTAL number of controllers.  Throws like Optimization's constructor.
  explicit ShardedOptimization(const OptimizationParams& params, std::size_t max_batch,
                               const std::vector<int>& devices = {}, bool strict_horizon = false);
  ~ShardedOptimization();
  ShardedOptimization(const ShardedOptimization&) = delete;
  ShardedOptimization& operator=(const ShardedOptimization&) = delete;

  // Optimization::StepBatch over all shards: states_soa is [4][B]; outputs are [N][B] etc. in the caller's order.
  [[nodiscard]] BatchOptimizationOutputs StepBatch(const std::vector<double>& states_soa,
                                                   const SingleCartPoleParams& dynamics_params, double b_x_set_point);
  // The same writing into caller-owned arrays (any of the output pointers may be null).
  void StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                     double b_x_set_point, double* u, double* predicted_states, std::int32_t* status,
                     std::int32_t* iterations, double* final_cost, double* final_equality_l1);
  // The general step: per-problem parameters / set-points / terminal rows (each optional) and the solution vectors
  // z [dim][B] as a further output (optimization.hpp: PerProblemInputs).
  void StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                     double b_x_set_point, const PerProblemInputs& per_problem, double* u, double* predicted_states,
                     std::int32_t* status, std::int32_t* iterations, double* final_cost, double* final_equality_l1,
                     double* solution);
  // Optimization::Reset on every shard.
  void Reset();
  // Optimization::SetPreviousSolution for B controllers (optimization.hpp:86-89): z_soa is [dim][B]; replaces every
  // shard's warm start.  GetSolution(B) is its reverse for the first B controllers that hold one.
  void SetPreviousSolution(const std::vector<double>& z_soa, std::size_t B);
  [[nodiscard]] std::vector<double> GetSolution(std::size_t B);
  // controllers [0, n) hold a previous solution.  A step with another batch size than the last one hands the warm start
  // over to the new split first (include/cpmpc.h: "Warm starts and the batch size"); nothing is ever misaligned.
  std::size_t PreviousSolutionBatch() const noexcept;
  // problems per chunk of every shard's host-pointer pipeline (Optimization::SetHostChunk)
  void SetHostChunk(std::size_t problems);
  std::size_t Dim() const noexcept;

  // the horizon exceeds cpmpc_max_parity_horizon() (include/cpmpc.h: cpmpc_horizon_beyond_parity), as Optimization::HorizonBeyondParity
  [[nodiscard]] bool HorizonBeyondParity() const noexcept;
  std::size_t NumShards() const noexcept;
  int DeviceOfShard(std::size_t shard) const noexcept;
  // columns [first, second) of a B-controller batch that `shard` solves
  std::pair<std::size_t, std::size_t> ShardRange(std::size_t shard, std::size_t B) const;
  const OptimizationParams& params() const noexcept { return params_; }

 private:
  OptimizationParams params_;
  std::size_t max_batch_;
  cpmpc_sharded* sharded_{nullptr};
};

}  // namespace pendulum
