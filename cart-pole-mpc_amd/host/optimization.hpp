// optimization.hpp -- pendulum::Optimization with the reference's signatures
// (optimization/optimization.hpp:12-108), implemented over the C-ABI of libcpmpc.so: every Step is a
// batch-of-one (or batch-of-B, see StepBatch) solve on the MI355X.  No CPU solver lives here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "structs.hpp"

struct cpmpc_solver;

namespace pendulum {

// optimization/optimization.hpp:12-53, field for field.
struct OptimizationParams {
  double control_dt{0.01};
  std::size_t window_length{40};
  std::size_t state_spacing{10};
  std::size_t max_iterations{8};
  double relative_exit_tol{1.0e-5};
  double absolute_first_derivative_tol{1.0e-6};
  double equality_penalty_initial{1.0};
  double u_guess_sinusoid_amplitude{10.0};
  double u_cost_weight{0.1};
  double u_derivative_cost_weight{0.1};
  double b_x_final_cost_weight{150.0};
  double th_final_cost_weight{-1.0};
  double b_x_dot_final_cost_weight{-1.0};
  double th_dot_final_cost_weight{-1.0};

  constexpr std::size_t NumStates() const noexcept { return window_length / state_spacing + 1; }
};

// Stands where mini_opt::NLSTerminationState stands (used at optimization_test.cc:44-46).
enum class NLSTerminationState : std::int32_t {
  NONE = 0,
  MAX_ITERATIONS = 1,
  SATISFIED_ABSOLUTE_TOL = 2,
  SATISFIED_RELATIVE_TOL = 3,
  SATISFIED_FIRST_ORDER_TOL = 4,
  QP_INDEFINITE = 5,
  USER_CALLBACK = 6,
  MAX_LAMBDA = 7,
  NON_FINITE = 8,
};

// Stands where mini_opt::NLSSolverOutputs stands (optimization.hpp:63): termination_state + ToString().
struct NLSSolverOutputs {
  NLSTerminationState termination_state{NLSTerminationState::NONE};
  int iterations{0};
  double final_cost{0.0};         // 1/2 |r|^2
  double final_equality_l1{0.0};  // |c|_1
  // the handle's horizon is beyond cpmpc_max_parity_horizon() (include/cpmpc.h: cpmpc_horizon_beyond_parity): ToString says so
  bool horizon_beyond_parity{false};
  std::string ToString() const;
};

// optimization/optimization.hpp:55-70
struct OptimizationOutputs {
  SingleCartPoleState initial_state;
  std::vector<double> previous_solution;
  NLSSolverOutputs solver_outputs;
  std::vector<double> u;
  std::vector<SingleCartPoleState> predicted_states;
};

// Batched outputs of StepBatch, structure-of-arrays with the batch index fastest.
struct BatchOptimizationOutputs {
  std::size_t batch{0};
  std::vector<double> u;                 // [N][B]
  std::vector<double> predicted_states;  // [N][4][B]
  std::vector<std::int32_t> status;      // [B]
  std::vector<std::int32_t> iterations;  // [B]
  std::vector<double> final_cost;        // [B]
  std::vector<double> final_equality_l1; // [B]
};

// Per-problem inputs of a batched step; a null pointer means "the shared argument of the call applies to every problem".
// The reference has one parameter set, one set-point and one set of terminal weights per Optimization object
// (optimization.hpp:73-108); a batch of controllers may differ in all three (domain randomisation, the UI's
// per-controller cost / constraint toggles, viz/src/application.ts:279-342).
struct PerProblemInputs {
  const double* dynamics_params = nullptr;   // [9][B], SingleCartPoleParams field order (structs.hpp:8-41)
  const double* set_points = nullptr;        // [B]
  const double* terminal_weights = nullptr;  // [4][B] in state order; >= 0 a cost row, < 0 an equality row
};

// Hybrid multiple-shooting MPC (optimization.hpp:73-108), solved on the GPU.
class Optimization {
 public:
  // Throws std::invalid_argument on the reference constructor's precondition failures
  // (optimization.cc:13-22) and std::runtime_error if no gfx950 device / library is usable.
  // Every horizon the reference's constructor accepts is accepted (optimization.cc:13-22).  strict_horizon: throw
  // std::runtime_error for window_length * control_dt beyond 1.0 s (cpmpc_max_parity_horizon), where eliminating the states
  // through the unstable plant no longer reproduces a full-space solve to 1e-5 on every cold start (include/cpmpc.h,
  // CPMPC_CREATE_STRICT_HORIZON); without it the library says so once per process on stderr and solves.
  explicit Optimization(const OptimizationParams& params, std::size_t max_batch = 1, int device = 0,
                        bool strict_horizon = false);
  ~Optimization();
  Optimization(const Optimization&) = delete;
  Optimization& operator=(const Optimization&) = delete;

  // Run an iteration of optimization and compute control outputs (optimization.cc:39-97).
  [[nodiscard]] OptimizationOutputs Step(const SingleCartPoleState& current_state,
                                         const SingleCartPoleParams& dynamics_params, double b_x_set_point);

  // B controllers in lock-step: states_soa is [4][B]; shares dynamics parameters and set-point.
  [[nodiscard]] BatchOptimizationOutputs StepBatch(const std::vector<double>& states_soa,
                                                   const SingleCartPoleParams& dynamics_params,
                                                   double b_x_set_point);

  // The same writing straight into caller-owned arrays ([N][B], [N][4][B], [B] ...; any output pointer may be null):
  // what the numpy entry point of pypendulum uses, no per-element conversion on either side.
  void StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                     double b_x_set_point, double* u, double* predicted_states, std::int32_t* status,
                     std::int32_t* iterations, double* final_cost, double* final_equality_l1);

  // The general batched step: per-problem inputs (each optional) and, when `solution` is not null, the solution vectors
  // z [dim][B] (MapKey order) as a further output.  Large batches run as a pipeline of chunks (include/cpmpc.h,
  // cpmpc_set_host_chunk); output arrays pinned with cpmpc_host_register are written by DMA.
  void StepBatchInto(const double* states_soa, std::size_t B, const SingleCartPoleParams& dynamics_params,
                     double b_x_set_point, const PerProblemInputs& per_problem, double* u, double* predicted_states,
                     std::int32_t* status, std::int32_t* iterations, double* final_cost, double* final_equality_l1,
                     double* solution);

  // Discard previous initial guess, which will reset the problem (optimization.hpp:83).
  void Reset();

  // Set the previous solution, used as guess on the next iteration (optimization.hpp:86-89).
  void SetPreviousSolution(const std::vector<double>& guess);
  // The same for B controllers: z_soa is [dim][B]; and its reverse (the warm start the next step will shift).
  void SetPreviousSolutionBatch(const std::vector<double>& z_soa, std::size_t B);
  [[nodiscard]] std::vector<double> GetSolutionBatch(std::size_t B);
  // problems per chunk of a pipelined host-pointer step (0 = never split)
  void SetHostChunk(std::size_t problems);
  std::size_t Dim() const;
  // window_length * control_dt exceeds cpmpc_max_parity_horizon(): solved as asked, a few cold starts in 10^4 may differ from a
  // full-space solve by more than 1e-5 (include/cpmpc.h: cpmpc_horizon_beyond_parity); also in every Step's solver_outputs
  [[nodiscard]] bool HorizonBeyondParity() const noexcept;

  const OptimizationParams& params() const noexcept { return params_; }

 private:
  OptimizationParams params_;
  std::size_t max_batch_;
  cpmpc_solver* solver_{nullptr};
  std::vector<double> previous_solution_;  // host mirror of the B = 1 warm start (for the outputs)
};

}  // namespace pendulum
