// pypendulum.cc -- Python module `pypendulum` with the names of the reference's nanobind wrapper
// (wrapper/wrapper.cc:40-98), bound with pybind11 over the C++ facade (nanobind is not available
// in this image).  Additions, all batched: Optimization(params, max_batch), .step_batch() (numpy arrays in and out,
// written by the C-ABI without per-element conversion), ShardedOptimization (several GPUs from one process),
// .reset(); Simulator.set_state(); and the JSON wire format of the reference's browser build (wasm.cc:19-65):
// X.to_json() / X.from_json(text) on every struct, OptimizationOutputs.{get_log, window_length, get_control,
// get_predicted_state} with the names of wasm.cc:86-105 in snake_case.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "json.hpp"
#include "optimization.hpp"
#include "sharded_optimization.hpp"
#include "simulator.hpp"

namespace py = pybind11;
using namespace pendulum;

// Batched outputs as numpy arrays (what step_batch returns): u [N, B], predicted_states [N, 4, B], the rest [B].
struct BatchArrays {
  std::size_t batch{0};
  py::array_t<double> u, predicted_states, final_cost, final_equality_l1;
  py::array_t<std::int32_t> status, iterations;
};

// states: a C-contiguous float64 array [4, B] (anything else is converted once by numpy).  The output arrays are
// allocated here and the C-ABI writes straight into them: no Python objects per element on either side (at
// B = 262 144 the list-based binding of round 2 boxed about 10^7 doubles per call).
template <typename Opt>
static BatchArrays StepBatchArrays(Opt& self, py::array_t<double, py::array::c_style | py::array::forcecast> states,
                                   const SingleCartPoleParams& dyn, double set_point, bool want_predicted) {
  if (states.ndim() != 2 || states.shape(0) != 4 || states.shape(1) < 1)
    throw std::invalid_argument("step_batch: states must be a [4, B] array");
  const std::size_t B = static_cast<std::size_t>(states.shape(1));
  const std::size_t N = self.params().window_length;
  BatchArrays out;
  out.batch = B;
  out.u = py::array_t<double>({N, B});
  if (want_predicted) out.predicted_states = py::array_t<double>({N, std::size_t{4}, B});
  out.status = py::array_t<std::int32_t>(B);
  out.iterations = py::array_t<std::int32_t>(B);
  out.final_cost = py::array_t<double>(B);
  out.final_equality_l1 = py::array_t<double>(B);
  const double* x = states.data();
  double* u = out.u.mutable_data();
  double* pred = want_predicted ? out.predicted_states.mutable_data() : nullptr;
  std::int32_t *st = out.status.mutable_data(), *it = out.iterations.mutable_data();
  double *fc = out.final_cost.mutable_data(), *fe = out.final_equality_l1.mutable_data();
  {
    py::gil_scoped_release release;  // the GPU round trip does not need the interpreter
    self.StepBatchInto(x, B, dyn, set_point, u, pred, st, it, fc, fe);
  }
  return out;
}

PYBIND11_MODULE(pypendulum, m) {
  m.doc() = "cart-pole MPC (MI355X-native hot path); API of gareth-cross/cart-pole-mpc's pypendulum";

  py::class_<SingleCartPoleParams>(m, "SingleCartPoleParams")
      .def(py::init<>())
      .def(py::init<double, double, double, double, double, double, double, double, double>(), py::arg("m_b"),
           py::arg("m_1"), py::arg("l_1"), py::arg("g"), py::arg("mu_b"), py::arg("v_mu_b"), py::arg("c_d_1"),
           py::arg("x_s"), py::arg("k_s"))
      .def_readwrite("m_b", &SingleCartPoleParams::m_b)
      .def_readwrite("m_1", &SingleCartPoleParams::m_1)
      .def_readwrite("l_1", &SingleCartPoleParams::l_1)
      .def_readwrite("g", &SingleCartPoleParams::g)
      .def_readwrite("mu_b", &SingleCartPoleParams::mu_b)
      .def_readwrite("v_mu_b", &SingleCartPoleParams::v_mu_b)
      .def_readwrite("c_d_1", &SingleCartPoleParams::c_d_1)
      .def_readwrite("x_s", &SingleCartPoleParams::x_s)
      .def_readwrite("k_s", &SingleCartPoleParams::k_s)
      .def("to_json", [](const SingleCartPoleParams& self) { return ToJson(self); })
      .def_static("from_json", &ParamsFromJson);

  py::class_<OptimizationParams>(m, "OptimizationParams")
      .def(py::init<>())
      .def_readwrite("control_dt", &OptimizationParams::control_dt)
      .def_readwrite("window_length", &OptimizationParams::window_length)
      .def_readwrite("state_spacing", &OptimizationParams::state_spacing)
      .def_readwrite("max_iterations", &OptimizationParams::max_iterations)
      .def_readwrite("relative_exit_tol", &OptimizationParams::relative_exit_tol)
      .def_readwrite("absolute_first_derivative_tol", &OptimizationParams::absolute_first_derivative_tol)
      .def_readwrite("equality_penalty_initial", &OptimizationParams::equality_penalty_initial)
      .def_readwrite("u_guess_sinusoid_amplitude", &OptimizationParams::u_guess_sinusoid_amplitude)
      .def_readwrite("u_cost_weight", &OptimizationParams::u_cost_weight)
      .def_readwrite("u_derivative_cost_weight", &OptimizationParams::u_derivative_cost_weight)
      .def_readwrite("b_x_final_cost_weight", &OptimizationParams::b_x_final_cost_weight)
      .def_readwrite("th_final_cost_weight", &OptimizationParams::th_final_cost_weight)
      .def_readwrite("b_x_dot_final_cost_weight", &OptimizationParams::b_x_dot_final_cost_weight)
      .def_readwrite("th_dot_final_cost_weight", &OptimizationParams::th_dot_final_cost_weight)
      .def("to_json", [](const OptimizationParams& self) { return ToJson(self); })
      .def_static("from_json", &OptimizationParamsFromJson);

  py::class_<SingleCartPoleState>(m, "SingleCartPoleState")
      .def(py::init<double, double, double, double>())
      .def_readwrite("b_x", &SingleCartPoleState::b_x)
      .def_readwrite("th_1", &SingleCartPoleState::th_1)
      .def_readwrite("b_x_dot", &SingleCartPoleState::b_x_dot)
      .def_readwrite("th_1_dot", &SingleCartPoleState::th_1_dot)
      .def("to_json", [](const SingleCartPoleState& self) { return ToJson(self); })
      .def_static("from_json", &StateFromJson);

  py::class_<OptimizationOutputs>(m, "OptimizationOutputs")
      .def("solver_summary", [](const OptimizationOutputs& self) { return self.solver_outputs.ToString(); })
      .def("get_log", [](const OptimizationOutputs& self) { return self.solver_outputs.ToString(); })  // wasm.cc:88-89
      .def("window_length", [](const OptimizationOutputs& self) { return self.u.size(); })              // wasm.cc:90-91
      .def("get_control",
           [](const OptimizationOutputs& self, std::size_t index) {                                     // wasm.cc:92-97
             if (index >= self.u.size()) throw py::index_error("control index out of range");
             return self.u[index];
           })
      .def("get_predicted_state",
           [](const OptimizationOutputs& self, std::size_t index) {                                     // wasm.cc:98-103
             if (index >= self.predicted_states.size()) throw py::index_error("state index out of range");
             return self.predicted_states[index];
           })
      .def_readonly("initial_state", &OptimizationOutputs::initial_state)
      .def_property_readonly("termination_state",
                             [](const OptimizationOutputs& self) {
                               return static_cast<int>(self.solver_outputs.termination_state);
                             })
      .def_readonly("previous_solution", &OptimizationOutputs::previous_solution)
      .def_readonly("u", &OptimizationOutputs::u)
      .def_readonly("predicted_states", &OptimizationOutputs::predicted_states)
      .def("to_json", [](const OptimizationOutputs& self) { return ToJson(self); })  // wasm.cc:104-105
      .def_static("from_json", &OptimizationOutputsFromJson);

  py::class_<BatchOptimizationOutputs>(m, "BatchOptimizationOutputs")
      .def_readonly("batch", &BatchOptimizationOutputs::batch)
      .def_readonly("u", &BatchOptimizationOutputs::u)
      .def_readonly("predicted_states", &BatchOptimizationOutputs::predicted_states)
      .def_readonly("status", &BatchOptimizationOutputs::status)
      .def_readonly("iterations", &BatchOptimizationOutputs::iterations)
      .def_readonly("final_cost", &BatchOptimizationOutputs::final_cost)
      .def_readonly("final_equality_l1", &BatchOptimizationOutputs::final_equality_l1);

  py::class_<Optimization>(m, "Optimization")
      .def(py::init<const OptimizationParams&>())
      .def(py::init<const OptimizationParams&, std::size_t, int, bool>(), py::arg("params"), py::arg("max_batch"),
           py::arg("device") = 0, py::arg("strict_horizon") = false)
      .def("step", &Optimization::Step)
      .def("step_batch", &StepBatchArrays<Optimization>, py::arg("states"), py::arg("dynamics_params"),
           py::arg("b_x_set_point"), py::arg("want_predicted") = true)
      .def("step_batch_lists", &Optimization::StepBatch)  // round 2's element-by-element form, kept for comparison
      .def("reset", &Optimization::Reset)
      .def("set_previous_solution", &Optimization::SetPreviousSolution)
      .def("set_previous_solution_batch", &Optimization::SetPreviousSolutionBatch, py::arg("z_soa"), py::arg("batch"))
      .def("get_solution_batch", &Optimization::GetSolutionBatch, py::arg("batch"))
      .def("set_host_chunk", &Optimization::SetHostChunk, py::arg("problems"))
      // the handle's horizon exceeds cpmpc_max_parity_horizon() (include/cpmpc.h): a per-object status, also in solver_summary()
      .def_property_readonly("horizon_beyond_parity", &Optimization::HorizonBeyondParity);

  py::class_<BatchArrays>(m, "BatchArrays")
      .def_readonly("batch", &BatchArrays::batch)
      .def_readonly("u", &BatchArrays::u)
      .def_readonly("predicted_states", &BatchArrays::predicted_states)
      .def_readonly("status", &BatchArrays::status)
      .def_readonly("iterations", &BatchArrays::iterations)
      .def_readonly("final_cost", &BatchArrays::final_cost)
      .def_readonly("final_equality_l1", &BatchArrays::final_equality_l1);

  py::class_<ShardedOptimization>(m, "ShardedOptimization")
      .def(py::init<const OptimizationParams&, std::size_t, const std::vector<int>&, bool>(), py::arg("params"),
           py::arg("max_batch"), py::arg("devices") = std::vector<int>{}, py::arg("strict_horizon") = false)
      .def("step_batch", &StepBatchArrays<ShardedOptimization>, py::arg("states"), py::arg("dynamics_params"),
           py::arg("b_x_set_point"), py::arg("want_predicted") = true)
      .def("reset", &ShardedOptimization::Reset)
      .def("set_previous_solution", &ShardedOptimization::SetPreviousSolution, py::arg("z_soa"), py::arg("batch"))
      .def("get_solution", &ShardedOptimization::GetSolution, py::arg("batch"))
      .def("previous_solution_batch", &ShardedOptimization::PreviousSolutionBatch)
      .def("num_shards", &ShardedOptimization::NumShards)
      .def("device_of_shard", &ShardedOptimization::DeviceOfShard)
      .def("shard_range", &ShardedOptimization::ShardRange)
      .def_property_readonly("horizon_beyond_parity", &ShardedOptimization::HorizonBeyondParity);

  py::class_<Vector2>(m, "Vector2")
      .def(py::init<double, double>())
      .def_readwrite("x", &Vector2::x)
      .def_readwrite("y", &Vector2::y)
      .def("to_json", [](const Vector2& self) { return ToJson(self); })
      .def_static("from_json", &Vector2FromJson)
      .def_static("list_from_json", &Vector2ListFromJson);
  m.def("get_default_optimization_params", []() { return OptimizationParams{}; });  // wasm.cc:118-119
  // wasm.cc:121-140: mini_opt's trace collector does not exist here; per-kernel timing is the batched API's
  // profile_enable/profile_read (cpmpc_profile_*), so the facade reports tracing as off, as a build without
  // MINI_OPT_TRACING does.
  m.def("is_tracing_enabled", []() { return false; });
  m.def("get_traces", []() { return std::string{}; });

  py::class_<Simulator>(m, "Simulator")
      .def(py::init<>())
      .def("step", &Simulator::Step)
      .def("get_state", &Simulator::GetState)
      .def("set_state", &Simulator::SetState);
}
