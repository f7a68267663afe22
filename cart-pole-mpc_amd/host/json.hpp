// json.hpp -- the JSON wire format of the API structs: what the reference's browser build exchanges
// with its UI and what OptimizationOutputs::toJson() emits (optimization/wasm.cc:19-28 field lists,
// wasm.cc:46-65 OptimizationOutputs, wasm.cc:105 toJson).  The reference serialises with
// nlohmann::json (absent here, an un-vendored dependency): objects print with their keys sorted,
// without whitespace, integers as integers and doubles in shortest round-trip form with ".0" appended
// to integral values.  ToJson reproduces that text; FromJson accepts any standard JSON spelling.
//
// "solver_outputs" differs by necessity: mini_opt's NLSSolverOutputs serialisation
// (mini_opt/serialization.hpp) is not available, so the object carries this repo's fields
// {final_cost, final_equality_l1, iterations, termination_state (name)}.
#pragma once
#include <string>
#include <vector>

#include "optimization.hpp"
#include "structs.hpp"

namespace pendulum {

std::string ToJson(const SingleCartPoleState& v);    // keys b_x, b_x_dot, th_1, th_1_dot (wasm.cc:19)
std::string ToJson(const SingleCartPoleParams& v);   // wasm.cc:20-21
std::string ToJson(const Vector2& v);                // wasm.cc:22
std::string ToJson(const OptimizationParams& v);     // wasm.cc:23-28
std::string ToJson(const NLSSolverOutputs& v);
std::string ToJson(const OptimizationOutputs& v);    // wasm.cc:56-62

// Throw std::invalid_argument on malformed text, a missing key or a value of the wrong type (the reference's
// json::parse(...).get<T>() throws nlohmann::json::exception in the same situations).
SingleCartPoleState StateFromJson(const std::string& text);
SingleCartPoleParams ParamsFromJson(const std::string& text);
Vector2 Vector2FromJson(const std::string& text);
std::vector<Vector2> Vector2ListFromJson(const std::string& text);  // the f_external argument, wasm.cc:78-80
OptimizationParams OptimizationParamsFromJson(const std::string& text);
OptimizationOutputs OptimizationOutputsFromJson(const std::string& text);  // wasm.cc:47-54

}  // namespace pendulum
