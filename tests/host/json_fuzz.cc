// json_fuzz.cc -- sanitizer + fuzz target for the host facade's JSON reader/writer (cart-pole-mpc_amd/host/json.cc),
// the one piece of the host layer that parses untrusted text (the UI's log format, optimization/wasm.cc:19-65).
// CPU build only:  g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all  json_fuzz.cc json.cc
// (tests/test_host_sanitizers.py builds and runs it).  No GPU library is linked: json.cc only needs the struct headers.
//
// 1. round trips: every struct -> ToJson -> FromJson -> ToJson is a fixed point, also for awkward doubles;
// 2. mutation fuzzing: valid documents with bytes flipped / inserted / deleted / spliced / truncated, plus hostile
//    hand-written inputs (deep nesting, huge numbers, lone surrogates, unterminated strings).  A reader may only
//    return a value or throw std::invalid_argument; anything else (another exception, a sanitizer report, a crash,
//    a hang) fails the run.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "json.hpp"

using namespace pendulum;

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return g_rng;
}

static long g_ok = 0, g_rejected = 0;

template <typename F>
static void feed(const std::string& text, F&& parse) {
  try {
    parse(text);
    ++g_ok;
  } catch (const std::invalid_argument&) {
    ++g_rejected;
  } catch (const std::exception& e) {
    std::printf("FAIL: unexpected exception type (%s) on input of %zu bytes: %.80s\n", e.what(), text.size(), text.c_str());
    std::exit(1);
  }
}

static void feed_all(const std::string& t) {
  feed(t, [](const std::string& s) { (void)StateFromJson(s); });
  feed(t, [](const std::string& s) { (void)ParamsFromJson(s); });
  feed(t, [](const std::string& s) { (void)Vector2FromJson(s); });
  feed(t, [](const std::string& s) { (void)Vector2ListFromJson(s); });
  feed(t, [](const std::string& s) { (void)OptimizationParamsFromJson(s); });
  feed(t, [](const std::string& s) { (void)OptimizationOutputsFromJson(s); });
}

static std::string mutate(const std::string& seed, const std::vector<std::string>& corpus) {
  std::string s = seed;
  const int n_ops = 1 + (int)(rnd() % 4);
  static const char kAlphabet[] = "{}[]\",:-+.eE0123456789 \t\n\\/utfalsn\x00\x7f\xc3\xff";
  for (int op = 0; op < n_ops; ++op) {
    const size_t pos = s.empty() ? 0 : (size_t)(rnd() % s.size());
    switch (rnd() % 7) {
      case 0: if (!s.empty()) s[pos] = (char)(rnd() & 0xff); break;                       // random byte
      case 1: if (!s.empty()) s[pos] = kAlphabet[rnd() % (sizeof kAlphabet - 1)]; break;  // structural byte
      case 2: s.insert(pos, 1, kAlphabet[rnd() % (sizeof kAlphabet - 1)]); break;         // insert
      case 3: if (!s.empty()) s.erase(pos, 1 + (size_t)(rnd() % 4)); break;               // delete a few
      case 4: s.resize(pos); break;                                                        // truncate
      case 5: {                                                                            // splice another document in
        const std::string& o = corpus[rnd() % corpus.size()];
        const size_t a = (size_t)(rnd() % (o.size() + 1));
        s.insert(pos, o.substr(a, (size_t)(rnd() % 64)));
        break;
      }
      case 6: if (!s.empty()) s.insert(pos, s.substr(pos, (size_t)(rnd() % 32))); break;  // duplicate a run
    }
    if (s.size() > 1u << 16) s.resize(1u << 16);
  }
  return s;
}

#define REQUIRE(cond)                                              \
  do {                                                             \
    if (!(cond)) {                                                 \
      std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main(int argc, char** argv) {
  const long iterations = argc > 1 ? std::atol(argv[1]) : 200000;

  // ---- 1. round trips ---------------------------------------------------------------------------------------
  const double awkward[] = {0.0, -0.0, 1.0, -1.5, 0.1, 1e-300, -1e300, 5e-324, 1.7976931348623157e308, 3.141592653589793,
                            123456789012345678.0, 1e21, 1e-7, 0.30000000000000004};
  std::vector<std::string> corpus;
  for (double a : awkward) {
    SingleCartPoleState st(a, -a, a * 0.5, 2.0 * a);
    const std::string t = ToJson(st);
    REQUIRE(ToJson(StateFromJson(t)) == t);
    corpus.push_back(t);
    Vector2 v2{a, -a};
    REQUIRE(ToJson(Vector2FromJson(ToJson(v2))) == ToJson(v2));
  }
  SingleCartPoleParams prm(1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0);
  REQUIRE(ToJson(ParamsFromJson(ToJson(prm))) == ToJson(prm));
  OptimizationParams op{};
  op.window_length = 40;
  REQUIRE(ToJson(OptimizationParamsFromJson(ToJson(op))) == ToJson(op));
  OptimizationOutputs out;
  out.initial_state = SingleCartPoleState(0.1, -1.5707963267948966, 0.0, 2.5);
  out.previous_solution = {1.0, -2.5, 1e-12};
  out.solver_outputs.iterations = 5;
  out.solver_outputs.termination_state = NLSTerminationState::SATISFIED_RELATIVE_TOL;
  out.solver_outputs.final_cost = 12.5;
  out.u = {0.0, -0.0, 299.99999999999994, -300.0};
  out.predicted_states.emplace_back(0.0, 1.0, 2.0, 3.0);
  out.predicted_states.emplace_back(-4.0, 5.5, -6.25, 7.125);
  REQUIRE(ToJson(OptimizationOutputsFromJson(ToJson(out))) == ToJson(out));
  corpus.push_back(ToJson(prm));
  corpus.push_back(ToJson(op));
  corpus.push_back(ToJson(out));
  corpus.push_back("[{\"x\":1.0,\"y\":-2.0},{\"x\":0.5,\"y\":0.25}]");

  // ---- 2. hostile inputs ------------------------------------------------------------------------------------------
  std::vector<std::string> hostile = {
      "", " ", "{", "}", "[", "]", "\"", "{\"b_x\":", "{\"b_x\":1,}", "nul", "tru", "-", "-.", "1e", "1e+", "0x10", "NaN", "Infinity",
      "{\"b_x\":1e999,\"b_x_dot\":-1e999,\"th_1\":0,\"th_1_dot\":0}", "{\"b_x\":\"1\"}", "{\"b_x\":[1]}", "{\"b_x\":{}}",
      "{\"b_x\":1,\"b_x\":2,\"b_x_dot\":0,\"th_1\":0,\"th_1_dot\":0}", "\"\\ud800\"", "\"\\udc00\\ud800\"", "\"\\u12\"", "\"\\x\"",
      "{\"window_length\":-1}", "{\"window_length\":1.5}", "{\"window_length\":18446744073709551616}", "{\"window_length\":1e30}",
      std::string(1 << 15, '['), std::string(1 << 15, '{'), std::string(1 << 14, '[') + std::string(1 << 14, ']'),
      "[" + std::string(1 << 15, '9') + "]", "\"" + std::string(1 << 15, '\\'), std::string("{\"a\":\0001}", 8),
      "\xef\xbb\xbf{}", "{\"u\":[1,2,", "{\"predicted_states\":[{\"b_x\":1}]}", "[[[[[[[[[[[[[[[[[[[[1]]]]]]]]]]]]]]]]]]]]"};
  for (const auto& h : hostile) feed_all(h);

  // ---- 3. mutation fuzzing -----------------------------------------------------------------------------------------
  for (long it = 0; it < iterations; ++it) {
    const std::string& seed = corpus[rnd() % corpus.size()];
    feed_all(mutate(seed, corpus));
  }
  std::printf("json fuzz: %ld iterations x 6 readers, %ld parsed, %ld rejected with std::invalid_argument, 0 other outcomes\n",
              iterations, g_ok, g_rejected);
  return 0;
}
