// json.cc -- see json.hpp.  A writer that prints what nlohmann::json::dump() prints for these structs and a
// small recursive-descent reader (objects, arrays, numbers, strings, true/false/null).
#include "json.hpp"

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>

namespace pendulum {
namespace {

// ---- writer ------------------------------------------------------------------------------------------------
// Shortest round-trip digits, then the layout rule nlohmann uses (detail/conversions/to_chars.hpp,
// format_buffer): with n = position of the decimal point relative to the first digit, fixed notation for
// -4 < n <= 15 (appending ".0" to integral values), otherwise d[.ddd]e±XX with at least two exponent digits.
// Non-finite values print as null.
void AppendDouble(std::string& out, double v) {
  if (!std::isfinite(v)) {
    out += "null";
    return;
  }
  if (v == 0.0) {
    out += std::signbit(v) ? "-0.0" : "0.0";
    return;
  }
  char buf[40];
  const auto res = std::to_chars(buf, buf + sizeof(buf), std::fabs(v), std::chars_format::scientific);
  std::string sci(buf, res.ptr);  // d[.ddd]e±XX, shortest digits
  const std::size_t epos = sci.find('e');
  std::string digits;
  for (std::size_t i = 0; i < epos; ++i)
    if (sci[i] != '.') digits += sci[i];
  const int exp10 = std::atoi(sci.c_str() + epos + 1);
  const int k = static_cast<int>(digits.size());
  const int n = exp10 + 1;  // digits before the decimal point
  if (v < 0) out += '-';
  if (k <= n && n <= 15) {  // integral: digits, zeros, ".0"
    out += digits;
    out.append(static_cast<std::size_t>(n - k), '0');
    out += ".0";
  } else if (0 < n && n <= 15) {  // dig.its
    out.append(digits, 0, static_cast<std::size_t>(n));
    out += '.';
    out.append(digits, static_cast<std::size_t>(n), std::string::npos);
  } else if (-4 < n && n <= 0) {  // 0.[000]digits
    out += "0.";
    out.append(static_cast<std::size_t>(-n), '0');
    out += digits;
  } else {  // d[.igits]e±XX
    out += digits[0];
    if (k > 1) {
      out += '.';
      out.append(digits, 1, std::string::npos);
    }
    out += 'e';
    const int e = n - 1;
    out += (e < 0) ? '-' : '+';
    const int ae = e < 0 ? -e : e;
    if (ae < 10) out += '0';
    out += std::to_string(ae);
  }
}

void AppendKey(std::string& out, const char* key, bool first) {
  if (!first) out += ',';
  out += '"';
  out += key;
  out += "\":";
}

void AppendDoubleList(std::string& out, const std::vector<double>& v) {
  out += '[';
  for (std::size_t i = 0; i < v.size(); ++i) {
    if (i) out += ',';
    AppendDouble(out, v[i]);
  }
  out += ']';
}

const char* TerminationName(NLSTerminationState s) {
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
  return "NONE";
}

// ---- reader ------------------------------------------------------------------------------------------------
struct Value {
  enum Kind { kNull, kBool, kNumber, kString, kArray, kObject } kind{kNull};
  double number{0.0};
  bool boolean{false};
  std::string text;
  std::vector<Value> items;
  std::map<std::string, Value> members;
};

class Reader {
 public:
  explicit Reader(const std::string& s) : s_(s) {}
  Value Parse() {
    Value v = ParseValue();
    SkipSpace();
    if (pos_ != s_.size()) Fail("trailing characters");
    return v;
  }

 private:
  [[noreturn]] void Fail(const char* what) const {
    throw std::invalid_argument(std::string("JSON: ") + what + " at offset " + std::to_string(pos_));
  }
  void SkipSpace() {
    while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r')) ++pos_;
  }
  bool Consume(char c) {
    SkipSpace();
    if (pos_ < s_.size() && s_[pos_] == c) {
      ++pos_;
      return true;
    }
    return false;
  }
  void Expect(const char* word) {
    for (const char* p = word; *p; ++p) {
      if (pos_ >= s_.size() || s_[pos_] != *p) Fail("unexpected token");
      ++pos_;
    }
  }
  std::string ParseString() {
    std::string out;
    while (true) {
      if (pos_ >= s_.size()) Fail("unterminated string");
      const char c = s_[pos_++];
      if (c == '"') return out;
      if (c != '\\') {
        out += c;
        continue;
      }
      if (pos_ >= s_.size()) Fail("unterminated escape");
      const char e = s_[pos_++];
      switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          if (pos_ + 4 > s_.size()) Fail("short \\u escape");
          unsigned cp = 0;
          for (int i = 0; i < 4; ++i) {  // exactly four hex digits (strtoul would also take signs, blanks, fewer digits)
            const char h = s_[pos_ + static_cast<std::size_t>(i)];
            unsigned d;
            if (h >= '0' && h <= '9') d = static_cast<unsigned>(h - '0');
            else if (h >= 'a' && h <= 'f') d = static_cast<unsigned>(h - 'a') + 10u;
            else if (h >= 'A' && h <= 'F') d = static_cast<unsigned>(h - 'A') + 10u;
            else Fail("bad \\u escape");
            cp = cp * 16u + d;
          }
          pos_ += 4;
          if (cp < 0x80) {
            out += static_cast<char>(cp);
          } else if (cp < 0x800) {
            out += static_cast<char>(0xC0 | (cp >> 6));
            out += static_cast<char>(0x80 | (cp & 0x3F));
          } else {
            out += static_cast<char>(0xE0 | (cp >> 12));
            out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F));
            out += static_cast<char>(0x80 | (cp & 0x3F));
          }
          break;
        }
        default: Fail("bad escape");
      }
    }
  }
  // The API structs nest three levels deep (outputs -> predicted_states[] -> state); anything much deeper is not one of
  // them, and an unbounded recursion on "[[[[..." is a stack overflow (found by tests/host/json_fuzz.cc).
  static constexpr int kMaxDepth = 32;
  struct DepthGuard {
    int& d;
    explicit DepthGuard(int& depth) : d(depth) { ++d; }
    ~DepthGuard() { --d; }
  };
  Value ParseValue() {
    const DepthGuard guard(depth_);
    if (depth_ > kMaxDepth) Fail("nesting too deep");
    SkipSpace();
    if (pos_ >= s_.size()) Fail("unexpected end");
    Value v;
    const char c = s_[pos_];
    if (c == '{') {
      ++pos_;
      v.kind = Value::kObject;
      if (Consume('}')) return v;
      do {
        SkipSpace();
        if (pos_ >= s_.size() || s_[pos_] != '"') Fail("expected a key");
        ++pos_;
        std::string key = ParseString();
        if (!Consume(':')) Fail("expected ':'");
        v.members[std::move(key)] = ParseValue();
      } while (Consume(','));
      if (!Consume('}')) Fail("expected '}'");
    } else if (c == '[') {
      ++pos_;
      v.kind = Value::kArray;
      if (Consume(']')) return v;
      do {
        v.items.push_back(ParseValue());
      } while (Consume(','));
      if (!Consume(']')) Fail("expected ']'");
    } else if (c == '"') {
      ++pos_;
      v.kind = Value::kString;
      v.text = ParseString();
    } else if (c == 't') {
      Expect("true");
      v.kind = Value::kBool;
      v.boolean = true;
    } else if (c == 'f') {
      Expect("false");
      v.kind = Value::kBool;
    } else if (c == 'n') {
      Expect("null");
    } else {
      // a JSON number starts with '-' or a digit (strtod alone would also take "+1", "0x10", "nan", "inf", blanks)
      if (!(c == '-' || (c >= '0' && c <= '9'))) Fail("expected a value");
      const std::size_t digit = pos_ + (c == '-' ? 1u : 0u);
      if (digit >= s_.size() || s_[digit] < '0' || s_[digit] > '9') Fail("expected a digit");
      if (s_[digit] == '0' && digit + 1 < s_.size() && (s_[digit + 1] == 'x' || s_[digit + 1] == 'X')) Fail("hexadecimal number");
      const char* begin = s_.c_str() + pos_;
      char* end = nullptr;
      v.number = std::strtod(begin, &end);
      if (end == begin) Fail("expected a value");
      pos_ += static_cast<std::size_t>(end - begin);
      v.kind = Value::kNumber;
    }
    return v;
  }

  const std::string& s_;
  std::size_t pos_{0};
  int depth_{0};
};

const Value& Member(const Value& obj, const char* key) {
  if (obj.kind != Value::kObject) throw std::invalid_argument("JSON: expected an object");
  const auto it = obj.members.find(key);
  if (it == obj.members.end()) throw std::invalid_argument(std::string("JSON: missing key '") + key + "'");
  return it->second;
}
double Number(const Value& v, const char* what) {
  if (v.kind == Value::kNull) return std::nan("");  // non-finite values are written as null
  if (v.kind != Value::kNumber) throw std::invalid_argument(std::string("JSON: '") + what + "' is not a number");
  return v.number;
}
double Number(const Value& obj, const char* key, int) { return Number(Member(obj, key), key); }
std::size_t Index(const Value& obj, const char* key) {
  const double d = Number(obj, key, 0);
  // (the upper bound keeps the conversion below defined: a double >= 2^64 cast to size_t is undefined behaviour)
  if (!(d >= 0.0) || d != std::floor(d) || !(d <= 9007199254740992.0))
    throw std::invalid_argument(std::string("JSON: '") + key + "' is not an unsigned integer");
  return static_cast<std::size_t>(d);
}
std::vector<double> NumberList(const Value& v, const char* what) {
  if (v.kind != Value::kArray) throw std::invalid_argument(std::string("JSON: '") + what + "' is not an array");
  std::vector<double> out;
  out.reserve(v.items.size());
  for (const Value& e : v.items) out.push_back(Number(e, what));
  return out;
}
SingleCartPoleState StateFrom(const Value& o) {
  return SingleCartPoleState(Number(o, "b_x", 0), Number(o, "th_1", 0), Number(o, "b_x_dot", 0), Number(o, "th_1_dot", 0));
}
Vector2 Vector2From(const Value& o) { return Vector2{Number(o, "x", 0), Number(o, "y", 0)}; }

}  // namespace

// ---- ToJson (keys in sorted order, as nlohmann's std::map-backed objects print) ----------------------------
std::string ToJson(const SingleCartPoleState& v) {
  std::string o = "{";
  AppendKey(o, "b_x", true), AppendDouble(o, v.b_x);
  AppendKey(o, "b_x_dot", false), AppendDouble(o, v.b_x_dot);
  AppendKey(o, "th_1", false), AppendDouble(o, v.th_1);
  AppendKey(o, "th_1_dot", false), AppendDouble(o, v.th_1_dot);
  return o + "}";
}

std::string ToJson(const SingleCartPoleParams& v) {
  std::string o = "{";
  AppendKey(o, "c_d_1", true), AppendDouble(o, v.c_d_1);
  AppendKey(o, "g", false), AppendDouble(o, v.g);
  AppendKey(o, "k_s", false), AppendDouble(o, v.k_s);
  AppendKey(o, "l_1", false), AppendDouble(o, v.l_1);
  AppendKey(o, "m_1", false), AppendDouble(o, v.m_1);
  AppendKey(o, "m_b", false), AppendDouble(o, v.m_b);
  AppendKey(o, "mu_b", false), AppendDouble(o, v.mu_b);
  AppendKey(o, "v_mu_b", false), AppendDouble(o, v.v_mu_b);
  AppendKey(o, "x_s", false), AppendDouble(o, v.x_s);
  return o + "}";
}

std::string ToJson(const Vector2& v) {
  std::string o = "{";
  AppendKey(o, "x", true), AppendDouble(o, v.x);
  AppendKey(o, "y", false), AppendDouble(o, v.y);
  return o + "}";
}

std::string ToJson(const OptimizationParams& v) {
  std::string o = "{";
  AppendKey(o, "absolute_first_derivative_tol", true), AppendDouble(o, v.absolute_first_derivative_tol);
  AppendKey(o, "b_x_dot_final_cost_weight", false), AppendDouble(o, v.b_x_dot_final_cost_weight);
  AppendKey(o, "b_x_final_cost_weight", false), AppendDouble(o, v.b_x_final_cost_weight);
  AppendKey(o, "control_dt", false), AppendDouble(o, v.control_dt);
  AppendKey(o, "equality_penalty_initial", false), AppendDouble(o, v.equality_penalty_initial);
  AppendKey(o, "max_iterations", false), o += std::to_string(v.max_iterations);
  AppendKey(o, "relative_exit_tol", false), AppendDouble(o, v.relative_exit_tol);
  AppendKey(o, "state_spacing", false), o += std::to_string(v.state_spacing);
  AppendKey(o, "th_dot_final_cost_weight", false), AppendDouble(o, v.th_dot_final_cost_weight);
  AppendKey(o, "th_final_cost_weight", false), AppendDouble(o, v.th_final_cost_weight);
  AppendKey(o, "u_cost_weight", false), AppendDouble(o, v.u_cost_weight);
  AppendKey(o, "u_derivative_cost_weight", false), AppendDouble(o, v.u_derivative_cost_weight);
  AppendKey(o, "u_guess_sinusoid_amplitude", false), AppendDouble(o, v.u_guess_sinusoid_amplitude);
  AppendKey(o, "window_length", false), o += std::to_string(v.window_length);
  return o + "}";
}

std::string ToJson(const NLSSolverOutputs& v) {
  std::string o = "{";
  AppendKey(o, "final_cost", true), AppendDouble(o, v.final_cost);
  AppendKey(o, "final_equality_l1", false), AppendDouble(o, v.final_equality_l1);
  AppendKey(o, "iterations", false), o += std::to_string(v.iterations);
  AppendKey(o, "termination_state", false), o += '"', o += TerminationName(v.termination_state), o += '"';
  return o + "}";
}

std::string ToJson(const OptimizationOutputs& v) {
  std::string o = "{";
  AppendKey(o, "initial_state", true), o += ToJson(v.initial_state);
  AppendKey(o, "predicted_states", false);
  o += '[';
  for (std::size_t i = 0; i < v.predicted_states.size(); ++i) {
    if (i) o += ',';
    o += ToJson(v.predicted_states[i]);
  }
  o += ']';
  AppendKey(o, "previous_solution", false), AppendDoubleList(o, v.previous_solution);
  AppendKey(o, "solver_outputs", false), o += ToJson(v.solver_outputs);
  AppendKey(o, "u", false), AppendDoubleList(o, v.u);
  return o + "}";
}

// ---- FromJson ----------------------------------------------------------------------------------------------
SingleCartPoleState StateFromJson(const std::string& text) { return StateFrom(Reader(text).Parse()); }

SingleCartPoleParams ParamsFromJson(const std::string& text) {
  const Value o = Reader(text).Parse();
  return SingleCartPoleParams(Number(o, "m_b", 0), Number(o, "m_1", 0), Number(o, "l_1", 0), Number(o, "g", 0),
                              Number(o, "mu_b", 0), Number(o, "v_mu_b", 0), Number(o, "c_d_1", 0), Number(o, "x_s", 0),
                              Number(o, "k_s", 0));
}

Vector2 Vector2FromJson(const std::string& text) { return Vector2From(Reader(text).Parse()); }

std::vector<Vector2> Vector2ListFromJson(const std::string& text) {
  const Value a = Reader(text).Parse();
  if (a.kind != Value::kArray) throw std::invalid_argument("JSON: expected an array of {x, y}");
  std::vector<Vector2> out;
  for (const Value& e : a.items) out.push_back(Vector2From(e));
  return out;
}

OptimizationParams OptimizationParamsFromJson(const std::string& text) {
  const Value o = Reader(text).Parse();
  OptimizationParams p;
  p.control_dt = Number(o, "control_dt", 0);
  p.window_length = Index(o, "window_length");
  p.state_spacing = Index(o, "state_spacing");
  p.max_iterations = Index(o, "max_iterations");
  p.relative_exit_tol = Number(o, "relative_exit_tol", 0);
  p.absolute_first_derivative_tol = Number(o, "absolute_first_derivative_tol", 0);
  p.equality_penalty_initial = Number(o, "equality_penalty_initial", 0);
  p.u_guess_sinusoid_amplitude = Number(o, "u_guess_sinusoid_amplitude", 0);
  p.u_cost_weight = Number(o, "u_cost_weight", 0);
  p.u_derivative_cost_weight = Number(o, "u_derivative_cost_weight", 0);
  p.b_x_final_cost_weight = Number(o, "b_x_final_cost_weight", 0);
  p.th_final_cost_weight = Number(o, "th_final_cost_weight", 0);
  p.b_x_dot_final_cost_weight = Number(o, "b_x_dot_final_cost_weight", 0);
  p.th_dot_final_cost_weight = Number(o, "th_dot_final_cost_weight", 0);
  return p;
}

OptimizationOutputs OptimizationOutputsFromJson(const std::string& text) {
  const Value o = Reader(text).Parse();
  OptimizationOutputs out;
  out.initial_state = StateFrom(Member(o, "initial_state"));
  out.previous_solution = NumberList(Member(o, "previous_solution"), "previous_solution");
  const Value& so = Member(o, "solver_outputs");
  out.solver_outputs.final_cost = Number(so, "final_cost", 0);
  out.solver_outputs.final_equality_l1 = Number(so, "final_equality_l1", 0);
  out.solver_outputs.iterations = static_cast<int>(Index(so, "iterations"));
  const Value& ts = Member(so, "termination_state");
  if (ts.kind != Value::kString) throw std::invalid_argument("JSON: 'termination_state' is not a string");
  bool found = false;
  for (int s = 0; s <= 8; ++s) {
    if (ts.text == TerminationName(static_cast<NLSTerminationState>(s))) {
      out.solver_outputs.termination_state = static_cast<NLSTerminationState>(s);
      found = true;
    }
  }
  if (!found) throw std::invalid_argument("JSON: unknown termination_state '" + ts.text + "'");
  out.u = NumberList(Member(o, "u"), "u");
  const Value& ps = Member(o, "predicted_states");
  if (ps.kind != Value::kArray) throw std::invalid_argument("JSON: 'predicted_states' is not an array");
  for (const Value& e : ps.items) out.predicted_states.push_back(StateFrom(e));
  return out;
}

}  // namespace pendulum
