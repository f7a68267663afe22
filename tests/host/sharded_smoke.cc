// sharded_smoke.cc -- TEST INFRASTRUCTURE: pendulum::ShardedOptimization (several shards, one process) against
// pendulum::Optimization (one handle) on the same batch.  On a one-GPU box the shards all live on device 0
// (`sharded_smoke 0 0 0`): a problem's arithmetic does not depend on the handle, the stream or the position in the
// batch it is solved at, so the results must be BITWISE those of the single handle, ragged split included.
// usage: sharded_smoke [device ...]   (no arguments: every visible device, one shard each).  Exit code 0 = all hold.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <vector>

#include "optimization.hpp"
#include "sharded_optimization.hpp"

using namespace pendulum;

static int failures = 0;
#define EXPECT(cond, ...)                         \
  do {                                            \
    if (!(cond)) {                                \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
      std::printf(__VA_ARGS__);                   \
      std::printf("\n");                          \
      ++failures;                                 \
    }                                             \
  } while (0)

template <typename T>
static bool same_bits(const std::vector<T>& a, const std::vector<T>& b) {
  return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
}

int main(int argc, char** argv) {
  std::vector<int> devices;
  for (int i = 1; i < argc; ++i) devices.push_back(std::atoi(argv[i]));
  OptimizationParams params{};
  params.max_iterations = 5;
  params.relative_exit_tol = 0.0;
  params.absolute_first_derivative_tol = 0.0;
  const SingleCartPoleParams dyn{1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0};
  const std::size_t N = params.window_length;

  for (const std::size_t B : {std::size_t{4133}, std::size_t{5}, std::size_t{64}}) {  // ragged, tiny, even
    std::mt19937_64 rng(B);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> x0(4 * B);
    for (std::size_t i = 0; i < B; ++i) {
      x0[0 * B + i] = 0.6 * U(rng);
      x0[1 * B + i] = M_PI * U(rng);
      x0[2 * B + i] = U(rng);
      x0[3 * B + i] = 3.0 * U(rng);
    }
    Optimization single(params, B);
    ShardedOptimization sharded(params, B, devices);
    const std::size_t n = sharded.NumShards();
    EXPECT(n >= 1, "no shards");
    // the split is contiguous, covers [0, B) and differs by at most one
    std::size_t next = 0, smallest = B, largest = 0;
    for (std::size_t s = 0; s < n; ++s) {
      const auto r = sharded.ShardRange(s, B);
      EXPECT(r.first == next && r.second >= r.first, "shard %zu covers [%zu, %zu), expected to start at %zu", s, r.first,
             r.second, next);
      next = r.second;
      smallest = std::min(smallest, r.second - r.first);
      largest = std::max(largest, r.second - r.first);
    }
    EXPECT(next == B && largest - smallest <= 1, "split does not cover the batch evenly");

    for (int tick = 0; tick < 3; ++tick) {  // cold start, then two warm-started steps (per-shard warm-start state)
      const BatchOptimizationOutputs a = single.StepBatch(x0, dyn, 0.05);
      const BatchOptimizationOutputs b = sharded.StepBatch(x0, dyn, 0.05);
      EXPECT(a.batch == B && b.batch == B && a.u.size() == N * B, "shapes");
      EXPECT(same_bits(a.u, b.u), "B=%zu tick %d: u differs between one handle and %zu shards", B, tick, n);
      EXPECT(same_bits(a.predicted_states, b.predicted_states), "B=%zu tick %d: predicted states differ", B, tick);
      EXPECT(same_bits(a.status, b.status) && same_bits(a.iterations, b.iterations), "B=%zu tick %d: status differs", B, tick);
      EXPECT(same_bits(a.final_cost, b.final_cost) && same_bits(a.final_equality_l1, b.final_equality_l1),
             "B=%zu tick %d: summaries differ", B, tick);
      for (std::size_t i = 0; i < B; ++i) x0[1 * B + i] += 1e-3;  // the measured state moves a little between ticks
    }
    single.Reset();
    sharded.Reset();
    const BatchOptimizationOutputs a = single.StepBatch(x0, dyn, 0.0);
    const BatchOptimizationOutputs b = sharded.StepBatch(x0, dyn, 0.0);
    EXPECT(same_bits(a.u, b.u), "B=%zu: u differs after Reset", B);
    std::printf("B=%zu over %zu shard(s): bitwise equal to the single handle (3 ticks + reset)\n", B, n);
  }
  // capacity and argument errors surface as exceptions, like Optimization's
  bool threw = false;
  try {
    ShardedOptimization tiny(params, 8, devices);
    std::vector<double> too_many(4 * 9, 0.0);
    (void)tiny.StepBatch(too_many, dyn, 0.0);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  EXPECT(threw, "a batch above the capacity must throw std::invalid_argument");
  if (failures == 0) std::printf("OK sharded\n");
  return failures == 0 ? 0 : 1;
}
