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
#include <thread>
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
  // ---- warm starts across a CHANGED batch size (round 4): the split of a batch depends on its size, so a sharded
  // optimizer hands the warm start over to the new split; a single handle keeps it in place.  Growing and shrinking
  // batches must stay bitwise equal to the single handle on the columns that are warm in both.
  {
    const std::size_t Bmax = 4133;
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> x_all(4 * Bmax);
    for (std::size_t i = 0; i < Bmax; ++i) {
      x_all[0 * Bmax + i] = 0.6 * U(rng);
      x_all[1 * Bmax + i] = M_PI * U(rng);
      x_all[2 * Bmax + i] = U(rng);
      x_all[3 * Bmax + i] = 3.0 * U(rng);
    }
    auto first_columns = [&](std::size_t B) {  // [4][B] of the first B problems
      std::vector<double> x(4 * B);
      for (int t = 0; t < 4; ++t) std::memcpy(&x[t * B], &x_all[t * Bmax], B * sizeof(double));
      return x;
    };
    Optimization single(params, Bmax);
    ShardedOptimization sharded(params, Bmax, devices);
    const std::size_t n = sharded.NumShards();
    // 3000 cold -> 4133 (first 3000 warm, the rest cold) -> 2000 (all warm) -> 2000 again
    for (const std::size_t B : {std::size_t{3000}, std::size_t{4133}, std::size_t{2000}, std::size_t{2000}}) {
      const std::vector<double> x = first_columns(B);
      const BatchOptimizationOutputs a = single.StepBatch(x, dyn, 0.02);
      const BatchOptimizationOutputs b = sharded.StepBatch(x, dyn, 0.02);
      EXPECT(same_bits(a.u, b.u) && same_bits(a.iterations, b.iterations) && same_bits(a.final_cost, b.final_cost),
             "hand-over: B=%zu differs between one handle and %zu shards", B, n);
      EXPECT(sharded.PreviousSolutionBatch() == B, "hand-over: %zu controllers warm after a step of %zu",
             sharded.PreviousSolutionBatch(), B);
    }
    // the warm starts themselves, read back: [dim][2000] of both
    EXPECT(same_bits(single.GetSolutionBatch(2000), sharded.GetSolution(2000)), "GetSolution differs after the hand-overs");
    // growing again: the single handle still holds columns 2000.. from the 4133-step, the sharded one dropped them
    // (documented): the first 2000 columns are warm in both and must agree, column by column
    {
      const std::vector<double> x = first_columns(Bmax);
      const BatchOptimizationOutputs a = single.StepBatch(x, dyn, 0.02);
      const BatchOptimizationOutputs b = sharded.StepBatch(x, dyn, 0.02);
      bool same = true;
      for (std::size_t k = 0; k < N && same; ++k)
        same = std::memcmp(&a.u[k * Bmax], &b.u[k * Bmax], 2000 * sizeof(double)) == 0;
      EXPECT(same, "hand-over: the columns that were warm in both differ after growing the batch again");
    }
    // SetPreviousSolution: a solution taken from one optimizer warm-starts a FRESH one of the other kind identically
    {
      const std::size_t B = 2500;
      const std::vector<double> z = single.GetSolutionBatch(B);
      Optimization single2(params, Bmax);
      ShardedOptimization sharded2(params, Bmax, devices);
      single2.SetPreviousSolutionBatch(z, B);
      sharded2.SetPreviousSolution(z, B);
      EXPECT(same_bits(sharded2.GetSolution(B), z), "SetPreviousSolution / GetSolution is not a round trip");
      const std::vector<double> x = first_columns(B);
      const BatchOptimizationOutputs a = single2.StepBatch(x, dyn, -0.03);
      const BatchOptimizationOutputs b = sharded2.StepBatch(x, dyn, -0.03);
      EXPECT(same_bits(a.u, b.u) && same_bits(a.status, b.status), "SetPreviousSolution: warm-started steps differ");
    }
    std::printf("warm-start hand-over across batch sizes 3000 -> 4133 -> 2000 -> 4133 and Set/GetSolution: bitwise equal (%zu shards)\n", n);

    // ---- per-problem parameters, set-points and terminal rows through the sharded host call, and the chunk pipeline ----
    {
      const std::size_t B = 3001;
      const std::vector<double> x = first_columns(B);
      std::vector<double> dynp(9 * B), sp(B), tw(4 * B);
      const auto d0 = dyn.ToArray();
      for (std::size_t i = 0; i < B; ++i) {
        for (int f = 0; f < 9; ++f) dynp[f * B + i] = d0[f] * (1.0 + 0.1 * U(rng));
        sp[i] = 0.2 * U(rng);
        tw[0 * B + i] = 100.0 + 50.0 * U(rng);          // b_x: always a cost row
        tw[1 * B + i] = (i % 3 == 0) ? 40.0 : -1.0;     // theta: a cost row for every third controller, else an equality
        tw[2 * B + i] = -1.0;
        tw[3 * B + i] = (i % 5 == 0) ? 5.0 : -1.0;
      }
      const PerProblemInputs pp{dynp.data(), sp.data(), tw.data()};
      const std::size_t dim = single.Dim();
      auto run = [&](auto& opt, std::vector<double>& u, std::vector<double>& pred, std::vector<std::int32_t>& st,
                     std::vector<double>& z) {
        u.assign(N * B, 0.0);
        pred.assign(4 * N * B, 0.0);
        st.assign(B, 0);
        z.assign(dim * B, 0.0);
        opt.StepBatchInto(x.data(), B, dyn, 0.0, pp, u.data(), pred.data(), st.data(), nullptr, nullptr, nullptr, z.data());
      };
      std::vector<double> u1, p1, z1, u2, p2, z2, u3, p3, z3;
      std::vector<std::int32_t> s1, s2, s3;
      Optimization a(params, B);
      a.SetHostChunk(0);  // one piece
      run(a, u1, p1, s1, z1);
      Optimization c(params, B);
      c.SetHostChunk(256);  // twelve chunks through three slots
      run(c, u3, p3, s3, z3);
      ShardedOptimization b(params, B, devices);
      b.SetHostChunk(128);  // every shard's slice in several chunks too: the shards' pipelines interleave
      run(b, u2, p2, s2, z2);
      EXPECT(same_bits(u1, u3) && same_bits(p1, p3) && same_bits(s1, s3) && same_bits(z1, z3),
             "per-problem inputs: the chunked host step differs from the unsplit one");
      EXPECT(same_bits(u1, u2) && same_bits(p1, p2) && same_bits(s1, s2) && same_bits(z1, z2),
             "per-problem inputs: %zu shards differ from one handle", b.NumShards());
      // and the per-problem inputs were really used: against the shared-parameter solve most controls must differ
      std::vector<double> u0(N * B);
      Optimization d(params, B);
      d.StepBatchInto(x.data(), B, dyn, 0.0, PerProblemInputs{}, u0.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      std::size_t differ = 0;
      for (std::size_t i = 0; i < B; ++i) differ += (u0[i] != u1[i]);
      EXPECT(differ > B / 2, "per-problem inputs had no effect (%zu of %zu first controls differ)", differ, B);
      std::printf("per-problem dyn / set-point / terminal rows: chunked == unsplit == %zu shards, bitwise\n", b.NumShards());
    }
  }

  // ---- two host threads, each with its own Optimization, stepping chunked batches at the same time: the library's
  // worker pool serves one parallel copy at a time and the calls must neither deadlock nor disturb each other's results
  {
    const std::size_t B = 6000;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> xa(4 * B), xb(4 * B);
    for (std::size_t i = 0; i < B; ++i) {
      xa[0 * B + i] = 0.6 * U(rng); xa[1 * B + i] = M_PI * U(rng); xa[2 * B + i] = U(rng); xa[3 * B + i] = 3.0 * U(rng);
      xb[0 * B + i] = 0.6 * U(rng); xb[1 * B + i] = M_PI * U(rng); xb[2 * B + i] = U(rng); xb[3 * B + i] = 3.0 * U(rng);
    }
    auto solve = [&](const std::vector<double>& x, std::vector<double>& u, std::vector<double>& pred, int reps) {
      Optimization opt(params, B);
      opt.SetHostChunk(512);
      u.assign(N * B, 0.0);
      pred.assign(4 * N * B, 0.0);
      for (int r = 0; r < reps; ++r) {
        opt.Reset();
        opt.StepBatchInto(x.data(), B, dyn, 0.01, PerProblemInputs{}, u.data(), pred.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
      }
    };
    std::vector<double> ua, pa, ub, pb, ua2, pa2, ub2, pb2;
    solve(xa, ua, pa, 1);
    solve(xb, ub, pb, 1);
    std::thread ta([&] { solve(xa, ua2, pa2, 6); });
    std::thread tb([&] { solve(xb, ub2, pb2, 6); });
    ta.join();
    tb.join();
    EXPECT(same_bits(ua, ua2) && same_bits(pa, pa2) && same_bits(ub, ub2) && same_bits(pb, pb2),
           "two threads stepping chunked batches concurrently changed a result");
    std::printf("two host threads x 6 chunked steps of %zu controllers: bitwise the serial results\n", B);
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
