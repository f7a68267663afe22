"""Host-side sanitizer builds (sanitizers on the CPU build only; the reference has an
EMSCRIPTEN_ENABLE_SANITIZE build and a leak-check hook, CMakeLists.txt:7,23-26, wasm.cc:140-144):
the JSON reader/writer of the facade -- the host code that parses untrusted text -- under ASan + UBSan with a
mutation fuzzer, and the oracle's C under the same sanitizers."""
import os
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "cart-pole-mpc_amd", "host")


def test_json_fuzz_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "json_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "host", "json_fuzz.cc"), os.path.join(HOST, "json.cc")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "60000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 other outcomes" in r.stdout, r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_oracle_under_asan_ubsan(tmp_path):
    """The oracle's C (the checker itself) through a cold re-plan, the plant and the generated dynamics under
    -fsanitize=address,undefined (oracle/Makefile: libcpmpc_oracle_asan.so), driven by a small C program."""
    src = tmp_path / "drive.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include "cpmpc_oracle.h"
int main(void) {
  orc_opt_params p; orc_default_opt_params(&p); p.max_iterations = 4;
  const double dyn[9] = {1.0, 0.1, 0.25, 9.81, 0.05, 0.1, 0.02, 0.8, 100.0};
  enum { B = 24 };
  double x0[4 * B], u[40 * B], pred[160 * B]; int status[B], iters[B];
  for (int b = 0; b < B; ++b) { x0[b] = 0.03 * b - 0.3; x0[B + b] = -1.5 + 0.2 * b; x0[2 * B + b] = 0.1 * b - 1; x0[3 * B + b] = 0.3 * b - 3; }
  orc_step_batch_cold(&p, NULL, dyn, 0.0, B, x0, u, pred, status, iters, 2);
  double st[4] = {0.79, 3.1, 1.0, 4.0}, fb[2] = {1, 0}, fm[2] = {0, -1}, f[4], Jx[16], Ju[4];
  for (int i = 0; i < 50; ++i) orc_sim_step(dyn, 0.0105, 3.0, fb, fm, st);
  orc_dynamics_generated(dyn, st, 1.0, fb, fm, f, Jx, Ju);
  orc_optimization* o = orc_opt_create(&p, NULL);
  double uu[40], pp[160], g[60], z[60]; orc_solver_summary s;
  for (int t = 0; t < 3; ++t) orc_opt_step(o, st, dyn, 0.1, uu, pp, g, z, &s);
  orc_opt_destroy(o);
  printf("ok %d %g %g\n", status[0], u[0], f[3]);
  return 0;
}
''')
    exe = str(tmp_path / "drive")
    orc_dir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-std=c99", "-g", "-O1", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + orc_dir, "-o", exe, str(src), os.path.join(orc_dir, "cpmpc_oracle.c"), "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2"))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-4000:]
    assert "runtime error" not in r.stderr, r.stderr[-4000:]
