"""Build libcpmpc.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libcpmpc.so")
# five translation units, compiled in parallel: the C-ABI (no device code) and the kernels of each (dtype, model) pair
UNITS = ["cpmpc_api", "engine_f32_single", "engine_f64_single", "engine_f32_double", "engine_f64_double"]
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]
# -fno-slp-vectorize: on gfx950 a v_pk_fma_f32 issues at ~1.8x the cost of a v_fma_f32 (tools/ubench/pk.hip), so the
# SLP vectoriser's packing plus its pairing moves is a net loss here (measured 91M -> 104M re-plans/s, 255 -> 189 VGPRs
# for the fused SQP kernel)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]
# per unit.  The float kernels are scheduled by LLVM's iterative ILP strategy: at two waves per SIMD they wait on their own
# dependent chains (SQ_WAIT_INST_ANY 30 % of wave-cycles) and that scheduler spaces them better -- same instructions, same
# results, 117.1 -> 119.8 M re-plans/s in two same-session A/Bs (round 4, tools/ab_both.sh; no spills instead of 4).  The
# double kernels (one wave per SIMD, at the register limit) lose 0.4 % with it and 1.5 - 4 % with max-ilp, iterative-minreg,
# iterative-maxocc and max-memory-clause, so they keep the default; amdgpu-schedule-metric-bias=0 changes nothing.
# Without the post-RA scheduler on top (it undoes part of that spacing): 119.9 -> 121.8 M; the double kernels lose 7.7 %
# without it, so again only the float units.
_F32_SCHED = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-enable-post-misched=0"]
# The double units: relaxed occupancy targets in the scheduler -- nothing for the fused kernel (50.0 -> 50.1, inside the
# noise), finalize_kernel 0.178 -> 0.162 ms at B = 262 144 in four of four A/B runs.
_F64_SCHED = ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=1"]
UNIT_FLAGS = {"engine_f32_single": _F32_SCHED, "engine_f32_double": _F32_SCHED,
              "engine_f64_single": _F64_SCHED, "engine_f64_double": _F64_SCHED}


def _deps():
    """Every file the library is compiled from: all of csrc/, the public header and this recipe (flags)."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h", ".inc"))]
    return files + [os.path.join(HERE, "..", "include", "cpmpc.h"), os.path.abspath(__file__)]



def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:  # one builder at a time (torchrun ranks)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB
        return _compile(verbose)


def build_variant(name, flags, force=False, verbose=False):
    """The same library with extra hipcc flags, in tools/_build/lib_<name>/ (used through CPMPC_LIB=<path>):
    `generated` = -DCPMPC_GENERATED_SINGLE=1 (kernels on the generated single-pendulum dynamics),
    `timing` = -DCPMPC_FUSED_TIMING (per-phase shader-clock counters, tools/phase_timing.py)."""
    out = os.path.join(HERE, "..", "tools", "_build", "lib_" + name, "libcpmpc.so")
    out = os.path.abspath(out)
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in _deps()):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    import fcntl
    with open(os.path.join(os.path.dirname(out), ".build.lock"), "w") as lock:  # one builder at a time, as build_lib
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in _deps()):
            return out
        _compile_units(out, list(flags), verbose)
    return out


def _compile_units(out, extra_flags, verbose):
    """hipcc -c every unit (in parallel: the kernels of one (dtype, model) pair take about a minute each), then link.
    Objects go next to the library; the library itself is replaced atomically."""
    from concurrent.futures import ThreadPoolExecutor

    objdir = os.path.join(os.path.dirname(out), "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def one(unit):
        obj = os.path.join(objdir, "%s.%d.o" % (unit, os.getpid()))
        # a variant that names a scheduling strategy itself replaces the unit's (the option may be given once)
        unit_flags = [] if any("amdgpu-sched-strategy" in f for f in extra_flags) else UNIT_FLAGS.get(unit, [])
        cmd = [hipcc] + HIPCC_FLAGS + unit_flags + extra_flags + ["-c", "-o", obj, os.path.join(CSRC, unit + ".hip")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        # hipcc's warnings (hundreds of "loop not unrolled" remarks for the run-time-spacing kernels) are shown only when
        # asked for or when the compilation fails
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or verbose:
            sys.stderr.write(r.stdout)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        return obj

    jobs = int(os.environ.get("CPMPC_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(one, UNITS))
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)  # atomic: a concurrent loader never sees a half-written library
    for o in objs:
        os.remove(o)
    return out


def _compile(verbose):
    return _compile_units(LIB, [], verbose)


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIB_DIR, "libpendulum_host.so")
HOST_SMOKE = os.path.join(LIB_DIR, "host_smoke")
SHARDED_SMOKE = os.path.join(LIB_DIR, "sharded_smoke")


def _pymod_path():
    import sysconfig
    return os.path.join(LIB_DIR, "pypendulum" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_host(force=False, verbose=False):
    """C++ facade (pendulum::Optimization / Simulator over the C-ABI), the pypendulum module
    (pybind11) and the C++ closed-loop smoke binary.  Plain g++: the facade only sees include/cpmpc.h."""
    import sysconfig

    build_lib()
    srcs = [os.path.join(HOST, f) for f in ("optimization.cc", "sharded_optimization.cc", "simulator.cc", "json.cc")]
    hdrs = [os.path.join(HOST, f) for f in ("optimization.hpp", "sharded_optimization.hpp", "simulator.hpp", "structs.hpp",
                                            "json.hpp")]
    cxx = os.environ.get("CXX", "g++")
    common = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]
    link = ["-L" + LIB_DIR, "-lcpmpc", "-Wl,-rpath,$ORIGIN"]

    def stale(target, deps):
        return force or not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target)
                                                          for d in deps + [LIB])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if stale(HOST_LIB, srcs + hdrs):
        run([cxx] + common + ["-shared", "-o", HOST_LIB] + srcs + link)
    # test infrastructure: C++ callers of the facade (closed loop through Optimization + Simulator; several shards
    # against one handle)
    for target, name in ((HOST_SMOKE, "facade_closed_loop.cc"), (SHARDED_SMOKE, "sharded_smoke.cc")):
        src = os.path.join(HERE, "..", "tests", "host", name)
        if stale(target, [src, HOST_LIB] + hdrs):
            run([cxx] + common + ["-pthread", "-I" + HOST, "-o", target, src, "-L" + LIB_DIR, "-lpendulum_host", "-lcpmpc",
                                  "-Wl,-rpath,$ORIGIN"])
    try:
        import pybind11
    except ImportError:
        return HOST_LIB  # facade built; Python binding needs pybind11
    mod = _pymod_path()
    mod_src = os.path.join(HOST, "pypendulum.cc")
    if stale(mod, [mod_src, HOST_LIB] + hdrs):
        run([cxx] + common + ["-shared", "-fvisibility=hidden", "-o", mod, mod_src,
                              "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
                              "-L" + LIB_DIR, "-lpendulum_host", "-lcpmpc", "-Wl,-rpath,$ORIGIN"])
    return mod


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    if "--variants" in sys.argv:
        print(build_variant("generated", ["-DCPMPC_GENERATED_SINGLE=1"], verbose=True))
