# -*- coding: utf-8 -*-
"""In-tree build of the gfx950 HIP library and the pybind11 drop-in modules.

    python -m celerite2_amd.build            # build what is stale
    python -m celerite2_amd.build --force

Outputs (git-ignored, but they travel to the GPU box with the tree):
    celerite2_amd/libcelerite2_amd.so                  C-ABI + HIP kernels (hipcc, --offload-arch=gfx950)
    celerite2_amd/driver.cpython-*.so                  pybind11, mirrors celerite2.driver
    celerite2_amd/backprop.cpython-*.so                pybind11, mirrors celerite2.backprop
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libcelerite2_amd.so")
HIP_SOURCES = ["c2_dispatch.hip", "c2_ops.hip", "c2_fused.hip", "c2_loglik.hip", "c2_loglik4.hip", "c2_loglik_q4.hip", "c2_loglik_t.hip", "c2_loglik_k2.hip", "c2_loglik_t6.hip", "c2_loglik_t4.hip", "c2_loglik_t2.hip", "c2_timepar.hip", "c2_timepar_grad.hip", "c2_timepar_grad32.hip", "c2_timepar_grad16.hip", "c2_sweep.hip", "c2_sweep_rev.hip", "c2_sweep_small.hip", "c2_sweep_small_rev.hip", "c2_solve_cols.hip", "c2_sweep_cols.hip", "c2_scan.hip", "c2_general.hip", "c2_general_tile.hip", "c2_mfma.hip", "c2_wide.hip", "c2_kron.hip", "c2_terms.hip", "c2_host.hip"]
HIP_HEADERS = ["c2_common.hpp", "c2_dispatch.hpp", "c2_loglik_helpers.hpp", "c2_rscatter.hpp", os.path.join(INCLUDE, "celerite2_amd.h")]
# sources that are #included by other sources (one compilation per width / chunk length): extra dependencies of those only
HIP_INCLUDED = {"c2_loglik_t.hip": ["c2_loglik_t2.hip", "c2_loglik_t4.hip", "c2_loglik_t6.hip"],
                "c2_timepar_grad.hip": ["c2_timepar_grad16.hip", "c2_timepar_grad32.hip"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# max-ilp machine scheduling: the kernels run at one wavefront per SIMD, so latency is hidden by ILP, not occupancy
# (measured +5 % on the fused gradient pair vs the default occupancy-driven strategy).
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "--amdgpu-sched-strategy=max-ilp"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    tt = os.path.getmtime(target)
    return any(os.path.getmtime(s) > tt for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False):
    """Compile every .hip source to an object in parallel (one hipcc per file), then link the shared library."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    if not (force or _stale(LIB, deps)):
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIPFLAGS if f != "-shared"]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        extra = [os.path.join(CSRC, inc) for inc, users in HIP_INCLUDED.items() if os.path.basename(src) in users]
        if force or _stale(obj, [src] + hdrs + extra):
            _run([HIPCC] + cflags + ["-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, srcs))
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


def ext_path(name):
    return os.path.join(HERE, name + sysconfig.get_config_var("EXT_SUFFIX"))


def build_pybind(force=False):
    import pybind11

    out = []
    for name in ("driver", "backprop"):
        src = os.path.join(CSRC, "py_%s.cpp" % name)
        target = ext_path(name)
        deps = [src, os.path.join(CSRC, "py_common.hpp"), os.path.join(INCLUDE, "celerite2_amd.h"), LIB]
        if force or _stale(target, deps):
            _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
                  "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], src,
                  "-L" + HERE, "-lcelerite2_amd", "-Wl,-rpath,$ORIGIN", "-o", target])
        out.append(target)
    return out


def build_all(force=False):
    """Build what is stale; prints one summary line (wall time, what was recompiled) so that a build log says
    unambiguously whether the compilers ran or everything was already up to date."""
    import time

    t0 = time.time()
    before = {f: os.path.getmtime(f) for f in [LIB] + [ext_path(n) for n in ("driver", "backprop")] if os.path.exists(f)}
    objdir = os.path.join(HERE, "build")
    obj_before = {f: os.path.getmtime(os.path.join(objdir, f)) for f in (os.listdir(objdir) if os.path.isdir(objdir) else [])}
    build_hip(force)
    build_pybind(force)
    rebuilt = [os.path.basename(f) for f in [LIB] + [ext_path(n) for n in ("driver", "backprop")]
               if before.get(f) != os.path.getmtime(f)]
    objs = [f for f in (os.listdir(objdir) if os.path.isdir(objdir) else [])
            if obj_before.get(f) != os.path.getmtime(os.path.join(objdir, f))]
    print("celerite2_amd.build: %.1f s; recompiled objects: %s; relinked: %s"
          % (time.time() - t0, ", ".join(sorted(objs)) or "none (up to date)", ", ".join(rebuilt) or "none (up to date)"),
          flush=True)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
