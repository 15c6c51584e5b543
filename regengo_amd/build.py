"""In-tree builds (no pip, no JIT cache): hipcc for the product library, g++ for the test-only host walker (the oracle's
C restatement is built by oracle/build.py: nothing in this package imports the oracle).  Everything lands next to the sources so that a `gpurun`
snapshot carries the binaries to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "regengo_amd", "csrc")
LIBDIR = os.path.join(ROOT, "regengo_amd", "lib")

PRODUCT_SOURCES = ["rgx_syntax.cc", "rgx_dfa.cc", "rgx_ref_engine.cc", "rgx_program.cc", "rgx_kernels.hip", "rgx_scan_exact.hip", "rgx_scan_sa.hip", "rgx_scan_us.hip", "rgx_scan_fc.hip", "rgx_tdfa.hip", "rgx_batch_tiny.hip", "rgx_replace.hip", "rgx_sharded.hip", "rgx_capi.cc"]
PRODUCT_LINK = ["-ldl", "-lpthread"]          # extra link arguments of the product library (RCCL itself is dlopen-ed: rgx_sharded.hip)
HOSTTEST_SOURCES = ["rgx_syntax.cc", "rgx_dfa.cc", "rgx_ref_engine.cc", "hosttest/rgx_hosttest.cc"]


def _hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or ""):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the product library cannot be built")


def _stamp(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _needs(target: str, deps, extra="") -> str | None:
    st = _stamp(deps, extra)
    sf = target + ".stamp"
    if os.path.exists(target) and os.path.exists(sf) and open(sf).read() == st:
        return None
    return st


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r.stdout


def _all_headers():
    hs = [os.path.join(ROOT, "include", "rgx.h")]
    for d, _, fs in os.walk(CSRC):
        hs += [os.path.join(d, f) for f in fs if f.endswith(".h") or f.endswith(".inc")]
    return hs


def product_lib_path() -> str:
    return os.path.join(LIBDIR, "librgx_hip.so")


def hosttest_lib_path() -> str:
    return os.path.join(LIBDIR, "librgx_hosttest.so")


def _compile_objects(srcs, flags, verbose=False):
    """One object per source, cached by a stamp over the source, every header and the flags; compiled in parallel (a
    HIP file takes most of a minute, the link a second: touching one kernel no longer rebuilds the other four)."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _all_headers()
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        st = _needs(o, [s] + hdrs, " ".join(flags))
        if st is not None:
            jobs.append((s, o, st))

    def one(job):
        s, o, st = job
        tmp = o + ".tmp%d" % os.getpid()
        log = _run([_hipcc()] + flags + ["-c", s, "-o", tmp])
        os.replace(tmp, o)
        open(o + ".stamp", "w").write(st)
        return log

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for log in ex.map(one, jobs):
                if verbose and log:
                    print(log)
    return objs


def build_product(verbose=False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    out = product_lib_path()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SOURCES]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC, "-Wno-unused-result", "-fvisibility=hidden", "-DRGX_BUILDING"]
    flags += os.environ.get("RGX_EXTRA_FLAGS", "").split()      # experiments only (e.g. -DRGX_US_PROFILE); part of the build stamp
    link = ["-shared", "-fPIC", "--offload-arch=gfx950"] + PRODUCT_LINK
    st = _needs(out, srcs + _all_headers(), " ".join(flags + link))
    if st is None:
        return out
    # several ranks of one job may get here at once: one builds (into a temporary, then an atomic rename), the rest wait
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if _needs(out, srcs + _all_headers(), " ".join(flags + link)) is None:
            return out
        # hipcc treats .cc as host C++ and .hip as HIP; one link step produces the .so
        objs = _compile_objects(srcs, flags, verbose)
        tmp = out + ".tmp%d" % os.getpid()
        log = _run([_hipcc()] + objs + link + ["-o", tmp])
        if verbose:
            print(log)
        os.replace(tmp, out)
        open(out + ".stamp", "w").write(st)
    return out


def build_hosttest() -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    out = hosttest_lib_path()
    srcs = [os.path.join(CSRC, s) for s in HOSTTEST_SOURCES]
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC]
    st = _needs(out, srcs + _all_headers(), " ".join(flags))
    if st is None:
        return out
    # the gloo workers of tests/test_dist_cpu.py import this at the same moment: one builds (temporary + atomic rename)
    import fcntl
    with open(os.path.join(LIBDIR, ".build_hosttest.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if _needs(out, srcs + _all_headers(), " ".join(flags)) is None:
            return out
        tmp = out + ".tmp%d" % os.getpid()
        _run(["g++"] + flags + srcs + ["-o", tmp])
        os.replace(tmp, out)
        open(out + ".stamp", "w").write(st)
    return out


def build_all(verbose=False):
    return {"product": build_product(verbose), "hosttest": build_hosttest()}


if __name__ == "__main__":
    print(build_all(verbose=True))
