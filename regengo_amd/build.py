"""In-tree builds (no pip, no JIT cache): hipcc for the product library, g++/gcc for the test-only
host walker and the oracle's C restatement.  Everything lands next to the sources so that a `gpurun`
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
ORACLE = os.path.join(ROOT, "oracle")

PRODUCT_SOURCES = ["rgx_syntax.cc", "rgx_dfa.cc", "rgx_program.cc", "rgx_kernels.hip", "rgx_scan_exact.hip", "rgx_scan_sa.hip", "rgx_scan_us.hip", "rgx_replace.hip", "rgx_capi.cc"]
HOSTTEST_SOURCES = ["rgx_syntax.cc", "rgx_dfa.cc", "hosttest/rgx_hosttest.cc"]


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


def build_product(verbose=False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    out = product_lib_path()
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SOURCES]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC, "-Wno-unused-result", "-fvisibility=hidden", "-DRGX_BUILDING"]
    flags += os.environ.get("RGX_EXTRA_FLAGS", "").split()      # experiments only (e.g. -DRGX_US_PROFILE); part of the build stamp
    st = _needs(out, srcs + _all_headers(), " ".join(flags))
    if st is None:
        return out
    # several ranks of one job may get here at once: one builds (into a temporary, then an atomic rename), the rest wait
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if _needs(out, srcs + _all_headers(), " ".join(flags)) is None:
            return out
        # hipcc treats .cc as host C++ and .hip as HIP; one link step produces the .so
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [_hipcc()] + flags + srcs + ["-o", tmp]
        log = _run(cmd)
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


BENCH_PATTERNS = [
    r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})",      # BASELINE configs C1/C2
    r"(?P<user>\w+)@(?P<domain>\w+)",                          # C3
]


def build_oracle() -> str:
    """The oracle's C restatement (checker / cpu_baseline only; never linked into the product): pre-generate and
    compile the pattern-specialised matchers bench.py and the GPU tests need, so they exist on the GPU box."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import gen_c
    for p in BENCH_PATTERNS:
        gen_c.CMatcher(p)
    return os.path.join(ORACLE, "_build")


def build_all(verbose=False):
    return {"product": build_product(verbose), "hosttest": build_hosttest(), "oracle": build_oracle()}


if __name__ == "__main__":
    print(build_all(verbose=True))
