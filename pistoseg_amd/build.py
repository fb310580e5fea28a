"""Builds libpistoseg_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object loaded with ctypes (pistoseg_amd/_lib.py)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpistoseg_hip.so")
DEBUG_LIB = os.path.join(HERE, "libpistoseg_hip_debug.so")
SOURCES = ["api.cpp", "conv_igemm.hip", "conv_wgrad.hip", "small_ops.hip", "pixel_ops.hip", "rfm_ops.hip", "sliding_ops.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_one(lib: str, objdir: str, extra_flags, force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    inc = os.path.join(HERE, "..", "include")
    headers = [os.path.join(CSRC, "ps_internal.h"), os.path.join(inc, "pistoseg_hip.h"), os.path.join(inc, "pistoseg_hip_debug.h")]
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc, *FLAGS, *extra_flags, "-x", "hip", "-c", src, "-o", obj]
            jobs.append((s, cmd))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for name, r in ex.map(run, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    sys.stderr.write(f"[build] {name}:\n{r.stderr}\n")
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {name}")
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in srcs]
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
    return lib


def build(force: bool = False, verbose: bool = True, debug: bool = True) -> str:
    """libpistoseg_hip.so = the product (no mutable tunables, no `ps_debug_*` symbols); libpistoseg_hip_debug.so = the same sources with
    -DPS_DEBUG_HOOKS for the parity suite's variant sweeps and the tools' ablations.  Returns the product library's path."""
    path = _build_one(LIB, os.path.join(HERE, "build"), [], force, verbose)
    if debug:
        _build_one(DEBUG_LIB, os.path.join(HERE, "build", "debug"), ["-DPS_DEBUG_HOOKS"], force, verbose)
    return path


def build_variant(tag: str, defines, verbose: bool = True, product: bool = False) -> str:
    """An A/B build with extra -D defines.  Debug library (default): pistoseg_amd/libpistoseg_hip_debug_<tag>.so, load it with
    PISTOSEG_HIP_DEBUG_LIB=<path>.  product=True: the product library's flags, pistoseg_amd/libpistoseg_hip_<tag>.so, load it with
    PISTOSEG_HIP_LIB=<path> (whole-step A/Bs through bench.py).  Optimisation harness only."""
    extra = [d if d.startswith("-") else f"-D{d}" for d in defines]  # NAME=VALUE -> -DNAME=VALUE; anything starting with '-' is a raw compiler flag
    if product:
        return _build_one(os.path.join(HERE, f"libpistoseg_hip_{tag}.so"), os.path.join(HERE, "build", "abp_" + tag), extra, False, verbose)
    lib = os.path.join(HERE, f"libpistoseg_hip_debug_{tag}.so")
    return _build_one(lib, os.path.join(HERE, "build", "ab_" + tag), ["-DPS_DEBUG_HOOKS", *extra], False, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
