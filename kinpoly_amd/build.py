"""In-tree native build: hipcc -> kinpoly_amd/libkinpoly_sim.so (gfx950 only).

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libkinpoly_sim.so")
SOURCES = ["kp_sim.hip"]
STAMP = LIB + ".flags"          # the flag list the library was built with (a change of flags rebuilds, like a change of a source)


def _dependencies():
    """every file the library is compiled from: csrc/*.hip|*.hpp and include/*.h (globbed, so that a new header is a dependency the day it appears)"""
    import glob
    inc = os.path.join(os.path.dirname(_HERE), "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(inc, "*.h")))


def kernel_source_sha256() -> str:
    """One fingerprint of what the device code is compiled from (csrc/*.hip|*.hpp + the flag list): profiles under profiles/ are stamped with it, and
    bench.py nulls PMC-derived figures whose stamp is not the current one (a profile of another kernel must not ride on a fresh driver record)."""
    import hashlib
    h = hashlib.sha256()
    for p in _dependencies():
        if p.startswith(CSRC):
            h.update(os.path.basename(p).encode()); h.update(open(p, "rb").read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()[:16]


# Optimisation flags of the product build (tools that compile instrumented variants of the library use the same list).  -O2 without the loop and SLP
# vectorisers: on the one-wavefront-per-env kernels their packed fp32 operations (v_pk_fma_f32 and friends need even-aligned register pairs) cost more
# in moves and register pressure than they save -- control-step launch 2.785 -> 2.695 ms (floor), 5.16 -> 5.03 ms (objects) against -O3 with both on
# (measured on the final kernels of round 3; -fno-unroll-loops on top loses 3 % on the object kernel).
OPT_FLAGS = ["-O2", "-fno-vectorize", "-fno-slp-vectorize"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libkinpoly_sim.so)")


def _flags():
    return [*OPT_FLAGS, *os.environ.get("KP_HIPCC_FLAGS", "").split()]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(p) > t for p in _dependencies()):
        return True
    try:
        return open(STAMP).read() != " ".join(_flags())
    except OSError:
        return True


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("KP_HIPCC_FLAGS", "").split()
    cmd = [_hipcc(), "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", *extra,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(" ".join(_flags()))
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
