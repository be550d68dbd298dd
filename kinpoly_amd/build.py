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
HEADERS = ["kp_model.hpp", "kp_device.hpp", "kp_step_kernel.hpp", "kp_obs_kernels.hpp", "kp_rollout_kernels.hpp",
           "../../include/kinpoly_sim.h"]


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


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("KP_HIPCC_FLAGS", "").split()
    cmd = [_hipcc(), "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", *extra,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
