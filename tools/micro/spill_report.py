"""Where do the register spills of the two control-step kernels sit?  Compiles kinpoly_amd/csrc to gfx950 assembly and counts scratch_load / scratch_store
instructions per phase of the substep loop (the phases are delimited by the s_memtime reads of the KP_PROFILE markers).  The object kernel needs ~200
spilled VGPRs (fp64 MPR); they are harmless in the prologue and in the collision pass, and cost 8 % when the allocator moves some into the
articulated-body loops (stable-PD / smooth phases) -- check this after touching register-heavy code.     python tools/micro/spill_report.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kinpoly_amd.build import OPT_FLAGS  # noqa: E402  (the product build's optimisation flags)
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *OPT_FLAGS, "-std=c++17", "--cuda-device-only", "-S", "-Wno-unused-value",
                           os.path.join(ROOT, "kinpoly_amd", "csrc", "kp_sim.hip"), "-o", asm])
    text = open(asm).read().split("\n")
starts = {i: m.group(1) for i, l in enumerate(text) if (m := re.match(r"^(_ZN2kp\w+):", l))}
order = sorted(starts)
names = ["load + forward pass", "prologue", "stable-PD", "kinematics", "collision", "constraints", "smooth solve", "Newton", "integrate"]
for k, a in enumerate(order):
    if "kp_step_queue_kernel" not in starts[a]:
        continue
    b = order[k + 1] if k + 1 < len(order) else len(text)
    body = text[a:b]
    marks = [i for i, l in enumerate(body) if "s_memtime" in l]
    per = [(names[j], sum("scratch_" in l for l in body[marks[j]:marks[j + 1]])) for j in range(min(9, len(marks) - 1))]
    print(("object kernel " if "ILb1E" in starts[a] else "floor kernel  ") + f"({sum('scratch_' in l for l in body)} scratch instructions):", ", ".join(f"{n} {c}" for n, c in per))
