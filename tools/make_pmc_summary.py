"""Fold the rocprofv3 passes of tools/profile_bench.sh into the JSON bench.py reads (profiles/r03/pmc_bench_<workload>.json) and copy
the kernel-stats CSV next to it.    python tools/make_pmc_summary.py <gpurun_out/r03_prof/WL> <workload> "<command>" """
import csv
import glob
import json
import os
import shutil
import sys

src, wl, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "gpurun_out", os.environ.get("KP_ROUND", "r05") + "_prof", "summary")
os.makedirs(dst, exist_ok=True)
KERNEL = "kp_step_queue_kernel"
N_SIMD, CLOCK_GHZ = 1024, 2.35          # 256 CUs x 4 SIMDs; shader clock under this kernel (tools/micro/queue_timeline.py)


def counters(tag):
    vals = {}
    for path in glob.glob(os.path.join(src, f"pmc_{tag}", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL in row.get("Kernel_Name", ""):
                    vals.setdefault(row["Counter_Name"], {}).setdefault(row.get("Dispatch_Id", "0"), 0.0)
                    vals[row["Counter_Name"]][row.get("Dispatch_Id", "0")] += float(row["Counter_Value"])
    return {k: sorted(v.values())[len(v) // 2] for k, v in vals.items() if v}      # median per launch (counter rows of one dispatch summed)


def launch_ms():
    for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(dst, os.environ.get("KP_ROUND", "r05") + f"_kernel_stats_{wl}.csv"))
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL in row.get("Name", ""):
                    return float(row["AverageNs"]) * 1e-6, int(row["Calls"])
    return None, 0


ms, calls = launch_ms()
fs, ws = counters("FETCH_SIZE"), counters("WRITE_SIZE")
iss = counters("SQ_INSTS_VALU+SQ_INSTS_SALU+SQ_INSTS_LDS")
cyc = counters("SQ_WAVE_CYCLES+SQ_BUSY_CYCLES+SQ_ACTIVE_INST_VALU")
sys.path.insert(0, ROOT)
from kinpoly_amd.build import kernel_source_sha256  # noqa: E402
out = {"kernel": KERNEL, "workload": wl, "command": cmd, "launch_ms": ms, "launches_in_stats_pass": calls, "kernel_source_sha256": kernel_source_sha256()}
if "FETCH_SIZE" in fs and "WRITE_SIZE" in ws:
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE counts 64 B per 128-B request: x2 (MI355X_MICROARCH.md, HBM section)
    out.update(FETCH_SIZE_KB_median=fs["FETCH_SIZE"], WRITE_SIZE_KB_median=ws["WRITE_SIZE"],
               hbm_bytes_per_launch=(2.0 * fs["FETCH_SIZE"] + ws["WRITE_SIZE"]) * 1024.0)
if "SQ_INSTS_VALU" in iss:
    out["issue"] = {"valu_insts_per_launch": iss.get("SQ_INSTS_VALU"), "salu_insts_per_launch": iss.get("SQ_INSTS_SALU"), "lds_insts_per_launch": iss.get("SQ_INSTS_LDS"),
                    "sq_wave_cycles_quad": cyc.get("SQ_WAVE_CYCLES"), "sq_busy_cycles_quad": cyc.get("SQ_BUSY_CYCLES"), "sq_active_inst_valu_quad": cyc.get("SQ_ACTIVE_INST_VALU")}
if "issue" in out and out["issue"].get("sq_active_inst_valu_quad") and ms:
    # SQ_ACTIVE_INST_VALU counts in units of 4 cycles summed over the SIMDs: fraction of the launch during which a SIMD issues VALU work
    out["issue"]["valu_active_frac_of_launch"] = out["issue"]["sq_active_inst_valu_quad"] * 4.0 / (N_SIMD * ms * 1e-3 * CLOCK_GHZ * 1e9)
json.dump(out, open(os.path.join(dst, f"pmc_bench_{wl}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
