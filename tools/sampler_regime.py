"""bench.py's `sampler_tracking_regime` block alone, optionally with a torch profiler table of one run.
    python tools/sampler_regime.py [--profile] [--calls 4] [--amp 0.05]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--profile", action="store_true"); ap.add_argument("--calls", type=int, default=4); ap.add_argument("--amp", type=float, default=0.05); ap.add_argument("--pool_depth", type=int, default=4); ap.add_argument("--blocking", action="store_true")
a = ap.parse_args()
r = bench.sampler_regime(0, 4, calls=a.calls, amp=a.amp, profile=a.profile, pool_depth=a.pool_depth, lagged=not a.blocking)
tab = r.pop("_profile_table", None)
print(json.dumps(r))
if tab:
    print(tab)
