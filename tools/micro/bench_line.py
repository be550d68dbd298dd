"""one-line summary of a bench.py JSON line read from stdin:   python bench.py ... | python tools/micro/bench_line.py [tag]"""
import json, sys
lines = [ln for ln in sys.stdin.read().splitlines() if ln.startswith("{")]
r = json.loads(lines[-1])
tag = " ".join(sys.argv[1:])
print(tag, r["config"].get("workload_id"), "value", round(r["value"]), "ms/step %.3f" % r["ms_per_step"], "launch %.3f" % r.get("roofline", {}).get("launch_ms", float("nan")),
      {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.get("launch_balance", {}).items()}, "ended/step %.4f" % r.get("episodes_ended_per_step_frac", float("nan")), flush=True)
