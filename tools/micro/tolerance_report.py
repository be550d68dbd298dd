"""Slack of the test tolerances: reads the file written by a `KP_TOL_REPORT=<file> python -m pytest tests -m gpu` run (tests/conftest.py) and prints, per
assert_allclose site, the tolerance in force (atol + rtol x max |desired|) against the largest error any call at that site measured."""
import collections
import sys

rows = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for line in open(sys.argv[1]):
    where, atol, rtol, err, mag = line.rstrip("\n").split("\t")
    r = rows[where]
    r[0] = max(r[0], float(err)); r[1] = float(atol) + float(rtol) * float(mag); r[2] = float(atol); r[3] += 1
for where, (err, tol, atol, n) in sorted(rows.items(), key=lambda kv: -(kv[1][1] / max(kv[1][0], 1e-300))):
    print(f"{where:34s} calls {n:5d}  tolerance {tol:9.2e} (atol {atol:g})  measured {err:9.2e}  slack x{tol / max(err, 1e-300):9.1f}")
