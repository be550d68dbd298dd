export TMPDIR=/tmp
T="timeout -s KILL"
$T 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warn\|warn\|sched_\|^$" | tail -12
