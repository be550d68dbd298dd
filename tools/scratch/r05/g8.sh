export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for nt in 64 128 256; do
timeout -s KILL 400 python bench.py --workload objects --threads-per-env $nt --steps 40 --warmup 15 --no-secondary --no-cpu-baseline --no-parity-live 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('objects NT=$nt value %.0f ms_per_step %.3f launch_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['launch_ms']), d['launch_balance'], d['roofline']['kernel'])"
done 2>&1 | tee gpurun_out/r05/objects_threads_per_env.log
