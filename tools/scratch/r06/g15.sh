#!/bin/bash
# round 6: does the lean layout's spare LDS let sub-batched env-steps on streams overlap the policy GEMMs with the other sub-batch's physics?  (random-init rollout)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
{
KP_PIPE_S=1 timeout -s KILL 240 python tools/pipeline_probe.py 2>&1 | grep "^S="
for slots in 1536 1280 1024 2048; do echo "queue_slots per sub-batch $slots"; KP_PIPE_S=2 KP_QUEUE_SLOTS=$slots timeout -s KILL 240 python tools/pipeline_probe.py 2>&1 | grep "^S="; done
for slots in 768 640; do echo "queue_slots per sub-batch $slots"; KP_PIPE_S=4 KP_QUEUE_SLOTS=$slots timeout -s KILL 240 python tools/pipeline_probe.py 2>&1 | grep "^S="; done
echo "hipGraph per sub-batch"; KP_PIPE_GRAPH=1 KP_PIPE_S=2 KP_QUEUE_SLOTS=1536 timeout -s KILL 240 python tools/pipeline_probe.py 2>&1 | grep "^S="
} 2>&1 | tee $O/pipeline_streams_lean.log
