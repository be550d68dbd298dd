#!/bin/bash
# objects workload as S sub-batches on S streams, with the queue kernel's slots split between them
export TMPDIR=/tmp; E=gpurun_out/r05; mkdir -p $E; O=$E/pipeline_objects_slots.log; : > $O
T="timeout -s KILL 150"
echo "== default slots" >> $O; $T python tools/micro/pipeline_objects.py objects 1,2 2>&1 | grep "S=" >> $O; echo rc $? >> $O
for q in 896 1024 1280; do echo "== KP_QUEUE_SLOTS=$q" >> $O; KP_QUEUE_SLOTS=$q $T python tools/micro/pipeline_objects.py objects 2 2>&1 | grep "S=" >> $O; echo rc $? >> $O; done
echo "== KP_QUEUE_SLOTS=640 S=3 (3 x 1024 envs)" >> $O; KP_PIPE_N=3072 KP_QUEUE_SLOTS=640 $T python tools/micro/pipeline_objects.py objects 3 2>&1 | grep "S=" >> $O; echo rc $? >> $O
echo "== KP_QUEUE_SLOTS=448 S=4" >> $O; KP_QUEUE_SLOTS=448 $T python tools/micro/pipeline_objects.py objects 4 2>&1 | grep "S=" >> $O; echo rc $? >> $O
cat $O
