#!/bin/bash
set -u
export TMPDIR=/tmp
E=gpurun_out/r04_b
mkdir -p $E
timeout -s KILL 900 python tools/launches_per_step.py objects $E/lps_objects > $E/launches_objects.csv 2> $E/launches_objects.err
timeout -s KILL 900 python tools/launches_per_step.py tracked $E/lps_tracked > $E/launches_tracked.csv 2> $E/launches_tracked.err
head -60 $E/launches_objects.csv | cut -c1-160
head -40 $E/launches_tracked.csv | cut -c1-160
