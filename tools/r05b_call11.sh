#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b11; rm -rf $o; mkdir -p $o
timeout 400 rocprofv3 --kernel-trace -f csv -d $o/tr -o t -- python bench.py --steps 20 --warmup 5 --windows 3 --no-extras --no-cpu-baseline > $o/bench.json 2> $o/err.txt
f=$(find $o/tr -name '*kernel_trace.csv' | head -1); ls -la $f
python tools/timeline_overlap.py $f | tee $o/overlap.json
python -c "
import json; d=json.load(open('$o/bench.json')); print(d['value'], d['breakdown']['ms_per_frame_by_leg'])"
rm -rf $o/tr
