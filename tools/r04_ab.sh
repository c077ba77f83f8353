#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "" "NS_CONV_CG=1" "NS_CONV_UT=2"; do
  echo "$v: $(env $v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'], round(d['sequential']['frames_per_s'],1))")"
done; done
