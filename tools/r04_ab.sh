#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in "" "NS_NGP_WGRAD_WGS=80" "NS_NGP_WGRAD_WGS=96" "NS_NGP_WGRAD_WGS=128"; do
  echo "$v: $(env $v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1 | cut -c1-40) | $(env $v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], round(d['sequential']['frames_per_s'],1))")"
done; done
