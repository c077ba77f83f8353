#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" 1; do
  echo "== trainer NS_NGP_WGRAD_ON_MAIN=$v"; NS_NGP_WGRAD_ON_MAIN=$v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2
done
q='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), [round(w["frames_per_s"],1) for w in d["windows"]], d["breakdown"]["ms_per_frame_by_leg"], d["extra"]["quality"])'
for v in "" 1; do
  echo "== bench NS_NGP_WGRAD_ON_MAIN=$v"; NS_NGP_WGRAD_ON_MAIN=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --quality-only 2>/dev/null | python -c "$q"
done
for v in 0 1; do
  echo "== bench NS_NGP_GRID_DECAY_ALL=$v"; NS_NGP_GRID_DECAY_ALL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --quality-only 2>/dev/null | python -c "$q"
  echo "== sphere NS_NGP_GRID_DECAY_ALL=$v"; NS_NGP_GRID_DECAY_ALL=$v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -1
  NS_NGP_GRID_DECAY_ALL=$v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -1
done
