#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NS_NGP_WGRAD=tr1 python tools/r04_wgrad_bench.py 2>&1 | tail -1
NS_NGP_WGRAD=tr2 python tools/r04_wgrad_bench.py 2>&1 | tail -1
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "mlp or converges or paired" 2>&1 | tail -2
for w in tr1 tr2; do for r in 1 ""; do
  echo "== trainer NS_NGP_WGRAD=$w NS_NGP_RAYS_ON_SIDE=$r"; NS_NGP_WGRAD=$w NS_NGP_RAYS_ON_SIDE=$r NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1
done; done
for w in tr1 tr2; do for r in 1 ""; do
  echo "== bench NS_NGP_WGRAD=$w NS_NGP_RAYS_ON_SIDE=$r"; NS_NGP_WGRAD=$w NS_NGP_RAYS_ON_SIDE=$r python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'])"
done; done
