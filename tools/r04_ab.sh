#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ngp_gpu.py tests/test_rccl_gpu.py -x -q -m gpu -k "optimiser_step or converges or paired or renders or pose_refinement or mlp_forward or pyngp or rccl or nerf_fusion" 2>&1 | tail -2
for i in 1 2; do
for v in "" "NS_NGP_MLP_STEP_UNFUSED=1"; do
  echo "$v: $(env $v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1 | cut -c1-40) | $(env $v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'], round(d['sequential']['frames_per_s'],1))")"
done; done
