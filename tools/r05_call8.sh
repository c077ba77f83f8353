#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do echo "step: $(timeout 200 python tools/r05_step_ablation.py 320 base 2>/dev/null | tail -1)"; done
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench total %.1f median %.1f seq %.1f legs %s'%(d['value'],d['windows_frames_per_s']['median'],d['sequential']['frames_per_s'],d['breakdown']['ms_per_frame_by_leg']))"; done
timeout 900 python -m pytest tests/test_ngp_gpu.py tests/test_rccl_gpu.py tests/test_bench_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
