#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_corr_gpu.py tests/test_update_op_gpu.py tests/test_slam_gpu.py tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in 1 ""; do
  echo "== tracker alone NS_LOOKUP_UNFUSED=$v"; NS_LOOKUP_UNFUSED=$v python tools/track_prof.py 60 2>&1 | tail -2
  echo "== bench NS_LOOKUP_UNFUSED=$v"; NS_LOOKUP_UNFUSED=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --quality-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'], d['extra']['quality']['ate_rmse_scene_units'])"
done
timeout 300 python bench.py --config c1280 --steps 2 --warmup 1 2>gpurun_out/c1280.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['cpu_baseline'], d['breakdown'])"; tail -3 gpurun_out/c1280.err
