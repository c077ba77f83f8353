#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_corr_gpu.py tests/test_ngp_gpu.py tests/test_bench_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -6
for v in 1 ""; do
  echo "== c1280 NS_LOOKUP_UNFUSED=$v"; NS_LOOKUP_UNFUSED=$v timeout 300 python bench.py --config c1280 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],3), d['config']['keyframe_centre_rmse_before_after'], d['breakdown']['ms_per_pass_by_leg'])"
done
