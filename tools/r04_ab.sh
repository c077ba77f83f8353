#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "optimiser_step or converges or paired" 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/bprof -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
grep -E "ngp_mlp_step_kernel|ngp_mlp_wgrad_tr|faccum|fscatter" /tmp/bprof/b_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
for i in 1 2 3; do
  echo "$(NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1 | cut -c1-40) | $(python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'], round(d['sequential']['frames_per_s'],1))")"
done
