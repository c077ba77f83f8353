#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/r04_bwd_ab.py 0.9 rays 2e-6 2>&1 | grep "gradient sigma\|bit-ident" | cut -c1-170
python tools/r04_bwd_ab.py 0.9 rays 1e-3 2>&1 | grep "gradient sigma\|bit-ident" | cut -c1-170
python tools/r04_bwd_ab.py 0.9 rays 2.0 2>&1 | grep "gradient sigma\|bit-ident" | cut -c1-170
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "fused or backward or converged" 2>&1 | tail -2
NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), [round(w['frames_per_s'],1) for w in d['windows']], d['breakdown']['ms_per_frame_by_leg'])"; done
