#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_corr_gpu.py -x -q -m gpu -k "altcorr" 2>&1 | tail -3
run() { python bench.py --config c1280 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), 'ms/pass; altcorr launch us', round(d['roofline']['avg_launch_us'],1), d['breakdown']['ms_per_pass_by_leg'], d['config']['state_checksums']['poses'])"; }
echo "fused LDS-staged (default):"; run
echo "fused direct (old):"; NS_ALTCORR_DIRECT=1 run
echo "plain LDS-staged:"; NS_LOOKUP_UNFUSED=1 run
