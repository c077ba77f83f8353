#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_corr_gpu.py tests/test_parity_c640_gpu.py -x -q -m gpu -k "volume or pyramid or pool or build" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --microbench corr_volume --reps 50 2>/dev/null | tail -1 | cut -c1-400; done
