#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_corr_gpu.py -x -q -m gpu -k "altcorr" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --microbench altcorr --reps 30 2>/dev/null | tail -1 | cut -c60-120
  NS_ALTCORR_DIRECT=1 python bench.py --microbench altcorr --reps 30 2>/dev/null | tail -1 | cut -c60-120
done
