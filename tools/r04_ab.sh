#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" 1 "" 1; do
  echo "== NS_ENC_FWD_NT=$v"; NS_ENC_FWD_NT=$v NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1
  NS_ENC_FWD_NT=$v timeout 120 python bench.py --microbench ngp_encode_fwd --reps 20 2>&1 | tail -1 | cut -c1-120
done
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "converged or encode_forward" 2>&1 | tail -2
