#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NS_NGP_WGRAD=staged python tools/r04_wgrad_bench.py 2>&1 | tail -2
NS_NGP_WGRAD=tr python tools/r04_wgrad_bench.py 2>&1 | tail -3
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "mlp" 2>&1 | tail -5
for w in staged tr; do
  echo "== trainer, NS_NGP_WGRAD=$w"
  NS_NGP_WGRAD=$w NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2
done
