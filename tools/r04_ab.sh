#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NS_NGP_WGRAD=staged python tools/r04_wgrad_bench.py 2>&1 | tail -1
NS_NGP_WGRAD=tr python tools/r04_wgrad_bench.py 2>&1 | tail -2
timeout 300 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu -k "mlp" 2>&1 | tail -2
echo "== trainer, pose chain on side"; NS_NGP_POSE_ON_SIDE=1 NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2
echo "== trainer, pose chain on side2"; NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2
echo "== trainer, no pose"; python tools/ngp_bench.py 800 320 2>&1 | tail -2
bash tools/r04_trace_ngp.sh 2>&1 | head -45
