#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== product (two waves per SIMD, 8 MFMAs per tap and wave)"; NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip | head -4
echo "== one wave per SIMD, 16 MFMAs per tap and wave"; NS_VARIANTS=1 NS_CONV_CG=1 NS_CONV_UT=4 NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip | head -4
done
NS_VARIANTS=1 NS_CONV_CG=1 NS_CONV_UT=4 timeout 200 python -m pytest tests/test_update_op_gpu.py -q -m gpu 2>&1 | tail -2
