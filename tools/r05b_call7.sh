#!/bin/bash
# convolution: source table in scalar registers (no per-chunk scalar loads from the argument segment) against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b7; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/new.so
for rep in 1 2; do
for v in new prev; do
  if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
  echo "== $v (rep $rep)"
  NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip
  timeout 120 python tools/small_conv_bench.py 2>&1 | grep " us " 
done
done 2>&1 | tee $o/conv_scalar_table.txt
cp /tmp/new.so $NEW
timeout 300 python -m pytest tests/test_conv_gpu.py tests/test_update_op_gpu.py tests/test_encoder_gpu.py tests/test_frontend_gpu.py -q -m gpu 2>&1 | tail -2
