#!/bin/bash
# round 5, second session, call 2: where in a chunk the next chunk's loads are issued / its slab is written (conv.hip CV_WTAPS, CV_EARLY_STORE)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b2; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/base.so
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
for rep in 1 2; do
for v in base wtaps3 wtaps1 wtaps9e1 wtaps3e1 wtaps1e1; do
  if [ $v = base ]; then cp /tmp/base.so $NEW; else cp tools/_bin/lib_$v.so $NEW; fi
  echo "== $v (rep $rep)"
  NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip
done
done 2>&1 | tee $o/conv_variants.txt
for v in wtaps3 wtaps1 wtaps3e1 wtaps1e1; do
  cp tools/_bin/lib_$v.so $NEW
  echo "== tests $v"; timeout 300 python -m pytest tests/test_conv_gpu.py tests/test_update_op_gpu.py tests/test_encoder_gpu.py -q -m gpu 2>&1 | tail -2
done 2>&1 | tee $o/conv_variant_tests.txt
cp /tmp/base.so $NEW
