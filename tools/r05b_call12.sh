#!/bin/bash
# gate convolution: how many of the next tap's fragment reads are issued behind each MFMA of the current tap (CV_DSPM)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b12; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/base.so
for rep in 1 2; do
for v in base ds3 ds6 ds1; do
  if [ $v = base ]; then cp /tmp/base.so $NEW; else cp tools/_bin/lib_$v.so $NEW; fi
  echo "== $v (rep $rep)"
  NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip | head -5
done
done 2>&1 | tee $o/conv_dspm.txt
cp /tmp/base.so $NEW
