#!/bin/bash
# timing ablations of the gate convolution's chunk loop (results of the ablated kernels are wrong by construction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b6; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/base.so
for v in base abl1 abl2 abl3 abl4 base; do
  if [ $v = base ]; then cp /tmp/base.so $NEW; else cp tools/_bin/lib_$v.so $NEW; fi
  echo "== $v"
  NS_CONV_BENCH_TORCH=0 timeout 120 python tools/conv_bench.py 2>&1 | grep hip | head -4
done 2>&1 | tee $o/conv_ablation.txt
cp /tmp/base.so $NEW
