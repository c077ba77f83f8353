#!/bin/bash
# scatter pass without scratch memory (the runs' cells in registers) against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b14; mkdir -p $o
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/new.so
timeout 600 python -m pytest tests/test_ngp_gpu.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
    echo "$v: $(timeout 200 python tools/r05_accum_cold.py 2>/dev/null | tail -1 | cut -c1-200)"
  done
done 2>&1 | tee $o/scatter_scratch.txt
for rep in 1 2 3; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
    echo "$v rep $rep: $(bash tools/bench_once.sh 2>&1 | grep total)"
  done
done 2>&1 | tee $o/bench_ab.txt
cp /tmp/new.so $NEW
