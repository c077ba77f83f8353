#!/bin/bash
# encode forward with 24-bit index multiplies / 32-bit output indexing against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/new.so
timeout 600 python -m pytest tests/test_ngp_gpu.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2 3; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
    echo "$v: $(timeout 120 python bench.py --microbench ngp_encode_fwd --reps 40 2>/dev/null | grep '^{' | tail -1 | cut -c1-130)"
  done
done
for rep in 1 2; do
  for v in new prev; do
    if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
    echo "$v rep $rep: $(bash tools/bench_once.sh 2>&1 | grep total)"
  done
done
cp /tmp/new.so $NEW
