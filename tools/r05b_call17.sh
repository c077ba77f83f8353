#!/bin/bash
# accumulate pass: four records per lane and round in the sparse record loop against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/new.so
timeout 300 python -m pytest tests/test_ngp_gpu.py -q -m gpu -x 2>&1 | tail -2
for v in new prev new prev; do
  if [ $v = new ]; then cp /tmp/new.so $NEW; else cp tools/_bin/lib_prev.so $NEW; fi
  echo "$v: $(timeout 200 python tools/r05_accum_cold.py 2>/dev/null | tail -1 | cut -c1-190)"
done
cp /tmp/new.so $NEW
