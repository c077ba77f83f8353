#!/bin/bash
# timing ablations of the accumulate pass (results wrong by construction): 1 = no records, no flush; 2 = records, no flush; 3 = flush only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NEW=nerf-slam_amd/lib/libnerfslam_hip.so; cp $NEW /tmp/base.so
for v in base fabl1 fabl2 fabl3 base; do
  if [ $v = base ]; then cp /tmp/base.so $NEW; else cp tools/_bin/lib_$v.so $NEW; fi
  echo "$v: $(timeout 200 python tools/r05_accum_cold.py 2>/dev/null | tail -1 | cut -c1-190)"
done
cp /tmp/base.so $NEW
