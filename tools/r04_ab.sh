#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in "" 1; do
  echo "NS_FB_SYNC=$s"
  NS_FB_SYNC=$s python tools/r04_bwd_ab.py 0.9 rays 2e-6 2>&1 | grep "gradient sigma\|bit-ident" | cut -c1-170
  NS_FB_SYNC=$s python tools/r04_bwd_ab.py 0.9 rays 2.0 2>&1 | grep "gradient sigma\|bit-ident" | cut -c1-170
  NS_FB_SYNC=$s NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 800 320 2>&1 | tail -2 | head -1
done
