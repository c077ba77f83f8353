#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "plain smooth: $(python bench.py --microbench altcorr_smooth --reps 30 2>/dev/null | tail -1 | cut -c1-150)"
for d in 0 1 2 3 4; do
  echo "enc dbg $d: $(NS_ALT_DBG=$d python bench.py --microbench altcorr_enc --reps 30 2>/dev/null | tail -1 | cut -c60-130)"
done
