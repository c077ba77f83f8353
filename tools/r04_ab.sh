#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for st in 0 1 2 3 4 0; do NS_FB_STAGE=$st python tools/r04_bwd_ab.py 0.9 rays 2e-6 2>&1 | grep "gradient sigma" | sed "s/^/STAGE=$st /"; done
for st in 0 1 2 3 4; do NS_FB_STAGE=$st python tools/r04_bwd_ab.py 0.9 rays 1e-3 2>&1 | grep "gradient sigma" | sed "s/^/STAGE=$st dense /"; done
