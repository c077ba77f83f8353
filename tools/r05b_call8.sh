#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b8; mkdir -p $o
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $o/gpu_tests.log; cat $o/gpu_tests.log
python - <<'PY'
import os, sys
sys.path.insert(0, "nerf-slam_amd")
maps = open("/proc/self/maps").read()
PY
