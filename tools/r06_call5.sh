#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06c5; mkdir -p $o
timeout 1500 python -m pytest tests/test_ngp_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 50 c640 > /dev/null 2>&1
grep "^\"ba_\|^\"void ba_" $o/prof/ba_kernel_stats.csv | grep -v solve_depth | cut -c1-120; rm -rf $o/prof
