#!/bin/bash
# round 6, call 1: the new parity / quality tests on the round-5 kernels + the baseline per-kernel table of the c1280-scale BA
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06c1; mkdir -p $o
timeout 900 python -m pytest tests/test_parity_c1280_full_gpu.py -x -q -m gpu 2>&1 | tail -15 > $o/full_scale.log; tail -5 $o/full_scale.log
timeout 600 python -m pytest tests/test_ngp_gpu.py -q -m gpu -s -k "fixed_point_table_gradient_trains" 2>&1 | grep -E "Q18_VS_F32|passed|failed|Error|assert" | tail -12 > $o/q18.log; cat $o/q18.log
timeout 300 python tools/ba_c1280_bench.py 10 > $o/ba_c1280_old.json 2> $o/ba_c1280_old.err; cat $o/ba_c1280_old.json; tail -3 $o/ba_c1280_old.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 10 > /dev/null 2>&1
cp $o/prof/ba_kernel_stats.csv $o/ba_c1280_old_kernel_stats.csv 2>/dev/null; rm -rf $o/prof
head -12 $o/ba_c1280_old_kernel_stats.csv
timeout 300 python tools/ba_c1280_bench.py 50 c640 > $o/ba_c640_old.json 2>/dev/null; cat $o/ba_c640_old.json
