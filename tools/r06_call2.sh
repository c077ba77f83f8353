#!/bin/bash
# round 6, call 2: the fused lineariser + Gram Schur kernel: BA parity tests, then timing at c1280 scale and at C640
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06c2; mkdir -p $o
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_parity_c640_gpu.py tests/test_parity_c1280_gpu.py tests/test_parity_c1280_full_gpu.py tests/test_parallel_ba.py -x -q -m gpu 2>&1 | tail -25 > $o/ba_tests.log; tail -8 $o/ba_tests.log
timeout 300 python tools/ba_c1280_bench.py 10 > $o/ba_c1280_new.json 2> $o/ba_c1280_new.err; cat $o/ba_c1280_new.json; tail -3 $o/ba_c1280_new.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 10 > /dev/null 2>&1
cp $o/prof/ba_kernel_stats.csv $o/ba_c1280_new_kernel_stats.csv 2>/dev/null; rm -rf $o/prof
grep "^\"ba_\|^\"void ba_" $o/ba_c1280_new_kernel_stats.csv | cut -c1-160
timeout 300 python tools/ba_c1280_bench.py 50 c640 > $o/ba_c640_new.json 2>/dev/null; cat $o/ba_c640_new.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 50 c640 > /dev/null 2>&1
grep "^\"ba_\|^\"void ba_" $o/prof/ba_kernel_stats.csv | cut -c1-160; rm -rf $o/prof
NS_VARIANTS=1 NS_BA_UNFUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ba -- python tools/ba_c1280_bench.py 50 c640 > $o/ba_c640_old_variants.json 2>/dev/null
grep "^\"ba_\|^\"void ba_" $o/prof/ba_kernel_stats.csv | cut -c1-160; rm -rf $o/prof
