#!/bin/bash
# copy the evidence of tools/r03_final.sh (gpurun_out/r03final/, scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
s=gpurun_out/r03final; d=profiles
cp $s/bench.json $d/r03_bench.json
cp $s/bench_prof.json $d/r03_bench_under_rocprof.json
cp $s/bench_kernel_stats.csv $d/r03_bench_kernel_stats.csv
cp $s/ngp_kernel_stats.csv $d/r03_ngp_kernel_stats.csv
cp $s/ngp.log $d/r03_ngp_bench.log
cp $s/traffic.json $d/r03_traffic.json
cp $s/bench_c1280.json $d/r03_bench_c1280.json
cp $s/c1280_kernel_stats.csv $d/r03_c1280_kernel_stats.csv
cp $s/bench_gpus2_one_device_gloo.json $d/r03_bench_gpus2_one_device_gloo.json
cp $s/bench_gpus3_one_device_gloo.json $d/r03_bench_gpus3_one_device_gloo.json
cp $s/bench_c1280_gpus2_one_device_gloo.json $d/r03_bench_c1280_gpus2_one_device_gloo.json
cp $s/ba_large.log $d/r03_ba_large_solve.log
[ -f gpurun_out/gputest.log ] && cp gpurun_out/gputest.log $d/r03_gpu_tests.log
ls -la $d | grep r03_
