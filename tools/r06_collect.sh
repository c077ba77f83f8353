#!/bin/bash
# copy the evidence of tools/r06_final.sh (gpurun_out/r06final/, scratch) into profiles/ (tracked)
s=gpurun_out/r06final; d=profiles
for f in bench.json bench_kernel_stats.csv bench_under_rocprof.json bench_c1280.json bench_c1280_gpus2_one_device_gloo.json \
         bench_gpus2_one_device_gloo.json bench_gpus3_one_device_gloo.json c1280_kernel_stats.csv gpu_tests.log ngp_kernel_stats.csv \
         rccl_one_rank.json traffic.json ba_traffic.json ba_c640.json ba_c640_kernel_stats.csv ba_c640_round5_kernels_stats.csv \
         ba_pmc_summary.txt bench_run2.json bench_run3.json; do
  [ -f $s/$f ] && cp $s/$f $d/r06_$f
done
grep -v amdgpu.ids $s/ngp.log > $d/r06_ngp_bench.log 2>/dev/null
ls -la $d/r06_*
