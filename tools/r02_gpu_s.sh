# round 2, pass s: MLP kernels with scalar row bases (occupancy 2 -> 4 waves/SIMD)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${NS_OUT:-r02s}; mkdir -p $o
timeout 200 python -m pytest tests/test_ngp_gpu.py -m gpu -q --timeout=100 -x > $o/pytest_ngp.log 2>&1; tail -4 $o/pytest_ngp.log
timeout 60 python tools/ngp_bench.py 200 300 2>&1 | grep "steps/s\|PSNR"
NS_NGP_EXTRINSICS=1 timeout 60 python tools/ngp_bench.py 200 300 2>&1 | grep "steps/s\|PSNR"
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- python tools/ngp_bench.py 100 300 > $o/ngp.log 2>&1; head -14 $o/ngp/ngp_kernel_stats.csv | cut -c1-120
timeout 150 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err; head -c 300 $o/bench.json; echo
