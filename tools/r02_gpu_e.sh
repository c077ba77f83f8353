# round 2: full check (tests, NeRF micro-bench, bench, kernel trace) -- every stage under a tight timeout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${NS_OUT:-r02k}; mkdir -p $o
timeout 240 python -m pytest tests -m gpu -q --timeout=100 -x > $o/pytest.log 2>&1; tail -15 $o/pytest.log
timeout 60 python tools/ngp_bench.py 200 300 > $o/ngp_sphere.log 2>&1; tail -3 $o/ngp_sphere.log
NS_NGP_EXTRINSICS=1 timeout 60 python tools/ngp_bench.py 200 300 > $o/ngp_sphere_extr.log 2>&1; tail -3 $o/ngp_sphere_extr.log
timeout 150 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 1000 $o/bench.err; head -c 600 $o/bench.json
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
head -32 $o/bprof/b_kernel_stats.csv | cut -c1-150
