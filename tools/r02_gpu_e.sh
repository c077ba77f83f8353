# round 2, pass e: parallel_run mode, dense LDS w/o scan, 2-stage camera gradient
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02i; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $o/pytest.log 2>&1; tail -25 $o/pytest.log
timeout 120 python tools/ngp_bench.py 200 300 > $o/ngp_sphere.log 2>&1; tail -3 $o/ngp_sphere.log
NS_NGP_EXTRINSICS=1 timeout 120 python tools/ngp_bench.py 200 300 > $o/ngp_sphere_extr.log 2>&1; tail -3 $o/ngp_sphere_extr.log
timeout 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 1500 $o/bench.err; head -c 1200 $o/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
head -32 $o/bprof/b_kernel_stats.csv | cut -c1-150
