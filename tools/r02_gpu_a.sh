# round 2, first GPU pass: parity suite, the product-pipeline bench, kernel trace of the bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02a; mkdir -p $o
timeout 600 python -m pytest tests -m gpu -q -x --timeout=500 > $o/pytest.log 2>&1; tail -15 $o/pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 1500 $o/bench.err; head -c 3000 $o/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
ls $o/bprof/* | head; f=$(ls $o/bprof/*/*kernel_stats.csv $o/bprof/*kernel_stats.csv 2>/dev/null | head -1); head -45 $f | cut -c1-150
