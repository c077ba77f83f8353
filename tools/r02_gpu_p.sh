# round 2, pass p: encoders on the MFMA convolution + altcorr fix + c1280 line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${NS_OUT:-r02p}; mkdir -p $o
timeout 120 python -m pytest tests/test_encoder_gpu.py tests/test_corr_gpu.py -m gpu -q --timeout=100 -x > $o/pytest_new.log 2>&1; tail -15 $o/pytest_new.log
timeout 60 python tools/enc_bench.py > $o/enc_bench.log 2>&1; tail -2 $o/enc_bench.log
timeout 240 python -m pytest tests -m gpu -q --timeout=100 -x > $o/pytest.log 2>&1; tail -8 $o/pytest.log
timeout 150 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 600 $o/bench.err; head -c 700 $o/bench.json
timeout 200 python bench.py --config c1280 --steps 2 --warmup 1 > $o/c1280.json 2> $o/c1280.err; grep -v "Gloo\|^$" $o/c1280.err | tail -12 | cut -c1-200; head -c 1500 $o/c1280.json
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
head -24 $o/bprof/b_kernel_stats.csv | cut -c1-150
