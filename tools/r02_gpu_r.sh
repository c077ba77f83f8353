# round 2, pass r: run-length dense backward, small-launch conv tiles, eager encoders, c1280 gauge
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${NS_OUT:-r02r}; mkdir -p $o
timeout 200 python -m pytest tests/test_ngp_gpu.py tests/test_conv_gpu.py tests/test_encoder_gpu.py tests/test_slam_gpu.py -m gpu -q --timeout=100 -x > $o/pytest_new.log 2>&1; tail -6 $o/pytest_new.log
for v in "" "NS_ENC_BWD_NO_RL=1" "NS_ENC_RL_PARTS=16,8" "NS_ENC_RL_PARTS=32,16" "NS_ENC_RL_PARTS=48,12"; do
  echo "== $v"; env $v timeout 60 python tools/ngp_bench.py 200 300 2>&1 | grep steps/s
done
timeout 60 python tools/enc_bench.py > $o/enc_bench.log 2>&1; tail -1 $o/enc_bench.log
timeout 240 python -m pytest tests -m gpu -q --timeout=100 -x --deselect tests/test_ngp_gpu.py --deselect tests/test_conv_gpu.py --deselect tests/test_encoder_gpu.py --deselect tests/test_slam_gpu.py > $o/pytest.log 2>&1; tail -4 $o/pytest.log
timeout 150 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 400 $o/bench.err; head -c 500 $o/bench.json; echo
timeout 200 python bench.py --config c1280 --steps 2 --warmup 1 > $o/c1280.json 2> $o/c1280.err; grep -v "Gloo\|^$" $o/c1280.err | tail -5 | cut -c1-200; head -c 300 $o/c1280.json; echo
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
head -30 $o/bprof/b_kernel_stats.csv | cut -c1-150
