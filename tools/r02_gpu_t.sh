cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02t; mkdir -p $o
for i in 1 2 3 4; do timeout 200 python -m pytest tests/test_slam_gpu.py tests/test_bench_pipeline_gpu.py tests/test_frontend_gpu.py -m gpu -q --timeout=100 -x > $o/thr_$i.log 2>&1; echo "loop $i rc $?"; tail -1 $o/thr_$i.log | cut -c1-100; done
