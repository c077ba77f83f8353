# the N > 1 bench topology on ONE device over gloo (what tests/test_bench_pipeline_gpu.py launches): evidence lines for profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02n3; mkdir -p $o
export NS_BENCH_DIST_BACKEND=gloo NS_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 3; do
  timeout 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 20 --warmup 5 > $o/bench_gpus$n.log 2> $o/bench_gpus$n.err
  grep "^{" $o/bench_gpus$n.log | tail -1 > $o/bench_gpus$n.json; head -c 400 $o/bench_gpus$n.json; echo
done
timeout 300 python -m pytest tests -m gpu -q --timeout=100 -x 2>&1 | tail -3
