#!/bin/bash
# Round-3 evidence on the final tree (one gpurun call): bench line, kernel tables, PMC passes over bench.py's OWN micro-benches,
# config #5, the one-device records of the N > 1 topologies.  Everything lands in gpurun_out/r03final/ and is copied to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03final; mkdir -p $o
REPS=20
timeout 600 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_prof.json 2> /dev/null
cp $o/bprof/b_kernel_stats.csv $o/bench_kernel_stats.csv 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- env NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 100 300 > $o/ngp.log 2>&1; grep steps/s $o/ngp.log
cp $o/ngp/ngp_kernel_stats.csv $o/ngp_kernel_stats.csv 2>/dev/null
for name in ngp_bwd ngp_fwd lookup volume conv; do
  case $name in ngp_bwd) mb="ngp_encode_bwd";; ngp_fwd) mb="ngp_encode_fwd";; lookup) mb="corr_lookup";; volume) mb="corr_volume";; conv) mb="conv_nhwc";; esac
  d=$o/pmc/$name; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace -f csv -d $d/trace -o t -- python bench.py --microbench $mb --reps $REPS > $d/trace.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $d/fetch -o f -- python bench.py --microbench $mb --reps $REPS > $d/fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $d/write -o w -- python bench.py --microbench $mb --reps $REPS > $d/write.log 2>&1
  tail -1 $d/trace.log
done
python tools/r03_traffic.py $o/pmc $o/bench_kernel_stats.csv $REPS $o/traffic.json
# config #5
timeout 300 python bench.py --config c1280 --steps 2 --warmup 1 > $o/bench_c1280.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/c1280 -o c -- python bench.py --config c1280 --steps 2 --warmup 1 > /dev/null 2>&1
cp $o/c1280/c_kernel_stats.csv $o/c1280_kernel_stats.csv 2>/dev/null
# N > 1 topologies on the ONE device over gloo (functional records)
for n in 2 3; do
  NS_BENCH_DIST_BACKEND=gloo NS_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 12 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $o/bench_gpus${n}_one_device_gloo.json
done
NS_BENCH_DIST_BACKEND=gloo NS_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29610 bench.py --gpus 2 --steps 2 --warmup 1 --config c1280 2>/dev/null | grep '^{' | tail -1 > $o/bench_c1280_gpus2_one_device_gloo.json
# large-system solve vs rocSOLVER at 6P = 1536 (what round 2 used)
timeout 200 python tools/ba_large_bench.py > $o/ba_large.log 2>&1; cat $o/ba_large.log
rm -rf $o/bprof $o/ngp $o/c1280; find $o/pmc -name "*agent_info.csv" -delete
ls -la $o; du -sh $o
python - <<PY
import json
d = json.load(open("$o/bench.json")); print(d["value"], [round(w["frames_per_s"], 1) for w in d["windows"]], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
PY
