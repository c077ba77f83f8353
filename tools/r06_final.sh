#!/bin/bash
# Round-6 evidence on ONE tree, in ONE gpurun call (the only per-round script: tools/r06_collect.sh copies what is to be judged
# into profiles/).  Order: the GPU suite; the kernel table of the product pipeline; PMC passes over bench.py's OWN micro-benches
# (the launches `avg_launch_us` times) -> traffic.json; PMC passes over the dense BA at config #5's scale (tools/ba_pmc.py);
# the one-rank RCCL record; THEN the bench lines that read those files (c640 headline x3, config #5 + its kernel table, the
# N > 1 topologies on the one device over gloo).  Everything lands in gpurun_out/r06final/.
# usage (from the container): H=$(git rev-parse HEAD); gpurun --timeout 3000 -- "NS_GIT_HEAD=$H bash tools/r06_final.sh"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06final; mkdir -p $o
REPS=20
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $o/gpu_tests.log; tail -2 $o/gpu_tests.log
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_under_rocprof.json 2> /dev/null
cp $o/bprof/b_kernel_stats.csv $o/bench_kernel_stats.csv 2>/dev/null
for name in ngp_bwd ngp_fwd mlp_fwd mlp_bwd mlp_wgrad lookup lookup_enc volume conv altcorr altcorr_enc; do
  case $name in ngp_bwd) mb="ngp_encode_bwd";; ngp_fwd) mb="ngp_encode_fwd";; mlp_fwd) mb="ngp_mlp_fwd";; mlp_bwd) mb="ngp_mlp_bwd";; mlp_wgrad) mb="ngp_mlp_wgrad";; lookup) mb="corr_lookup_coop";; lookup_enc) mb="corr_lookup_enc";; volume) mb="corr_volume";; conv) mb="conv_nhwc";; altcorr) mb="altcorr";; altcorr_enc) mb="altcorr_enc";; esac
  d=$o/pmc/$name; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace -f csv -d $d/trace -o t -- python bench.py --microbench $mb --reps $REPS > $d/trace.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $d/fetch -o f -- python bench.py --microbench $mb --reps $REPS > $d/fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d $d/write -o w -- python bench.py --microbench $mb --reps $REPS > $d/write.log 2>&1
  case $name in mlp_*|conv) timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -f csv -d $d/mfma -o m -- python bench.py --microbench $mb --reps $REPS > $d/mfma.log 2>&1;; esac
  grep '^{' $d/trace.log | tail -1 | cut -c1-200
done
python tools/traffic.py $o/pmc $o/bench_kernel_stats.csv $REPS $o/traffic.json > $o/traffic_summary.txt 2>&1; tail -15 $o/traffic_summary.txt
cp $o/traffic.json profiles/r06_traffic.json
# the dense BA at config #5's scale and at C640: kernel trace + counters (tools/ba_pmc.py)
bash tools/r06_ba_pmc.sh final > $o/ba_pmc_summary.txt 2>&1; cp gpurun_out/r06pmcfinal/ba_pmc.json $o/ba_traffic.json; cp $o/ba_traffic.json profiles/r06_ba_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/bac640 -o ba -- python tools/ba_c1280_bench.py 50 c640 > $o/ba_c640.json 2>/dev/null
cp $o/bac640/ba_kernel_stats.csv $o/ba_c640_kernel_stats.csv 2>/dev/null
NS_VARIANTS=1 NS_BA_UNFUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/bac640o -o ba -- python tools/ba_c1280_bench.py 50 c640 > $o/ba_c640_round5_kernels.json 2>/dev/null
cp $o/bac640o/ba_kernel_stats.csv $o/ba_c640_round5_kernels_stats.csv 2>/dev/null
# the RCCL path with one rank (+ the timed self-exchange); bench.py's `predicted` block reads it
timeout 300 python tests/rccl_worker.py 29655 2>/dev/null | grep '^{' | tail -1 > $o/rccl_one_rank.json; cp $o/rccl_one_rank.json profiles/r06_rccl_one_rank.json
# config #5: its kernel table first (the line quotes the shares), then the line
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $o/c1280 -o c -- python bench.py --config c1280 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $o/c1280/c_kernel_stats.csv $o/c1280_kernel_stats.csv 2>/dev/null; cp $o/c1280_kernel_stats.csv profiles/r06_c1280_kernel_stats.csv 2>/dev/null
timeout 500 python bench.py --config c1280 --steps 2 --warmup 1 > $o/bench_c1280.json 2> $o/bench_c1280.err
# the headline (its breakdown feeds the `predicted` block of later runs) + two more runs of the same tree
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err
cp $o/bench.json profiles/r06_bench.json
for r in 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_run$r.json 2>/dev/null; done
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- env NS_NGP_EXTRINSICS=1 python tools/ngp_bench.py 160 320 > $o/ngp.log 2>&1; grep steps/s $o/ngp.log
cp $o/ngp/ngp_kernel_stats.csv $o/ngp_kernel_stats.csv 2>/dev/null
# N > 1 topologies on the ONE device over gloo (functional records)
for n in 2 3; do
  NS_BENCH_DIST_BACKEND=gloo NS_BENCH_ONE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 12 --warmup 2 2>/dev/null | grep '^{' | tail -1 > $o/bench_gpus${n}_one_device_gloo.json
done
NS_BENCH_DIST_BACKEND=gloo NS_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29610 bench.py --gpus 2 --steps 2 --warmup 1 --config c1280 2>/dev/null | grep '^{' | tail -1 > $o/bench_c1280_gpus2_one_device_gloo.json
# (the raw counter dumps are ~60 MB: over gpurun's 64-MiB merge limit the WHOLE directory is dropped; the json files hold what is used)
mkdir -p $o/pmc_logs; for d in $o/pmc/*; do cp $d/trace.log $o/pmc_logs/$(basename $d).log 2>/dev/null; done
rm -rf $o/bprof $o/ngp $o/c1280 $o/pmc $o/bac640 $o/bac640o gpurun_out/r06pmcfinal
ls -la $o; du -sh $o
python - <<PY
import json
d = json.load(open("$o/bench.json")); r = d["roofline"]
print(d["value"], d["windows_frames_per_s"], r["kernel"], round(r["frac"], 3), r.get("frac_in_step"), r["traffic"], r.get("traffic_source"), d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
for k, v in r["other"].items():
    print(" ", k, round(v["avg_launch_us"], 1), v.get("in_step_us") and round(v["in_step_us"], 1), round(v["frac"], 4), v.get("traffic"), v.get("hbm_utilisation_rocprof"), v.get("frac_line_granular"))
print(r.get("clause_60pct"))
for r_ in (2, 3):
    try: print("run", r_, json.load(open("$o/bench_run%d.json" % r_))["value"])
    except Exception as e: print("run", r_, "failed", e)
c = json.load(open("$o/bench_c1280.json")); rr = c["roofline"]
print("c1280", c["value"], rr["kernel"], round(rr["frac"], 3))
for k, v in rr["other"].items():
    print(" ", k, round(v["avg_launch_us"], 1), round(v["frac"], 3), v.get("traffic_over_algorithmic"), v.get("mfma_busy"))
PY
