#!/bin/bash
# round 5, first GPU call: new parity tests, fetch-granularity micro-experiment (time + counters), step ablation, baseline bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05a; mkdir -p $o
timeout 900 python -m pytest tests/test_parity_c1280_gpu.py tests/test_parity_c640_gpu.py -x -q -m gpu 2>&1 | tail -15 > $o/tests.log; tail -3 $o/tests.log
# fetch granularity: timings, then counters (separate passes)
timeout 120 tools/_bin/fetch_gran 2048 > $o/fetch_gran_time.jsonl 2> $o/fetch_gran.err; cat $o/fetch_gran_time.jsonl
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -f csv -d $o/fg_rd -o r -- tools/_bin/fetch_gran 2048 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $o/fg_fs -o f -- tools/_bin/fetch_gran 2048 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_BUBBLE_sum TCC_REQ_sum TCC_MISS_sum -f csv -d $o/fg_bb -o b -- tools/_bin/fetch_gran 2048 > /dev/null 2>&1
python - <<PY > $o/fetch_gran_counters.json
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$o/fg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: v for c, v in cs.items()} for k, cs in out.items()}, indent=1))
PY
head -c 6000 $o/fetch_gran_counters.json
rm -rf $o/fg_rd $o/fg_fs $o/fg_bb
# step ablation
timeout 400 python tools/r05_step_ablation.py 320 > $o/ablation.jsonl 2> $o/ablation.err; cat $o/ablation.jsonl; tail -3 $o/ablation.err
NS_VARIANTS=1 NS_NGP_WGRAD_WGS=256 timeout 200 python tools/r05_step_ablation.py 320 base,no_pose_chain >> $o/ablation.jsonl 2>> $o/ablation.err; tail -2 $o/ablation.jsonl
# baseline bench of the round-4 tree on this box
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_base.json 2> $o/bench_base.err
python - <<PY
import json
d = json.load(open("$o/bench_base.json"))
print("bench", d["value"], d["windows_frames_per_s"], d.get("sequential", {}).get("frames_per_s"), d.get("breakdown", {}).get("ms_per_frame_by_leg"))
PY
