#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r06c4; mkdir -p $o
timeout 900 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 600 $o/bench.err
python - <<PY
import json
d = json.load(open("$o/bench.json"))
print("value", d["value"], d["counts"].get("mean_active_edges_per_update"))
print(json.dumps(d["extra"].get("sensitivity"), indent=0)[:2500])
print(json.dumps(d["extra"].get("predicted_scaling"))[:1500])
print(json.dumps(d["roofline"].get("clause_60pct")))
PY
timeout 900 python bench.py --config c1280 --steps 2 --warmup 1 --no-cpu-baseline > $o/bench_c1280.json 2> $o/bench_c1280.err; tail -c 600 $o/bench_c1280.err
python - <<PY
import json
d = json.load(open("$o/bench_c1280.json"))
r = d["roofline"]
print(d["value"], r["kernel"], r["frac"], r["avg_launch_us"])
for k, v in r["other"].items():
    print(" ", k, round(v["avg_launch_us"], 1), round(v["frac"], 3))
print(d["dense_ba"], d["breakdown"])
PY
