# round 2 evidence: bench line, kernel stats of the bench, NeRF trainer stats + PMC traffic.  Every stage under a timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02final; mkdir -p $o
timeout 170 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err; head -c 400 $o/bench.json; echo
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- python tools/ngp_bench.py 100 300 > $o/ngp.log 2>&1; grep steps/s $o/ngp.log
run() { name=$1; shift; NS_NGP_EXTRINSICS=1 timeout 90 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $o/pmc_$name -o $name -- python tools/ngp_bench.py 20 300 > $o/pmc_$name.log 2>&1 || tail -3 $o/pmc_$name.log; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in sorted(glob.glob("$o/pmc_*/**/*counter_collection.csv", recursive=True)) + sorted(glob.glob("$o/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "ngp_" not in k: continue
        k = k[k.index("ngp_"):][:34]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            v = v[len(v) // 2:]
            res[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(res, open("$o/ngp_pmc.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k, {c: round(x["mean"]) for c, x in d.items()})
PY
