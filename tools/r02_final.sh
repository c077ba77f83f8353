# round 2 evidence of the final tree: tests, bench lines (c640, c1280), kernel stats, NeRF trainer stats, PMC traffic, encoders.
# Every stage under a timeout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02final; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests -m gpu -q --timeout=100 -x > $o/pytest.log 2>&1; tail -3 $o/pytest.log
timeout 170 python bench.py --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err; head -c 400 $o/bench.json; echo
timeout 200 python bench.py --config c1280 --steps 2 --warmup 1 > $o/c1280.json 2> $o/c1280.err; head -c 300 $o/c1280.json; echo
timeout 60 python tools/enc_bench.py > $o/enc_bench.json 2> $o/enc_bench.err; cat $o/enc_bench.json
timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $o/bprof.log 2>&1
timeout 90 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- python tools/ngp_bench.py 100 300 > $o/ngp.log 2>&1; grep steps/s $o/ngp.log
NS_NGP_EXTRINSICS=1 timeout 60 python tools/ngp_bench.py 200 300 2>&1 | grep steps/s
run() { name=$1; shift; NS_NGP_EXTRINSICS=1 timeout 150 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $o/pmc_$name -o $name -- python tools/ngp_bench.py 6 40 > $o/pmc_$name.log 2>&1 || tail -3 $o/pmc_$name.log; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
files = sorted(set(glob.glob("$o/pmc_*/**/*counter_collection.csv", recursive=True)))
for f in files:
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
rm -rf $o/pmc_fetch $o/pmc_write
