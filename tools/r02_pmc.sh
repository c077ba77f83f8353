cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r02final; mkdir -p $o
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
timeout 120 python -m pytest tests/test_ngp_gpu.py -m gpu -q -x --timeout=100 -k "ingest" 2>&1 | tail -5
