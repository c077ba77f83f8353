#!/bin/bash
# PMC passes (each in its own run, kernel-trace only) for the NeRF trainer.  usage: tools/pmc_ngp.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $out/$name -o $name -- python tools/ngp_bench.py 20 300 > $out/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("_Z"):
            k = k[4:] if k[3].isdigit() else k
        if "ngp_" not in k: continue
        agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            v = v[len(v) // 2:]          # steady state (second half of the launches)
            res[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
json.dump(res, open("$out/ngp_pmc.json", "w"), indent=1)
for k, d in res.items():
    print(k, {c: round(x["mean"]) for c, x in d.items()})
PY
