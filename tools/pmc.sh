#!/bin/bash
# PMC passes (each in its own run, kernel-trace only) for tools/microbench.py <what>.
# usage: tools/pmc.sh <what> <outdir>
what=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $out/$name -o $name -- python tools/microbench.py $what 3 > $out/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run ta TA_BUSY_avr TA_BUSY_max TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
find $out -name "*counter_collection.csv" | head
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "at::" in k or "rocclr" in k: continue
        print(f.split("/")[-1][:12], k, {c: (sum(v)/len(v), len(v)) for c, v in d.items()})
PY
