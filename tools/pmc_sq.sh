#!/bin/bash
# instruction-mix counters for tools/microbench.py <what>; usage: tools/pmc_sq.sh <what> <outdir>
what=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $out/$name -o $name -- python tools/microbench.py $what 3 > $out/$name.log 2>&1 || tail -3 $out/$name.log; }
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run c SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INSTS_MFMA
run d SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "at::" in k or "rocclr" in k or "Cijk" in k: continue
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
