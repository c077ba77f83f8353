cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_altcorr; mkdir -p $o
python tools/altcorr_bench2.py | head -1
NS_ALTCORR_NO_XCD=1 python tools/altcorr_bench2.py | head -1
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $o/$name -o $name -- python tools/altcorr_bench2.py > $o/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run mfma SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$o/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "altcorr" not in k: continue
        agg[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
