# SQ counters of the gate convolution (bench.py's own micro-bench): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_pmc_conv; rm -rf $o; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $o/p -o s -- python bench.py --microbench conv_nhwc --reps 10 > $o/log.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $o/q -o s -- python bench.py --microbench conv_nhwc --reps 10 >> $o/log.txt 2>&1
python - <<PY
import csv, collections
for sub in ("p", "q"):
    try: rows = list(csv.DictReader(open("$o/%s/s_counter_collection.csv" % sub)))
    except Exception as e: print(sub, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "conv_nhwc" not in k: continue
        print(k, {n: round(x / cnt[(k, n)]) for n, x in v.items()})
PY
tail -2 $o/log.txt
