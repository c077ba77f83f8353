#!/bin/bash
# SQ counters of the gate convolution (tools/conv_one.py: E = 48, 448 -> 256, 3x3): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b5; rm -rf $o; mkdir -p $o
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $o/p -o s -- python tools/conv_one.py > $o/log.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $o/q -o s -- python tools/conv_one.py >> $o/log.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 -f csv -d $o/r -o s -- python tools/conv_one.py >> $o/log.txt 2>&1
python - <<PY
import csv, collections, glob
for sub in ("p", "q", "r"):
    fs = glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print(sub, "no file"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        if "conv_nhwc" not in k: continue
        print(sub, k, {n: round(x / cnt[(k, n)]) for n, x in v.items()})
    fs = glob.glob("$o/%s/**/*kernel_trace.csv" % sub, recursive=True)
    if fs:
        d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(fs[0])) if "conv_nhwc" in r["Kernel_Name"]]
        print(sub, "durations us", [round(x / 1e3, 1) for x in d])
PY
tail -3 $o/log.txt
rm -rf $o/p $o/q $o/r
