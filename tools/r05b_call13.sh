#!/bin/bash
# SQ counters of the table gradient's two passes (bench.py --microbench ngp_encode_bwd, a trained step's samples): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r05b13; rm -rf $o; mkdir -p $o
run() { n=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $o/$n -o s -- python bench.py --microbench ngp_encode_bwd --reps 20 > $o/$n.log 2>&1; }
run p SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run q SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run r SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
python - <<PY
import csv, collections, glob, json
out = {}
for sub in ("p", "q", "r"):
    fs = glob.glob("$o/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs: print(sub, "no file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "fscatter" in k or "faccum" in k:
            agg["scatter" if "fscatter" in k else "accumulate"][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = [x[1] for x in sorted(v)][-20:]
            out.setdefault(k, {})[c] = round(sum(v) / len(v))
    fs = glob.glob("$o/%s/**/*kernel_trace.csv" % sub, recursive=True)
    if fs and sub == "p":
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "fscatter" in k or "faccum" in k:
                per["scatter" if "fscatter" in k else "accumulate"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in per.items(): out[k]["duration_us"] = round(sum(v[-20:]) / 20, 1)
print(json.dumps(out, indent=1))
json.dump(out, open("$o/table_gradient_sq.json", "w"), indent=1)
PY
rm -rf $o/p $o/q $o/r
