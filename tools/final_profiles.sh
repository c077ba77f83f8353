cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/final; mkdir -p $o
timeout 400 python bench.py > $o/bench.json 2> $o/bench.err; tail -c 300 $o/bench.err
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $o/ngp -o ngp -- python tools/ngp_bench.py 100 300 > $o/ngp.log 2>&1; grep steps/s $o/ngp.log
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $o/uop -o u -- python tools/update_op_prof.py 10 > /dev/null 2>&1
run() { name=$1; shift; timeout 100 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $o/pmc_$name -o $name -- python tools/conv_one.py > $o/pmc_$name.log 2>&1 || tail -3 $o/pmc_$name.log; }
run d SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS
run c SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU
python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$o/pmc_*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_nhwc" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res.update({c: round(sum(v) / len(v)) for c, v in agg.items()})
json.dump(res, open("$o/conv_pmc.json", "w"), indent=1); print(res)
d = json.load(open("$o/bench.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("conv_nets"), d["cpu_baseline"]["value"])
PY
