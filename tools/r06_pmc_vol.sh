#!/bin/bash
# counters of the volume build (tools/vol_levels_bench.py, E = 10 at 60x80, 4 levels): arms "band" (levels 2/3 through the LDS run
# buffers) and "noband" (round 5's per-group stores); separate rocprofv3 --pmc passes (no trace domains beside --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NS_VARIANTS=1 LEVELS=4
for arm in band noband; do
  out=gpurun_out/r06vol/$arm; mkdir -p $out
  [ $arm = noband ] && export NS_VOL_NO_BAND=1
  run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $out/$name -o $name -- python tools/vol_levels_bench.py 5 > $out/$name.log 2>&1 || tail -3 $out/$name.log; }
  run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES
  run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES
  run c SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INSTS_MFMA
  run d SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
  run e WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  run f FETCH_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  python - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if "corr_volume" not in k: continue
        res.update({c: round(sum(v)/len(v)) for c, v in d.items()})
print("$arm", json.dumps(res))
json.dump(res, open("$out/../$arm.json", "w"), indent=1)
PY
  find $out -name "*.csv" -size +200k -delete
done
