# SQ counters of the NeRF step's kernels (one pass, 8 SQ slots): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_pmc_ngp; rm -rf $o; mkdir -p $o
NS_NGP_EXTRINSICS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d $o/p -o s -- python tools/ngp_bench.py 32 64 > $o/log.txt 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$o/p/s_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"]
print("%-44s %6s %12s  wait_any wait_inst active  mfma_busy/wavecyc*4 valu lds wait_lds" % ("kernel", "calls", "wave_cyc/call"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not k.startswith(("ngp", "_Z", "void ngp")): continue
    w = v["SQ_WAVE_CYCLES"] or 1
    print("%-44s %6d %12.0f  %7.2f %7.2f %7.2f  %7.3f %7.2f %7.2f %7.2f" % (k, cnt[k], w / max(cnt[k], 1), v["SQ_WAIT_ANY"] / w, v["SQ_WAIT_INST_ANY"] / w, v["SQ_ACTIVE_INST_ANY"] / w, v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * w), v["SQ_ACTIVE_INST_VALU"] / w, v["SQ_ACTIVE_INST_LDS"] / w, v["SQ_WAIT_INST_LDS"] / w))
PY
