cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_ngp; mkdir -p $o
NS_NGP_EXTRINSICS=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o ngp -- python tools/ngp_bench.py 160 320 > $o/bench.log 2>&1
grep "steps/s" $o/bench.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$o/prof/ngp_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "ngp_encode_fwd_kernel" in r["Kernel_Name"]]
a,b=idx[-40],idx[-38]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:8.1f} {e:8.1f} {e-s:7.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:60]}")
PY
head -30 $o/prof/ngp_kernel_stats.csv | cut -c1-150
