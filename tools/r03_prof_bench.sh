cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_benchprof; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench.json 2> $o/bench.err
python - <<PY
import csv, json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1]); print("value", d["value"], "seq", d["sequential"]["frames_per_s"], d["breakdown"]["ms_per_frame_by_leg"])
rows=list(csv.DictReader(open("$o/prof/b_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%")
# one steady NeRF step timeline from the trace
tr=list(csv.DictReader(open("$o/prof/b_kernel_trace.csv")))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(tr) if "ngp_encode_fwd_kernel" in r["Kernel_Name"]]
a,b=idx[-200],idx[-199]
t0=int(tr[a]["Start_Timestamp"])
for r in tr[a:b+1]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:8.1f} {e:8.1f} {e-s:7.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:60]}")
PY
