cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_benchprof; mkdir -p $o
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $o/bprof -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench_prof.json 2> $o/err.log
tail -c 300 $o/err.log
cp $o/bprof/b_kernel_stats.csv $o/bench_kernel_stats.csv
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$o/bench_kernel_stats.csv")))
for r in rows[:32]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:8.2f}ms avg {float(r['AverageNs'])/1e3:7.1f}us {r['Percentage']}")
d=json.load(open("$o/bench_prof.json")); print(d["value"], d["breakdown"]["ms_per_frame_by_leg"])
# timeline of one mapper step inside the pipeline
rows=list(csv.DictReader(open("$o/bprof/b_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "ngp_encode_fwd_kernel" in r["Kernel_Name"]]
a,b=idx[-60],idx[-58]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:8.1f} {e:8.1f} {e-s:7.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:60]}")
PY
rm -rf $o/bprof
