cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04_track; mkdir -p $o
python tools/track_prof.py 40 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $o/prof -o t -- python tools/track_prof.py 40 > $o/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$o/prof/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:8.2f} ms {100*float(r['TotalDurationNs'])/tot:6.2f}%")
# launches in one non-keyframe frame: take the trace between two consecutive enc_stem_im2col kernels near the end
tr=list(csv.DictReader(open("$o/prof/t_kernel_trace.csv")))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(tr) if "enc_stem_im2col" in r["Kernel_Name"]]
best=None
for a,b in zip(idx[-30:-1], idx[-29:]):
    if best is None or b-a < best[1]-best[0]: best=(a,b)
a,b=best
t0=int(tr[a]["Start_Timestamp"])
print("---- one non-keyframe frame:", b-a, "launches")
for r in tr[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:8.1f} {e-s:7.1f}  {r['Kernel_Name'][:80]}")
PY
python - <<PY
import csv
o="gpurun_out/r04_track"
tr=list(csv.DictReader(open(o+"/prof/t_kernel_trace.csv")))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
# one UPDATE: between two consecutive corr_lookup_coop kernels near the end (E = 48 edges)
idx=[i for i,r in enumerate(tr) if "corr_lookup_coop" in r["Kernel_Name"]]
a,b=idx[-3],idx[-2]
t0=int(tr[a]["Start_Timestamp"])
print("---- one update:", b-a, "launches", (int(tr[b]["Start_Timestamp"])-t0)/1e3, "us")
for r in tr[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:8.1f} {e-s:7.1f}  {r['Kernel_Name'][:90]}")
PY
rm -rf gpurun_out/r04_track/prof
