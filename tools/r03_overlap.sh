# do the tracker's kernels overlap the mapper's inside the --parallel_run pipeline?  kernel trace of bench.py, then per hardware
# queue: which kernels ran there, and how much of the tracker's kernel time ran while a NeRF kernel was running
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03_overlap; rm -rf $o; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace -f csv -d $o/p -o b -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/bench.json 2>/dev/null
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$o/p/b_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
T0, T1 = ev[0][0], ev[-1][1]
# steady state: the last 35 % of the trace (the timed windows)
lo = T0 + int(0.55 * (T1 - T0)); hi = T0 + int(0.80 * (T1 - T0))
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
isn = lambda n: "ngp_" in n
q = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for s, e, qi, n in ev:
    k = q[qi]
    if isn(n): k[0] += 1; k[1] += (e - s) / 1e3
    else: k[2] += 1; k[3] += (e - s) / 1e3
print("window ms", (hi - lo) / 1e6)
for qi, k in sorted(q.items()): print("queue", qi, "ngp launches %d (%.1f ms)  other launches %d (%.1f ms)" % (k[0], k[1] / 1e3, k[2], k[3] / 1e3))
# union of NeRF-busy intervals, then the tracker's kernel time inside / outside it
iv = sorted((s, e) for s, e, qi, n in ev if isn(n)); merged = []
for s, e in iv:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
import bisect
starts = [m[0] for m in merged]
def inside(s, e):
    t = 0; i = max(bisect.bisect_right(starts, s) - 1, 0)
    while i < len(merged) and merged[i][0] < e:
        t += max(0, min(e, merged[i][1]) - max(s, merged[i][0])); i += 1
    return t
tin = sum(inside(s, e) for s, e, qi, n in ev if not isn(n)); tall = sum(e - s for s, e, qi, n in ev if not isn(n))
nb = sum(m[1] - m[0] for m in merged)
print("NeRF-busy union %.1f ms of %.1f;  tracker kernel time %.1f ms, of which %.1f ms while a NeRF kernel was running" % (nb / 1e6, (hi - lo) / 1e6, tall / 1e6, tin / 1e6))
alliv = sorted((s, e) for s, e, qi, n in ev); m2 = []
for s, e in alliv:
    if m2 and s <= m2[-1][1]: m2[-1][1] = max(m2[-1][1], e)
    else: m2.append([s, e])
print("any-kernel-busy union %.1f ms (idle %.1f ms)" % (sum(b - a for a, b in m2) / 1e6, ((hi - lo) - sum(b - a for a, b in m2)) / 1e6))
PY
