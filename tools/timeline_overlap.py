"""How much of the timed pipeline's GPU time is the two legs running AT THE SAME TIME?  Reads a rocprofv3 --kernel-trace CSV of
bench.py and classifies every kernel as mapper (ngp_*) or tracker (everything else), then reports, over the busiest 60 % of
the run (the timed windows), per leg: busy time (union of its kernels' intervals), the overlap of the two unions, the time
neither runs, and per kernel class how much longer its launches take while a kernel of the other leg is running.
usage: python tools/timeline_overlap.py <kernel_trace.csv>"""
import collections
import csv
import json
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(u):
    return sum(b - a for a, b in u)


def intersect(u, v):
    i = j = 0
    s = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b:
            s += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return s


def overlap_with(iv, u):
    """per interval: fraction of it covered by the union u"""
    import bisect
    starts = [a for a, _ in u]
    out = []
    for a, b in iv:
        k = max(0, bisect.bisect_right(starts, a) - 1)
        c = 0
        while k < len(u) and u[k][0] < b:
            lo, hi = max(a, u[k][0]), min(b, u[k][1])
            if lo < hi:
                c += hi - lo
            k += 1
        out.append(c / max(1, b - a))
    return out


rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
t0, t1 = ks[0][0], ks[-1][1]
lo, hi = t0 + int(0.10 * (t1 - t0)), t1        # (set-up cut off; the bins below find the pipeline)
ks = [k for k in ks if k[0] >= lo and k[1] <= hi]
is_map = lambda n: "ngp_" in n
mp = [(a, b) for a, b, n in ks if is_map(n)]
tr = [(a, b) for a, b, n in ks if not is_map(n)]
um, ut = union(mp), union(tr)
both = intersect(um, ut)
span = hi - lo
# per 50-ms bin: which bins belong to the steady pipeline (the mapper trains in them)?
BIN = 50_000_000
def clip(u, a, b):
    return [[max(x, a), min(y, b)] for x, y in u if y > a and x < b]
bins = []
t = lo
while t + BIN <= hi:
    cm, ct = clip(um, t, t + BIN), clip(ut, t, t + BIN)
    bm, bt, bb = total(cm) / BIN, total(ct) / BIN, intersect(cm, ct) / BIN
    bins.append((bm, bt, bb, 1.0 - (bm + bt - bb)))
    t += BIN
act = [b for b in bins if b[0] >= 0.35]
steady = {"bins_of_50_ms": len(bins), "bins_with_the_mapper_training": len(act)}
if act:
    for i, k in enumerate(("mapper_busy", "tracker_busy", "both_at_once", "neither")):
        v = sorted(b[i] for b in act)
        steady[k] = {"mean": round(sum(v) / len(v), 3), "min": round(v[0], 3), "median": round(v[len(v) // 2], 3), "max": round(v[-1], 3)}
steady["bins"] = [[round(x, 2) for x in b] for b in bins]       # [mapper, tracker, both, neither] per 50 ms, in time order
par = [b for b in bins if b[0] >= 0.35 and b[2] >= 0.25]             # the --parallel_run windows: both legs live at once
if par:
    steady["parallel_windows"] = {k: round(sum(b[i] for b in par) / len(par), 3) for i, k in
                                  enumerate(("mapper_busy", "tracker_busy", "both_at_once", "neither"))}
    steady["parallel_windows"]["bins"] = len(par)
res = {"steady_pipeline_bins": steady, "span_ms": span / 1e6, "mapper_busy": total(um) / span, "tracker_busy": total(ut) / span, "both_at_once": both / span,
       "neither": 1.0 - (total(um) + total(ut) - both) / span,
       "tracker_kernel_time_inside_mapper_busy": None, "kernels": len(ks)}
frac = overlap_with(tr, um)
dur = [b - a for a, b in tr]
res["tracker_kernel_time_inside_mapper_busy"] = sum(f * d for f, d in zip(frac, dur)) / max(1, sum(dur))
# slow-down of the commonest tracker kernels when overlapped
by = collections.defaultdict(lambda: [[], []])
for (a, b, n), f in zip([k for k in ks if not is_map(k[2])], frac):
    by[n.split("(")[0][:48]][0 if f < 0.1 else (1 if f > 0.9 else 0 if False else 1 if f > 0.9 else 0)].append((b - a) / 1e3) if (f < 0.1 or f > 0.9) else None
tab = []
for n, (alone, over) in by.items():
    if len(alone) >= 20 and len(over) >= 20:
        tab.append((n, len(alone), sum(alone) / len(alone), len(over), sum(over) / len(over)))
tab.sort(key=lambda t: -t[3] * t[4])
res["us_alone_vs_overlapped"] = [{"kernel": n, "n_alone": a, "us_alone": round(x, 1), "n_overlapped": b, "us_overlapped": round(y, 1)} for n, a, x, b, y in tab[:12]]
print(json.dumps(res, indent=1))
