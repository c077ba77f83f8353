#!/usr/bin/env python3
"""profiles/r03_traffic.json from the rocprofv3 passes of tools/r03_final.sh over `bench.py --microbench NAME` -- the SAME
launches whose `avg_launch_us` bench.py reports (round-2 verdict: "make every roofline frac / traffic reproducible from
profiles/ on the same workload").  Per bench.py roofline entry:
  traffic_bytes          2 x FETCH_SIZE + WRITE_SIZE per call, summed over the kernels of the call (KB counters; FETCH_SIZE counts
                         64-B units of 128-B requests on gfx950: x 2, MI355X_MICROARCH.md), last `reps` launches of each kernel
  traffic_by_kernel      the same per kernel
  l2_hit_rate            TCC_HIT / (TCC_HIT + TCC_MISS) over those launches
  rocprof_avg_launch_us  sum of the kernels' average durations in the --kernel-trace pass of the same command
  in_pipeline_avg_us     the same kernels' average durations INSIDE the timed pipeline (profiles/r03_bench_kernel_stats.csv)
usage: python tools/r03_traffic.py <dir with NAME/{fetch,write,trace}> <bench_kernel_stats.csv> <reps> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

ENTRIES = {
    "ngp_encode_bwd[2^18]": ("ngp_bwd", ["ngp_enc_fscatter_kernel", "ngp_enc_faccum_kernel", "ngp_encode_bwd_dense_rl_kernel",
                                         "ngp_enc_dense_reduce_kernel"]),
    "ngp_encode_fwd_kernel[2^18]": ("ngp_fwd", ["ngp_encode_fwd_kernel"]),
    "corr_lookup_coop_kernel[E=48]": ("lookup", ["corr_lookup_coop_kernel"]),
    "corr_volume_tiled_kernel[E=10]": ("volume", ["corr_volume_tiled_kernel"]),
    "conv_nhwc_kernel<3x3,448->256>[E=48]": ("conv", ["conv_nhwc_kernel<3, 4, 4, 2>"]),
}


def counters(d, reps):
    """{kernel substring -> {counter -> mean over the last `reps` launches}}"""
    out = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, cs in per.items():
            for c, v in cs.items():
                v = [x[1] for x in sorted(v)][-reps:]
                out[k][c] = sum(v) / len(v)
    return out


def durations(d, reps):
    out = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        for k, v in per.items():
            v = [x[1] for x in sorted(v)][-reps:]
            out[k] = sum(v) / len(v)
    return out


def pick(table, sub):
    for k, v in table.items():
        if sub in k:
            return v
    return None


def main():
    root, stats, reps, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    pipe = {}
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            pipe[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    res = {}
    for name, (sub, kernels) in ENTRIES.items():
        d = os.path.join(root, sub)
        if not os.path.isdir(d):
            continue
        cnt, dur = counters(d, reps), durations(d, reps)
        by, tot, hit, miss, us, inpipe = {}, 0.0, 0.0, 0.0, 0.0, 0.0
        for k in kernels:
            c = pick(cnt, k)
            if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            b = 1024.0 * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"])
            by[k] = {"bytes": int(b), "fetch_kb_x2": round(2 * c["FETCH_SIZE"]), "write_kb": round(c["WRITE_SIZE"])}
            tot += b
            hit += c.get("TCC_HIT_sum", 0.0)
            miss += c.get("TCC_MISS_sum", 0.0)
            u = pick(dur, k)
            if u is not None:
                by[k]["rocprof_avg_us"] = round(u, 1)
                us += u
            p = pick(pipe, k)
            if p is not None:
                by[k]["in_pipeline_avg_us"] = round(p[0], 1)
                inpipe += p[0]
        if not by:
            continue
        res[name] = {"traffic_bytes": int(tot), "traffic_by_kernel": by, "l2_hit_rate": round(hit / max(hit + miss, 1.0), 3),
                     "rocprof_avg_launch_us": round(us, 1), "in_pipeline_avg_us": round(inpipe, 1) if inpipe else None,
                     "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) and "
                             "--kernel-trace over `python bench.py --microbench %s --reps %d`; last %d launches of every kernel" % (sub, reps, reps)}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps({k: (v["traffic_bytes"], v["rocprof_avg_launch_us"], v["in_pipeline_avg_us"], v["l2_hit_rate"]) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
