#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the rocprofv3 passes of tools/r05_final.sh (r04_final.sh before it) over `bench.py --microbench NAME` -- the SAME
launches whose `avg_launch_us` bench.py reports (round-2 verdict: "make every roofline frac / traffic reproducible from
profiles/ on the same workload").  Per bench.py roofline entry:
  traffic_bytes          2 x FETCH_SIZE + WRITE_SIZE per call, summed over the kernels of the call (KB counters; FETCH_SIZE counts
                         64-B units of 128-B requests on gfx950: x 2, MI355X_MICROARCH.md), last `reps` launches of each kernel
  traffic_by_kernel      the same per kernel
  l2_hit_rate            TCC_HIT / (TCC_HIT + TCC_MISS) over those launches
  rocprof_avg_launch_us  sum of the kernels' average durations in the --kernel-trace pass of the same command
  in_pipeline_avg_us     the same kernels' average durations INSIDE the timed pipeline (profiles/rNN_bench_kernel_stats.csv)
  mfma_busy              (MFMA-bound entries) SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) of the kernels, raw counters beside it
  samples                the marched samples of the trainer step the NeRF micro-benches ran on (bench.py scales traffic per sample)
`_meta`: git HEAD and the sha256 of the library the passes ran on -- bench.py prints `traffic_stale` when it runs another one.
usage: python tools/traffic.py <dir with NAME/{fetch,write,trace[,mfma]}> <bench_kernel_stats.csv> <reps> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

ENTRIES = {
    "ngp_encode_bwd[step samples]": ("ngp_bwd", ["ngp_enc_fscatter_direct_kernel", "ngp_enc_faccum_kernel"]),
    "ngp_encode_fwd_kernel[step samples]": ("ngp_fwd", ["ngp_encode_fwd_kernel"]),
    "ngp_mlp_fwd_kernel[step samples]": ("mlp_fwd", ["ngp_mlp_fwd_kernel"]),
    "ngp_mlp_bwd_kernel[step samples]": ("mlp_bwd", ["ngp_mlp_bwd_kernel"]),
    "ngp_mlp_wgrad_tr_kernel[step samples]": ("mlp_wgrad", ["ngp_mlp_wgrad_tr_kernel", "ngp_mlp_wgrad_reduce_kernel"]),
    "corr_lookup_coop_kernel[E=48]": ("lookup", ["corr_lookup_coop_kernel"]),
    "corr_lookup_enc_kernel[E=48]": ("lookup_enc", ["corr_lookup_enc_kernel"]),
    "corr_volume_tiled_kernel[E=10]": ("volume", ["corr_volume_tiled_kernel"]),
    "conv_nhwc_kernel<3x3,448->256>[E=48]": ("conv", ["conv_nhwc_kernel<3, 4, 4, 2>"]),
    "altcorr_tile_mfma_lds_kernel[E=48, 160x90]": ("altcorr", ["altcorr_tile_mfma"]),
    "altcorr_tile_enc_lds_kernel[E=48, 160x90]": ("altcorr_enc", ["altcorr_tile_enc"]),
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
        mf = {}
        for k in kernels:
            c = pick(cnt, k)
            if c is not None and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                mf[k] = {kk: c.get(kk) for kk in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
                if c.get("SQ_BUSY_CU_CYCLES"):
                    mf[k]["mfma_busy_over_4x_busy_cu_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"]), 4)
        samples = None
        tl = os.path.join(d, "trace.log")
        if os.path.exists(tl):
            for ln in open(tl):
                if ln.startswith("{"):
                    samples = json.loads(ln).get("samples")
        res[name] = {"traffic_bytes": int(tot), "traffic_by_kernel": by, "l2_hit_rate": round(hit / max(hit + miss, 1.0), 3),
                     "rocprof_avg_launch_us": round(us, 1), "in_pipeline_avg_us": round(inpipe, 1) if inpipe else None,
                     "samples": samples, "mfma_busy": mf or None,
                     "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes) and "
                             "--kernel-trace over `python bench.py --microbench %s --reps %d`; last %d launches of every kernel" % (sub, reps, reps)}
    import hashlib
    import subprocess
    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root_dir, "nerf-slam_amd", "lib", "libnerfslam_hip.so")
    head = os.environ.get("NS_GIT_HEAD") or ""
    if not head:
        try:
            head = subprocess.run(["git", "-C", root_dir, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            head = ""
    res["_meta"] = {"git_head": head or None, "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
                    "note": "git_head = the commit the tree was at when tools/r05_final.sh was sent to the GPU box (passed in through "
                            "NS_GIT_HEAD: the box has no .git); lib_sha256 = the library the passes ran"}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps({k: (v["traffic_bytes"], v["rocprof_avg_launch_us"], v["in_pipeline_avg_us"], v["l2_hit_rate"], v["samples"]) for k, v in res.items() if k != "_meta"}, indent=1))


if __name__ == "__main__":
    main()
