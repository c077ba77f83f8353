#!/usr/bin/env python3
"""Per-kernel counter summary of the rocprofv3 passes over tools/ba_c1280_bench.py (tools/r06_ba_pmc.sh): durations from the
--kernel-trace pass, HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE counts 64-B units of 128-B requests on
gfx950: x 2, MI355X_MICROARCH.md section HBM), L2 hit rate, SQ wave-cycle split, MFMA busy -- means over the last `reps`
launches of each BA kernel.  usage: ba_pmc.py <dir with trace/ fetch/ write/ sq/ sq2/> <reps> <bench json> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ["ba_edge_table_kernel", "ba_linearize_slot_kernel", "ba_schur_gram_kernel", "ba_schur_reduce_kernel", "ba_finalize_kernel",
           "ba_solve_depth_kernel", "ba_linearize_kernel", "ba_accum_kernel", "ba_schur_kernel"]


def counters(d, reps):
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
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        for k, v in per.items():
            v = [x[1] for x in sorted(v)][-reps:]
            out[k] = sum(v) / len(v)
    return out


def pick(table, sub):
    for k, v in table.items():
        if k.startswith(sub) or ("void " + sub) in k or (" " + sub) in k:
            return v
    return None


def main():
    root, reps, bench, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    dur = durations(os.path.join(root, "trace"), reps)
    cs = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        if os.path.isdir(os.path.join(root, sub)):
            for k, v in counters(os.path.join(root, sub), reps).items():
                cs.setdefault(k, {}).update(v)
    res = {}
    for name in KERNELS:
        d = pick(dur, name)
        c = pick(cs, name) or {}
        if d is None:
            continue
        e = {"rocprof_avg_launch_us": d}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["traffic_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            e["fetch_bytes_x2"] = 2.0 * c["FETCH_SIZE"] * 1024.0
            e["write_bytes"] = c["WRITE_SIZE"] * 1024.0
            e["hbm_tb_per_s"] = e["traffic_bytes"] / d / 1e6
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            e["wave_cycles_split"] = {k2: c[k1] / wc for k1, k2 in (("SQ_WAIT_ANY", "parked_on_waitcnt_or_barrier"), ("SQ_WAIT_INST_ANY", "issue_stalled"),
                                                                   ("SQ_ACTIVE_INST_ANY", "issuing")) if k1 in c}
            e["raw_sq"] = {k: c[k] for k in c if k.startswith("SQ_") or k.startswith("GRBM")}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"] > 0:
            e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
        if "GRBM_GUI_ACTIVE" in c:
            e["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / d / 1e3
        res[name] = e
    if os.path.exists(bench):
        b = json.loads([l for l in open(bench) if l.startswith("{")][-1])
        res["_bench"] = b
        alg = b["algorithmic_bytes"]
        for name, key in (("ba_linearize_slot_kernel", "linearise_accumulate_contract_bytes"), ("ba_schur_gram_kernel", "schur_min_bytes"),
                          ("ba_solve_depth_kernel", "depth_update_bytes")):
            if name in res:
                e = res[name]
                e["algorithmic_bytes"] = alg[key]
                e["frac_of_8TBs_algorithmic"] = alg[key] / e["rocprof_avg_launch_us"] / 1e6 / 8.0
                if "traffic_bytes" in e:
                    e["traffic_over_algorithmic"] = e["traffic_bytes"] / alg[key]
    res["_meta"] = {"git_head": os.environ.get("NS_GIT_HEAD"), "reps": reps}
    json.dump(res, open(dst, "w"), indent=1)
    for k, e in res.items():
        if k.startswith("_"):
            continue
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items() if a != "raw_sq"})


if __name__ == "__main__":
    main()
