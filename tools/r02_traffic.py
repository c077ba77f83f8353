"""profiles/r02_traffic.json from the PMC passes of tools/r02_final.sh (gpurun_out/r02final/ngp_pmc.json; KB per launch):
HBM traffic per launch = 2 x FETCH_SIZE (gfx950: the counter counts 64-B units of 128-B requests, MI355X_MICROARCH.md) +
WRITE_SIZE.  The lookup / volume entries are round 1's measurements (kernels unchanged), carried over."""
import json, os, sys
root = os.path.join(os.path.dirname(__file__), "..")
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "r02final", "ngp_pmc.json")
pmc = json.load(open(src))
old = json.load(open(os.path.join(root, "profiles", "r02_traffic.json")))


def kb(prefix):
    for k, v in pmc.items():
        if k.startswith(prefix):
            return 2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"], v
    return 0.0, None


parts = ["ngp_zero_ints", "ngp_enc_bin_count", "ngp_enc_bin_scatter", "ngp_enc_bin_accum", "ngp_encode_bwd_dense_rl", "ngp_enc_dense_reduce"]
vals = [(p, kb(p)[0]) for p in parts]
out = {k: v for k, v in old.items() if k.startswith("corr_")}
out["ngp_encode_bwd[2^18]"] = {
    "traffic_bytes": int(1024 * sum(v for _, v in vals)),
    "note": "rocprofv3 --pmc (separate passes, tools/r02_final.sh: `tools/ngp_bench.py` sphere scene, pose refinement on, ~2.3e5 "
            "samples per step): sum over the call's launches of 2 x FETCH_SIZE + WRITE_SIZE: "
            + ", ".join("%s %.0f MB" % (p, v / 1024) for p, v in vals) + "; profiles/r02_ngp_pmc.json"}
a, av = kb("ngp_adam_kernel")
out["ngp_adam_kernel[hash grid]"] = {"traffic_bytes": int(1024 * a), "note": "same passes; mean over the grid and the (tiny) MLP launch"}
f, fv = kb("ngp_encode_fwd_kernel")
hit = fv["TCC_HIT_sum"]["mean"] / max(fv["TCC_HIT_sum"]["mean"] + fv["TCC_MISS_sum"]["mean"], 1.0)
out["ngp_encode_fwd_kernel[2^18]"] = {"traffic_bytes": int(1024 * f), "note": "same passes; L2 hit rate %.0f %%" % (100 * hit)}
json.dump(out, open(os.path.join(root, "profiles", "r02_traffic.json"), "w"), indent=1)
print(json.dumps({k: v["traffic_bytes"] for k, v in out.items()}, indent=1))
