"""Where does the mapper's optimiser step lose time to its own side streams?  (round 5, timing experiment only.)

The table gradient's two kernels take 50 + 57 us alone and 87 + 108 us inside the step (profiles/r04_bench.json); this tool
re-times the WHOLE step (HIP-graph replay, as the product runs it) with individual launches of the side streams turned into
no-ops, by wrapping the ctypes library object the trainer calls through -- the product code is not touched and none of the
ablated variants trains correctly (they are timing probes, nothing else).
usage: python tools/r05_step_ablation.py [steps]      -> one JSON line per variant, ms per step (median of 5 blocks)"""
import faulthandler, json
faulthandler.enable()
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd")]
import torch

import nerfslam.ngp as ngp_mod
from nerfslam.ngp import NgpConfig, NgpNerf

import importlib.util
spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(root, "tools", "ngp_scene.py"))
sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)

real_lib = ngp_mod.lib


class Proxy:
    def __init__(self, L, skip):
        self._L, self._skip = L, set(skip)

    def __getattr__(self, name):
        f = getattr(self._L, name)
        if name in self._skip:
            return lambda *a, **k: 0
        return f


VARIANTS = {
    "base": [],
    "no_wgrad": ["ns_ngp_mlp_wgrad_partials_n"],
    "no_wgrad_no_mlp_step": ["ns_ngp_mlp_wgrad_partials_n", "ns_ngp_mlp_step_fused"],
    "no_pose_chain": ["ns_ngp_encode_jacobian_dot_n", "ns_ngp_camera_gradient_2stage", "ns_ngp_camera_step_ctl"],
    "no_side_work_at_all": ["ns_ngp_mlp_wgrad_partials_n", "ns_ngp_mlp_step_fused", "ns_ngp_encode_jacobian_dot_n",
                            "ns_ngp_camera_gradient_2stage", "ns_ngp_camera_step_ctl"],
    "no_table_gradient": ["ns_ngp_encode_backward_fused_n"],
    "no_table_gradient_no_side": ["ns_ngp_encode_backward_fused_n", "ns_ngp_mlp_wgrad_partials_n", "ns_ngp_mlp_step_fused",
                                  "ns_ngp_encode_jacobian_dot_n", "ns_ngp_camera_gradient_2stage", "ns_ngp_camera_step_ctl"],
}

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 320
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None        # e.g. "base": NS_NGP_WGRAD_WGS=256 python ... 320 base
dev = torch.device("cuda:0")
work = torch.cuda.Stream(device=dev)
for name, skip in VARIANTS.items():
    if only and name not in only:
        continue
    for wgs in (os.environ.get("NS_NGP_WGRAD_WGS"),):          # (read once per process by the library: vary it per process)
        ngp_mod.lib = real_lib
        net = NgpNerf(NgpConfig(optimize_extrinsics=True), dev, seed=0)
        net.set_images(*sc.sphere_scene())
        with torch.cuda.stream(work):
            for _ in range(20):                                   # 320 real steps: a trained scene's sample set
                net.train_steps(16, return_loss=False)
            torch.cuda.synchronize()
            # from here on the variant: fresh graphs with the no-op launches
            L = real_lib()
            ngp_mod.lib = (lambda P=Proxy(L, skip): P)
            net._graphs = [None, None]
            net._pair = None
            net._chains = {}
            for _ in range(4):
                net.train_steps(16, return_loss=False)
            torch.cuda.synchronize()
            blocks = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(steps // 16):
                    net.train_steps(16, return_loss=False)
                torch.cuda.synchronize()
                blocks.append(1e3 * (time.perf_counter() - t0) / (steps // 16 * 16))
        blocks.sort()
        print(json.dumps({"variant": name + (f" wgs={wgs}" if wgs else ""), "ms_per_step_median": round(blocks[2], 4),
                          "min": round(blocks[0], 4), "max": round(blocks[-1], 4), "samples": int(net.last_samples)}), flush=True)
        del net
        torch.cuda.empty_cache()
