"""debug: poison the caching allocator with NaNs, then run the NeRF convergence test and report the first
buffer that goes non-finite."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "nerf-slam_amd"), os.path.join(root, "tests"), root]
import torch
dev = torch.device("cuda:0")
junk = [torch.full((1 << 28,), float("nan"), device=dev) for _ in range(8)]   # 8 GiB of NaN
hj = [torch.full((1 << 28,), float("nan"), device=dev, dtype=torch.float16) for _ in range(4)]
del junk, hj
from nerfslam import ngp
orig = ngp.NgpNerf.train_step
state = {"n": 0}
def wrapped(self):
    out = orig(self)
    state["n"] += 1
    if state["n"] <= 3:
        N = self.last_samples; N8 = (N + 7) // 8 * 8
        chk = {"s_pos": self.s_pos[:N8], "s_dir": self.s_dir[:N8], "s_dt": self.s_dt[:N], "s_t": self.s_t[:N], "s_feat": self.s_feat[:N8],
               "s_out": self.s_out[:N8], "s_dout": self.s_dout[:N8], "s_dfeat": self.s_dfeat[:N8], "partial": self.partial,
               "mlp_grad": self.mlp_grad, "grid_grad": self.grid_grad, "mlp_master": self.mlp_master, "grid_master": self.grid_master,
               "mlp_half": self.mlp_half, "grid_half": self.grid_half, "density_grid": self.density_grid, "loss": self.loss_tensor}
        for i, a in enumerate(self.act):
            chk[f"act{i}"] = a.view(-1)[:a.shape[0] * N8]
        for i, a in enumerate(self.dact):
            chk[f"dact{i}"] = a.view(-1)[:a.shape[0] * N8]
        bad = {k: int((~torch.isfinite(v.float())).sum()) for k, v in chk.items()}
        print("step", state["n"], "N", N, {k: v for k, v in bad.items() if v}, flush=True)
        if state["n"] == 1:
            nz = (~torch.isfinite(self.s_feat[:N8].float())).nonzero()
            print("levels(feature idx) hist", torch.bincount(nz[:, 1], minlength=32).tolist())
            us = torch.unique(nz[:, 0]); print("bad samples", us.numel(), us[:5].tolist(), us[-5:].tolist())
            rs, rn = self.ray_start.long(), self.ray_n.long()
            print("sum ray_n", int(rn.sum()), "max end", int((rs + rn).max()), "rays n>0", int((rn > 0).sum()), "counter", self.counter.tolist())
            sm = nz[:8:4, 0]
            print("samples", sm.tolist(), "pos", self.s_pos[sm].tolist(), "unit", self.to_unit(self.s_pos[sm]).tolist())
            print("pos range", self.s_pos[:N].min(0).values.tolist(), self.s_pos[:N].max(0).values.tolist())
            again = self.encode(self.to_unit(self.s_pos[:N8]))
            print("re-encode nonfinite", int((~torch.isfinite(again.float())).sum()))
    return out
ngp.NgpNerf.train_step = wrapped
import test_ngp_gpu
try:
    test_ngp_gpu.test_training_converges_on_a_synthetic_scene(dev)
    print("PASS")
except AssertionError as e:
    print("FAIL", str(e)[:200])
