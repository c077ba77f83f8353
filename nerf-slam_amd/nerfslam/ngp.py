"""NeRF mapping backend on the HIP kernels of csrc/ngp.hip + csrc/ngp_mlp.hip.

`NgpNerf` is the trainer that sits behind the `pyngp.Testbed` surface the reference drives
(/root/reference/fusion/nerf_fusion.py:57-101, 285-303, 388-424).  The algorithm is the published
instant-ngp NeRF (multiresolution hash grid -> fully-fused MLPs -> occupancy-grid ray marching ->
volume-rendering loss -> Adam) with the NeRF-SLAM fork's per-pixel depth + depth-covariance
supervision (nerf_fusion.py:100-101, 285-289).  The fork itself is un-vendored: configuration
values below are this project's own statement of the published defaults.

torch is used for allocation, random ray selection and a few elementwise glue ops; every
per-sample / per-parameter pass runs in the HIP kernels.
"""
import ctypes as C
import math
from dataclasses import dataclass

import torch

from ._lib import NerfSlamHipError, check, lib, ptr, stream_ptr

MLP_SHAPES = [(64, 32), (16, 64), (64, 32), (64, 64), (16, 64)]
MLP_OFFS = [0, 2048, 3072, 5120, 9216]
MLP_TOTAL = 10240


@dataclass
class NgpConfig:
    # hash grid (Mueller et al. 2022 defaults; finest resolution 2048 * aabb_scale)
    n_levels: int = 16
    log2_hashmap: int = 19
    base_res: int = 16
    aabb_scale: int = 4                  # nerf_fusion.py:68
    # occupancy grid / marching
    grid_size: int = 128
    max_steps_per_ray: int = 1024
    cone_angle: float = 1.0 / 256.0
    # training
    n_rays: int = 4096                   # initial rays per batch; adapted to fill max_samples
    max_rays: int = 1 << 16
    max_samples: int = 1 << 18
    lr: float = 1e-2
    beta1: float = 0.9
    beta2: float = 0.99
    eps: float = 1e-15
    l2_mlp: float = 1e-6
    loss_scale: float = 128.0
    depth_lambda: float = 1.0            # nerf_fusion.py:100
    grid_update_every: int = 16
    grid_decay: float = 0.95
    min_optical_thickness: float = 0.01
    near: float = 0.05
    wgrad_ksplit: int = 256
    optimize_extrinsics: bool = False    # nerf_fusion.py:99 sets it; refine c2w of the training views
    extrinsic_lr_pos: float = 1e-4       # scene units per step (Adam)
    extrinsic_lr_rot: float = 1e-4       # radians per step (Adam)
    grad_fixed_scale: float = 262144.0   # hash-grid gradients accumulate as packed Q18 fixed point (0: f32 atomics)

    @property
    def per_level_scale(self):
        return math.exp(math.log(2048.0 * self.aabb_scale / self.base_res) / (self.n_levels - 1))

    @property
    def n_cascades(self):
        return 1 + int(math.log2(self.aabb_scale))

    @property
    def min_step(self):
        return math.sqrt(3.0) / self.max_steps_per_ray

    @property
    def max_step(self):
        return self.min_step * self.aabb_scale * 8.0


def pack_fixed(g0, g1, scale):
    """host-side twin of csrc/ngp.hip:pack_fixed (tests, diagnostics): int64 word = round(g0 S) + (round(g1 S) << 32)"""
    import numpy as np
    lo = np.rint(np.asarray(g0, np.float64) * scale).astype(np.int64)
    hi = np.rint(np.asarray(g1, np.float64) * scale).astype(np.int64)
    return lo + (hi << 32)


def unpack_fixed(words, scale):
    import numpy as np
    w = np.asarray(words, np.int64)
    lo = (w & 0xffffffff).astype(np.uint32).view(np.int32).astype(np.int64)
    hi = (w - lo) >> 32
    return lo / scale, hi / scale


class NgpNerf:
    def __init__(self, cfg=None, device="cuda:0", seed=1337, group=None, world=None, rank=None):
        """group / world / rank: REPLICATED trainers: every replica holds the full model and the same image
        set, samples its own rays (seed + rank) and the gradients are summed over the replicas before Adam -- one
        all-reduce of the hash-grid gradient (packed fixed-point words add exactly as int64) and one of the MLP gradient."""
        self.cfg = cfg or NgpConfig()
        self.device = torch.device(device)
        self.group = group
        if world is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized() and group is not None
            world, rank = (dist.get_world_size(group), dist.get_rank(group)) if on else (1, 0)
        self.world, self.rank = int(world), int(rank or 0)
        # replicas start from the SAME parameters and keep the same occupancy-grid sampling sequence (base seed); only the
        # ray selection differs (seed + rank), so the summed gradient is the gradient of an R-times larger batch
        base_seed, seed = int(seed), int(seed) + self.rank
        c, dev = self.cfg, self.device
        off = (C.c_uint32 * (c.n_levels + 1))()
        check(lib().ns_ngp_grid_layout(c.n_levels, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale), None,
                                       None, off), "ngp_grid_layout")
        self.n_grid = int(off[c.n_levels]) * 2
        g = torch.Generator(device="cpu").manual_seed(base_seed)
        f = dict(dtype=torch.float32, device=dev)
        self.grid_master = (torch.rand(self.n_grid, generator=g) * 2e-4 - 1e-4).to(dev)
        w = []
        for (o, i) in MLP_SHAPES:  # Xavier uniform
            lim = math.sqrt(6.0 / (o + i))
            w.append((torch.rand(o * i, generator=g) * 2 - 1) * lim)
        self.mlp_master = torch.cat(w).to(dev)
        self.grid_half = self.grid_master.half()
        self.mlp_half = self.mlp_master.half()
        self.grid_grad, self.mlp_grad = torch.zeros(self.n_grid, **f), torch.zeros(MLP_TOTAL, **f)
        self.grid_m1, self.grid_m2 = torch.zeros(self.n_grid, **f), torch.zeros(self.n_grid, **f)
        self.mlp_m1, self.mlp_m2 = torch.zeros(MLP_TOTAL, **f), torch.zeros(MLP_TOTAL, **f)
        G, nc = c.grid_size, c.n_cascades
        self.density_grid = torch.zeros(nc * G ** 3, **f)
        self.bits = torch.full((nc * G ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
        self.step = 0
        self.loss = float("nan")
        self.gen = torch.Generator(device=dev).manual_seed(base_seed)
        self.seed = int(seed)
        # training views
        self.images = self.depths = self.depth_covs = self.c2w = None
        self.intr = None
        self.n_images = 0
        # scratch sized for max_samples
        S = c.max_samples
        h = dict(dtype=torch.float16, device=dev)
        self.s_pos, self.s_dir = torch.empty((S, 3), **f), torch.empty((S, 3), **f)
        self.s_dt, self.s_t = torch.empty(S, **f), torch.empty(S, **f)
        self.s_feat, self.s_out, self.s_dout = torch.empty((S, 32), **h), torch.empty((S, 4), **h), torch.empty((S, 4), **h)
        self.s_dfeat = torch.empty((S, 32), **h)
        self.act = [torch.empty((u, S), **h) for u in (64, 32, 64, 64)]         # h1T cinT h3T h4T
        self.dact = [torch.empty((u, S), **h) for u in (16, 64, 64, 16, 64)]    # d5T d4T d3T ddT d1T
        self.partial = torch.zeros((c.wgrad_ksplit, MLP_TOTAL), **f)
        self.counter = torch.zeros(3, dtype=torch.int32, device=dev)
        ws_bytes = lib().ns_ngp_encode_backward_workspace_bytes(*self._grid_args())
        self.enc_ws = torch.zeros(max(ws_bytes // 4, 1), **f)   # replicated coarse-level gradient tables (kept zeroed)
        self.rays_per_batch = c.n_rays
        self.samples_requested = 0

    # ------------------------------------------------------------------------------------------
    def _grid_args(self):
        c = self.cfg
        return (c.n_levels, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))

    def to_unit(self, pos):
        """NGP scene coordinates -> [0,1]^3 over the render box [0.5 - s/2, 0.5 + s/2]^3."""
        s = float(self.cfg.aabb_scale)
        return ((pos - (0.5 - 0.5 * s)) / s).contiguous()

    def encode(self, pos_unit, out=None):
        """-> features UNIT-MAJOR [32, N] f16 (the layout the MLP kernels read); `out`: flat scratch to write into."""
        N = pos_unit.shape[0]
        out = out.view(-1)[:32 * N].view(32, N) if out is not None else torch.empty((32, N), dtype=torch.float16, device=self.device)
        check(lib().ns_ngp_encode_forward(*self._grid_args(), ptr(pos_unit), ptr(self.grid_half), ptr(out), 1, C.c_long(N),
                                          stream_ptr()), "ngp_encode_forward")
        return out

    # ------------------------------------------------------------------------------------------
    def set_images(self, images, depths, depth_covs, c2w, intr):
        """images [n,H,W,4] f32 linear premultiplied RGBA, depths / depth_covs [n,H,W] f32 (depth <= 0:
        unsupervised pixel), c2w [n,3,4] f32 (camera-to-world in NGP scene coordinates), intr (fx,fy,cx,cy)."""
        dev = self.device
        self.images = images.to(dev, torch.float32).contiguous()
        self.depths = depths.to(dev, torch.float32).contiguous()
        self.depth_covs = depth_covs.to(dev, torch.float32).contiguous()
        self.c2w = c2w.to(dev, torch.float32).contiguous()
        self.intr = [float(v) for v in intr]
        self.n_images = self.images.shape[0]

    def _rays(self, img_idx, u, v):
        fx, fy, cx, cy = self.intr
        d = torch.stack([(u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy, torch.ones_like(u)], -1)
        R, t = self.c2w[img_idx, :, :3], self.c2w[img_idx, :, 3]
        d = torch.einsum("rij,rj->ri", R, d)
        d = d / d.norm(dim=-1, keepdim=True)
        return t.contiguous(), d.contiguous()

    def _t_range(self, o, d):
        s = float(self.cfg.aabb_scale)
        lo, hi = 0.5 - 0.5 * s, 0.5 + 0.5 * s
        inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
        t0, t1 = (lo - o) * inv, (hi - o) * inv
        tmin = torch.minimum(t0, t1).amax(-1).clamp(min=self.cfg.near)
        tmax = torch.maximum(t0, t1).amin(-1)
        return torch.stack([tmin, torch.maximum(tmax, tmin)], -1).contiguous()

    def march(self, o, d, tr, unit=False):
        c = self.cfg
        s = float(c.aabb_scale)
        lo, inv = (0.5 - 0.5 * s, 1.0 / s) if unit else (0.0, 1.0)
        R = o.shape[0]
        self.counter.zero_()
        self.ray_start = torch.empty(R, dtype=torch.int32, device=self.device)
        self.ray_n = torch.empty(R, dtype=torch.int32, device=self.device)
        check(lib().ns_ngp_march(ptr(self.bits), c.grid_size, c.n_cascades, ptr(o), ptr(d), ptr(tr), R,
                                 C.c_float(c.cone_angle), C.c_float(c.min_step), C.c_float(c.max_step),
                                 C.c_float(lo), C.c_float(inv), c.max_steps_per_ray, C.c_long(c.max_samples), ptr(self.counter), ptr(self.ray_start),
                                 ptr(self.ray_n), ptr(self.s_pos), ptr(self.s_dir), ptr(self.s_dt), ptr(self.s_t),
                                 stream_ptr()), "ngp_march")
        # the one host read-back of a step (instant-ngp reads its ray counter too): end of the reserved ranges
        cnt = self.counter.tolist()
        self.samples_requested = cnt[0]     # > max_samples: some rays of this batch received no samples
        return cnt[2]

    def train_step(self):
        if self.n_images == 0:
            return None
        c, dev = self.cfg, self.device
        with torch.cuda.device(dev):
            n, H, W = self.images.shape[:3]
            R = self.rays_per_batch
            f = dict(dtype=torch.float32, device=dev)
            o, d, tr = torch.empty((R, 3), **f), torch.empty((R, 3), **f), torch.empty((R, 2), **f)
            gt_rgb, gt_depth, gt_cov = torch.empty((R, 3), **f), torch.empty(R, **f), torch.empty(R, **f)
            ray_img = torch.empty(R, dtype=torch.int32, device=dev) if c.optimize_extrinsics else None
            s = float(c.aabb_scale)
            fx, fy, cx, cy = self.intr
            seed = (self.seed * 0x9E3779B1 + self.step * 0x85EBCA77) & 0xFFFFFFFF
            check(lib().ns_ngp_sample_rays(ptr(self.images), ptr(self.depths), ptr(self.depth_covs), ptr(self.c2w), n, H, W,
                                           C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                           C.c_float(0.5 - 0.5 * s), C.c_float(0.5 + 0.5 * s), C.c_float(c.near),
                                           C.c_uint32(seed), R, ptr(o), ptr(d), ptr(tr), ptr(gt_rgb), ptr(gt_depth),
                                           ptr(gt_cov), ptr(ray_img), stream_ptr()), "ngp_sample_rays")
            N = self.march(o, d, tr, unit=True)   # positions come back in unit-cube coordinates
            # keep the sample budget filled without refusing rays (instant-ngp adapts its rays per batch likewise)
            want = R * 0.9 * c.max_samples / max(self.samples_requested, 1)
            self.rays_per_batch = int(min(max(want, 256), c.max_rays)) // 128 * 128
            if N == 0:
                if self.world > 1:
                    # replicas stay in lockstep: contribute a zero gradient to this step's all-reduce and take the (shared)
                    # optimiser step like the others -- returning here would leave the peers alone in the collective
                    if c.optimize_extrinsics:
                        self._grow_camera_state(self.n_images)
                    self._allreduce_gradients()
                    if c.optimize_extrinsics:
                        self._camera_step()
                    self._optimizer_step()
                else:
                    self.step += 1
                self.last_samples, self.last_rays = 0, R
                return 0.0
            N8 = (N + 7) // 8 * 8  # the weight-gradient GEMM reads 16-byte runs
            if N8 > N:
                self.s_pos[N:N8] = 0.5
                self.s_dir[N:N8] = 0.0
                self.s_dt[N:N8] = 0.0
            pos_unit = self.s_pos[:N8]
            # forward
            featT = self.encode(pos_unit, self.s_feat)
            acts = [a.view(-1)[:a.shape[0] * N8].view(a.shape[0], N8) for a in self.act]
            dacts = [a.view(-1)[:a.shape[0] * N8].view(a.shape[0], N8) for a in self.dact]
            dfeatT = self.s_dfeat.view(-1)[:32 * N8].view(32, N8)
            check(lib().ns_ngp_mlp_forward(ptr(self.mlp_half), ptr(featT), ptr(self.s_dir), ptr(self.s_out),
                                           *[ptr(a) for a in acts], C.c_long(N8), stream_ptr()), "ngp_mlp_forward")
            out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
            out_depth = torch.empty(R, dtype=torch.float32, device=dev)
            loss = torch.zeros(1, dtype=torch.float32, device=dev)
            if N8 > N:
                self.s_dout[N:N8] = 0
            check(lib().ns_ngp_composite(ptr(self.s_out), ptr(self.s_dt), ptr(self.s_t), ptr(self.ray_start),
                                         ptr(self.ray_n), R, ptr(gt_rgb), ptr(gt_depth), ptr(gt_cov),
                                         C.c_float(c.depth_lambda), C.c_float(c.loss_scale), ptr(out_rgb),
                                         ptr(out_depth), ptr(loss), ptr(self.s_dout), stream_ptr()), "ngp_composite")
            # backward
            check(lib().ns_ngp_mlp_backward(ptr(self.mlp_half), ptr(self.s_dout), ptr(featT), *[ptr(a) for a in acts],
                                            ptr(dfeatT), *[ptr(a) for a in dacts], ptr(self.partial),
                                            c.wgrad_ksplit, ptr(self.mlp_grad), C.c_long(N8), stream_ptr()),
                  "ngp_mlp_backward")
            check(lib().ns_ngp_encode_backward(*self._grid_args(), ptr(pos_unit), ptr(dfeatT), 1,
                                               ptr(self.grid_grad), ptr(self.enc_ws), C.c_float(c.grad_fixed_scale), C.c_long(N8),
                                               stream_ptr()), "ngp_encode_backward")
            if c.optimize_extrinsics:
                self._camera_backward(pos_unit, dfeatT, d, ray_img, N8, R)
            if self.world > 1:
                self._allreduce_gradients()
            if c.optimize_extrinsics:
                self._camera_step()     # after the all-reduce: every replica applies the SAME pose update
            self._optimizer_step()
            self.loss_tensor = loss / (self.ray_n >= 0).sum().clamp(min=1)
            self.last_samples, self.last_rays = N, R
        return self.loss_tensor

    def _optimizer_step(self):
        c = self.cfg
        self.step += 1
        for (m, hp, g, m1, m2, l2, fx) in (
                (self.grid_master, self.grid_half, self.grid_grad, self.grid_m1, self.grid_m2, 0.0, c.grad_fixed_scale),
                (self.mlp_master, self.mlp_half, self.mlp_grad, self.mlp_m1, self.mlp_m2, c.l2_mlp, 0.0)):
            check(lib().ns_ngp_adam(ptr(m), ptr(hp), ptr(g), ptr(m1), ptr(m2), C.c_long(m.numel()), self.step,
                                    C.c_float(c.lr), C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps),
                                    C.c_float(l2), C.c_float(c.loss_scale * self.world), C.c_float(fx), stream_ptr()),
                  "ngp_adam")
        if self.step % c.grid_update_every == 0:
            self.update_density_grid()

    def _allreduce_gradients(self):
        """sum over the replicas; Adam then divides by loss_scale * world (mean gradient)"""
        import torch.distributed as dist
        g = self.grid_grad.view(torch.int64) if self.cfg.grad_fixed_scale > 0 else self.grid_grad
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(self.mlp_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.cfg.optimize_extrinsics and getattr(self, "cam_grad", None) is not None:
            dist.all_reduce(self.cam_grad, op=dist.ReduceOp.SUM, group=self.group)   # 24 B per training view
        self.bytes_allreduced = getattr(self, "bytes_allreduced", 0) + g.numel() * g.element_size() + self.mlp_grad.numel() * 4

    def _camera_backward(self, pos_unit, dfeatT, rays_d, ray_img, N, R):
        """pose refinement: sample-position gradients through the encoding -> per-image 6-dof gradient -> Adam on c2w"""
        c, dev = self.cfg, self.device
        n = self.n_images
        self._grow_camera_state(n)
        dpos = torch.empty((N, 3), dtype=torch.float32, device=dev)
        check(lib().ns_ngp_encode_backward_input(*self._grid_args(), ptr(pos_unit), ptr(self.grid_half), ptr(dfeatT), ptr(dpos),
                                                 C.c_long(N), stream_ptr()), "ngp_encode_backward_input")
        check(lib().ns_ngp_camera_gradient(ptr(dpos), ptr(self.s_t), ptr(rays_d), ptr(self.ray_start), ptr(self.ray_n),
                                           ptr(ray_img), C.c_float(1.0 / float(c.aabb_scale)), ptr(self.cam_grad), R,
                                           stream_ptr()), "ngp_camera_gradient")

    def _grow_camera_state(self, n):
        """per-view Adam moments of the pose refinement: GROWN when keyframes are added (a reset would restart the bias
        correction of every existing view each time the tracker sends a keyframe)"""
        f = dict(dtype=torch.float32, device=self.device)
        if getattr(self, "cam_grad", None) is None:
            self.cam_grad, self.cam_m1, self.cam_m2 = torch.zeros((n, 6), **f), torch.zeros((n, 6), **f), torch.zeros((n, 6), **f)
        elif self.cam_grad.shape[0] < n:
            k = n - self.cam_grad.shape[0]
            self.cam_grad = torch.cat([self.cam_grad, torch.zeros((k, 6), **f)])
            self.cam_m1 = torch.cat([self.cam_m1, torch.zeros((k, 6), **f)])
            self.cam_m2 = torch.cat([self.cam_m2, torch.zeros((k, 6), **f)])
        elif self.cam_grad.shape[0] > n:
            self.cam_grad, self.cam_m1, self.cam_m2 = self.cam_grad[:n].contiguous(), self.cam_m1[:n].contiguous(), self.cam_m2[:n].contiguous()

    def _camera_step(self):
        """Adam + Rodrigues retraction of every training view's c2w from cam_grad (summed over the replicas when world > 1:
        the gradient scale loss_scale * world makes it the mean, like the model gradients)"""
        c, n = self.cfg, self.n_images
        check(lib().ns_ngp_camera_step(ptr(self.c2w), ptr(self.cam_grad), ptr(self.cam_m1), ptr(self.cam_m2), n, self.step + 1,
                                       C.c_float(c.extrinsic_lr_pos), C.c_float(c.extrinsic_lr_rot), C.c_float(c.beta1),
                                       C.c_float(c.beta2), C.c_float(c.eps), C.c_float(c.loss_scale * self.world), stream_ptr()),
              "ngp_camera_step")

    # ------------------------------------------------------------------------------------------
    def density_at(self, pos_scene):
        """sigma at scene positions [N,3] (encode + density half of the network)."""
        N = pos_scene.shape[0]
        if N % 2:  # a lane of the MLP kernel owns two adjacent samples
            return self.density_at(torch.cat([pos_scene, pos_scene[-1:]], 0))[:N]
        out = torch.empty((N, 4), dtype=torch.float16, device=self.device)
        feat = self.encode(self.to_unit(pos_scene))
        dirs = torch.zeros((N, 3), dtype=torch.float32, device=self.device)
        nul = C.c_void_p(0)
        check(lib().ns_ngp_mlp_forward(ptr(self.mlp_half), ptr(feat), ptr(dirs), ptr(out), nul, nul, nul, nul,
                                       C.c_long(N), stream_ptr()), "ngp_mlp_forward")
        return out[:, 3].float().exp()

    def update_density_grid(self, n_cells=1 << 18):
        """EMA of the density sampled at jittered cell centres of a random subset of cells; a cell is
        occupied when density * min_step exceeds min(mean, threshold) (instant-ngp's rule)."""
        c, dev = self.cfg, self.device
        G, nc = c.grid_size, c.n_cascades
        total = nc * G ** 3
        cells = torch.randint(0, total, (min(n_cells, total),), device=dev, generator=self.gen)
        mip = cells // (G ** 3)
        r = cells % (G ** 3)
        xyz = torch.stack([r % G, (r // G) % G, r // (G * G)], -1).float()
        jit = torch.rand(xyz.shape, device=dev, generator=self.gen)
        scale = (2.0 ** mip.float())[:, None]
        pos = ((xyz + jit) / G - 0.5) * scale + 0.5
        dens = self.density_at(pos.contiguous()) * c.min_step
        self.density_grid.mul_(c.grid_decay)
        self.density_grid[cells] = torch.maximum(self.density_grid[cells], dens)
        thr = min(float(self.density_grid.mean()), c.min_optical_thickness)
        occ = (self.density_grid > thr).view(-1, 8).to(torch.uint8)
        wts = (2 ** torch.arange(8, device=dev, dtype=torch.uint8))
        self.bits = (occ * wts).sum(-1).to(torch.uint8).contiguous()

    @torch.no_grad()
    def render(self, c2w, H, W, intr=None, chunk=4096):
        """-> (rgb [H,W,3], depth [H,W]) of one view (inference path of B7)."""
        dev, c = self.device, self.cfg
        fx, fy, cx, cy = intr or self.intr
        vv, uu = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        u, v = uu.reshape(-1).float(), vv.reshape(-1).float()
        d = torch.stack([(u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy, torch.ones_like(u)], -1)
        c2w = c2w.to(dev, torch.float32)
        d = d @ c2w[:, :3].t()
        d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
        o = c2w[:, 3].expand_as(d).contiguous()
        rgb = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
        dep = torch.empty((H * W,), dtype=torch.float32, device=dev)
        nul = C.c_void_p(0)
        with torch.cuda.device(dev):
            s = 0
            while s < H * W:
                oo, dd = o[s:s + chunk].contiguous(), d[s:s + chunk].contiguous()
                N = self.march(oo, dd, self._t_range(oo, dd))
                R = oo.shape[0]
                if self.samples_requested > c.max_samples and R > 1:
                    chunk = max(1, R // 2)   # a ray was refused: every pixel must be rendered, retry with fewer rays
                    continue
                orgb = torch.zeros((R, 3), dtype=torch.float32, device=dev)
                odep = torch.zeros(R, dtype=torch.float32, device=dev)
                if N > 0:
                    Ne = N + (N & 1)
                    if Ne > N:
                        self.s_pos[N:Ne], self.s_dir[N:Ne] = 0.5, 0.0
                    featT = self.encode(self.to_unit(self.s_pos[:Ne]), self.s_feat)
                    check(lib().ns_ngp_mlp_forward(ptr(self.mlp_half), ptr(featT), ptr(self.s_dir), ptr(self.s_out),
                                                   nul, nul, nul, nul, C.c_long(Ne), stream_ptr()), "ngp_mlp_forward")
                    check(lib().ns_ngp_composite(ptr(self.s_out), ptr(self.s_dt), ptr(self.s_t), ptr(self.ray_start),
                                                 ptr(self.ray_n), R, nul, nul, nul, C.c_float(0), C.c_float(1), ptr(orgb),
                                                 ptr(odep), nul, nul, stream_ptr()), "ngp_composite")
                rgb[s:s + R], dep[s:s + R] = orgb, odep
                s += R
        return rgb.view(H, W, 3), dep.view(H, W)
