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
import os
from dataclasses import dataclass

import torch

from ._lib import NerfSlamHipError, check, lib, ptr, stream_ptr, variant_env

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
    # round 6: the occupancy refresh rides on the LAST step before an update, on the ray stream ahead of the next step's rays, from
    # parameters one step older than the eager refresh saw -- instead of ~150 us of refresh + ~110 us of re-marching between two
    # graph replays once per 16 steps (5 % of the mapping leg).  False: refresh eagerly after the step, re-march the next rays.
    refresh_in_step: bool = True
    steps_per_graph: int = 8             # optimiser steps replayed per HIP-graph launch of train_steps() (even; <= grid_update_every).
                                         # Pipeline, two runs per arm in one call: 2: 134.5 / 135.8, 4: 138.8 / 137.6, 8: 137.5 / 139.9,
                                         # 16: 137.2 / 140.0 frames/s (profiles/r06_ab_records.json); 8 = 2.8 ms between two polls of
                                         # the packet queue
    state_record_floats: int = 6         # the table's optimiser state: one record of 6 floats (24 B: master.xy, m1.xy, m2.xy) per
                                         # entry -- or 8 (round 5's 32-byte record; its 8 unused bytes were a quarter of the flush's
                                         # traffic once a step touches nearly every line of the state: 2^22 gradient resolution) --
                                         # told to the library through the `_rec` entry points (include/nerfslam_hip.h)
    wgrad_after_scatter: bool = False    # A/B hook: the MLP weight-gradient chain forks AFTER the scatter pass instead of beside it
    grid_rule: str = "subset"            # occupancy refresh: "subset" (2^18 uniform cells per update) | "ngp" (instant-ngp's rule)
    grid_decay: float = 0.95
    grid_decay_all: bool = True          # subset rule: EVERY cell decays on every update, as in instant-ngp's rule (max(prev * decay,
                                         # new), new = 0 for cells not drawn), at grid_decay ** (cells drawn / half the grid): the same
                                         # fading per re-evaluation as instant-ngp, which redraws half the grid per update.  False:
                                         # only the drawn cells decay (a once-occupied cell then never fades unless it is drawn)
    min_optical_thickness: float = 0.01
    near: float = 0.05
    optimize_extrinsics: bool = False    # nerf_fusion.py:99 sets it; refine c2w of the training views
    extrinsic_lr_pos: float = 1e-4       # scene units per step (Adam)
    extrinsic_lr_rot: float = 1e-4       # radians per step (Adam)
    # hash-grid gradients accumulate as packed fixed point, 2^-22 of the 128x loss-scaled gradient per unit (0: f32 atomics).
    # Rounds 2-5 used 2^18: a corner contribution below 1.5e-8 of the real gradient was dropped -- ~90 % of them in a converged
    # scene -- which cost 1.3-2.2 dB of PSNR and 0.25-0.7 mm of depth against the f32 gradient (tools/q_scale_sweep.py, three
    # seeds: profiles/r06_ab_records.json); at 2^22 the step costs the same 0.27 ms and trains within the seeds' spread of f32.
    # A single contribution saturates at |g| >= 2^24 / scale = 4 (scaled), an entry's sum at 2^31 / scale = 512.
    grad_fixed_scale: float = 4194304.0
    use_graph: bool = True               # replay the training step from a HIP graph (False: same launch sequence, eager)

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
    def __init__(self, cfg=None, device="cuda:0", seed=1337, group=None, world=None, rank=None, replicated=None):
        """group / world / rank: REPLICATED trainers: every replica holds the full model and the same image
        set, samples its own rays (seed + rank) and the gradients are summed over the replicas before Adam -- one
        all-reduce of the hash-grid gradient (packed fixed-point words add exactly as int64) and one of the MLP gradient.
        replicated=True runs the replicated trainers' launch sequence (gradient buffer, collectives, sharded Adam, parameter
        gather) even in a group of ONE: the RCCL path of the step is then executable on a one-GPU box (tests/test_rccl_gpu.py)."""
        self.cfg = cfg or NgpConfig()
        self.device = torch.device(device)
        self.group = group
        if world is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized() and group is not None
            world, rank = (dist.get_world_size(group), dist.get_rank(group)) if on else (1, 0)
        self.world, self.rank = int(world), int(rank or 0)
        self.replicated = self.world > 1 if replicated is None else bool(replicated) or self.world > 1
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
        # Optimiser state of the table: ONE record per entry, [master.xy | m1.xy | m2.xy] (24 bytes; round 5: 32 with 8 unused)
        # (csrc/ngp.hip: adam_entry_stride).  grid_master / grid_m1 / grid_m2 are [entries, 2] VIEWS of it; the library is told the
        # record size (`_rec` entry points), and an entry the step touches costs one 128-byte line instead of three.
        self.state_rec = int(c.state_record_floats)
        if self.state_rec not in (6, 8):
            raise ValueError("NgpConfig.state_record_floats is 6 or 8")
        self.grid_state, self.grid_master, self.grid_m1, self.grid_m2 = self.new_grid_state(self.n_grid // 2, dev, self.state_rec)
        if variant_env("NS_ADAM_SEPARATE"):      # A/B: rounds 2-4's three dense arrays (same arithmetic, same bits)
            self.grid_state = None
            self.state_rec = 2
            self.grid_master, self.grid_m1, self.grid_m2 = (torch.zeros((self.n_grid // 2, 2), **f) for _ in range(3))
        self.grid_master.copy_((torch.rand(self.n_grid, generator=g) * 2e-4 - 1e-4).view(-1, 2))
        w = []
        for (o, i) in MLP_SHAPES:  # Xavier uniform
            lim = math.sqrt(6.0 / (o + i))
            w.append((torch.rand(o * i, generator=g) * 2 - 1) * lim)
        self.mlp_master = torch.cat(w).to(dev)
        # Replicated trainers exchange WHAT A STEP TOUCHED (round 6; rounds 3-5: the dense packed gradient, 100 MB per trainer and
        # step through an all-to-all, then the 25-MB f16 table through an all-gather -- a one-rank replicated step cost 2.07 x the
        # one-trainer step before a byte reached xGMI).  The table gradient's flush appends (entry, packed integer sum) pairs to
        # `_emit_list` -- at most one per entry: ~0.9 M of the default grid's 12.6 M in a converged scene -- the lists' agreed-length
        # prefixes are all-gathered (one link per pair of trainers on the xGMI mesh), and EVERY trainer adds the lists per entry
        # (64-bit integer atomics: exact, order-free) and applies Adam to every touched entry (ns_ngp_sparse_table_update): the same
        # sums, the same update, the same bits everywhere -- no parameter travels back.  `grid_grad`, viewed as one int64 word per
        # entry, is the accumulator (zero between steps).
        n_entries = self.n_grid // 2
        self.grid_half = self.grid_master.reshape(-1).half()
        self.mlp_half = self.mlp_master.half()
        self.grid_grad, self.mlp_grad = torch.zeros(self.n_grid, **f), torch.zeros(MLP_TOTAL, **f)
        self.sparse_exchange = self.replicated and c.grad_fixed_scale > 0
        if self.sparse_exchange:
            from .parallel import list_class
            self._emit_list = torch.zeros((list_class(n_entries), 2), dtype=torch.int64, device=dev)   # (worst case: every entry touched)
            self._emit_count = torch.zeros(1, dtype=torch.int32, device=dev)
            self._lists = None               # [world, class, 2] int64, grown on demand
            self.wire_log = []               # pairs exchanged per step (bench.py: rccl_per_trainer)
        self.mlp_m1, self.mlp_m2 = torch.zeros(MLP_TOTAL, **f), torch.zeros(MLP_TOTAL, **f)
        G, nc = c.grid_size, c.n_cascades
        self.density_grid = torch.zeros(nc * G ** 3, **f)
        self.bits = torch.full((nc * G ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
        self.step = 0
        self.loss = float("nan")
        self._static = False
        self.gen = torch.Generator(device=dev).manual_seed(base_seed)
        self.seed = int(seed)
        self.base_seed = base_seed
        # training views
        self.images = self.depths = self.depth_covs = self.c2w = None
        self.intr = None
        self.n_images = 0
        # scratch sized for max_samples
        S = c.max_samples
        h = dict(dtype=torch.float16, device=dev)
        # sample arrays of the RENDER path (render() / march()); the training step has its own two sets (_alloc_static)
        self.s_pos, self.s_dir = torch.empty((S, 3), **f), torch.empty((S, 3), **f)
        self.s_dt, self.s_t = torch.empty(S, **f), torch.empty(S, **f)
        # (zero-initialised, not torch.empty: the training step skips the unused tail of the budget, and what it skips must
        # never hold NaN bits -- see csrc/ngp.hip:ngp_encode_fwd_kernel)
        self.s_feat, self.s_out = torch.zeros((S, 32), **h), torch.zeros((S, 4), **h)
        self.s_dfeat = torch.zeros((S, 32), **h)
        # (rounds 3-4 also kept [units][S] buffers of every hidden activation / gradient here -- 235 MB -- for the superseded
        #  weight-gradient kernel over STORED activations; the product recomputes them on chip, csrc/ngp_mlp_wgrad.hip)
        self.mlp_wgs = 512               # slabs of partial weight gradients the workspace holds (384 .. 1024 measured: within 2 %)
        self.relu_masks = torch.zeros(6 * S, dtype=torch.int32, device=dev)     # one bit per hidden unit and sample (csrc/ngp_mlp.hip)
        # Table gradient.  One trainer: the round-3 path (csrc/ngp.hip: no count pass, Adam applied to the touched entries in
        # the flush of the accumulation; the gradient buffer is not used).  Replicated trainers: gradient buffer + all-reduce
        # + streaming Adam, so the binned path that writes the buffer keeps its own workspace.
        self.fused_adam = not self.replicated and c.grad_fixed_scale > 0
        self.enc_ws_bytes = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*self._grid_args(), C.c_long(c.max_samples)))
        self.fused_ws = c.grad_fixed_scale > 0 and self.enc_ws_bytes > 0
        if self.fused_ws:
            self.enc_ws = torch.zeros(self.enc_ws_bytes // 8 + 1, dtype=torch.int64, device=dev)   # zeroed once: overflow counter
        else:
            self.fused_adam = False
            ws_bytes = lib().ns_ngp_encode_backward_workspace_bytes(*self._grid_args(), C.c_long(c.max_samples))
            self.enc_ws = torch.zeros(max(ws_bytes // 4, 1), **f)   # record queues / counters of round 2's binned encode backward
        self.rays_per_batch = c.n_rays
        self.samples_requested = 0

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def new_grid_state(n_entries, device, record_floats=8):
        """(records [n_entries, record_floats] f32, and the master / m1 / m2 views [n_entries, 2] into them) -- the interleaved
        optimiser state: m1 == master + 2 floats, m2 == master + 4 floats.  record_floats = 8 is what the entry points WITHOUT
        `_rec` recognise from the pointers (rounds 5's layout: the default here, for their callers); 6 (the trainer's) has to be
        passed to the `_rec` entry points."""
        assert record_floats in (6, 8)
        rec = torch.zeros((int(n_entries), int(record_floats)), dtype=torch.float32, device=device)
        return rec, rec[:, 0:2], rec[:, 2:4], rec[:, 4:6]

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
    def set_images(self, images, depths, depth_covs, c2w, intr, n_images=None):
        """images [n,H,W,4] f32 linear premultiplied RGBA, depths / depth_covs [n,H,W] f32 (depth <= 0:
        unsupervised pixel), c2w [n,3,4] f32 (camera-to-world in NGP scene coordinates), intr (fx,fy,cx,cy).
        n_images: number of VALID leading views when the tensors are whole pre-allocated slot arrays (their addresses then
        never change as keyframes arrive, and the captured training step stays valid)."""
        dev = self.device
        self.images = images.to(dev, torch.float32).contiguous()
        self.depths = depths.to(dev, torch.float32).contiguous()
        self.depth_covs = depth_covs.to(dev, torch.float32).contiguous()
        self.c2w = c2w.to(dev, torch.float32).contiguous()
        self.intr = [float(v) for v in intr]
        self.n_images = self.images.shape[0] if n_images is None else int(n_images)
        if getattr(self, "_static", False):
            self.set_views(self.n_images)

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
        # (rendering path: its own counters / ray tables / sample arrays -- the training step's are part of captured graphs and
        # hold the rays marched ahead for the next step)
        self.rm_counter = torch.zeros(3, dtype=torch.int32, device=self.device)
        self.rm_start = torch.empty(R, dtype=torch.int32, device=self.device)
        self.rm_n = torch.empty(R, dtype=torch.int32, device=self.device)
        check(lib().ns_ngp_march(ptr(self.bits), c.grid_size, c.n_cascades, ptr(o), ptr(d), ptr(tr), R,
                                 C.c_float(c.cone_angle), C.c_float(c.min_step), C.c_float(c.max_step),
                                 C.c_float(lo), C.c_float(inv), c.max_steps_per_ray, C.c_long(c.max_samples), ptr(self.rm_counter), ptr(self.rm_start),
                                 ptr(self.rm_n), ptr(self.s_pos), ptr(self.s_dir), ptr(self.s_dt), ptr(self.s_t),
                                 stream_ptr()), "ngp_march")
        cnt = self.rm_counter.tolist()      # host read-back: rendering sizes its launches by the sample count
        self.samples_requested = cnt[0]     # > max_samples: some rays of this batch received no samples
        return cnt[2]

    # ------------------------------------------------------------------------------------------
    # Training step = a FIXED launch sequence with fixed arguments: the per-step scalars (ray count, seed,
    # step number, number of training views) live in the device control block `ctl`, every per-sample kernel runs over the
    # whole sample budget (slots the marcher did not fill carry a zero upstream gradient), and nothing is read back.  The
    # sequence is therefore captured ONCE in a HIP graph and replayed -- the eager form of the same sequence is what the
    # replicated-trainer mode runs (its all-reduces sit between the backward pass and the optimiser).
    def _alloc_static(self):
        c, dev = self.cfg, self.device
        f = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        h = dict(dtype=torch.float16, device=dev)
        S = c.max_samples
        Rc = self.ray_cap = int(min(c.max_rays, 16384))
        import struct
        fbits = lambda x: struct.unpack("<i", struct.pack("<f", x))[0]
        # TWO sets of {control block, march counters, ray tables, sample arrays, loss gradient}: step k trains on one while the
        # rays of step k + 1 are sampled and marched into the other (side stream, from the START of step k's launch sequence)
        self.sets = []
        for _ in range(2):
            t = dict(r_o=torch.zeros((Rc, 3), **f), r_d=torch.zeros((Rc, 3), **f), r_tr=torch.zeros((Rc, 2), **f),
                     r_rgb=torch.zeros((Rc, 3), **f), r_depth=torch.zeros(Rc, **f), r_cov=torch.ones(Rc, **f),
                     r_img=torch.zeros(Rc, **i32), ray_start=torch.zeros(Rc, **i32), ray_n=torch.full((Rc,), -1, **i32),
                     # unused sample slots must hold finite inputs: the per-sample kernels run over all of them
                     s_pos=torch.full((S, 3), 0.5, **f), s_dir=torch.zeros((S, 3), **f), s_dt=torch.zeros(S, **f),
                     s_t=torch.zeros(S, **f), s_dout=torch.zeros((S, 4), **h), counter=torch.zeros(3, **i32),
                     loss=torch.zeros(Rc, **f),       # per ray (summed when the loss is read)
                     order=torch.zeros(Rc // 16 + 4, dtype=torch.int64, device=dev),     # the marcher's ordered-range words (+ out / ticket counters)
                     ctl=torch.tensor([self.step, min(self.rays_per_batch, Rc), self.seed & 0x7FFFFFFF, max(self.n_images, 1),
                                       fbits(1.0 - c.beta1 ** (self.step + 1)), fbits(1.0 - c.beta2 ** (self.step + 1)), 0, 0], **i32))
            self.sets.append(t)
        self.cur = 0                                     # set of the step about to be trained
        self.out_rgb, self.out_depth = torch.zeros((Rc, 3), **f), torch.zeros(Rc, **f)
        self.dpos = torch.zeros((c.max_samples, 3), **f)
        self.ray_g = torch.zeros((Rc, 6), **f)
        self.last = torch.zeros(4, **i32)
        self._graphs, self._graph_key, self._pair, self._chains = [None, None], None, None, {}
        self._graphs_r, self._pair_r, self._refresh_seen = [None, None], None, False     # the variants that carry the occupancy refresh
        self._grid_scratch()
        self._side = torch.cuda.Stream(device=dev)       # next step's ray marching, then weight / pose gradients
        self._side2 = torch.cuda.Stream(device=dev)      # dense levels of the table gradient
        self._primed = False
        self._static = True

    @property
    def ctl(self):
        """control block of the step about to be trained (device int32[8]: step, rays, seed, views, Adam c1 / c2 bits)"""
        return self.sets[self.cur]["ctl"]

    def set_views(self, n):
        for t in self.sets:
            t["ctl"][3] = max(int(n), 1)

    def _step_key(self):
        c = self.cfg
        return (self.images.data_ptr(), self.depths.data_ptr(), self.depth_covs.data_ptr(), self.c2w.data_ptr(),
                tuple(self.images.shape[1:3]), tuple(self.intr), c.depth_lambda, c.optimize_extrinsics,
                None if getattr(self, "cam_grad", None) is None else self.cam_grad.data_ptr(), self.bits.data_ptr())

    def _enqueue_rays(self, t):
        """sample the rays of the step set `t` belongs to (its control block names step, seed and ray count) and march them:
        fills the set's ray tables, sample arrays and march counters (which must be zero)"""
        c, L = self.cfg, lib()
        S, Rc = c.max_samples, self.ray_cap
        n_cap, H, W = self.images.shape[:3]
        s = float(c.aabb_scale)
        fx, fy, cx, cy = self.intr
        st, ctl = stream_ptr(), ptr(t["ctl"])
        # (no per-step clears: the marcher writes ray_n of every ray of the batch, the composite pass the loss gradient of every
        #  marched sample, and the MLP backward kernels mask the <= 7 slots between the sample count and its multiple of 8)
        check(L.ns_ngp_sample_rays_ctl(ptr(self.images), ptr(self.depths), ptr(self.depth_covs), ptr(self.c2w), n_cap, H, W,
                                       C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                       C.c_float(0.5 - 0.5 * s), C.c_float(0.5 + 0.5 * s), C.c_float(c.near), C.c_uint32(0), Rc,
                                       ptr(t["r_o"]), ptr(t["r_d"]), ptr(t["r_tr"]), ptr(t["r_rgb"]), ptr(t["r_depth"]),
                                       ptr(t["r_cov"]), ptr(t["r_img"]), ctl, 0, st), "ngp_sample_rays")
        # (ranges in workgroup order: the batch is the same on every run, csrc/ngp.hip: ngp_march_kernel's ordered mode)
        check(L.ns_ngp_march_ordered(ptr(self.bits), c.grid_size, c.n_cascades, ptr(t["r_o"]), ptr(t["r_d"]), ptr(t["r_tr"]), Rc,
                                     C.c_float(c.cone_angle), C.c_float(c.min_step), C.c_float(c.max_step),
                                     C.c_float(0.5 - 0.5 * s), C.c_float(1.0 / s), c.max_steps_per_ray, C.c_long(S), ptr(t["counter"]),
                                     ptr(t["ray_start"]), ptr(t["ray_n"]), ptr(t["s_pos"]), ptr(t["s_dir"]), ptr(t["s_dt"]),
                                     ptr(t["s_t"]), ctl, None if variant_env("NS_MARCH_UNORDERED") else ptr(t["order"]), st), "ngp_march")

    @property
    def mlp_frags(self):
        """the fragment table the NEXT optimiser step reads (complete and current: written by the last step's optimiser)"""
        return self.mlp_frags2[self.cur]

    def _enqueue_step(self, x, phase="all", refresh=False):
        """one optimiser step on set `x` on the current stream (+ two side streams); no host synchronisation, no allocation.
        phase="pre" (replicated trainers): everything up to the gradient exchange; returns the closure that enqueues what follows it.

        main  : encode (+ Jacobian rows) -> MLP forward (bit masks) -> composite (loss per ray) -> [fork 1] activation gradients
                -> [fork 2] table gradient of the HASHED levels with Adam in its flush (scatter, accumulate)
        side  : [from the start] control block of step k + 1, its rays sampled and marched into the OTHER set (the marcher is
                latency bound and reads only the occupancy bits and the images); [1] MLP weight gradients from a recomputed
                forward, reduce, MLP Adam, fragment pack; [2] pose refinement: Jacobian dot, camera gradient, pose step
        side2 : [2] table gradient of the DENSE levels (LDS-atomic bound) with Adam in its reduce
        Under capture the ORDER of the calls below decides which hardware queue a branch gets: the main stream's
        kernels are enqueued first after every fork, the ray branch first at the start of the step."""
        c, dev = self.cfg, self.device
        L = lib()
        S, Rc = c.max_samples, self.ray_cap
        s = float(c.aabb_scale)
        X, Y = self.sets[x], self.sets[1 - x]
        st = stream_ptr()
        ctl = ptr(X["ctl"])
        main = torch.cuda.current_stream()
        probe = getattr(self, "_probe", None)

        def mark(name):
            """probe_steps(): a timing event on the CURRENT stream; kernel `name` is what ran since the previous mark of that
            stream.  (Never under capture: the probed step runs eagerly.)"""
            if probe is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                probe.append((torch.cuda.current_stream().cuda_stream, name, ev))
        # (under capture the ORDER of these calls decides which hardware queue a branch gets: with the forward pass enqueued
        #  before this branch, the executor put the branch behind side2's kernels on one queue and the step took 0.58 ms; the
        #  same swap in the second step of a paired graph only: 0.43 -> 0.50 ms)
        # (round 4: the next step's rays go to the THIRD stream, which is idle until the pose chain forks after the activation
        #  gradients; on `side` the march (~95 us of latency-bound work in a handful of waves) held the weight-gradient kernel back
        #  until the scatter had filled the CUs, and that kernel needs whole SIMDs.  The A/B switches that moved this branch and
        #  the pose chain back to `side` are gone: with the rays on one side stream and the pose step on the other,
        #  the next step's ray sampling read c2w unordered against camera_step's write.  Both sit on `side2`, in stream order.)
        ray_stream = self._side2
        ray_stream.wait_stream(main)
        with torch.cuda.stream(ray_stream):
            check(L.ns_ngp_step_prepare(ptr(X["ctl"]), ptr(Y["ctl"]), ptr(X["counter"]), ptr(Y["counter"]), ptr(self.last),
                                        C.c_float(0.9), C.c_long(S), 256, Rc, C.c_float(c.beta1), C.c_float(c.beta2), None,
                                        stream_ptr()), "ngp_step_prepare")
            ev_params_read = None
            if refresh:
                # the occupancy refresh of the update that follows this step, AHEAD of the next step's rays (they are marched on
                # the refreshed bits: nothing to march again) and beside this step's forward / backward passes; it reads the table
                # and the MLP as the previous step left them -- this step's optimiser kernels wait for `ev_params_read`
                mark(None)
                ev_params_read = torch.cuda.Event()
                self._enqueue_grid_refresh(stream_ptr(), ctl=ptr(X["ctl"]), after_params_read=ev_params_read.record)
                mark("occupancy refresh (in step)")
            mark(None)
            self._enqueue_rays(Y)
            mark("ngp_sample_rays + ngp_march (next step's rays)")
        # sample count of THIS step = end of the marcher's reserved ranges (device memory): the per-sample kernels are launched
        # over the whole budget S (fixed grids, fixed row strides) and skip the tail the marcher did not fill
        n_dev = C.c_void_p(X["counter"].data_ptr() + 8)
        featT = self.s_feat.view(-1)[:32 * S].view(32, S)
        jac = None
        if c.optimize_extrinsics:
            # the forward pass also writes d(feature)/d(position): the pose refinement's input gradient is then a dot product
            if getattr(self, "s_jac", None) is None:
                self.s_jac = torch.zeros((6 * c.n_levels, S), dtype=torch.float16, device=dev)
            jac = self.s_jac
        mark(None)
        check(L.ns_ngp_encode_forward_j_n(*self._grid_args(), ptr(X["s_pos"]), ptr(self.grid_half), ptr(featT), 1, ptr(jac),
                                          C.c_long(S), n_dev, st), "ngp_encode_forward")
        mark("ngp_encode_fwd_kernel")
        # MLP backward: bit-mask activation gradients WITHOUT their five gradient stores on this stream + the weight
        # gradients recomputed on chip on a side stream next to the table gradient.  (The forms this replaced -- everything in one
        # kernel on this stream; a weight-gradient kernel over stored activations -- live on as reference kernels of
        # tests/test_ngp_gpu.py::test_mlp_forward_backward only.)
        if getattr(self, "partial_fused", None) is None:     # (first step after construction: eager)
            self.partial_fused = torch.zeros((self.mlp_wgs, MLP_TOTAL), dtype=torch.float32, device=dev)
            # the weights in MFMA operand order (forward + transposed fragments), TWO tables: step k reads table k & 1 and its
            # optimiser writes table (k + 1) & 1 -- the MLP's optimiser runs on a side stream while the activation-gradient
            # kernel of the same step may still be reading the table on the main one (round 4, first form: ONE table, ordered
            # only by the weight-gradient kernel in front of the optimiser taking longer than that kernel; making the optimiser
            # wait for the main stream's event instead cost the pipeline 10 %: 133-134 -> 117-121 frames/s, step alone unchanged)
            nfr = int(L.ns_ngp_mlp_fragment_table_bytes()) // 2
            self.mlp_frags2 = torch.zeros((2, nfr), dtype=torch.float16, device=dev)
            for k in (0, 1):
                check(L.ns_ngp_mlp_pack_fragments(ptr(self.mlp_half), ptr(self.mlp_frags2[k]), st), "ngp_mlp_pack_fragments")
        fr_r, fr_w = self.mlp_frags2[x], self.mlp_frags2[1 - x]      # read by this step / written for the next one
        check(L.ns_ngp_mlp_forward_f_n(ptr(fr_r), ptr(featT), ptr(X["s_dir"]), ptr(self.s_out), ptr(self.relu_masks),
                                       C.c_long(S), n_dev, st), "ngp_mlp_forward")
        mark("ngp_mlp_fwd_kernel")
        check(L.ns_ngp_composite_rays(ptr(self.s_out), ptr(X["s_dt"]), ptr(X["s_t"]), ptr(X["ray_start"]), ptr(X["ray_n"]), Rc,
                                      ptr(X["r_rgb"]), ptr(X["r_depth"]), ptr(X["r_cov"]), C.c_float(c.depth_lambda),
                                      C.c_float(c.loss_scale), ptr(self.out_rgb), ptr(self.out_depth), None, ptr(X["loss"]),
                                      ptr(X["s_dout"]), ctl, st), "ngp_composite")
        mark("ngp_composite_kernel")
        single = not self.replicated
        pose = c.optimize_extrinsics

        # ---- the pieces ----
        def adam(m, hp, g, m1, m2, l2, fxs, stream):
            rec = self.state_rec if m is self.grid_master else 2          # (the MLP's state: three dense arrays)
            check(L.ns_ngp_adam_rec_ctl(ptr(m), ptr(hp), ptr(g), ptr(m1), ptr(m2), rec, C.c_long(m.numel()), 0, C.c_float(c.lr),
                                        C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps), C.c_float(l2),
                                        C.c_float(c.loss_scale * self.world), C.c_float(fxs), ctl, stream), "ngp_adam")
        mlp = (self.mlp_master, self.mlp_half, self.mlp_grad, self.mlp_m1, self.mlp_m2, c.l2_mlp, 0.0)

        def mlp_adam(stream):
            adam(*mlp, stream)
            # the fragment table follows the weights (read by the next step's forward / backward kernels)
            check(L.ns_ngp_mlp_pack_fragments(ptr(self.mlp_half), ptr(fr_w), stream), "ngp_mlp_pack_fragments")

        def camera_step(stream):
            check(L.ns_ngp_camera_step_ctl(ptr(self.c2w), ptr(self.cam_grad), ptr(self.cam_m1), ptr(self.cam_m2),
                                           self.cam_grad.shape[0], 0, C.c_float(c.extrinsic_lr_pos), C.c_float(c.extrinsic_lr_rot),
                                           C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps),
                                           C.c_float(c.loss_scale * self.world), ctl, stream), "ngp_camera_step")

        def pose_gradient(stream):
            """input gradient of the encoding (a dot product with the Jacobian rows the forward pass wrote) -> per-ray / per-image
            6-dof gradients.  (The form that gathered the table a second time -- and had to be ordered against the optimiser
            rewriting it -- is gone; its kernel stays a reference of tests/test_ngp_gpu.py.)"""
            check(L.ns_ngp_encode_jacobian_dot_n(*self._grid_args(), ptr(jac), ptr(self.s_dfeat), ptr(self.dpos),
                                                 C.c_long(S), n_dev, stream), "ngp_encode_jacobian_dot")
            n_cam = self.cam_grad.shape[0]
            check(L.ns_ngp_camera_gradient_2stage(ptr(self.dpos), ptr(X["s_t"]), ptr(X["r_d"]), ptr(X["ray_start"]),
                                                  ptr(X["ray_n"]), ptr(X["r_img"]), C.c_float(1.0 / s), ptr(self.cam_grad), Rc,
                                                  ctl, ptr(self.ray_g) if n_cam <= 4096 else None, n_cam, stream), "ngp_camera_gradient")

        def table_gradient(parts, stream):
            # one trainer: Adam in the flushes (no gradient buffer); replicated trainers: the touched entries into the list
            fa = self.fused_adam
            if self.sparse_exchange:
                check(L.ns_ngp_encode_backward_fused_emit_n(*self._grid_args(), ptr(X["s_pos"]), ptr(self.s_dfeat), ptr(self.enc_ws),
                                                            C.c_size_t(self.enc_ws_bytes), C.c_float(c.grad_fixed_scale), C.c_long(S),
                                                            n_dev, ptr(self._emit_list), ptr(self._emit_count), parts, stream),
                      "ngp_encode_backward_fused_emit")
                return
            check(L.ns_ngp_encode_backward_fused_rec_n(*self._grid_args(), ptr(X["s_pos"]), ptr(self.s_dfeat),
                                                       None if fa else ptr(self.grid_grad), ptr(self.enc_ws),
                                                       C.c_size_t(self.enc_ws_bytes), C.c_float(c.grad_fixed_scale), C.c_long(S), n_dev,
                                                       ptr(self.grid_master) if fa else None, ptr(self.grid_half) if fa else None,
                                                       ptr(self.grid_m1) if fa else None, ptr(self.grid_m2) if fa else None,
                                                       self.state_rec, 0, C.c_float(c.lr), C.c_float(c.beta1), C.c_float(c.beta2),
                                                       C.c_float(c.eps), C.c_float(c.loss_scale * self.world), ctl, parts, stream),
                  "ngp_encode_backward_fused")

        # ---- THREE streams (a HIP graph runs its branches on a handful of hardware queues: a fourth concurrent branch shared a
        #      queue with the main one and the table gradient waited behind the pose refinement, +100 us).  Every fork costs the
        #      main stream ~10 us; ONE two-way fork after the activation gradients cost it 29 us and started all three branches'
        #      heaviest kernels at the same instant (0.46 -> 0.50 ms), so the two forks stay apart:
        #   main  : [fork 1] activation gradients [fork 2] table gradient (scatter, accumulate + Adam)
        #   side  : [1] MLP weight gradients + the MLP's optimiser step
        #   side2 : (the next step's rays, from the start of the step) [2] the pose refinement's chain + the pose step
        fork1 = torch.cuda.Event()
        # (round 4, measured again: ONE fork point after the activation gradients for both side branches: 0.280-0.283 ->
        #  0.287-0.294 ms per step)
        fork1.record(main)
        mark(None)
        check(L.ns_ngp_mlp_dgrad_f_n(ptr(fr_r), ptr(X["s_dout"]), ptr(self.relu_masks), ptr(self.s_dfeat),
                                     C.c_long(S), n_dev, st), "ngp_mlp_dgrad")
        mark("ngp_mlp_bwd_kernel")
        # (round 4, measured and not kept: the weight gradients IN LINE on the main stream ahead of the scatter, so that the
        #  scatter does not share the CUs with a kernel that takes whole SIMDs: step 0.288 -> 0.312 ms, 130 -> 121 frames/s)
        def weight_gradients(after):
            with torch.cuda.stream(self._side):
                st1 = stream_ptr()
                self._side.wait_event(after)
                mark(None)
                # (the optimiser that follows writes the OTHER fragment table: the activation-gradient kernel on the main stream
                #  may still be reading this one)
                check(L.ns_ngp_mlp_wgrad_partials_n(ptr(fr_r), ptr(featT), ptr(X["s_dir"]), ptr(X["s_dout"]),
                                                    ptr(self.partial_fused), self.mlp_wgs, C.c_long(S), n_dev, st1),
                      "ngp_mlp_wgrad_partials")
                mark("ngp_mlp_wgrad_tr_kernel")
                slabs = int(L.ns_ngp_mlp_wgrad_slabs(self.mlp_wgs, C.c_long(S)))
                if ev_params_read is not None:
                    self._side.wait_event(ev_params_read)    # (the MLP's optimiser rewrites the weights the refresh evaluates)
                if single:
                    # the MLP's optimiser step in one launch: slab reduce + Adam + f16 copy + both fragment tables (the zero rows
                    # of the tables were written once by the pack above); bit-identical to reduce + ns_ngp_adam_ctl + pack
                    # (tests/test_ngp_gpu.py::test_mlp_optimiser_step_in_one_launch_is_bit_identical)
                    check(L.ns_ngp_mlp_step_fused(ptr(self.partial_fused), slabs, ptr(self.mlp_grad), ptr(self.mlp_master),
                                                  ptr(self.mlp_half), ptr(self.mlp_m1), ptr(self.mlp_m2), ptr(fr_w), 0,
                                                  C.c_float(c.lr), C.c_float(c.beta1), C.c_float(c.beta2), C.c_float(c.eps),
                                                  C.c_float(c.l2_mlp), C.c_float(c.loss_scale * self.world), ctl, st1),
                          "ngp_mlp_step_fused")
                    mark("ngp_mlp_step_kernel")
                else:       # replicated trainers: the summed gradient is all-reduced first, the optimiser follows the exchange (post())
                    check(L.ns_ngp_mlp_reduce(ptr(self.partial_fused), slabs, ptr(self.mlp_grad), st1), "ngp_mlp_reduce")
                    mark("ngp_mlp_wgrad_reduce_kernel")
        if not c.wgrad_after_scatter:
            weight_gradients(fork1)
        fork = torch.cuda.Event()
        fork.record(main)
        # (enqueue order matters under capture although the dependencies do not change: the graph executor keeps the FIRST
        #  successor created for a node on that node's hardware queue and hands later ones to other queues -- enqueued after the
        #  side branches, the scatter landed on the side stream's queue BEHIND the pose refinement, 0.49 ms)
        if self.fused_ws:
            mark(None)
            if self.sparse_exchange:
                self._emit_count.zero_()
            table_gradient(1, st)
            mark("ngp_enc_fscatter_direct_kernel")
            if c.wgrad_after_scatter:      # A/B: the weight gradients beside the accumulate pass only
                ev_sc = torch.cuda.Event()
                ev_sc.record(main)
                weight_gradients(ev_sc)
            if ev_params_read is not None:
                main.wait_event(ev_params_read)          # (the flush rewrites the table the refresh encodes from)
            table_gradient(2, st)
            mark("ngp_enc_faccum_kernel")
        else:       # (no workspace / f32-atomic configuration of the tests: grad_fixed_scale = 0)
            check(L.ns_ngp_encode_backward_n(*self._grid_args(), ptr(X["s_pos"]), ptr(self.s_dfeat), 1, ptr(self.grid_grad),
                                             ptr(self.enc_ws), C.c_size_t(self.enc_ws_bytes), C.c_float(c.grad_fixed_scale),
                                             C.c_long(S), n_dev, st),
                  "ngp_encode_backward")
        # (round 4: every dense level goes through the bins by default -- no owner-computes pass)
        dense_pass = self.fused_ws and int(L.ns_ngp_encode_backward_fused_dense_levels(*self._grid_args())) > 0
        # the pose refinement's chain (Jacobian dot, camera gradient, reduce, pose step: ~45 us of small kernels) runs on the third
        # stream, next to the weight-gradient chain of `side` instead of behind it, and in stream order behind the ray sampling
        # of the next step that reads the poses it rewrites
        use_side2 = True                      # (the next step's rays were enqueued there at the start of the step)
        if dense_pass or pose:
            with torch.cuda.stream(self._side2):
                self._side2.wait_event(fork)
                if dense_pass:
                    table_gradient(4, stream_ptr())
                    table_gradient(8, stream_ptr())
                if pose:
                    pose_gradient(stream_ptr())
                    if single:
                        camera_step(stream_ptr())
        if use_side2:
            main.wait_stream(self._side2)
        main.wait_stream(self._side)
        if not single:
            def post(stream):
                """what follows the gradient exchange: pose step, Adam on THIS trainer's shard of the table (summed gradient in
                `_gshard`), the MLP's Adam.  Fixed arguments: captured as the second graph of the replicated step."""
                if pose:   # after the all-reduce: every replica applies the SAME pose update
                    camera_step(stream)
                if not self.sparse_exchange:     # (f32 gradient: dense all-reduce, streaming Adam; the list form's update follows
                    #                                     the exchange, outside the captured part: its length changes per step)
                    adam(self.grid_master, self.grid_half, self.grid_grad, self.grid_m1, self.grid_m2, 0.0, c.grad_fixed_scale, stream)
                mlp_adam(stream)
            if phase == "pre":          # (replicated step from two graphs: the caller runs the collectives between them)
                return post
            self._exchange_gradients(x, lambda: post(st))
        elif not self.fused_adam:
            adam(self.grid_master, self.grid_half, self.grid_grad, self.grid_m1, self.grid_m2, 0.0, c.grad_fixed_scale, st)

    def train_steps(self, n, return_loss=True):
        """`n` optimiser steps.  Same as n calls of train_step(); two steps at a time are replayed from ONE graph where that is
        possible (both single-step graphs captured, nothing reallocated, no occupancy update between the two): the gap between
        two graph launches (join, launch, fork: 25-35 us on the device) is paid once per pair."""
        c = self.cfg
        i = 0
        chain = max(2, int(c.steps_per_graph)) // 2 * 2      # steps per chained graph (even).  Trainer alone, 4 / 8 / 16 against 2:
                       # 0.438 -> 0.434 / 0.427 / 0.433 ms, inside the box-to-box spread; in the PIPELINE the mapper thread's replays
                       # compete with the tracker thread for the interpreter (cfg.steps_per_graph)
        while i < n:
            left = c.grid_update_every - self.step % c.grid_update_every          # steps until the next occupancy update
            first_ride = left <= chain and c.refresh_in_step and c.grid_rule == "subset" and not getattr(self, "_refresh_seen", False)
            if n - i >= 2 and self._pair_ready() and not first_ride:      # (the first refresh-carrying step runs singly, eagerly)
                m = min(chain, n - i, left) // 2 * 2
                # the chain that ends on an update carries the refresh in its last step (cfg.refresh_in_step)
                rides = m == left and c.refresh_in_step and c.grid_rule == "subset" and self._refresh_seen
                with torch.cuda.device(self.device):
                    g = (self._pair_r if rides else self._pair) if m == 2 else self._chains.get((m, rides))
                    if g is None:
                        from ._lib import capture_lock, graph_capture
                        with capture_lock:
                            torch.cuda.synchronize(self.device)
                            g = torch.cuda.CUDAGraph()
                            with graph_capture(g, capture_error_mode="thread_local"):
                                for k in range(m):
                                    self._enqueue_step(k & 1, refresh=rides and k == m - 1)
                        if m == 2 and rides:
                            self._pair_r = g
                        elif m == 2:
                            self._pair = g
                        else:
                            self._chains[(m, rides)] = g
                    g.replay()
                    self.step += m
                    if self.step % c.grid_update_every == 0 and not rides:
                        self.update_density_grid()
                        self._primed = False
                i += m
            else:
                self.train_step(return_loss=False)
                i += 1
        return self.loss_tensor if (return_loss and self.n_images > 0) else None

    def _pair_ready(self):
        c = self.cfg
        if self.replicated or not c.use_graph or self.n_images == 0:
            return False
        if not getattr(self, "_static", False) or not self._primed or self.cur != 0:
            return False
        if self._graphs[0] is None or self._graphs[1] is None:
            return False
        if c.optimize_extrinsics:
            self._grow_camera_state(self.images.shape[0])
        if self._graph_key != self._step_key():
            return False
        return self.step % c.grid_update_every <= c.grid_update_every - 2

    def train_step(self, return_loss=True):
        if self.n_images == 0:
            return None
        c, dev = self.cfg, self.device
        with torch.cuda.device(dev):
            if not getattr(self, "_static", False):
                self._alloc_static()
            if c.optimize_extrinsics:
                self._grow_camera_state(self.images.shape[0])
            x = self.cur
            if not self._primed:
                # no predecessor marched this step's rays (first step), or they were marched on an occupancy grid that has been
                # updated since: march them (again: same control block, same seed, same pixels) before the step
                X = self.sets[x]
                X["counter"].zero_()
                X["loss"].zero_()
                self._enqueue_rays(X)
                self._primed = True
            rides = self._refresh_rides()      # this step carries the occupancy refresh of the update that follows it
            if self.replicated and c.use_graph and not variant_env("NS_NGP_REPL_EAGER"):
                self._replicated_step(x, rides)
            elif self.replicated or not c.use_graph or getattr(self, "_probe", None) is not None:
                self._enqueue_step(x, refresh=rides)
            else:
                key = self._step_key()
                if self._graph_key != key:
                    # (re)capture: the first steps after a (re)allocation run eagerly -- they are also the warm-up
                    self._graphs, self._graph_key, self._pair, self._chains = [None, None], key, None, {}
                    self._graphs_r, self._pair_r = [None, None], None
                    self._eager_left = 2
                graphs = self._graphs_r if rides else self._graphs
                if self._eager_left > 0 or (rides and not self._refresh_seen):      # (the refresh's kernels run eagerly once too)
                    self._eager_left = max(self._eager_left - 1, 0)
                    self._enqueue_step(x, refresh=rides)
                elif graphs[x] is None:
                    from ._lib import capture_lock, graph_capture
                    with capture_lock:          # the tracker thread takes the same lock around its host read-backs
                        torch.cuda.synchronize(dev)
                        g = torch.cuda.CUDAGraph()
                        with graph_capture(g, capture_error_mode="thread_local"):
                            self._enqueue_step(x, refresh=rides)
                    graphs[x] = g               # (capturing does not execute: the step runs with the first replay)
                    g.replay()
                else:
                    graphs[x].replay()
            if rides:
                self._refresh_seen = True
            self.cur = 1 - x
            self.step += 1
            if self.step % c.grid_update_every == 0 and not rides:
                self.update_density_grid()
                # The next step's rays were marched ahead (side branch of this step's launch sequence) on the grid as it was
                # BEFORE this update: drop them, the next step marches its rays again on the updated grid (same seed, same
                # pixels).  Training on the stale march one step in 16 made the result a coin toss: PSNR after 500 steps on
                # the sphere scene 19-31 dB in a third of the runs instead of 34-36 dB (10 of 10 runs with this line; 10 of
                # 10 on the tree before the rays moved ahead).  Cost: one eager sample + march per update, < 1 % of a step.
                # (round 6, cfg.refresh_in_step: the refresh ran INSIDE the step, ahead of the next step's rays: nothing to redo)
                self._primed = False
        return self.loss_tensor if return_loss else None

    def probe_steps(self, n=8):
        """-> {kernel: mean microseconds over `n` optimiser steps run EAGERLY with timing events between the kernels of every
        stream}: each kernel's duration IN the step -- on the step's own samples, with the other streams' kernels running next to
        it as they do in the replayed graph (the events themselves cost a stream a few us per mark: durations, not the step time,
        are what this is for).  bench.py prints them beside the stand-alone launch times of the same kernels."""
        acc, cnt = {}, {}
        for _ in range(n):
            self._probe = []
            try:
                self.train_step(return_loss=False)
                torch.cuda.synchronize(self.device)
                last = {}
                for stream, name, ev in self._probe:
                    if name is not None and stream in last:
                        acc[name] = acc.get(name, 0.0) + 1e3 * last[stream].elapsed_time(ev)
                        cnt[name] = cnt.get(name, 0) + 1
                    last[stream] = ev
            finally:
                self._probe = None
        return {k: acc[k] / cnt[k] for k in acc}

    def _replicated_step(self, x, rides=False):
        """One step of a replicated trainer from TWO HIP graphs with the collectives between them:
            graph A: forward, backward, table gradient -> the list of touched entries, MLP / pose gradients (three streams)
            eager  : all-gather of the lists' lengths (started; read below), all-reduce of the MLP and pose gradients
            graph B: pose step, MLP Adam + fragment pack -- on the device while the host waits for the lengths (the step's one
                     host synchronisation: an event behind the 4-byte all-gather)
            eager  : all-gather of the lists, the per-entry sums + Adam on every touched entry (ns_ngp_sparse_table_update)
        The collectives are stream-ordered behind graph A and ahead of graph B (RCCL waits for the current stream); nothing
        synchronises with the host (under gloo the staging copies do).  Eager steps first, as in the one-trainer path."""
        key = self._step_key()
        if self._graph_key != key:
            self._graphs, self._graph_key, self._pair, self._chains = [None, None], key, None, {}
            self._graphs_r, self._pair_r = [None, None], None
            self._eager_left = 2
        if self._eager_left > 0 or (rides and not self._refresh_seen):      # (the refresh's kernels run eagerly once too)
            self._eager_left = max(self._eager_left - 1, 0)
            self._enqueue_step(x, refresh=rides)
            return
        graphs = self._graphs_r if rides else self._graphs
        if graphs[x] is None:
            from ._lib import capture_lock, graph_capture
            with capture_lock:
                torch.cuda.synchronize(self.device)
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with graph_capture(ga, capture_error_mode="thread_local"):
                    post = self._enqueue_step(x, phase="pre", refresh=rides)
                with graph_capture(gb, capture_error_mode="thread_local"):
                    post(stream_ptr())
            graphs[x] = (ga, gb)
        ga, gb = graphs[x]
        ga.replay()
        self._exchange_gradients(x, gb.replay)

    @property
    def loss_tensor(self):
        """mean per-ray loss of the last step (device scalar): sum of the per-ray losses the step left in its set / their
        number (recorded by ns_ngp_step_prepare at the start of that step)"""
        return self.sets[1 - self.cur]["loss"].sum(dim=0, keepdim=True) / self.last[3].clamp(min=1).float()

    @property
    def last_samples(self):
        return int(self.last[2].item()) if getattr(self, "_static", False) else 0

    @property
    def last_rays(self):
        return int(self.last[3].item()) if getattr(self, "_static", False) else 0

    @property
    def samples_requested_last(self):
        return int(self.last[0].item()) if getattr(self, "_static", False) else 0

    def _exchange_gradients(self, x, between=None):
        """replicated trainers, after the gradients of a step: the lists of touched table entries are all-gathered and applied
        (every trainer: the same integer sums, the same Adam update), the small MLP / pose gradients are all-reduced; Adam divides
        by loss_scale * world (mean gradient).  `between()` enqueues what depends on the small all-reduces only (the pose step and
        the MLP's optimiser: graph B): it runs on the device while the host waits for the lists' lengths -- the one host
        synchronisation of the step, on an event behind a 4-byte-per-trainer all-gather -- and enqueues the list exchange."""
        import torch.distributed as dist
        R = self.world
        c = self.cfg
        handle = None
        if self.sparse_exchange:
            from .parallel import gather_counts_begin, gather_counts_end, gather_lists, list_class
            handle = gather_counts_begin(self._emit_count, self.group)
        else:
            dist.all_reduce(self.grid_grad, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(self.mlp_grad, op=dist.ReduceOp.SUM, group=self.group)
        if self.cfg.optimize_extrinsics and getattr(self, "cam_grad", None) is not None:
            dist.all_reduce(self.cam_grad, op=dist.ReduceOp.SUM, group=self.group)   # 24 B per training view
        if between is not None:
            between()
        if self.sparse_exchange:
            counts_dev, counts = gather_counts_end(handle)
            n_pairs = list_class(max(counts))
            if self._lists is None or self._lists.numel() < R * n_pairs * 2:
                self._lists = torch.zeros((R * n_pairs * 2,), dtype=torch.int64, device=self.device)
            recv = self._lists[:R * n_pairs * 2].view(R, n_pairs, 2)
            wire = gather_lists(self._emit_list, recv, n_pairs, self.group) + (R - 1) * 4
            self.wire_log.append(n_pairs)
            with torch.cuda.device(self.device):
                check(lib().ns_ngp_sparse_table_update_rec(ptr(recv), ptr(counts_dev), R, C.c_long(n_pairs), C.c_long(max(counts)),
                                                       ptr(self.grid_grad), ptr(self.grid_master), ptr(self.grid_half), ptr(self.grid_m1),
                                                       ptr(self.grid_m2), self.state_rec, 0, C.c_float(c.lr), C.c_float(c.beta1), C.c_float(c.beta2),
                                                       C.c_float(c.eps), C.c_float(c.loss_scale * self.world), C.c_float(c.grad_fixed_scale),
                                                       ptr(self.sets[x]["ctl"]), stream_ptr()), "ngp_sparse_table_update")
        else:
            wire = 2 * (R - 1) * self.grid_grad.numel() * 4 // R
        self.bytes_allreduced = getattr(self, "bytes_allreduced", 0) + wire + (2 * (R - 1) * self.mlp_grad.numel() * 4) // R

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

    def _grid_scratch(self, n=None):
        """scratch of the subset refresh (cells, points, features, network output, directions, partial sums), allocated once"""
        c, dev = self.cfg, self.device
        total = c.n_cascades * c.grid_size ** 3
        n = min(int(n if n is not None else (1 << 18)), total) & ~1
        if getattr(self, "_grid_ws", None) is None:
            self._grid_ws = {}                 # per draw size: the default size's buffers are part of captured graphs and never move
        ws = self._grid_ws.get(n)
        if ws is None:
            f = dict(dtype=torch.float32, device=dev)
            ws = self._grid_ws[n] = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty((n, 3), **f),
                                     torch.zeros((32, n), dtype=torch.float16, device=dev), torch.zeros((n, 4), dtype=torch.float16, device=dev),
                                     torch.zeros((n, 3), **f), torch.zeros(256, dtype=torch.float64, device=dev))
            if not c.grid_decay_all and getattr(self, "_grid_tmp", None) is None:
                self._grid_tmp = torch.zeros(total, **f)
        return ws

    def _enqueue_grid_refresh(self, st, seed=None, ctl=None, n_cells=None, after_params_read=None):
        """the subset refresh's launch sequence on stream `st`: draw cells + jittered points, encode, density network, decay / max /
        mean / bit packing.  `ctl`: the seed is completed on the device from the step's control block (the captured form);
        `after_params_read()` is called behind the last kernel that reads the table / the MLP."""
        c = self.cfg
        G, nc = c.grid_size, c.n_cascades
        total = nc * G ** 3
        cells, pos, feat, out, dirs, part = self._grid_scratch(n_cells)
        n = cells.shape[0]
        s_box = float(c.aabb_scale)
        L, nul = lib(), C.c_void_p(0)
        if ctl is not None:
            seed0 = (self.base_seed * 0x9E3779B1 + 0x27D4EB2F) & 0xFFFFFFFF                                # same on every replica
            check(L.ns_ngp_grid_cells_ctl(G, nc, C.c_uint32(seed0), n, C.c_float(0.5 - 0.5 * s_box), C.c_float(0.5 + 0.5 * s_box),
                                          ptr(cells), ptr(pos), ctl, st), "ngp_grid_cells")
        else:
            check(L.ns_ngp_grid_cells(G, nc, C.c_uint32(seed), n, C.c_float(0.5 - 0.5 * s_box), C.c_float(0.5 + 0.5 * s_box), ptr(cells),
                                      ptr(pos), st), "ngp_grid_cells")
        check(L.ns_ngp_encode_forward(*self._grid_args(), ptr(pos), ptr(self.grid_half), ptr(feat), 1, C.c_long(n), st),
              "ngp_encode_forward")
        check(L.ns_ngp_mlp_forward(ptr(self.mlp_half), ptr(feat), ptr(dirs), ptr(out), nul, nul, nul, nul, C.c_long(n), st),
              "ngp_mlp_forward")
        if after_params_read is not None:
            after_params_read()
        if c.grid_decay_all:
            # (fading all cells at instant-ngp's 0.95 with 4 % of the grid drawn would empty the grid 12 x
            #  faster than the rule it stands in for; not fading the undrawn cells at all leaves floaters for ever)
            decay_all = float(c.grid_decay) ** min(1.0, n / (0.5 * total))
            check(L.ns_ngp_grid_update(ptr(out), ptr(cells), n, C.c_float(c.min_step), C.c_float(decay_all),
                                       C.c_float(c.min_optical_thickness), ptr(self.density_grid), C.c_long(total), ptr(part),
                                       ptr(self.bits), st), "ngp_grid_update")
        else:
            # decay only what was re-evaluated: a draw of 4 % of the grid per update must not fade the other 96 %
            check(L.ns_ngp_grid_update_sampled(ptr(out), ptr(cells), n, C.c_float(c.min_step), C.c_float(c.grid_decay),
                                               C.c_float(c.min_optical_thickness), ptr(self.density_grid), ptr(self._grid_tmp),
                                               C.c_long(total), ptr(part), ptr(self.bits), st), "ngp_grid_update_sampled")

    def _refresh_rides(self):
        """does the step about to run carry the occupancy refresh (cfg.refresh_in_step: the last step before an update)"""
        c = self.cfg
        return bool(c.refresh_in_step and c.grid_rule == "subset" and getattr(self, "_static", False)
                    and (self.step + 1) % c.grid_update_every == 0)

    def update_density_grid(self, n_cells=None):
        """Occupancy update, instant-ngp's rule (testbed_nerf.cu `update_density_grid_nerf`, called from its training loop
        every 16 steps): during the first 256 steps EVERY cell of every cascade is re-evaluated, afterwards G^3/4 cells per
        cascade drawn uniformly plus G^3/4 per cascade drawn among the currently OCCUPIED cells (up to 8 tries per sample);
        density at a jittered point of the cell, grid = max(decay * grid, new), a cell is occupied when density * min_step
        exceeds min(mean, threshold).  That is `grid_rule = "ngp"`.  The default "subset" draws
        2^18 cells uniformly per update instead: as implemented here (torch glue around the density evaluation) the
        published rule costs 0.35 ms per step (3.1 M density evaluations every 16 steps: 0.57 -> 0.92 ms) and did not train
        better on the sphere scene (PSNR after 500 steps 27.1-37.9 dB over 6 runs against 34.3-36.3 dB).  `n_cells`: a uniform
        draw of that many cells (tests).  The subset rule runs on the HIP kernels of csrc/ngp.hip ("occupancy-grid refresh":
        cells + jittered points, encode + density network, decay / max / mean / bit packing: 7 launches instead of ~35 torch
        ones)."""
        c, dev = self.cfg, self.device
        G, nc = c.grid_size, c.n_cascades
        G3 = G ** 3
        total = nc * G3
        if n_cells is None and c.grid_rule == "subset":
            n_cells = 1 << 18
        if n_cells is not None:
            # the subset rule on the HIP kernels (csrc/ngp.hip, "occupancy-grid refresh"): 7 launches, no allocation
            self._grid_updates = getattr(self, "_grid_updates", 0) + 1
            seed = (self.base_seed * 0x9E3779B1 + self._grid_updates * 0x85EBCA77 + 0x27D4EB2F) & 0xFFFFFFFF   # same on every replica
            self._enqueue_grid_refresh(stream_ptr(), seed=seed, n_cells=n_cells)
            return
        if n_cells is not None:
            cells = torch.randint(0, total, (min(int(n_cells), total),), device=dev, generator=self.gen)
        elif self.step < 256:
            cells = torch.arange(total, device=dev)
        else:
            n = (G3 // 4) * nc
            uni = torch.randint(0, total, (n,), device=dev, generator=self.gen)
            tries = torch.randint(0, total, (n, 8), device=dev, generator=self.gen, dtype=torch.int32)
            hit = (self.bits[(tries >> 3).long()] >> (tries & 7).to(torch.uint8)) & 1
            first = hit.to(torch.int32).argmax(dim=1, keepdim=True)            # first occupied try (try 0 when none is)
            cells = torch.cat([uni, tries.gather(1, first)[:, 0].long()])
            del tries, hit
        new = torch.zeros(total, dtype=torch.float32, device=dev)
        for s0 in range(0, cells.shape[0], 1 << 20):                           # bounded scratch: 2^20 points at a time
            cc = cells[s0:s0 + (1 << 20)]
            mip = cc // G3
            r = cc % G3
            xyz = torch.stack([r % G, (r // G) % G, r // (G * G)], -1).float()
            jit = torch.rand(xyz.shape, device=dev, generator=self.gen)
            scale = (2.0 ** mip.float())[:, None]
            pos = ((xyz + jit) / G - 0.5) * scale + 0.5
            dens = self.density_at(pos.contiguous()) * c.min_step
            new.scatter_reduce_(0, cc, dens, "amax", include_self=True)        # (max: independent of the order of duplicates)
        torch.maximum(self.density_grid.mul_(c.grid_decay), new, out=self.density_grid)
        thr = self.density_grid.mean().clamp(max=c.min_optical_thickness)      # device scalar: no read-back
        occ = (self.density_grid > thr).view(-1, 8).to(torch.uint8)
        wts = (2 ** torch.arange(8, device=dev, dtype=torch.uint8))
        self.bits.copy_((occ * wts).sum(-1).to(torch.uint8))                   # in place: the captured step reads this buffer

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
                    check(lib().ns_ngp_composite(ptr(self.s_out), ptr(self.s_dt), ptr(self.s_t), ptr(self.rm_start),
                                                 ptr(self.rm_n), R, nul, nul, nul, C.c_float(0), C.c_float(1), ptr(orgb),
                                                 ptr(odep), nul, nul, stream_ptr()), "ngp_composite")
                rgb[s:s + R], dep[s:s + R] = orgb, odep
                s += R
        return rgb.view(H, W, 3), dep.view(H, W)
