"""The tracking hot path of ONE keyframe step as a fixed kernel chain (SURVEY.md 8(a) rows A1-A13 at the BASELINE shape:
640x480 -> 80x60 grid, 48 active + 48 inactive edges, P=10 poses, K'=13 depth maps), independent of the conv nets:

  motion filter      1-edge correlation pyramid build + one 4-level lookup
  proximity factors  2 x frame_distance over 125 pairs + 2 x 1 pair
  new edges          correlation pyramid build for 10 new edges
  6 x update()       reprojection + motion features of the 48 active edges, their 4-level lookup, then BA itrs=2:
                       2 x [reduced camera matrix, device Cholesky solve + pose retraction, depth back-substitution],
                     the depth/pose covariance block, the paired convex 8x upsampling of the updated keyframes

bench.py reports it as `extra.hot_path_chain` (round 1's headline; NOT the tracked+mapped metric) and uses it for the
kernel-level roofline entries of the correlation kernels and for the CPU baseline (`cpu_baseline`: the oracle on a
bounded sample of the same chain)."""
import os
import time

import numpy as np
import torch

HT, WD, CH = 60, 80, 128
HW = HT * WD
NBUF = 16
E_ACTIVE, E_INACTIVE, E_NEW = 48, 48, 10
KF0, KF1 = 6, 16
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
TILED = os.environ.get("NS_BENCH_ROWMAJOR") is None  # volumes in the 8x8-tiled layout (the frontend's); set to compare


def quat_exp(w):
    th = np.linalg.norm(w)
    return np.concatenate([np.sin(th / 2) * w / max(th, 1e-12), [np.cos(th / 2)]])


def make_graph(rng):
    """48 active edges among frames [6,16), 48 inactive edges among frames [3,16) (both ends >= kf0-3,
    visual_frontend.py:420-424); no duplicates, no self loops."""
    def pick(lo, hi, n, seen):
        es = []
        for i in range(lo, hi):
            for j in range(max(lo, i - 2), min(hi, i + 3)):
                if i != j and (i, j) not in seen and len(es) < n:
                    es.append((i, j)); seen.add((i, j))
        while len(es) < n:
            i, j = (int(x) for x in rng.integers(lo, hi, 2))
            if i != j and (i, j) not in seen:
                es.append((i, j)); seen.add((i, j))
        return es
    seen = set()
    act = pick(KF0, KF1, E_ACTIVE, seen)
    ina = pick(KF0 - 3, KF1, E_INACTIVE, seen)
    allv = ina + act  # torch.cat([inactive, active]) (visual_frontend.py:421-422)
    ii = np.array([e[0] for e in allv], np.int64)
    jj = np.array([e[1] for e in allv], np.int64)
    return ii, jj


class HotPath:
    def __init__(self, dev, seed=0):
        from nerfslam import ba_plan
        from nerfslam.corr import CorrBlock
        self.dev = dev
        rng = np.random.default_rng(seed)
        g = torch.Generator(device="cpu").manual_seed(seed)
        poses = np.zeros((NBUF, 7), np.float32)
        for k in range(NBUF):
            poses[k, :3] = rng.normal(0, 0.05, 3)
            poses[k, 3:] = quat_exp(rng.normal(0, 0.02, 3))
        self.cTw0 = torch.from_numpy(poses).to(dev)
        from nerfslam import se3
        self.wTb0 = se3.inv(self.cTw0.double()).float().contiguous()
        self.disps0 = torch.empty((NBUF, HT, WD)).uniform_(0.2, 2.0, generator=g).to(dev)
        self.cTw, self.wTb, self.disps = self.cTw0.clone(), self.wTb0.clone(), self.disps0.clone()
        self.disps_sens = torch.zeros_like(self.disps)
        W = WD * 8.0
        self.intr = (torch.tensor([0.5 * W, 0.5 * W, (W - 1) / 2, (HT * 8.0 - 1) / 2]) / 8.0).to(dev)
        self.extr = torch.tensor([0, 0, 0, 0, 0, 0, 1.0]).to(dev)
        self.fmaps = torch.randn((NBUF, CH, HT, WD), generator=g).half().to(dev)
        ii, jj = make_graph(rng)
        self.ii_h, self.jj_h = ii, jj
        self.ii, self.jj = torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev)
        self.M = ii.shape[0]
        self.plan = ba_plan.BaPlan(ii, jj, KF0, KF1, dev)
        self.K = self.plan.K
        # targets = reprojection + noise, weights ~ U(0,1), damping as visual_frontend.py:428
        c, _ = self._reproject(self.ii, self.jj)
        self.targets = (c + 0.5 * torch.randn(c.shape, generator=g).to(dev)).contiguous()
        self.weights = torch.rand((self.M, 2, HT, WD), generator=g).to(dev)
        self.eta = (0.2 * torch.empty((self.K, HT, WD)).uniform_(1e-4, 2e-2, generator=g) + 1e-7).to(dev)
        # persistent 48-edge pyramid, coordinates of the active edges in the frontend's layout
        ai, aj = self.ii[E_INACTIVE:], self.jj[E_INACTIVE:]
        # feature bank as the frontend keeps it (nerfslam/frontend.py:set_keyframe): channels-last f16, pre-divided by 4
        self.feat_bank = (self.fmaps.reshape(NBUF, CH, HW) / 4.0).transpose(1, 2).contiguous()
        self.corr48 = CorrBlock.from_pyramid(CorrBlock.build_pyramid(self.feat_bank, self.feat_bank, ai.contiguous(), aj.contiguous(),
                                                                     E_ACTIVE, HT, WD, tiled=TILED), tiled=TILED, hw=(HT, WD))
        gy, gx = torch.meshgrid(torch.arange(HT), torch.arange(WD), indexing="ij")
        grid = torch.stack([gx, gy], -1).float()
        c48 = grid[None, None] + torch.empty((1, E_ACTIVE, HT, WD, 2)).uniform_(-8, 8, generator=g)
        oob = torch.rand((1, E_ACTIVE, HT, WD), generator=g) < 0.05          # 5 % of the lookups fall outside
        c48[oob] += (torch.randint(0, 2, (int(oob.sum()), 2), generator=g).float() * 2 - 1) * (HT + WD)
        self.coords48 = c48.to(dev)
        self.coords1 = self.coords48[:, :1].contiguous()
        self.new_i = torch.from_numpy(rng.integers(KF0, KF1, E_NEW)).to(dev)
        self.new_j = torch.from_numpy(rng.integers(KF0, KF1, E_NEW)).to(dev)
        pi, pj = np.meshgrid(np.arange(KF1 - 5, KF1), np.arange(0, KF1 + 9)[:25] % KF1, indexing="ij")
        self.fd_i = torch.from_numpy(pi.reshape(-1).astype(np.int64)).to(dev)
        self.fd_j = torch.from_numpy(pj.reshape(-1).astype(np.int64)).to(dev)
        self.fd1_i = torch.tensor([KF1 - 3], device=dev)
        # update-operator glue of every update() (visual_frontend.py:379-386, 445-446, 909-918): reprojection of the active
        # edges, motion features, convex upsampling of the updated keyframes' inverse depths and depth covariances
        self.ai, self.aj = ai.contiguous(), aj.contiguous()
        self.kx = torch.unique(self.ai)
        self.target_a = self.targets[E_INACTIVE:].permute(0, 2, 3, 1).contiguous()   # the frontend's [E,ht,wd,2]
        self.coords_a = torch.empty((E_ACTIVE, HT, WD, 2), device=dev)
        self.motion = torch.empty((E_ACTIVE, 4, HT, WD), device=dev)
        self.upmask = torch.randn((self.kx.shape[0], HT, WD, 576), generator=g).half().to(dev)  # the mask head's f16 logits, channels-last as nerfslam.update_op writes them
        self.depth_cov = torch.rand((NBUF, HT, WD), generator=g).to(dev)
        self.disps_up = torch.zeros((NBUF, 8 * HT, 8 * WD), device=dev)
        self.depth_cov_up = torch.zeros((NBUF, 8 * HT, 8 * WD), device=dev)
        self.fd1_j = torch.tensor([KF1 - 2], device=dev)
        self.CorrBlock, self.ba_plan = CorrBlock, ba_plan
        self.ev = None  # optional per-op event recorder

    def _reproject(self, ii, jj):
        """targets for the synthetic problem (float64 torch, setup only)."""
        from nerfslam import se3
        gy, gx = torch.meshgrid(torch.arange(HT, device=self.dev), torch.arange(WD, device=self.dev), indexing="ij")
        fx, fy, cx, cy = self.intr.double()
        X = torch.stack([(gx - cx) / fx, (gy - cy) / fy, torch.ones_like(gx, dtype=torch.float64),
                         torch.zeros_like(gx, dtype=torch.float64)], -1)[None].repeat(ii.shape[0], 1, 1, 1)
        X[..., 3] = self.disps[ii].double()
        G = se3.mul(self.cTw[jj].double(), se3.inv(self.cTw[ii].double()))
        Y = se3.act(G[:, None, None], X)
        z = Y[..., 2].clamp(min=0.25)
        c = torch.stack([fx * Y[..., 0] / z + cx, fy * Y[..., 1] / z + cy], 1)
        return c.float(), z

    # ---- the ops of one step -------------------------------------------------------------------
    def _t(self, name, fn):
        if self.ev is None:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        self.ev.setdefault(name, []).append((s, e))
        return r

    def op_build(self, i, j):
        """correlation pyramids of new edges straight from the feature bank (frontend.py:add_factors)"""
        pyr = self.CorrBlock.build_pyramid(self.feat_bank, self.feat_bank, i, j, i.shape[0], HT, WD, tiled=TILED)
        return self.CorrBlock.from_pyramid(pyr, tiled=TILED, hw=(HT, WD))

    def op_set_keyframe(self, k):
        """the incoming frame's features enter the bank (frontend.py:set_keyframe)"""
        self.feat_bank[k] = (self.fmaps[k].reshape(CH, HW) / 4.0).t()

    def op_lookup48(self):
        return self.corr48(self.coords48)

    def op_update_glue_pre(self):
        from nerfslam._lib import check, lib, ptr, stream_ptr
        L = lib()
        check(L.ns_reproject(ptr(self.cTw), ptr(self.disps), ptr(self.intr), ptr(self.ai), ptr(self.aj), ptr(self.coords_a),
                             None, E_ACTIVE, HT, WD, stream_ptr()), "reproject")
        check(L.ns_motion_features(ptr(self.coords_a), ptr(self.target_a), ptr(self.motion), E_ACTIVE, HT, WD,
                                   stream_ptr()), "motion_features")

    def op_upsample(self):
        import ctypes as C
        from nerfslam._lib import check, lib, ptr, stream_ptr
        check(lib().ns_cvx_upsample_keyframes_nhwc(ptr(self.disps), ptr(self.depth_cov), ptr(self.kx), ptr(self.upmask),
                                                   ptr(self.disps_up), ptr(self.depth_cov_up), self.kx.shape[0], HT, WD,
                                                   C.c_float(1.0), stream_ptr()), "cvx_upsample_keyframes_nhwc")

    def op_ba_iteration(self, want_cov):
        import droid_backends
        bp = self.ba_plan
        H, v, Q, E, w = self._t("rcm", lambda: bp.reduced_camera_matrix(
            self.plan, self.cTw, self.disps, self.intr, self.extr, self.disps_sens, self.targets, self.weights,
            self.eta, self.ii, self.jj))
        sol = self._t("solve", lambda: bp.ba_solve(H, v, KF0, KF1, self.wTb, self.cTw, self.extr,
                                                   prior_pose=None, want_cov=want_cov))
        self._t("depth", lambda: bp.solve_depth(self.plan, sol["dx"], self.disps, Q, E, w, clamp_min=0.001))
        return sol, Q, E

    def step(self):
        import droid_backends
        # new keyframe slot seeded from saved state (visual_frontend.py:626-635); keeps the synthetic
        # problem stationary across steps
        self.cTw.copy_(self.cTw0); self.wTb.copy_(self.wTb0); self.disps.copy_(self.disps0)
        self._t("set_keyframe", lambda: self.op_set_keyframe(KF1 - 1))
        # motion filter (visual_frontend.py:976-1007)
        blk = self._t("build1", lambda: self.op_build(self.new_i[:1], self.new_j[:1]))
        self._t("lookup1", lambda: blk(self.coords1))
        # proximity factors (visual_frontend.py:712-775, 611)
        for a, b in ((self.fd_i, self.fd_j), (self.fd_j, self.fd_i), (self.fd1_i, self.fd1_j), (self.fd1_j, self.fd1_i)):
            self._t("frame_distance", lambda: droid_backends.frame_distance(self.cTw, self.disps, self.intr, a, b, 0.3))
        # correlation volumes of the new edges (visual_frontend.py:838-844)
        self._t("build10", lambda: self.op_build(self.new_i, self.new_j))
        # iters1 + iters2 updates (visual_frontend.py:607-621)
        for _ in range(6):
            self._t("reproject+motion", self.op_update_glue_pre)
            self._t("lookup48", self.op_lookup48)
            self.op_ba_iteration(False)
            sol, Q, E = self.op_ba_iteration(True)
            self._t("cov", lambda: self.ba_plan.depth_cov(self.plan, sol["Linv"], Q, E, HW))
            self._t("upsample", self.op_upsample)


ALG_BYTES = {
    # per (edge, level, pixel) 64 taps*2 + 49 outputs*2 + 8 coords = 234 B
    "lookup48": E_ACTIVE * 4 * HW * 234,
    # per edge: read 2*HW*128*2, write HW^2*2*(1+1/4+1/16+1/64)
    "build10": E_NEW * (2 * HW * CH * 2 + int(HW * HW * 2 * (1 + 0.25 + 0.0625 + 0.015625))),
}


def cpu_baseline(hp):
    """The oracle (a C port of the reference kernels; OpenMP over the host's cores for the volume build, the lookup and the
    per-edge linearisation, the rest single-threaded -- `cores` reports the thread count) on a bounded sample of the
    same workload: one 1-edge pyramid build, one 48-edge 4-level lookup, one BA linearisation + Schur reduction + depth
    back-substitution at M=96; extrapolated to the op counts of one keyframe step of the chain."""
    import oracle
    t = {}
    f = hp.fmaps.cpu().numpy()
    i0, j0 = int(hp.new_i[0]), int(hp.new_j[0])
    t0 = time.time(); oracle.corr_pyramid(f[i0:i0 + 1], f[j0:j0 + 1]); t["build_per_edge"] = time.time() - t0
    pyr = [p.cpu().numpy() for p in hp.corr48.untiled()]
    c = np.ascontiguousarray(hp.coords48[0].cpu().numpy().transpose(0, 3, 1, 2))
    t0 = time.time()
    for l in range(4):
        oracle.corr_index_forward(pyr[l], c / np.float32(2 ** l), 3)
    t["lookup48"] = time.time() - t0
    a = [x.cpu().numpy() for x in (hp.cTw0, hp.disps0, hp.intr, hp.extr, hp.disps_sens, hp.targets, hp.weights, hp.eta)]
    t0 = time.time()
    H, v, Q, E, w, kx = oracle.reduced_camera_matrix(*a, hp.ii_h, hp.jj_h, KF0, KF1)
    dx = np.zeros((KF1 - KF0, 6), np.float32)
    oracle.solve_depth(dx, a[1], Q, E, w, hp.ii_h, hp.jj_h, KF0, KF1)
    t["ba_iteration"] = time.time() - t0
    step = (1 + E_NEW) * t["build_per_edge"] + (6 + 1.0 / E_ACTIVE) * t["lookup48"] + 12 * t["ba_iteration"]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"value": 1.0 / step, "unit": "keyframe steps/s (tracking hot-path chain only: no conv nets, no mapping)", "cores": cores, "kind": "port",
            "seconds_per_keyframe_step": step,
            "sample": "oracle (C port, OpenMP over %d host cores for the volume build, the lookup and the per-edge linearisation; "
                      "accumulation / Schur / depth update single-threaded): 1-edge pyramid build %.2fs, 48-edge 4-level lookup %.2fs, "
                      "one M=96 BA linearisation+Schur+depth %.2fs; extrapolated to one step = 11 builds, "
                      "6 lookups, 12 BA iterations" % (cores, t["build_per_edge"], t["lookup48"], t["ba_iteration"])}


