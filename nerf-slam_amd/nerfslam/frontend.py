"""TrackingFrontend -- host driver of the tracking hot path, mirroring RaftVisualFrontend
(/root/reference/slam/visual_frontends/visual_frontend.py; lines cited per method).

What is kept from the reference: the keyframe buffers and their layout (:162-237), the factor-graph
bookkeeping (nerfslam.factor_graph, index-identical), the update() sequence reproject -> motion features
-> correlation lookup -> update operator -> dense BA (itrs=2) -> covariances (:370-470), the keyframe
distance tests (:778-799) and the SLAM -> mapper packet (:1337-1391).

What is different: every numerical step is a HIP kernel launched through the C ABI and nothing in
update()/ba() synchronises with the host -- the edge lists live on the host already, the BA plan is
cached per graph version, the reduced camera system is solved on the device.

The two learned components of DROID-SLAM (feature/context encoders and the ConvGRU update operator,
networks/droid_net.py) are OUT of this project's hot-path scope and their
weights (`droid.pth`) are missing from the reference tree; they are injected as callables:
    feature_fn(image [3,H,W] float) -> fmap [128, H/8, W/8]
    update_op(corr [1,E,196,ht,wd], motion [1,E,4,ht,wd], ii, jj) -> (delta [1,E,ht,wd,2], weight [1,E,ht,wd,2],
                                                                          damping [n_unique_ii, ht, wd]
                                                                          [, upmask [n_unique_ii, 576, ht, wd] or channels-last [n_unique_ii, ht, wd, 576] f16])
"""
import ctypes as C

import os

import numpy as np
import torch

from . import ba_plan, se3
from ._lib import capture_lock, check, lib, ptr, stream_ptr, variant_env
from .corr import CorrPool
from .factor_graph import FactorGraph


class TrackingFrontend:
    def __init__(self, buffer, H, W, intrinsics, device="cuda:0", feature_fn=None, update_op=None, max_factors=48,
                 compute_covariances=True):
        self.device = dev = torch.device(device)
        self.buffer, self.H, self.W = buffer, H, W
        self.ht, self.wd = H // 8, W // 8
        self.HW = self.ht * self.wd
        f = dict(dtype=torch.float32, device=dev)
        ident = torch.tensor([0, 0, 0, 0, 0, 0, 1.0], **f)
        # buffers (:162-237)
        self.cam0_T_world = ident.repeat(buffer, 1)
        self.world_T_body = ident.repeat(buffer, 1)
        self.cam0_T_body = ident.clone()                       # vio_slam.py:87
        self.world_T_body_cov = torch.eye(6, **f).repeat(buffer, 1, 1) * 1e-4
        self.cam0_idepths = torch.ones((buffer, self.ht, self.wd), **f)
        self.cam0_idepths_sensed = torch.zeros((buffer, self.ht, self.wd), **f)
        self.cam0_idepths_cov = torch.ones((buffer, self.ht, self.wd), **f)
        self.cam0_depths_cov = torch.ones((buffer, self.ht, self.wd), **f)
        self.damping = 1e-6 * torch.ones((buffer, self.ht, self.wd), **f)
        self.intr8 = (torch.as_tensor(intrinsics, dtype=torch.float32) / 8.0).to(dev)   # :274
        self.feat_bank = torch.zeros((buffer, self.HW, 128), dtype=torch.float16, device=dev)  # channels-last, /4
        self.images = torch.zeros((buffer, 3, H, W), dtype=torch.uint8, device=dev)
        # full-resolution inverse depth / depth covariance by convex upsampling with the update operator's mask (:191-192,
        # :445-446); keyframes that never received a mask are upsampled bilinearly in get_viz_out
        self.cam0_idepths_up = torch.zeros((buffer, H, W), **f)
        self.cam0_depths_cov_up = torch.ones((buffer, H, W), **f)
        self.has_up = torch.zeros(buffer, dtype=torch.bool, device=dev)
        self.viz_idx = torch.zeros(buffer, dtype=torch.bool, device=dev)
        gy, gx = torch.meshgrid(torch.arange(self.ht, device=dev), torch.arange(self.wd, device=dev), indexing="ij")
        self.coords0 = torch.stack([gx, gy], -1).float()        # [ht,wd,2]
        # graph + per-edge payloads
        self.graph = FactorGraph(max_factors=max_factors)
        self.graph.sort_device = dev     # the age permutation of add_factors is sorted where the reference sorts it (:826)
        self.ii = self.jj = torch.zeros(0, dtype=torch.long, device=dev)
        # correlation volumes: a slot-addressed pool (nerfslam.corr.CorrPool); `corr` is None until the first edge has one
        self.corr = None
        self.slots = np.zeros(0, np.int32)           # pool slot of every active edge, in the graph's edge order
        self.slots_dev = torch.zeros(0, dtype=torch.int32, device=dev)
        self._free_slots = []
        self._edge_cache = None
        self.target = torch.zeros((0, self.ht, self.wd, 2), **f)
        self.weight = torch.zeros((0, self.ht, self.wd, 2), **f)
        self.target_inactive = torch.zeros((0, self.ht, self.wd, 2), **f)
        self.weight_inactive = torch.zeros((0, self.ht, self.wd, 2), **f)
        self.kf_idx = 0
        self.prior_pose = None            # frame-0 prior (:1089-1095)
        self.feature_fn, self.update_op = feature_fn, update_op
        self.compute_covariances = compute_covariances
        self._plan, self._plan_key = None, None
        self.n_updates = 0
        self.n_update_edges = 0        # active edges summed over the updates (bench.py: mean edges per update)
        self.beta = 0.3
        self.keyframe_thresh, self.frontend_thresh = 4.0, 16.0
        self.frontend_window, self.frontend_radius, self.frontend_nms, self.max_age = 25, 2, 1, 25

    # ---------------------------------------------------------------------------------------------
    def set_keyframe(self, k, image, fmap=None):
        """store frame k (:309-318): image uint8 [3,H,W]; features from feature_fn unless given."""
        self.images[k] = image.to(self.device)
        if fmap is None:
            fmap = self.feature_fn(image.to(self.device).float())
        # channels-last, pre-divided by 4 in half exactly as corr.py:67-68 scales its operands
        self.feat_bank[k] = (fmap.to(self.device).half().reshape(128, self.HW) / 4.0).t().contiguous()

    def _sync_edges(self):
        self.ii = torch.from_numpy(self.graph.ii).to(self.device)
        self.jj = torch.from_numpy(self.graph.jj).to(self.device)
        self.slots_dev = torch.from_numpy(np.ascontiguousarray(self.slots, np.int32)).to(self.device)
        self._edge_cache = None      # device-side index tensors derived from the edge list (rebuilt lazily in update())

    def _take_slots(self, n):
        if self.corr is None:
            self.corr = CorrPool(self.ht, self.wd, max(self.graph.max_factors + 16, 2 * n), self.device)
            self._free_slots = list(range(self.corr.capacity - 1, -1, -1))
        if len(self._free_slots) < n:        # init / global passes may exceed max_factors: grow (copies the live volumes once)
            old = self.corr.capacity
            self.corr.grow(max(2 * old, old + n))
            self._free_slots = list(range(self.corr.capacity - 1, old - 1, -1)) + self._free_slots
        return np.asarray([self._free_slots.pop() for _ in range(n)], np.int32)

    def reproject(self, ii, jj):
        """(:909-918) -> coords [E,ht,wd,2]."""
        E = ii.shape[0]
        coords = torch.empty((E, self.ht, self.wd, 2), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().ns_reproject(ptr(self.cam0_T_world), ptr(self.cam0_idepths), ptr(self.intr8), ptr(ii), ptr(jj),
                                     ptr(coords), None, E, self.ht, self.wd, stream_ptr()), "reproject")
        return coords

    def motion_features(self, coords1, target):
        """(:379-386) -> [E,4,ht,wd]"""
        E = coords1.shape[0]
        out = torch.empty((E, 4, self.ht, self.wd), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().ns_motion_features(ptr(coords1), ptr(target.contiguous()), ptr(out), E, self.ht, self.wd, stream_ptr()),
                  "motion_features")
        return out

    def distance(self, ii, jj, bidirectional=True):
        """(:778-799)"""
        import droid_backends
        ii = torch.as_tensor(ii, dtype=torch.long, device=self.device).reshape(-1).contiguous()
        jj = torch.as_tensor(jj, dtype=torch.long, device=self.device).reshape(-1).contiguous()
        d1 = droid_backends.frame_distance(self.cam0_T_world, self.cam0_idepths, self.intr8, ii, jj, self.beta)
        if not bidirectional:
            return d1
        d2 = droid_backends.frame_distance(self.cam0_T_world, self.cam0_idepths, self.intr8, jj, ii, self.beta)
        return 0.5 * (d1 + d2)

    # ---------------------------------------------------------------------------------------------
    def add_factors(self, ii, jj, remove=False):
        """(:806-862): de-duplicate, evict by age when over max_factors, build correlation pyramids for the new
        edges straight from the feature bank (one fused launch), initialise targets with the reprojection."""
        ni, nj, removed = self.graph.add(ii, jj, remove=remove, have_volumes=self.corr is not None)
        if removed is not None:
            self._drop_payload(removed, store=True)
        if ni.shape[0] == 0:
            return
        di, dj = torch.from_numpy(ni).to(self.device), torch.from_numpy(nj).to(self.device)
        new_slots = self._take_slots(ni.shape[0])
        self.corr.build(self.feat_bank, self.feat_bank, di, dj, torch.from_numpy(new_slots).to(self.device))
        self.slots = np.concatenate([self.slots, new_slots])
        tgt = self.reproject(di, dj)
        self.target = torch.cat([self.target, tgt], 0)
        self.weight = torch.cat([self.weight, torch.zeros_like(tgt)], 0)
        self._sync_edges()

    def _drop_payload(self, mask_h, store):
        """mask_h: HOST bool array over the active edges (the edge lists live on the host: no read-back)"""
        mh = np.asarray(mask_h, bool)
        mask = torch.from_numpy(mh).to(self.device)
        if store:
            self.target_inactive = torch.cat([self.target_inactive, self.target[mask]], 0)
            self.weight_inactive = torch.cat([self.weight_inactive, self.weight[mask]], 0)
        keep = ~mask
        self.target, self.weight = self.target[keep], self.weight[keep]
        if self.slots.shape[0] == mh.shape[0]:       # volumes stay where they are: the slots return to the free list
            self._free_slots.extend(int(v) for v in self.slots[mh])
            self.slots = self.slots[~mh]
        self._sync_edges()

    def rm_factors(self, mask, store=False):
        """(:868-892); mask: host bool array over the active edges."""
        mask = np.asarray(mask, bool)
        self.graph.remove(mask, store=store)
        self._drop_payload(mask, store)

    def add_neighborhood_factors(self, kf0, kf1, radius=3):
        ii, jj = FactorGraph.neighborhood_edges(kf0, kf1, radius, self.graph.stereo)
        self.add_factors(ii, jj)

    def add_proximity_factors(self, kf0=0, kf1=0, rad=2, nms=2, thresh=16.0, remove=False):
        """(:712-775): distances on the device (one D2H of the distance vector), selection on the host."""
        t = self.kf_idx + 1
        I, J = np.meshgrid(np.arange(kf0, t), np.arange(kf1, t), indexing="ij")
        with capture_lock:      # (host read-back: not while the mapper thread captures its step graphs, see _lib.capture_lock)
            d = self.distance(I.reshape(-1), J.reshape(-1)).cpu().numpy()
        es = self.graph.proximity_edges(d, self.kf_idx, kf0, kf1, rad, nms, thresh)
        if es:
            e = np.asarray(es, np.int64)
            with capture_lock:  # (the age permutation of add() is sorted on the device and read back)
                self.add_factors(e[:, 0], e[:, 1], remove)

    # ---------------------------------------------------------------------------------------------
    def _edges(self):
        """host + device index tensors derived from the current edge lists, built once per graph change"""
        c = self._edge_cache
        if c is None:
            g, dev = self.graph, self.device
            kx = np.unique(g.ii)
            kf0 = max(0, int(g.ii.min()))
            ii_h, jj_h, m = g.ba_edges(kf0)
            m_d = torch.from_numpy(m).to(dev)
            Mi, E = int(m.sum()), int(g.ii.shape[0])
            # BA inputs [M,2,ht,wd]: inactive edges first (visual_frontend.py:420-424); their part is filled once here
            tgt = torch.empty((Mi + E, 2, self.ht, self.wd), dtype=torch.float32, device=dev)
            wgt = torch.empty_like(tgt)
            if Mi:
                tgt[:Mi] = self.target_inactive[m_d].permute(0, 3, 1, 2)
                wgt[:Mi] = self.weight_inactive[m_d].permute(0, 3, 1, 2)
            c = self._edge_cache = dict(kx=kx, kx_d=torch.from_numpy(kx).to(dev), kf0=kf0, ii_h=ii_h, jj_h=jj_h, Mi=Mi,
                                        tgt=tgt, wgt=wgt, ii_list=g.ii.tolist(), jj_list=g.jj.tolist())
        return c

    def update(self, itrs=2):
        """one update-operator + dense-BA step (:370-470).  No host synchronisation: the edge lists are host arrays, every
        index tensor the step needs is cached per graph change, and the update operator gets the host copies of (ii, jj)."""
        c = self._edges()
        coords1 = self.reproject(self.ii, self.jj)                                    # [E,ht,wd,2]
        motion = self.motion_features(coords1, self.target)
        enc = getattr(getattr(self.update_op, "__self__", None), "corr_encoder", None)
        if enc is not None and not variant_env("NS_LOOKUP_UNFUSED"):
            corr = self.corr.lookup_encoded(coords1[None], self.slots_dev, enc)       # lookup + Conv2d(196,128,1) + ReLU, one launch
        else:
            corr = self.corr.lookup(coords1[None], self.slots_dev)                    # [1,E,196,ht,wd]
        if getattr(self.update_op, "host_indices", False):
            res = self.update_op(corr, motion[None], self.ii, self.jj, ii_host=c["ii_list"], jj_host=c["jj_list"])
        else:
            res = self.update_op(corr, motion[None], self.ii, self.jj)
        delta, weight, damping = res[:3]
        upmask = res[3] if len(res) > 3 else None
        self.target = coords1 + delta[0].float()
        self.weight = weight[0].float()
        self.damping[c["kx_d"]] = damping
        Mi = c["Mi"]
        c["tgt"][Mi:] = self.target.permute(0, 3, 1, 2)
        c["wgt"][Mi:] = self.weight.permute(0, 3, 1, 2)
        out = self.ba(c["tgt"], c["wgt"], c["ii_h"], c["jj_h"], c["kf0"], itrs=itrs)
        if upmask is not None:
            self.upsample(c["kx_d"], upmask)
        self.graph.age += 1
        self.viz_idx[c["kf0"]:self.kf_idx + 1] = True
        self.n_updates += 1
        self.n_update_edges += int(self.target.shape[0])
        return out

    def ba(self, target, weight, ii_h, jj_h, kf0, kf1=None, itrs=2, lm=0.0, ep=0.0, compute_covariances=None):
        """dense bundle adjustment (:1071-1232) without leaving the device."""
        if kf1 is None:
            kf1 = int(max(ii_h.max(), jj_h.max())) + 1
        key = (self.graph.version, kf0, kf1, ii_h.shape[0])
        if self._plan_key != key:
            self._plan, self._plan_key = ba_plan.BaPlan(ii_h, jj_h, kf0, kf1, self.device), key
            self._ii_ba = torch.from_numpy(ii_h).to(self.device)
            self._jj_ba = torch.from_numpy(jj_h).to(self.device)
            self._kx_ba = torch.from_numpy(self._plan.kx_host).to(self.device)
        plan = self._plan
        kx = self._kx_ba
        damping = (0.2 * self.damping[kx] + 1e-7).contiguous()                        # :428
        prior = self.prior_pose if (kf0 == 0 and self.prior_pose is not None) else None
        sol = None
        cov = self.compute_covariances if compute_covariances is None else compute_covariances
        for it in range(itrs):
            H, v, Q, E, w = ba_plan.reduced_camera_matrix(plan, self.cam0_T_world, self.cam0_idepths, self.intr8,
                                                          self.cam0_T_body, self.cam0_idepths_sensed, target, weight,
                                                          damping, self._ii_ba, self._jj_ba)
            last = it == itrs - 1
            sol = ba_plan.ba_solve(H, v, kf0, kf1, self.world_T_body, self.cam0_T_world, self.cam0_T_body,
                                   prior_pose=prior, ep=ep, lm=lm, want_cov=cov and last)
            ba_plan.solve_depth(plan, sol["dx"], self.cam0_idepths, Q, E, w, clamp_min=0.001)   # :1161-1162
        if cov and sol["Linv"] is not None:
            z = ba_plan.depth_cov(plan, sol["Linv"], Q, E, self.HW).view(-1, self.ht, self.wd)      # :1191-1219
            self.world_T_body_cov[kf0:kf1] = sol["sigma_g"]
            self.cam0_idepths_cov[kx] = z
            self.cam0_depths_cov[kx] = z / self.cam0_idepths[kx] ** 4                          # :1229
        return sol

    def upsample(self, kx, upmask):
        """convex 8x upsampling of the inverse depths and depth covariances of keyframes kx (:445-446)"""
        n = kx.shape[0]
        kx = kx.to(torch.int64).contiguous()
        if upmask.dim() == 4 and tuple(upmask.shape) == (n, self.ht, self.wd, 576) and upmask.dtype == torch.float16:
            # channels-last logits, as nerfslam.update_op writes them: no transposition
            with torch.cuda.device(self.device):
                check(lib().ns_cvx_upsample_keyframes_nhwc(ptr(self.cam0_idepths), ptr(self.cam0_depths_cov), ptr(kx),
                                                           ptr(upmask.contiguous()), ptr(self.cam0_idepths_up),
                                                           ptr(self.cam0_depths_cov_up), n, self.ht, self.wd, C.c_float(1.0),
                                                           stream_ptr()), "cvx_upsample_keyframes_nhwc")
            self.has_up[kx] = True
            return
        mask = upmask.reshape(n, 576, self.ht, self.wd).contiguous()
        dt = 1 if mask.dtype == torch.float16 else 2
        if mask.dtype not in (torch.float16, torch.float32):
            mask, dt = mask.float(), 2
        with torch.cuda.device(self.device):  # both maps, in place in the keyframe buffers, one pass over the mask
            check(lib().ns_cvx_upsample_keyframes(ptr(self.cam0_idepths), ptr(self.cam0_depths_cov), ptr(kx), ptr(mask), dt,
                                                  ptr(self.cam0_idepths_up), ptr(self.cam0_depths_cov_up), n, self.ht,
                                                  self.wd, C.c_float(1.0), stream_ptr()), "cvx_upsample_keyframes")
        self.has_up[kx] = True

    # ---------------------------------------------------------------------------------------------
    def get_viz_out(self):
        """SLAM -> mapper packet (:1337-1391): the dirty keyframes, as DEVICE tensors (the reference moves them to
        the CPU for --multi_gpu and pickles them; here nerfslam.transport ships them over RCCL)."""
        idx, = torch.where(self.viz_idx)
        if idx.numel() == 0:
            return None
        up = lambda x: torch.nn.functional.interpolate(x[:, None], size=(self.H, self.W), mode="bilinear",
                                                       align_corners=False)[:, 0]
        out = {"cam0_poses": self.cam0_T_world[idx], "world_T_body": self.world_T_body[idx],
               "world_T_body_cov": self.world_T_body_cov[idx], "cam0_idepths": self.cam0_idepths[idx],
               "cam0_idepths_up": torch.where(self.has_up[idx, None, None], self.cam0_idepths_up[idx], up(self.cam0_idepths[idx])),
               "cam0_idepths_sensed": self.cam0_idepths_sensed[idx],
               "cam0_idepths_cov": self.cam0_idepths_cov[idx], "cam0_depths_cov": self.cam0_depths_cov[idx],
               "cam0_depths_cov_up": torch.where(self.has_up[idx, None, None], self.cam0_depths_cov_up[idx],
                                                 up(self.cam0_depths_cov[idx])),
               "cam0_images": self.images[idx],
               "cam0_intrinsics": (self.intr8 * 8.0)[None].repeat(idx.numel(), 1), "viz_idx": idx,
               "kf_idx": self.kf_idx, "is_last_frame": False}
        self.viz_idx[:] = False
        return out
