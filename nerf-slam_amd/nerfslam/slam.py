"""TrackingSLAM -- per-frame state machine around TrackingFrontend; the call surface of the reference's
VioSLAM (/root/reference/slam/vio_slam.py:78-127: `slam(batch) -> [state, viz_out] | False`,
`stop_condition()`) and the frame logic of RaftVisualFrontend.forward
(/root/reference/slam/visual_frontends/visual_frontend.py:240-368, 577-688, 1255-1335):

    frame 0 -> keyframe 0;  motion filter;  warm-up until `keyframe_warmup` keyframes;  initialize
    (neighbourhood edges r=3, 8 updates, proximity edges, 8 updates);  per keyframe: age-out edges,
    proximity edges, iters1 updates, keyframe distance test (reject -> rm_keyframe), iters2 updates,
    seed the next keyframe;  at buffer end / last frame: global BA passes (7 and 12 steps) on the
    on-the-fly correlation (AltCorrBlock).

The learned networks are injected through `args.networks` (see nerfslam.frontend docstring):
    networks.features(image_u8 [3,H,W]) -> fmap [128,H/8,W/8]
    networks.update(corr, motion, ii, jj) -> (delta, weight, damping)
    networks.motion(corr [1,1,196,ht,wd], last_kf) -> delta [1,1,ht,wd,2]   (motion filter, :978-1008)
    optional hooks: networks.begin_keyframe(k, image_u8), networks.remove_keyframe(k)   (context features, hidden states)
nerfslam.droid_nets.DroidNetworks provides all of them on top of the DROID architecture (random or loaded weights).
The GTSAM Values / NonlinearFactorGraph the reference returns empty (:248-250) are returned as None.
"""
import os

import numpy as np
import torch

from ._lib import capture_lock, variant_env
from .corr import AltCorrBlock, CorrBlock
from .frontend import TrackingFrontend


class TrackingSLAM:
    def __init__(self, name, args, device):
        self.name, self.args, self.device = name, args, torch.device(device)
        self.net = getattr(args, "networks", None)
        if self.net is None:
            raise RuntimeError("TrackingSLAM: args.networks (features / update / motion callables) is required; the "
                               "DROID weights are not part of this project")
        self.buffer = args.buffer
        self.global_ba = getattr(args, "slam", True) and getattr(args, "global_ba", True)
        self.keyframe_warmup, self.motion_filter_thresh = 8, 2.4           # :93-95
        self.keyframe_thresh = None                                        # None: the frontend's default (4.0, :110)
        self.iters1, self.iters2 = 4, 2                                    # :101-102
        self.backend_thresh, self.backend_radius, self.backend_nms = 22.0, 2, 3
        self.fe = None
        self.last_k, self.last_kf = None, 0
        self.is_initialized, self.stop = False, False
        self.kf_to_frame = {}
        self.stats = {"frames": 0, "candidates": 0, "rejected": 0, "updates": 0}   # bookkeeping only (bench / logs)
        self.leg_ms = None     # set to a dict: backend() attributes its time per leg (adds device synchronisations: bench only)

    # the reference's nn.Module call
    def __call__(self, batch):
        out = self._frontend(batch["data"])
        return False if out is False else [None, out]

    def stop_condition(self):
        return self.stop

    # ---------------------------------------------------------------------------------------------
    def _store(self, k, data, fmap):
        fe = self.fe
        fe.set_keyframe(fe.kf_idx, data["image_u8"], fmap)
        if hasattr(self.net, "begin_keyframe"):
            self.net.begin_keyframe(fe.kf_idx, data["image_u8"])
        depth = data.get("depth")
        if depth is not None:   # EXTENSION: the reference allocates cam0_idepths_sensed (:190) but never fills it (monocular
                                # only); a dataset that carries depth seeds the sensed-depth prior of the BA here
            d = torch.as_tensor(depth, dtype=torch.float32, device=self.device)[3::8, 3::8] * float(data.get("depth_scale", 1.0))
            fe.cam0_idepths_sensed[fe.kf_idx] = torch.where(d > 0, 1.0 / d, torch.zeros_like(d))
        self.kf_to_frame[fe.kf_idx] = k

    def _enough_motion(self, fmap):
        """one update-operator evaluation on the identity lookup between the last keyframe and this frame."""
        fe = self.fe
        f1 = fe.feat_bank[self.last_kf][None]                              # channels-last half, already / 4
        f2 = (fmap.to(self.device).half().reshape(128, fe.HW) / 4.0).t().contiguous()[None]
        pyr = CorrBlock.build_pyramid(f1, f2, None, None, 1, fe.ht, fe.wd, tiled=True)
        corr = CorrBlock.from_pyramid(pyr, tiled=True, hw=(fe.ht, fe.wd))(fe.coords0[None, None])
        delta = self.net.motion(corr, self.last_kf)
        with capture_lock:      # host read-back (see _lib.capture_lock)
            return float(delta.norm(dim=-1).mean()) > self.motion_filter_thresh

    def _frontend(self, data):
        k = int(data["k"][0])
        img = torch.as_tensor(data["images"][0], device=self.device)[..., :3].permute(2, 0, 1).contiguous()
        depths = data.get("depths")
        calib0 = data["calibs"][0] if "calibs" in data else None
        data = dict(data, image_u8=img, depth=depths[0] if depths is not None and len(depths) > 0 else None,
                    depth_scale=float(getattr(calib0, "depth_scale", 1.0) or 1.0))     # vio_slam.py: gt_depths * calib.depth_scale
        last_frame = bool(data.get("is_last_frame", False))
        if self.fe is None:
            assert k == 0
            intr = np.asarray(data["calibs"][0].camera_model.numpy() if hasattr(data["calibs"][0], "camera_model")
                              else data["calibs"][0], np.float32)
            self.fe = TrackingFrontend(self.buffer, img.shape[1], img.shape[2], intr, self.device,
                                       feature_fn=self.net.features, update_op=self.net.update)
            if self.keyframe_thresh is not None:
                self.fe.keyframe_thresh = self.keyframe_thresh
            self._store(k, data, None)
            self.fe.prior_pose = self.fe.world_T_body[0].clone()           # frame-0 prior (:1089-1095, :1234-1253)
            self.last_k, self.last_kf = k, 0
            self.fe.viz_idx[0] = True
            viz = self._viz(last_frame)
            self.fe.kf_idx += 1
            return viz
        fe = self.fe
        self.stats["frames"] += 1
        fmap = self.net.features(img)
        if not self._enough_motion(fmap):
            if last_frame:
                fe.kf_idx -= 1
                self.terminate()
                return self._viz(True)
            return None
        self._store(k, data, fmap)
        self.stats["candidates"] += 1
        if not self.is_initialized:
            if fe.kf_idx >= self.keyframe_warmup:
                self._initialize()
        elif not self._track():
            self.rm_keyframe(fe.kf_idx - 1)
            self.stats["rejected"] += 1
            return None
        self.last_k, self.last_kf = k, fe.kf_idx
        viz = self._viz(last_frame)
        if fe.kf_idx + 1 >= self.buffer or last_frame:
            self.terminate()
            return self._viz(True)
        fe.kf_idx += 1
        return viz

    def _viz(self, last):
        out = self.fe.get_viz_out()
        if out is None:
            out = {}
        out["is_last_frame"] = last
        out["kf_idx_to_f_idx"] = dict(self.kf_to_frame)
        return out

    # ---------------------------------------------------------------------------------------------
    def _seed_next(self, window):
        fe, k = self.fe, self.fe.kf_idx
        if k + 1 >= self.buffer:
            return
        for buf in (fe.cam0_T_world, fe.world_T_body, fe.world_T_body_cov):
            buf[k + 1] = buf[k]
        lo = k + 1 - window
        fe.cam0_idepths[k + 1] = fe.cam0_idepths[lo:k + 1].mean()
        fe.cam0_idepths_cov[k + 1] = fe.cam0_idepths_cov[lo:k + 1].mean() if window > 1 else fe.cam0_idepths_cov[k]
        fe.cam0_depths_cov[k + 1] = fe.cam0_depths_cov[lo:k + 1].mean() if window > 1 else fe.cam0_depths_cov[k]

    def _initialize(self):
        """:641-688"""
        fe = self.fe
        fe.add_neighborhood_factors(0, fe.kf_idx, radius=3)
        for _ in range(8):
            fe.update()
        fe.add_proximity_factors(0, 0, rad=2, nms=2, thresh=fe.frontend_thresh, remove=False)
        for _ in range(8):
            fe.update()
        self._seed_next(4)
        self.is_initialized = True
        fe.viz_idx[:fe.kf_idx + 1] = True
        fe.rm_factors(fe.graph.ii < (self.keyframe_warmup - 4), store=True)

    def _track(self):
        """:577-638"""
        fe, k = self.fe, self.fe.kf_idx
        if fe.corr is not None:
            fe.rm_factors(fe.graph.age > fe.max_age, store=True)
        fe.add_proximity_factors(kf0=k - 4, kf1=max(k + 1 - fe.frontend_window, 0), rad=fe.frontend_radius,
                                 nms=fe.frontend_nms, thresh=fe.frontend_thresh, remove=True)
        s = fe.cam0_idepths_sensed[k]
        fe.cam0_idepths[k] = torch.where(s > 0, s, fe.cam0_idepths[k])
        for _ in range(self.iters1):
            fe.update()
        with capture_lock:      # host read-back (see _lib.capture_lock)
            reject = float(fe.distance([k - 2], [k - 1])) < fe.keyframe_thresh
        if reject:
            return False
        for _ in range(self.iters2):
            fe.update()
        self._seed_next(1)
        return True

    def rm_keyframe(self, k):
        """:530-574: slide frame k+1 over k in every buffer, drop / renumber the edges that touch it."""
        fe = self.fe
        for buf in (fe.images, fe.cam0_T_world, fe.world_T_body, fe.world_T_body_cov, fe.cam0_idepths, fe.cam0_idepths_cov,
                    fe.cam0_depths_cov, fe.cam0_idepths_sensed, fe.feat_bank, fe.cam0_idepths_up, fe.cam0_depths_cov_up,
                    fe.has_up):
            buf[k] = buf[k + 1]
        if hasattr(self.net, "remove_keyframe"):
            self.net.remove_keyframe(k)
        if k + 1 in self.kf_to_frame:       # (the reference leaves its kf -> frame table stale here)
            self.kf_to_frame[k] = self.kf_to_frame.pop(k + 1)
        keep_inactive, drop_active = fe.graph.remove_keyframe(k)
        ki = torch.from_numpy(keep_inactive).to(self.device)
        fe.target_inactive, fe.weight_inactive = fe.target_inactive[ki], fe.weight_inactive[ki]
        fe._drop_payload(drop_active, store=False)

    # ---------------------------------------------------------------------------------------------
    def _leg(self, name):
        """context manager: with `leg_ms` set, synchronise on both sides and add the elapsed time to leg `name`"""
        import contextlib
        import time
        if self.leg_ms is None:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def timed():
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            yield
            torch.cuda.synchronize(self.device)
            self.leg_ms[name] = self.leg_ms.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)
        return timed()

    def backend(self, steps, group=None):
        """global BA over all keyframes with on-the-fly correlation (:1255-1300, :474-527).

        group (torch.distributed process group, every rank holding the SAME keyframe buffer): the pass is SHARDED BY SOURCE
        FRAME over the group's ranks (nerfslam.parallel): a rank correlates / runs the update operator /
        linearises only the edges whose source frame it owns -- the edges of a source frame share its depth map, so every
        Schur contribution is local and additive -- one all-reduce of the reduced camera system ((6P)^2 + 6P floats) and one of
        the depth-map updates per BA iteration make poses and depths identical on all ranks.  The reference has no such mode
        (its only split is tracker | mapper)."""
        fe, t = self.fe, self.fe.kf_idx
        if not bool(torch.any(fe.cam0_idepths_sensed)):                    # normalize (:1302-1307)
            s = fe.cam0_idepths[:t].mean()
            fe.cam0_idepths[:t] /= s
            fe.cam0_T_world[:t, :3] *= s
        g = fe.graph
        saved = g.max_factors
        g.reset(max_factors=16 * t)
        fe.corr, fe.damping = None, 1e-6 * torch.ones_like(fe.cam0_idepths)      # (drops the volume pool: the global
        fe.slots, fe._free_slots = np.zeros(0, np.int32), []                       #  passes correlate on the fly)
        fe.target = fe.weight = fe.target_inactive = fe.weight_inactive = torch.zeros((0, fe.ht, fe.wd, 2), device=self.device)
        I, J = np.meshgrid(np.arange(0, t + 1), np.arange(0, t + 1), indexing="ij")
        d = fe.distance(I.reshape(-1), J.reshape(-1)).cpu().numpy()
        es = np.asarray(g.proximity_edges(d, t, 0, 0, self.backend_radius, self.backend_nms, self.backend_thresh), np.int64)
        if steps and es.shape[0]:
            ii_h, jj_h, _ = g.add(es[:, 0], es[:, 1])
            self.last_backend_edges = int(ii_h.shape[0])
            ii, jj = torch.from_numpy(ii_h).to(self.device), torch.from_numpy(jj_h).to(self.device)
            # (half, like the reference's features_imgs: AltCorrBlock then keeps a half pyramid and correlates on the matrix cores)
            fmaps = (fe.feat_bank * 4.0).transpose(1, 2).reshape(1, self.buffer, 128, fe.ht, fe.wd)
            corr_op = AltCorrBlock(fmaps)
            target, weight = fe.reproject(ii, jj), torch.zeros((ii.shape[0], fe.ht, fe.wd, 2), device=self.device)
            sba, own = None, np.ones(ii_h.shape[0], bool)
            if group is not None:
                import torch.distributed as dist
                from .parallel import ShardedBA
                if dist.get_world_size(group) > 1:
                    sba = ShardedBA(ii_h, jj_h, 0, int(max(ii_h.max(), jj_h.max())) + 1, self.device, group=group)
                    own = np.zeros(ii_h.shape[0], bool)
                    own[sba.mine] = True
                    self.last_backend_edges_mine = int(own.sum())
            for _ in range(steps):
                with self._leg("reproject + motion features"):
                    coords1 = fe.reproject(ii, jj)
                    motion = fe.motion_features(coords1, target)
                for lo in range(0, int(jj_h.max()) + 1, 8):                # windows of 8 source frames (:494-499)
                    vh = np.nonzero(own & (ii_h >= lo) & (ii_h < lo + 8))[0]   # (the edge lists are host arrays: the window
                    if vh.shape[0] == 0:                                   #  selection costs no device read-back)
                        continue
                    v = torch.from_numpy(vh).to(self.device)
                    iv, jv = ii[v], jj[v]
                    with self._leg("on-the-fly correlation (altcorr)"):
                        # (networks that advertise `corr_encoder` get the correlation with the encoder's 1x1 convolution + ReLU
                        #  already applied, in one launch: the 196 f32 planes per edge are never written)
                        enc = getattr(self.net, "corr_encoder", None)
                        if enc is not None and corr_op.half and not variant_env("NS_LOOKUP_UNFUSED"):
                            corr = corr_op.encoded(coords1[None, v], iv, jv, enc)
                        else:
                            corr = corr_op(coords1[None, v], iv, jv)
                    with self._leg("update operator"):
                        if getattr(self.net.update, "host_indices", False):
                            res = self.net.update(corr, motion[None, v], iv, jv, ii_host=ii_h[vh].tolist(), jj_host=jj_h[vh].tolist())
                        else:
                            res = self.net.update(corr, motion[None, v], iv, jv)
                    with self._leg("targets / damping / upsampling"):
                        delta, w, damping = res[:3]
                        target[v], weight[v] = coords1[v] + delta[0].float(), w[0].float()
                        kxv = torch.from_numpy(np.unique(ii_h[vh])).to(self.device)
                        fe.damping[kxv] = damping
                        if len(res) > 3:
                            fe.upsample(kxv, res[3])
                with self._leg("dense BA (2 iterations)"):
                    tg, wg = target.permute(0, 3, 1, 2).contiguous(), weight.permute(0, 3, 1, 2).contiguous()
                    if getattr(self, "keep_backend_ba_inputs", False):     # (bench.py: the BA roofline entries re-run this linearisation)
                        self.last_backend_ba = (tg, wg, ii_h, jj_h)
                    if sba is None:
                        fe.ba(tg, wg, ii_h, jj_h, kf0=0, itrs=2, compute_covariances=False)   # :523-526 (its lm / ep arguments are
                    else:                                                                      #  dead: ba() never reads them)
                        kx_all = torch.from_numpy(sba.kx_all).to(self.device)
                        eta = (0.2 * fe.damping[kx_all] + 1e-7).reshape(kx_all.shape[0], -1).contiguous()   # :428, rows like kx_all
                        for _it in range(2):
                            sba.iteration(fe.cam0_T_world, fe.cam0_idepths, fe.intr8, fe.cam0_T_body, fe.cam0_idepths_sensed, tg,
                                          wg, eta, fe.world_T_body, prior_pose=fe.prior_pose)
            if sba is not None:
                self._replicate_source_frame_state(ii_h, own, group)
        g.reset(max_factors=saved)
        fe._sync_edges()
        fe.viz_idx[:t] = True

    def _replicate_source_frame_state(self, ii_h, own, group):
        """End of a sharded backend pass: the update operator's per-source-frame outputs -- damping and the convex-upsampled
        depth / covariance maps -- exist only on the rank that owns the source frame; poses and low-resolution depths are already
        identical (ShardedBA.iteration).  Every source frame has exactly one owner, so a sum of the owner-masked rows IS the
        exchange: three all-reduces per pass, after which every rank again holds the SAME keyframe buffer (the precondition of
        the next pass, and what the mapper is handed)."""
        import torch.distributed as dist
        fe = self.fe
        src = np.unique(ii_h)
        mine = np.isin(src, np.unique(ii_h[own]))
        idx = torch.from_numpy(src).to(self.device)
        m = torch.from_numpy(mine).to(self.device)
        for buf in (fe.damping, fe.cam0_idepths_up, fe.cam0_depths_cov_up):
            # a SELECT, not a multiply: a stale non-finite value in a non-owner's row (inf * 0 = NaN) must not reach the sum
            rows = torch.where(m[:, None, None], buf[idx], torch.zeros((), dtype=buf.dtype, device=self.device))
            dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=group)
            buf[idx] = rows
        # a frame has upsampled maps only if its OWNER upsampled it (an update operator without an upmask leaves has_up unset
        # and get_viz_out falls back to the bilinear maps): the flag travels owner-masked like the rows above
        up = (fe.has_up[idx] & m).to(torch.int32)
        dist.all_reduce(up, op=dist.ReduceOp.SUM, group=group)
        fe.has_up[idx] = up > 0

    def terminate(self):
        """:1309-1335"""
        if self.global_ba:
            for steps in (7, 12, 0):
                self.backend(steps)
        self.stop = True
