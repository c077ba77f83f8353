"""NerfFusion -- the mapper-side driver (reference: /root/reference/fusion/nerf_fusion.py:29-307).

Consumes the SLAM -> fusion packet (visual_frontend.py:1364-1382), converts it the way the reference's
`process_slam` does (:140-235) -- but on the DEVICE: camera-to-world from cam0_T_world, sRGB -> linear
(:204-215, utils/utils.py:136-139), depth = 1/idepth_up, depth covariance pass-through, the four mask
policies (:173-183) -- and feeds `pyngp.Testbed.nerf.training.update_training_images`; training runs on
spins without a usable packet exactly like the reference (:249-253)."""
import numpy as np
import torch

from . import se3


def srgb_to_linear(img):
    """utils/utils.py:136-139"""
    limit = 0.04045
    return torch.where(img > limit, torch.pow((img + 0.055) / 1.055, 2.4), img / 12.92)


class NerfFusion:
    def __init__(self, name, args, device):
        import pyngp as ngp
        self.name, self.args, self.device = name, args, torch.device(device)
        self.iters_if_none, self.total_iters = 1, 0                           # :51-54
        self.stop_iters = getattr(args, "stop_iters", 25000)
        dev_index = self.device.index if self.device.index is not None else 0
        # args.trainer_group: process group of the replicated mapper GPUs under --multi_gpu (None: one trainer)
        self.ngp = ngp.Testbed(ngp.TestbedMode.Nerf, dev_index, group=getattr(args, "trainer_group", None))
        self.ngp.create_empty_nerf_dataset(args.buffer, 1.0, np.array([np.inf] * 3), 4, None)    # :67-72
        self.ngp.nerf.training.n_images_for_training = 0
        self.ngp.reload_network_from_file(getattr(args, "network", "") or "base.json")
        self.ngp.shall_train = True
        self.ngp.nerf.training.optimize_extrinsics = True
        self.ngp.nerf.training.depth_supervision_lambda = 1.0
        self.ngp.nerf.training.depth_loss_type = ngp.LossType.L2
        self.mask_type = getattr(args, "mask_type", "ours")
        self.anneal, self.anneal_every_iters, self.annealing_rate = False, 200, 0.95
        self.fit_volume_once()

    def process_slam(self, packet):
        """:140-235.  Returns False after ingesting (the reference then skips training on this spin).
        Intentional difference: the reference drops the packet flagged `is_last_frame` (:143-147) although it carries the
        poses / depths refined by the final global BA; here it is ingested like any other (the mapper ends on the
        globally optimised trajectory)."""
        if packet is None or "cam0_poses" not in packet:
            return True
        dev = self.device
        idx = packet["viz_idx"].to(dev)
        poses = packet["cam0_poses"].to(dev)
        images = packet["cam0_images"].to(dev)
        idepths_up = packet["cam0_idepths_up"].to(dev)
        depths_cov_up = packet["cam0_depths_cov_up"].to(dev)
        intr = packet["cam0_intrinsics"][0].to(dev)
        n, _, H, W = images.shape
        if self.mask_type == "raw":                                    # :173-183
            depths_cov_up = torch.ones_like(depths_cov_up)
        elif self.mask_type == "no_depth":
            idepths_up = -torch.ones_like(idepths_up)
        elif self.mask_type == "ours_w_thresh":
            # :177-179: threshold = depths_cov_up.quantile(0.50) (linear interpolation between the two middle order
            # statistics; computed from a sort because torch.quantile refuses more than 2^24 elements)
            flat = depths_cov_up.flatten().float().sort().values
            m = flat.numel()
            thr = 0.5 * (flat[(m - 1) // 2] + flat[m // 2])
            idepths_up = torch.where(depths_cov_up.sqrt() > thr, -torch.ones_like(idepths_up), idepths_up)
        c2w = se3.matrix(se3.inv(poses.float()))[:, :3, :4].contiguous()  # :198-203 (inverse of cam_T_world)
        rgb = srgb_to_linear(images.float().permute(0, 2, 3, 1) / 255.0)
        rgba = torch.cat([rgb, torch.ones((n, H, W, 1), device=dev)], -1).contiguous()   # alpha 1: premultiplied == rgb
        depths = (1.0 / idepths_up)[..., None].contiguous()             # :205 (negative where masked)
        covs = depths_cov_up[..., None].contiguous()
        fx, fy, cx, cy = (float(v) for v in intr)
        self.send_data(idx.tolist(), c2w, rgba, depths, covs, [W, H], [cx, cy], [fx, fy])
        return False

    def send_data(self, frame_ids, poses, images, depths, depths_cov, resolution, principal_point, focal_length):
        """:267-289"""
        self.ngp.nerf.training.update_training_images(frame_ids, poses, images, depths, depths_cov, resolution,
                                                      principal_point, focal_length, 1.0, 1.0)

    def fuse(self, packets):
        """:238-262"""
        fit = True
        if packets:
            pkt = packets.get("slam") if isinstance(packets, dict) else None
            if pkt is not None:
                viz = pkt[1] if isinstance(pkt, (list, tuple)) else pkt
                if viz is not None and viz.get("is_last_frame") and "cam0_poses" not in viz:
                    return None
                fit = self.process_slam(viz)
        if fit:
            self.fit_volume()
        return True

    def evaluate(self, stride=2):
        """PSNR / depth-L1 of the rendered training views against what the tracker uploaded (nerf_fusion.py:388-470 does this
        against ground-truth frames it never stores -- `ref_frames` stays empty there)"""
        import pyngp as ngp
        from . import eval as ev
        n = self.ngp.nerf.training.n_images_for_training
        train, mode = self.ngp.shall_train, self.ngp.render_mode
        self.ngp.shall_train = False
        ps, l1 = [], []
        for i in range(0, n, stride):
            ref_rgb, ref_depth = self.ngp.training_view(i)
            H, W = ref_depth.shape
            self.ngp.set_camera_to_training_view(i)
            self.ngp.render_mode = ngp.Shade
            est = self.ngp.render(W, H, 1, True)[..., :3]
            self.ngp.render_mode = ngp.Depth
            dep = self.ngp.render(W, H, 1, True)[..., 0]
            ps.append(ev.psnr(est, ref_rgb))
            m = ref_depth > 0
            if m.any() and dep[m].mean() > 0:
                l1.append(ev.depth_l1_cm(dep[m], ref_depth[m]))
        self.ngp.shall_train, self.ngp.render_mode = train, mode
        return {"psnr": float(np.mean(ps)) if ps else float("nan"), "depth_l1_cm": float(np.mean(l1)) if l1 else float("nan"),
                "views": len(ps)}

    def stop_condition(self):
        return self.total_iters > self.stop_iters

    def fit_volume(self):
        for _ in range(self.iters_if_none):
            self.fit_volume_once()

    def fit_volume_once(self):
        """:298-307"""
        self.ngp.frame()
        self.total_iters = self.ngp.training_step
        if self.anneal and self.total_iters % self.anneal_every_iters == 0:
            self.ngp.nerf.training.depth_supervision_lambda *= self.annealing_rate
