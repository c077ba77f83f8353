"""pyngp -- the attribute surface of instant-ngp's Python module that NeRF-SLAM's mapper drives
(/root/reference/fusion/nerf_fusion.py:57-101, 285-303, 388-424), backed by the HIP
trainer `nerfslam.ngp.NgpNerf`.  Only what `NerfFusion` touches is provided; GUI-only knobs are
accepted and ignored.

The real module is an un-vendored fork (ToniRV/instant-ngp @ feature/nerf_slam); entry points marked
[fork] below exist only there and their semantics are inferred from the call sites.
"""
import enum
import time

import numpy as np
import torch

from nerfslam.ngp import NgpConfig, NgpNerf


class TestbedMode(enum.Enum):
    Nerf = 0
    Sdf = 1
    Image = 2
    Volume = 3


class LossType(enum.Enum):
    L2 = 0
    L1 = 1
    Huber = 2


class RenderMode(enum.Enum):
    Shade = 0
    Depth = 1


Shade, Depth = RenderMode.Shade, RenderMode.Depth


class BoundingBox:
    def __init__(self, lo, hi):
        self.min, self.max = np.asarray(lo, np.float32), np.asarray(hi, np.float32)


class _Dataset:
    def __init__(self):
        self.n_images = 0
        self.aabb_scale = 1
        self.scale = 1.0
        self.offset = np.array([0.5, 0.5, 0.5], np.float32)
        self.paths = []


class _Training:
    def __init__(self, owner):
        self._o = owner
        self.n_images_for_training = 0
        self.extrinsic_learning_rate = 1e-4
        self.depth_loss_type = LossType.L2
        self.near_distance = 0.05
        self.density_grid_decay = 0.95
        self.dataset = _Dataset()

    @property
    def optimize_extrinsics(self):
        """camera-pose refinement of the training views (nerf_fusion.py:99,123)"""
        return self._o._cfg.optimize_extrinsics

    @optimize_extrinsics.setter
    def optimize_extrinsics(self, v):
        self._o._cfg.optimize_extrinsics = bool(v)

    @property
    def depth_supervision_lambda(self):
        return self._o._cfg.depth_lambda

    @depth_supervision_lambda.setter
    def depth_supervision_lambda(self, v):
        self._o._cfg.depth_lambda = float(v)

    def update_training_images(self, frame_ids, poses, images, depths, depths_cov, resolution, principal_point,
                               focal_length, depth_scale, depth_cov_scale):
        """[fork] (nerf_fusion.py:285-289): (re)upload the listed keyframes.  poses [n,3,4] camera-to-world,
        images [n,H,W,4] linear premultiplied RGBA f32, depths / depths_cov [n,H,W,1] f32.  numpy arrays or torch
        tensors (device tensors skip the host bounce the reference pays, nerf_fusion.py:217-226)."""
        self._o._ingest(frame_ids, poses, images, depths, depths_cov, resolution, principal_point, focal_length,
                        depth_scale, depth_cov_scale)


class _NerfState:
    def __init__(self, owner):
        self.training = _Training(owner)
        self.visualize_cameras = False
        self.rendering_min_transmittance = 1e-4


class Testbed:
    def __init__(self, mode=TestbedMode.Nerf, device=0, group=None):
        """group: torch.distributed process group of REPLICATED trainers (the --multi_gpu split with more than one mapper
        GPU); None: a single trainer, as the reference has."""
        if mode != TestbedMode.Nerf:
            raise NotImplementedError("only TestbedMode.Nerf is used by NeRF-SLAM")
        self._group = group
        self._device = torch.device("cuda", int(device))
        self._cfg = NgpConfig()
        self._net = None
        self._slots = 0
        self.nerf = _NerfState(self)
        self.shall_train = True
        self.dynamic_res = False
        self.dynamic_res_target_fps = 15
        self.camera_smoothing = False
        self.display_gui = False
        self.visualize_unit_cube = False
        self.background_color = [0.0, 0.0, 0.0, 1.0]
        self.snap_to_pixel_centers = True
        self.render_mode = Shade
        self.camera_matrix = np.eye(4, dtype=np.float32)[:3]
        self.elapsed_training_time = 0.0
        self.training_step = 0
        self._loss_t = None
        self.steps_per_frame = 16
        self._t0 = time.time()

    @property
    def loss(self):
        return float("nan") if self._loss_t is None else float(self._loss_t)

    # -- dataset -------------------------------------------------------------------------------
    def create_empty_nerf_dataset(self, n_images, nerf_scale=1.0, nerf_offset=None, aabb_scale=4, render_aabb=None):
        """[fork] (nerf_fusion.py:67-72): allocate `n_images` training slots."""
        self._cfg.aabb_scale = int(aabb_scale)
        self._slots = int(n_images)
        ds = self.nerf.training.dataset
        ds.n_images, ds.aabb_scale, ds.scale = self._slots, int(aabb_scale), float(nerf_scale)
        off = np.asarray(nerf_offset if nerf_offset is not None else [0.5, 0.5, 0.5], np.float32)
        ds.offset = np.where(np.isfinite(off), off, 0.5).astype(np.float32)  # the reference passes inf ("not needed")
        self._net = NgpNerf(self._cfg, self._device, group=self._group)
        self._imgs = self._deps = self._covs = self._c2w = None

    def reload_network_from_file(self, path=None):
        """The reference loads configs/nerf/base.json of the un-vendored fork (nerf_fusion.py:58-61,90); the
        network here is the fixed configuration of NgpConfig.  Re-initialises the parameters."""
        if self._net is not None:
            self._net = NgpNerf(self._cfg, self._device, group=self._group)
            self._push_images()

    def init_window(self, *a, **k):
        pass

    def _ingest(self, frame_ids, poses, images, depths, depths_cov, resolution, principal_point, focal_length,
                depth_scale, depth_cov_scale):
        dev = self._device
        t = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(dev, torch.float32)
        ids = [int(i) for i in frame_ids]
        poses, images = t(poses), t(images)
        depths, covs = t(depths).reshape(len(ids), *images.shape[1:3]), t(depths_cov).reshape(len(ids), *images.shape[1:3])
        H, W = images.shape[1:3]
        if self._imgs is None or self._imgs.shape[1:3] != (H, W):
            f = dict(dtype=torch.float32, device=dev)
            self._imgs = torch.zeros((self._slots, H, W, 4), **f)
            self._deps = torch.zeros((self._slots, H, W), **f)
            self._covs = torch.ones((self._slots, H, W), **f)
            self._c2w = torch.zeros((self._slots, 3, 4), **f)
        ds = self.nerf.training.dataset
        idx = torch.as_tensor(ids, device=dev)
        self._imgs[idx] = images
        self._deps[idx] = depths * float(depth_scale)
        self._covs[idx] = covs * float(depth_cov_scale)
        c2w = poses.clone()
        c2w[:, :, 3] = c2w[:, :, 3] * ds.scale + torch.as_tensor(ds.offset, device=dev)
        self._c2w[idx] = c2w
        fl = np.broadcast_to(np.asarray(focal_length, np.float32).reshape(-1), (2,))
        pp = np.asarray(principal_point, np.float32).reshape(-1)
        # principal point arrives normalised to [0,1] in instant-ngp's convention when < 1.5, pixels otherwise
        cx, cy = (pp[0] * W, pp[1] * H) if pp.max() <= 1.5 else (pp[0], pp[1])
        self._intr = (float(fl[0]), float(fl[1]), float(cx), float(cy))
        self.nerf.training.n_images_for_training = max(self.nerf.training.n_images_for_training, max(ids) + 1)
        self._push_images()

    def _push_images(self):
        n = self.nerf.training.n_images_for_training
        if self._net is not None and self._imgs is not None and n > 0:
            # whole slot arrays + the number of valid views: addresses stay put as keyframes arrive (captured training step)
            self._net.set_images(self._imgs, self._deps, self._covs, self._c2w, self._intr, n_images=n)

    # -- training / rendering ----------------------------------------------------------------------
    def frame(self):
        """One instant-ngp "frame" (nerf_fusion.py:299): a slice of training steps (no GUI here)."""
        if self.shall_train and self._net is not None and self._net.n_images > 0:
            t0 = time.time()
            loss = self._net.train_steps(self.steps_per_frame)
            if loss is not None:
                self._loss_t = loss                      # device scalar; converted when `loss` is read (no per-frame sync)
            self.training_step = self._net.step
            self.elapsed_training_time += time.time() - t0
        return True

    def training_view(self, i):
        """(linear rgb [H,W,3], depth [H,W]) of training slot i as uploaded (numpy) -- evaluation reference"""
        return self._imgs[int(i), ..., :3].cpu().numpy(), self._deps[int(i)].cpu().numpy()

    def apply_camera_smoothing(self, *a, **k):
        pass

    def set_camera_to_training_view(self, i):
        self.camera_matrix = self._c2w[int(i)].cpu().numpy()

    def render(self, width, height, spp=1, linear=True, fps=0.0, **k):
        """-> [height, width, 4] float32 (Shade: linear rgb + alpha 1; Depth: depth replicated)."""
        c2w = torch.as_tensor(np.asarray(self.camera_matrix, np.float32)[:3])
        sx, sy = width / self._imgs.shape[2], height / self._imgs.shape[1]
        fx, fy, cx, cy = self._intr
        rgb, dep = self._net.render(c2w, int(height), int(width), (fx * sx, fy * sy, cx * sx, cy * sy))
        out = torch.ones((height, width, 4), dtype=torch.float32, device=rgb.device)
        out[..., :3] = rgb if self.render_mode == Shade else dep[..., None]
        return out.cpu().numpy()
