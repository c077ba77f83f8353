"""Evaluation metrics of the headline benchmark (BASELINE.json: "PSNR + ATE-RMSE vs ref"): image PSNR and the depth L1
the reference's NerfFusion.eval_gt_traj reports (/root/reference/fusion/nerf_fusion.py:388-470, scale-matched mean
absolute depth error in cm, truncated at 2 m), and the absolute trajectory error after a least-squares Sim(3) / SE(3)
alignment (Umeyama 1991; what evo's `ape --align [--correct_scale]` computes -- the reference evaluates with evo
offline).  numpy / torch only; no HIP kernels involved."""
import numpy as np
import torch


def mse2psnr(mse):
    return -10.0 * np.log10(max(float(mse), 1e-20))


def psnr(est, ref):
    """est / ref: arrays or tensors of equal shape with values in [0,1]"""
    est = torch.as_tensor(np.asarray(est) if not isinstance(est, torch.Tensor) else est).double().cpu()
    ref = torch.as_tensor(np.asarray(ref) if not isinstance(ref, torch.Tensor) else ref).double().cpu()
    return mse2psnr(((est - ref) ** 2).mean().item())


def depth_l1_cm(est_depth, ref_depth, truncate=2.0):
    """nerf_fusion.py:452-457: scale est to the reference's mean, mean |diff| truncated at `truncate` m, in cm"""
    est, ref = np.asarray(est_depth, np.float64), np.asarray(ref_depth, np.float64)
    scale = ref.mean() / est.mean()
    return float(np.minimum(np.abs(scale * est - ref), truncate).mean() * 100.0)


def umeyama(src, dst, with_scale=True):
    """least-squares similarity  dst ~ s R src + t  (points [n,3]) -> (s, R, t)"""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    U, D, Vt = np.linalg.svd(xd.T @ xs / src.shape[0])
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float((D * np.diag(S)).sum() / (xs ** 2).sum(1).mean()) if with_scale else 1.0
    return s, R, mu_d - s * R @ mu_s


def ate_rmse(est_xyz, gt_xyz, correct_scale=True):
    """absolute trajectory error (RMSE of the position residuals after alignment); monocular runs use correct_scale"""
    s, R, t = umeyama(est_xyz, gt_xyz, with_scale=correct_scale)
    res = (s * (R @ np.asarray(est_xyz, np.float64).T).T + t) - np.asarray(gt_xyz, np.float64)
    return float(np.sqrt((res ** 2).sum(1).mean()))


def camera_centres(cam_T_world):
    """world positions of the cameras from world->camera poses [n,7] = [t, q(xyzw)] (the tracker's state)"""
    from . import se3
    return se3.inv(torch.as_tensor(cam_T_world).double())[:, :3].cpu().numpy()
