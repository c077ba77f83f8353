"""metrics of the headline benchmark (PSNR, depth L1, ATE-RMSE with Umeyama alignment)"""
import numpy as np
import torch


def test_psnr_and_depth_l1():
    from nerfslam import eval as ev
    a = np.full((4, 5, 3), 0.5)
    assert abs(ev.psnr(a + 0.1, a) - 20.0) < 1e-9 and ev.psnr(a, a) > 150
    d = np.random.default_rng(0).uniform(1, 3, (6, 7))
    assert ev.depth_l1_cm(2.0 * d, d) < 1e-9                      # global scale is factored out
    assert 0.0 < ev.depth_l1_cm(d + 0.3 * np.sin(d), d) < 200.0


def test_ate_is_invariant_to_similarity_and_measures_noise():
    from nerfslam import eval as ev
    rng = np.random.default_rng(1)
    gt = np.cumsum(rng.normal(0, 0.1, (50, 3)), 0)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    est = 2.5 * (q @ gt.T).T + np.array([1.0, -2.0, 0.5])
    assert ev.ate_rmse(est, gt) < 1e-10
    assert ev.ate_rmse(est, gt, correct_scale=False) > 0.1            # SE(3) alignment cannot absorb the scale
    noisy = est + rng.normal(0, 0.05, est.shape)
    r = ev.ate_rmse(noisy, gt)
    assert 0.5 * 0.05 * np.sqrt(3) / 2.5 < r < 1.5 * 0.05 * np.sqrt(3) / 2.5
    s, R, t = ev.umeyama(gt, est)
    assert abs(s - 2.5) < 1e-9 and np.abs(R - q).max() < 1e-9


def test_camera_centres_from_tracker_poses():
    from nerfslam import eval as ev, se3
    T = torch.tensor([[0.3, -0.2, 0.1, 0.0, 0.0, np.sin(0.4), np.cos(0.4)]])
    c = ev.camera_centres(T)
    back = se3.act(T.double(), torch.tensor([[c[0, 0], c[0, 1], c[0, 2], 1.0]]).double())
    assert back[0, :3].abs().max().item() < 1e-12                 # the camera centre maps to the camera origin
