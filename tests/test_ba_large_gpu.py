"""The BA back end beyond one workgroup's LDS (csrc/ba_solve_large.hip, the chunked covariance kernel of csrc/ba_solve.hip):
  * frontend windows with covariances above 18 poses (6P > 108) and any window above 32 poses (6P > 192)
  * the global BA over the whole buffer (visual_frontend.py:1255-1295): 6P = 1536 for config #5's 256 keyframes
  * depth covariances for windows of 33 poses and more (the round-2 tree skipped them with a warning)
Round 2 sent the first two to rocSOLVER through torch.linalg; nothing in nerfslam/ leaves the hand-written path now.
Reference semantics: oracle.ba_solve_retract / oracle.ba_covariances (float64 numpy restatements of :1123-1230).
"""
import numpy as np
import pytest
import torch

import synth
from test_parity_c640_gpu import T, _close, _rcm

pytestmark = pytest.mark.gpu


def _spd(n, rng, cond=1e3):
    """f32 SPD system whose two triangles differ by rounding (the HessianFactors keep the upper one, :1127-1134)"""
    A = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(A)
    ev = np.geomspace(1.0, cond, n)
    H = (Q * ev) @ Q.T
    H32 = H.astype(np.float32)
    H32 = np.triu(H32) + np.tril((H * (1 + 1e-6 * rng.standard_normal((n, n)))).astype(np.float32), -1)
    return H32


@pytest.mark.parametrize("P,want_cov,with_prior", [(20, True, True), (40, True, False), (33, True, True), (11, True, False),
                                                   (256, False, True), (256, True, False)])
def test_large_solve_vs_float64(oracle_mod, dev, P, want_cov, with_prior):
    from nerfslam import ba_plan
    rng = np.random.default_rng(1000 + P)
    n, kf0 = 6 * P, 2
    kf1 = kf0 + P
    H = _spd(n, rng)
    v = rng.standard_normal((n, 1)).astype(np.float32)
    wTb = np.zeros((kf1 + 1, 7), np.float32)
    for k in range(kf1 + 1):
        wTb[k, :3] = rng.normal(0, 0.3, 3)
        wTb[k, 3:] = synth.quat_exp(rng.normal(0, 0.2, 3))
    extr = np.array([0.01, -0.02, 0.03, *synth.quat_exp(np.array([0.02, 0.01, -0.03]))], np.float32)
    prior = wTb[kf0].copy() if with_prior else None
    if with_prior:
        prior[:3] += 2e-3
    delta, wTb_new, cTw_new, Hfull = oracle_mod.ba_solve_retract(H, v, wTb, extr, kf0, kf1, prior_pose=prior)
    wd_, cd_ = T(wTb, dev), torch.zeros((kf1 + 1, 7), device=dev)
    sol = ba_plan._ba_solve_large(T(H, dev), T(v, dev), kf0, kf1, wd_, cd_, T(extr, dev), None if prior is None else T(prior, dev),
                                  1e-4, 0.0, 0.0, True, want_cov, want_hfull=True)
    assert sol["info"].item() == 0
    _close(sol["Hfull"], Hfull, 1e-12, "Hfull")
    _close(sol["dx"], delta.astype(np.float32), 2e-6, "dx")
    _close(wd_[kf0:kf1], wTb_new.astype(np.float32), 1e-6, "world_T_body")
    _close(cd_[kf0:kf1], cTw_new.astype(np.float32), 1e-6, "cam_T_world")
    assert torch.equal(wd_[:kf0].cpu(), torch.from_numpy(wTb[:kf0])) and torch.equal(wd_[kf1:].cpu(), torch.from_numpy(wTb[kf1:]))
    if want_cov:
        L = np.linalg.cholesky(Hfull)
        Linv = np.linalg.inv(L)
        _close(sol["Linv"], Linv.astype(np.float32), 1e-5, "Linv")
        assert torch.equal(torch.triu(sol["Linv"], 1), torch.zeros_like(sol["Linv"]))
        sg = Linv.T @ Linv
        _close(sol["sigma_g"], np.stack([sg[6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(P)]).astype(np.float32), 1e-5, "sigma_g")
    # the routed entry point picks this path by size
    if 6 * P > (ba_plan.MAX_SMALL_SYSTEM_COV if want_cov else ba_plan.MAX_SMALL_SYSTEM):
        wd2, cd2 = T(wTb, dev), torch.zeros((kf1 + 1, 7), device=dev)
        sol2 = ba_plan.ba_solve(T(H, dev), T(v, dev), kf0, kf1, wd2, cd2, T(extr, dev), prior_pose=None if prior is None else T(prior, dev),
                                want_cov=want_cov)
        assert torch.equal(sol2["dx"], sol["dx"]) and torch.equal(wd2, wd_)


def test_large_solve_equals_lds_solve(dev):
    """same system through both kernels (6P = 60): two orderings of the same f64 arithmetic"""
    from nerfslam import ba_plan
    rng = np.random.default_rng(7)
    H, v = T(_spd(60, rng), dev), T(rng.standard_normal((60, 1)).astype(np.float32), dev)
    a = ba_plan.ba_solve(H, v, 0, 10, retract=False, want_cov=True, want_hfull=True, ep=0.1, lm=1e-4)
    b = ba_plan._ba_solve_large(H, v, 0, 10, None, None, None, None, 1e-4, 0.1, 1e-4, False, True, want_hfull=True)
    assert torch.equal(a["Hfull"], b["Hfull"])
    for k, tol in (("dx", 1e-6), ("Linv", 1e-6), ("sigma_g", 1e-6)):
        _close(b[k], a[k].cpu().numpy(), tol, k)


def test_large_solve_reports_indefinite(dev):
    from nerfslam import ba_plan
    n = 300
    H = torch.eye(n, device=dev)
    H[137, 137] = -1.0
    wTb = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]], device=dev).repeat(50, 1)
    cTw = wTb.clone()
    sol = ba_plan.ba_solve(H, torch.ones((n, 1), device=dev), 0, 50, wTb, cTw, wTb[0].clone(), want_cov=True)
    assert sol["info"].item() == 138 and sol["dx"].abs().max().item() == 0
    assert torch.equal(wTb, cTw) and sol["Linv"].abs().max().item() == 0 and sol["sigma_g"].abs().max().item() == 0


@pytest.mark.parametrize("cfg", [dict(ht=12, wd=16, P=36, M=260, seed=31, kf0=2, extra_fixed=2),
                                 dict(ht=6, wd=8, P=180, M=1100, seed=32, kf0=0, extra_fixed=0)], ids=["P36", "P180"])
def test_depth_covariances_of_large_windows(oracle_mod, dev, cfg):
    """windows whose L^-1 does not fit LDS whole: the chunked covariance kernel (one / two row chunks) and the large solve
    against oracle.ba_covariances (pinned by tests/test_oracle_pins.py::test_covariance_block_*)"""
    from nerfslam import ba_plan
    p = synth.make_problem(**cfg)
    ref, got, d = _rcm(oracle_mod, dev, p)
    kf0, kf1 = p["kf0"], p["kf1"]
    H = got[0].clone()
    H += 1e-3 * torch.eye(H.shape[0], device=dev) * H.abs().max()
    sol = ba_plan.ba_solve(H, got[1], kf0, kf1, retract=False, want_cov=True, want_hfull=True)
    assert sol["info"].item() == 0 and 6 * (kf1 - kf0) > ba_plan.MAX_SMALL_SYSTEM
    Hfull = sol["Hfull"].cpu().numpy()
    sig, zc, kx = oracle_mod.ba_covariances(Hfull, got[3].cpu().numpy(), got[2].cpu().numpy(), p["ii"], p["jj"], kf0, kf1, p["HW"])
    _close(sol["sigma_g"], sig.astype(np.float32), 2e-3, "sigma_g")
    plan = ba_plan.BaPlan(p["ii"], p["jj"], kf0, kf1, dev)
    z = ba_plan.depth_cov(plan, sol["Linv"], got[2], got[3], p["HW"])
    assert z.shape == zc.shape
    _close(z, zc.astype(np.float32), 2e-3, "z_cov")
    rel = np.abs(z.cpu().numpy() - zc) / np.abs(zc)
    assert np.median(rel) < 1e-4


def test_frontend_covariances_with_a_34_pose_window(oracle_mod, dev):
    """TrackingFrontend.ba() with a window the round-2 tree skipped covariances for (P >= 33): covariances are written"""
    from nerfslam.frontend import TrackingFrontend
    ht, wd, nkf = 12, 16, 34
    p = synth.make_problem(ht=ht, wd=wd, P=nkf, M=230, seed=33, noise=0.05)
    fe = TrackingFrontend(nkf + 2, ht * 8, wd * 8, p["intr"] * 8.0, dev, compute_covariances=True)
    fe.cam0_T_world[:p["poses"].shape[0]] = T(p["poses"], dev)
    fe.world_T_body[:p["poses"].shape[0]] = T(np.stack([oracle_mod.se3_inv64(q) for q in p["poses"]]).astype(np.float32), dev)
    fe.cam0_idepths[:p["disps"].shape[0]] = T(p["disps"], dev)
    fe.prior_pose = fe.world_T_body[0].clone()
    before = fe.cam0_idepths_cov.clone()
    sol = fe.ba(T(p["targets"], dev), T(p["weights"], dev), p["ii"], p["jj"], 0, itrs=2)
    assert sol["info"].item() == 0 and sol["Linv"].shape == (6 * nkf, 6 * nkf)
    kx = np.unique(p["ii"])
    assert (fe.cam0_idepths_cov[T(kx, dev)] != before[T(kx, dev)]).all() and torch.isfinite(fe.cam0_idepths_cov).all()
    assert (fe.world_T_body_cov[:nkf].diagonal(dim1=1, dim2=2) > 0).all()
