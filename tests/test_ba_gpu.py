"""GPU parity of the BA / geometry kernels against the CPU oracle (through the C ABI shim).

Tolerances (fp32 path; stated per SURVEY.md 8c / north_star "within a stated fp32 tolerance"):
  per-pixel quantities (E, C, b, Q, w):  |got-ref| <= 2e-5 * max|ref|   (different but equivalent
        evaluation order of the same float formulas: matrix form of the adjoints, FMA contraction)
  pixel-reduced quantities (Hs, vs, H, v): |got-ref| <= 2e-4 * max|ref| (fp32 partial sums in a
        different order; the oracle reduces in double)
  solved quantities (dx, poses, disps):   stated at each test.
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def close(got, ref, rel, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


PROBLEMS = [dict(ht=12, wd=16, P=5, M=24, seed=0),
            dict(ht=9, wd=13, P=4, M=14, seed=1, kf0=3, extra_fixed=2, sensed_frac=0.3),
            dict(ht=30, wd=40, P=8, M=60, seed=2, kf0=3, extra_fixed=3),
            dict(ht=43, wd=77, P=6, M=30, seed=3, sensed_frac=0.5),
            dict(ht=60, wd=80, P=10, M=96, seed=4, kf0=6, extra_fixed=3, sensed_frac=0.1)]   # BASELINE C640: M=96, P=10


@pytest.mark.parametrize("cfg", PROBLEMS)
def test_projective_transform_k1(oracle_mod, dev, cfg):
    from nerfslam import ba_plan
    p = synth.make_problem(**cfg)
    ref = oracle_mod.projective_transform(p["targets"], p["weights"], p["poses"], p["disps"], p["intr"], p["extr"],
                                          p["ii"], p["jj"])
    got = ba_plan.projective_transform(T(p["targets"], dev), T(p["weights"], dev), T(p["poses"], dev),
                                       T(p["disps"], dev), T(p["intr"], dev), T(p["extr"], dev), T(p["ii"], dev),
                                       T(p["jj"], dev))
    for k in ("Eiz", "Ejz", "Cii", "bz"):
        close(got[k], ref[k], 2e-5, k)
    for k in ("Hs", "vs"):
        close(got[k], ref[k], 2e-4, k)


def test_projective_transform_extrinsics_and_stereo(oracle_mod, dev):
    """non-identity cam_T_body (incl. the reference's in-place adjoint quirk, droid_kernels.cu:380-381)
    and a stereo self-edge (ii == jj, :249-259, 367, 432)."""
    from nerfslam import ba_plan
    p = synth.make_problem(ht=10, wd=12, P=4, M=12, seed=5)
    p["extr"] = np.array([0.065, -0.02, -0.008, 0.0077, -0.0105, -0.7018, 0.7123], np.float32)
    p["extr"][3:] /= np.linalg.norm(p["extr"][3:])
    p["jj"][0] = p["ii"][0]
    ref = oracle_mod.projective_transform(p["targets"], p["weights"], p["poses"], p["disps"], p["intr"], p["extr"],
                                          p["ii"], p["jj"])
    got = ba_plan.projective_transform(T(p["targets"], dev), T(p["weights"], dev), T(p["poses"], dev),
                                       T(p["disps"], dev), T(p["intr"], dev), T(p["extr"], dev), T(p["ii"], dev),
                                       T(p["jj"], dev))
    for k in ("Eiz", "Ejz", "Cii", "bz"):
        close(got[k], ref[k], 2e-5, k)
    for k in ("Hs", "vs"):
        close(got[k], ref[k], 2e-4, k)
    assert np.abs(ref["Eiz"][0]).max() == 0 and got["Eiz"][0].abs().max().item() == 0


def _rcm_both(oracle_mod, dev, p):
    import droid_backends
    ref = oracle_mod.reduced_camera_matrix(p["poses"], p["disps"], p["intr"], p["extr"], p["disps_sens"], p["targets"],
                                           p["weights"], p["eta"], p["ii"], p["jj"], p["kf0"], p["kf1"])
    d = {k: T(p[k], dev) for k in ("poses", "disps", "intr", "extr", "disps_sens", "targets", "weights", "eta", "ii",
                                   "jj")}
    got = droid_backends.reduced_camera_matrix(d["poses"], d["poses"], d["disps"], d["intr"], d["extr"],
                                               d["disps_sens"], d["targets"], d["weights"], d["eta"], d["ii"], d["jj"],
                                               p["kf0"], p["kf1"])
    return ref, got, d


@pytest.mark.parametrize("cfg", PROBLEMS)
def test_reduced_camera_matrix(oracle_mod, dev, cfg):
    p = synth.make_problem(**cfg)
    ref, got, _ = _rcm_both(oracle_mod, dev, p)
    H, v, Q, E, w = got
    rH, rv, rQ, rE, rw, kx = ref
    P = p["kf1"] - p["kf0"]
    assert H.shape == (6 * P, 6 * P) and v.shape == (6 * P, 1)
    assert Q.shape == rQ.shape and E.shape == rE.shape and w.shape == rw.shape
    close(E, rE, 2e-5, "E")
    close(Q, rQ, 2e-5, "Q")
    close(w, rw, 2e-5, "w")
    close(H, rH, 2e-4, "H")
    close(v, rv, 2e-4, "v")
    Hn = H.cpu().numpy()
    assert np.abs(Hn - Hn.T).max() <= 1e-6 * np.abs(Hn).max()


@pytest.mark.parametrize("cfg", PROBLEMS[:3])
def test_solve_depth(oracle_mod, dev, cfg):
    import droid_backends
    p = synth.make_problem(**cfg)
    ref, got, d = _rcm_both(oracle_mod, dev, p)
    P = p["kf1"] - p["kf0"]
    dx = (np.random.default_rng(9).standard_normal((P, 6)) * 1e-2).astype(np.float32)
    want = oracle_mod.solve_depth(dx, p["disps"], ref[2], ref[3], ref[4], p["ii"], p["jj"], p["kf0"], p["kf1"])
    disps = d["disps"].clone()
    out = droid_backends.solve_depth(T(dx, dev), disps, got[2], got[3], got[4], d["ii"], d["jj"], p["kf0"], p["kf1"])
    assert out is None
    # |dz| can be large where Q is large; compare the update relative to its own magnitude
    close(disps - d["disps"], want - p["disps"], 1e-4, "dz")


@pytest.mark.parametrize("cfg", PROBLEMS[:3])
@pytest.mark.parametrize("with_prior", [False, True])
def test_ba_solve_retract(oracle_mod, dev, cfg, with_prior):
    from nerfslam import ba_plan
    p = synth.make_problem(**cfg)
    ref, got, d = _rcm_both(oracle_mod, dev, p)
    kf0, kf1 = p["kf0"], p["kf1"]
    nb = p["poses"].shape[0]
    # world_T_body = cam_T_world^-1 (identity extrinsics)
    wTb = np.stack([oracle_mod.se3_inv64(q) for q in p["poses"]]).astype(np.float32)
    prior = wTb[kf0].copy() if with_prior else None
    if with_prior:
        prior[:3] += 1e-3
    H = got[0].clone()
    H += 1e-3 * torch.eye(H.shape[0], device=dev) * H.abs().max()  # gauge: keep the test system well conditioned
    delta, wTb_new, cTw_new, Hfull = oracle_mod.ba_solve_retract(H.cpu().numpy(), got[1].cpu().numpy(), wTb, p["extr"],
                                                                 kf0, kf1, prior_pose=prior)
    wd_, cd_ = T(wTb, dev), T(p["poses"], dev).clone()
    sol = ba_plan.ba_solve(H, got[1], kf0, kf1, wd_, cd_, T(p["extr"], dev),
                           prior_pose=None if prior is None else T(prior, dev), want_cov=True, want_hfull=True)
    assert sol["info"].item() == 0
    close(sol["Hfull"], Hfull, 1e-12, "Hfull")
    close(sol["dx"], delta.astype(np.float32), 1e-4, "dx")
    close(wd_[kf0:kf1], wTb_new.astype(np.float32), 1e-5, "world_T_body")
    close(cd_[kf0:kf1], cTw_new.astype(np.float32), 1e-5, "cam_T_world")
    assert torch.equal(wd_[:kf0].cpu(), torch.from_numpy(wTb[:kf0])) and nb > kf1 - 1
    # covariances (visual_frontend.py:1164-1230)
    plan = ba_plan.BaPlan(p["ii"], p["jj"], kf0, kf1, dev)
    L = np.linalg.cholesky(Hfull.astype(np.float32).astype(np.float64))
    Linv = np.linalg.inv(L)
    sg = Linv.T @ Linv
    close(sol["sigma_g"], np.stack([sg[6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(kf1 - kf0)]).astype(np.float32),
          2e-3, "sigma_g")
    try:
        sig, zc, kx = oracle_mod.ba_covariances(Hfull, got[3].cpu().numpy(), got[2].cpu().numpy(), p["ii"], p["jj"],
                                                kf0, kf1, p["HW"])
    except ValueError:
        return  # graph shape the reference's covariance block cannot handle either
    z = ba_plan.depth_cov(plan, sol["Linv"], got[2], got[3], p["HW"])
    close(z, zc.astype(np.float32), 2e-3, "z_cov")


def test_ba_solve_reports_indefinite(dev):
    from nerfslam import ba_plan
    H = -torch.eye(12, device=dev)
    v = torch.ones((12, 1), device=dev)
    sol = ba_plan.ba_solve(H, v, 0, 2, retract=False)
    assert sol["info"].item() > 0 and sol["dx"].abs().max().item() == 0


def test_frame_distance(oracle_mod, dev):
    import droid_backends
    p = synth.make_problem(ht=30, wd=40, P=7, M=20, seed=4, trans_sigma=0.3)
    n = p["poses"].shape[0]
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ii, jj = ii.reshape(-1).astype(np.int64), jj.reshape(-1).astype(np.int64)
    p["disps"][1] = 0.01  # everything far away: translation-only flow ~ 0
    p["poses"][2, 2] = 5.0  # camera 2 moved far forward: many points behind -> 1000 sentinel path
    ref = oracle_mod.frame_distance(p["poses"], p["disps"], p["intr"], ii, jj, 0.3)
    a = (T(p["poses"], dev), T(p["disps"], dev), T(p["intr"], dev), T(ii, dev), T(jj, dev), 0.3)
    got = droid_backends.frame_distance(*a)
    assert (ref == 1000.0).any() and (ref < 1000.0).any()
    np.testing.assert_array_equal(got.cpu().numpy() == 1000.0, ref == 1000.0)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    # bit-reproducible run to run (argsort of these distances decides the factor-graph indices)
    assert torch.equal(got, droid_backends.frame_distance(*a))


def test_dead_ops_api_parity(oracle_mod, dev):
    """projmap / iproj / depth_filter / solve_poses: exported by the reference, unused by its live path."""
    import droid_backends
    p = synth.make_problem(ht=14, wd=18, P=6, M=16, seed=6)
    poses, disps, intr = T(p["poses"], dev), T(p["disps"], dev), T(p["intr"], dev)
    ii, jj = T(p["ii"], dev), T(p["jj"], dev)
    c, v = droid_backends.projmap(poses, disps, intr, ii, jj)
    rc, rv = oracle_mod.projmap(p["poses"], p["disps"], p["intr"], p["ii"], p["jj"])
    close(c, rc, 1e-5, "projmap coords")
    assert (v.cpu().numpy() == rv).mean() > 0.999
    close(droid_backends.iproj(poses, disps, intr), oracle_mod.iproj(p["poses"], p["disps"], p["intr"]), 1e-5, "iproj")
    inds = np.arange(p["poses"].shape[0], dtype=np.int64)
    th = np.full(inds.shape, 0.05, np.float32)
    cnt = droid_backends.depth_filter(poses, disps, intr, T(inds, dev), T(th, dev)).cpu().numpy()
    rcnt = oracle_mod.depth_filter(p["poses"], p["disps"], p["intr"], inds, th)
    assert (cnt == rcnt).mean() > 0.999
    dx = (np.random.default_rng(1).standard_normal((3, 6)) * 0.05).astype(np.float32)
    dx[1, 3:] = 1e-6  # small-angle branch
    pp = poses.clone()
    assert droid_backends.solve_poses(pp, T(dx, dev), 1, 4) is None
    close(pp, oracle_mod.pose_retr(p["poses"], dx, 1, 4), 1e-6, "pose_retr")


@pytest.mark.parametrize("motion_only", [False, True])
def test_dead_ba_op_both_modes(oracle_mod, dev, motion_only):
    """droid_backends.ba (src/droid.cpp:133-165 -> ba_cuda, droid_kernels.cu:1441-1568; dead in NeRF-SLAM's live path): one
    iteration against the oracle's pieces put together the way ba_cuda does -- reduced camera matrix (or, motion_only, the
    plain pose block of K1's per-edge blocks), (ep + lm diag) damping, float64 solve, Exp(dx) * T retraction, depth update."""
    import droid_backends
    p = synth.make_problem(ht=12, wd=16, P=5, M=24, seed=8)
    lm, ep = 1e-4, 0.1
    kf0, kf1 = int(p["kf0"]), int(p["kf1"])
    P = kf1 - kf0
    t = {k: T(p[k], dev) for k in ("poses", "disps", "intr", "extr", "disps_sens", "targets", "weights", "eta", "ii", "jj")}
    poses, disps = t["poses"].clone(), t["disps"].clone()
    dx, dz = droid_backends.ba(poses, poses.clone(), disps, t["intr"], t["extr"], t["disps_sens"], t["targets"], t["weights"],
                               t["eta"], t["ii"], t["jj"], kf0, kf1, 1, lm, ep, motion_only)
    if motion_only:
        o = oracle_mod.projective_transform(p["targets"], p["weights"], p["poses"], p["disps"], p["intr"], p["extr"], p["ii"], p["jj"])
        A, b = np.zeros((6 * P, 6 * P)), np.zeros(6 * P)
        ri = np.concatenate([p["ii"], p["ii"], p["jj"], p["jj"]]) - kf0
        ci = np.concatenate([p["ii"], p["jj"], p["ii"], p["jj"]]) - kf0
        for blk, (r, c) in zip(o["Hs"].reshape(-1, 6, 6).astype(np.float64), zip(ri, ci)):
            if 0 <= r < P and 0 <= c < P:
                A[6 * r:6 * r + 6, 6 * c:6 * c + 6] += blk
        for vec, r in zip(o["vs"].reshape(-1, 6).astype(np.float64), np.concatenate([p["ii"], p["jj"]]) - kf0):
            if 0 <= r < P:
                b[6 * r:6 * r + 6] += vec
        A = np.tril(A) + np.tril(A, -1).T                                    # SimplicialLLT reads the lower triangle
        new_disps = p["disps"]
        assert dz is None
    else:
        H, v, Q, E, w, kx = oracle_mod.reduced_camera_matrix(p["poses"], p["disps"], p["intr"], p["extr"], p["disps_sens"],
                                                            p["targets"], p["weights"], p["eta"], p["ii"], p["jj"], kf0, kf1)
        A, b = H.astype(np.float64), v.astype(np.float64).reshape(-1)
        A = np.triu(A) + np.triu(A, 1).T
    A = A + np.diag(ep + lm * np.diag(A))
    rdx = np.linalg.solve(A, b).reshape(P, 6).astype(np.float32)
    close(dx, rdx, 2e-3, "dx")
    close(poses, oracle_mod.pose_retr(p["poses"], rdx, kf0, kf1), 1e-4, "retracted poses")
    if not motion_only:
        new_disps = oracle_mod.solve_depth(rdx, p["disps"], Q, E, w, p["ii"], p["jj"], kf0, kf1)
        assert dz is not None and dz.shape[0] == kx.shape[0]
    close(disps, new_disps, 2e-3, "disparities")
