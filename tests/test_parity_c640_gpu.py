"""Parity at the BASELINE shape (configs[1]: 640x480 -> 60x80 grid, E=48 active edges, M=96, P=10) of the kernels
the tracking step spends its time in, against the CPU oracle -- VERDICT r01 "parity holes":

  * ns_reproject (runs in every update(); reference networks/geom/projective_ops.py:98-145 through
    visual_frontend.py:909-918) vs oracle.reproject (itself pinned to the reference's own output by
    tests/test_oracle_pins.py::test_golden_reproject), on the golden inputs and on the C640 problem;
  * corr_lookup_coop_kernel (correlation_kernels.cu:20-70 x 4 levels) bit-exact at E=48, 60x80, tiled and row-major
    volumes, with 5 % of the coordinates far out of bounds;
  * level 0 of corr_volume_tiled_kernel (corr.py:63-72) vs the exactly-accumulated oracle at 60x80;
  * solve_depth / ba_solve + retraction / covariances on the C640 problem (M=96, P=10).
"""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C640 = dict(ht=60, wd=80, P=10, M=96, seed=4, kf0=6, extra_fixed=3, sensed_frac=0.1)


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _reproject_hip(dev, poses, disps, intr, ii, jj):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    n, (_, ht, wd) = ii.shape[0], disps.shape
    coords = torch.full((n, ht, wd, 2), float("nan"), device=dev)
    valid = torch.full((n, ht, wd), float("nan"), device=dev)
    a = [T(x, dev) for x in (poses, disps, intr, ii, jj)]     # named: a temporary would be freed (and its block reused) before the launch
    check(lib().ns_reproject(*[ptr(x) for x in a], ptr(coords), ptr(valid), n, ht, wd, stream_ptr()), "reproject")
    return coords.cpu().numpy(), valid.cpu().numpy()


def test_reproject_kernel_on_golden_inputs(oracle_mod, dev):
    """the reference's own output (tools/gen_golden.py ran projective_ops.projective_transform) AND the oracle"""
    g = np.load(os.path.join(GOLD, "projective_transform.npz"))
    coords, valid = _reproject_hip(dev, g["poses"], g["disps"], g["intr"], g["ii"], g["jj"])
    rc, rv = oracle_mod.reproject(g["poses"], g["disps"], g["intr"], g["ii"], g["jj"])
    np.testing.assert_array_equal(valid, rv)
    np.testing.assert_array_equal(valid, g["valid"][..., 0])
    scale = np.abs(g["coords"]).max()
    assert np.abs(coords - rc).max() <= 2e-5 * scale
    assert np.abs(coords - g["coords"]).max() <= 2e-5 * scale


@pytest.mark.parametrize("case", ["c640", "behind_camera", "self_edges"])
def test_reproject_kernel_c640(oracle_mod, dev, case):
    """coords within 2e-5 of max|coords| wherever the depth is not clamped, valid mask exact.
    MIN_DEPTH: the torch path the frontend mirrors uses 0.2 (projective_ops.py:9), the BA kernels 0.25 -- the cases
    with points behind / close to the camera would expose a mix-up."""
    p = synth.make_problem(**C640)
    poses, disps = p["poses"].copy(), p["disps"].copy()
    ii, jj = p["ii"], p["jj"].copy()
    if case == "behind_camera":
        poses[8, 2] += 3.0            # frame 8 far forward: many points of other frames fall behind it or inside MIN_DEPTH
        disps[7] = 4.5                # depth 0.22: between the two MIN_DEPTH constants after a small motion
    if case == "self_edges":
        jj[:5] = ii[:5]
    coords, valid = _reproject_hip(dev, poses, disps, p["intr"], ii, jj)
    rc, rv = oracle_mod.reproject(poses, disps, p["intr"], ii, jj)
    assert np.isfinite(coords).all() and np.isfinite(valid).all()           # every output element is written
    np.testing.assert_array_equal(valid, rv)
    if case == "behind_camera":
        assert (rv == 0).mean() > 0.02 and (rv == 1).mean() > 0.5
    ok = rv > 0
    err = np.abs(coords - rc)[ok].max()
    assert err <= 2e-5 * np.abs(rc[ok]).max(), err
    # invalid pixels still carry the reference's clamped-depth coordinates (the lookup reads them): loose bound, they
    # are dominated by 1/clamp amplification
    if (~ok).any():   # (a pixel whose depth rounds to opposite sides of the 0.1 clamp may legitimately jump: allow 0.1 % of them)
        bad = np.abs(coords - rc)[~ok].max(-1) > 1e-3 * np.maximum(1.0, np.abs(rc[~ok]).max(-1))
        assert bad.mean() <= 1e-3, bad.mean()


@pytest.mark.parametrize("tiled", [True, False])
def test_coop_lookup_bitexact_at_baseline_shape(oracle_mod, dev, tiled):
    """E=48 edges of a 16-frame feature bank at 60x80: the fused cooperative lookup vs the oracle's four per-level
    lookups on the SAME volumes (device build, fetched back), bit for bit; 5 % far-out-of-bounds coordinates, a few
    border-straddling ones, NaN / inf."""
    from nerfslam.corr import CorrBlock
    ht, wd, E, nfr = 60, 80, 48, 16
    g = torch.Generator().manual_seed(640)
    bank = (torch.randn((nfr, ht * wd, 128), generator=g) / 4.0).half().to(dev)
    ii = torch.randint(0, nfr, (E,), generator=g).to(dev)
    jj = torch.randint(0, nfr, (E,), generator=g).to(dev)
    pyr = CorrBlock.build_pyramid(bank, bank, ii, jj, E, ht, wd, tiled=tiled)
    blk = CorrBlock.from_pyramid(pyr, tiled=tiled, hw=(ht, wd))
    vols = [v.cpu().numpy() for v in (blk.untiled() if tiled else blk.corr_pyramid)]
    assert [v.shape for v in vols] == [(E, ht, wd, ht >> l, wd >> l) for l in range(4)]
    coords = _c640_lookup_coords(E, ht, wd)
    out = blk(torch.from_numpy(coords).to(dev)[None])[0].cpu().numpy()
    assert out.shape == (E, 196, ht, wd)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    cf = np.where(np.isfinite(cf), cf, np.float32(-1e5))        # the reference floor()s non-finite coordinates to "outside"
    for l in range(4):
        ref = oracle_mod.corr_index_forward(vols[l], cf / np.float32(2 ** l), 3).reshape(E, 49, ht, wd)
        got = out[:, 49 * l:49 * (l + 1)]
        same = (got.view(np.uint16) == ref.view(np.uint16)) | ((got == 0) & (ref == 0))
        assert same.all(), f"level {l}: {int((~same).sum())} of {same.size} differ"
    oob = (coords[..., 0] < -8) | (coords[..., 0] > wd + 8) | (coords[..., 1] < -8) | (coords[..., 1] > ht + 8)
    assert 0.03 < oob.mean() < 0.08
    assert (out.transpose(0, 2, 3, 1)[oob] == 0).all()


def _c640_lookup_coords(E, ht, wd):
    """the coordinate set of the bit-exactness test above: 5 % far out of bounds, border-straddling, NaN / inf"""
    _, coords = synth.lookup_inputs(E, ht, wd, seed=48, oob_frac=0.05, levels=0)
    coords[0, 0, 0] = [-2.5, -2.25]
    coords[-1, -1, -1] = [wd + 1.5, ht + 0.75]
    coords[3, 10, 10] = [wd - 0.5, ht - 0.5]
    coords[5, 1, 1] = [1e9, 3.0]
    coords[5, 1, 2] = [3.0, -np.inf]
    coords[7, 2, 2] = [np.nan, 1.0]
    return coords


def test_fused_lookup_encoder_at_baseline_shape(oracle_mod, dev):
    """corr_lookup_enc_kernel -- the PRODUCT DEFAULT of the c640 path (lookup of correlation_kernels.cu:20-70 x 4 levels +
    Conv2d(196,128,1) + ReLU of droid_net.py:83-87 in one launch) -- on the exact E=48 / 60x80 / 5 %-OOB / border / NaN / inf
    coordinate set the unfused kernel is held to above, through slot-permuted pooled volumes.
    Chain of evidence: the ORACLE's four per-level lookups of the device-built volumes (bit-exact to the unfused kernel, asserted
    again here) -> the layer evaluated in float64 on those 196 planes -> the fused output within one f16 rounding of it
    (f32 MFMA accumulation, one rounding to half; same bound as test_lookup_fused_with_the_correlation_encoder).  Pixels whose
    window lies wholly outside the volume must give exactly relu(bias)."""
    from nerfslam.corr import CorrPool
    from nerfslam.update_op import CorrEncoderWeights
    ht, wd, E, nfr = 60, 80, 48, 16
    g = torch.Generator().manual_seed(640)
    bank = (torch.randn((nfr, ht * wd, 128), generator=g) / 4.0).half().to(dev)
    ii = torch.randint(0, nfr, (E,), generator=g).to(dev)
    jj = torch.randint(0, nfr, (E,), generator=g).to(dev)
    pool = CorrPool(ht, wd, E + 5, dev)
    slots = torch.randperm(E + 5, generator=g)[:E].to(torch.int32).to(dev)
    pool.build(bank, bank, ii, jj, slots)
    coords = _c640_lookup_coords(E, ht, wd)
    cd = torch.from_numpy(coords).to(dev)[None]
    W = (torch.randn((128, 196, 1, 1), generator=g) / 14.0).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    fused = pool.lookup_encoded(cd, slots, CorrEncoderWeights(W, b)).c1
    assert fused.shape == (E, ht, wd, 128) and fused.dtype == torch.float16 and torch.isfinite(fused).all()
    # oracle lookup of the same volumes
    vols = [v.cpu().numpy() for v in pool.block(slots).corr_pyramid]
    assert [v.shape for v in vols] == [(E, ht, wd, ht >> l, wd >> l) for l in range(4)]
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    cf = np.where(np.isfinite(cf), cf, np.float32(-1e5))
    ref = np.concatenate([oracle_mod.corr_index_forward(vols[l], cf / np.float32(2 ** l), 3).reshape(E, 49, ht, wd)
                          for l in range(4)], 1)
    look = pool.lookup(cd, slots)[0].cpu().numpy()
    assert ((look.view(np.uint16) == ref.view(np.uint16)) | ((look == 0) & (ref == 0))).all()
    x = torch.from_numpy(ref.astype(np.float64)).permute(0, 2, 3, 1)                       # [E,ht,wd,196]
    want = torch.relu(x @ W.half().double().cpu().reshape(128, 196).t() + b.double().cpu())
    scale = float(want.abs().max())
    err = (fused.cpu().double() - want).abs()
    assert float(err.max()) <= 1.5e-3 * scale + 1e-3, (float(err.max()), scale)
    # per-element: within one f16 ulp of the float64 value (+ the f32 accumulation error of 196 products)
    ulp = np.spacing(want.numpy().astype(np.float16)).astype(np.float64)
    assert (err.numpy() <= ulp + 2e-3).all()
    oob = (coords[..., 0] < -8) | (coords[..., 0] > wd + 8) | (coords[..., 1] < -8) | (coords[..., 1] > ht + 8)
    oob |= ~np.isfinite(coords).all(-1)
    assert 0.03 < oob.mean() < 0.08
    rb = torch.relu(b).half().cpu()
    assert torch.equal(fused.cpu()[torch.from_numpy(oob)], rb.expand(int(oob.sum()), 128))


@pytest.mark.parametrize("tiled", [True, False])
def test_volume_level0_vs_oracle_at_baseline_shape(oracle_mod, dev, tiled):
    """corr_volume_tiled_kernel / corr_volume_pyramid_kernel level 0 at 60x80 (2 edges incl. a self pair): f32 MFMA
    accumulation + one rounding to half -> within 1 half-ulp (+2e-6) of the exactly-accumulated oracle; pools exact."""
    import ctypes as C
    from nerfslam.corr import CorrBlock
    ht, wd = 60, 80
    rng = np.random.default_rng(60)
    f = rng.standard_normal((3, 128, ht, wd)).astype(np.float16)
    bank = (torch.from_numpy(f).to(dev).reshape(3, 128, ht * wd) / 4.0).transpose(1, 2).contiguous()
    ii, jj = torch.tensor([0, 2], device=dev), torch.tensor([1, 2], device=dev)
    pyr = CorrBlock.build_pyramid(bank, bank, ii, jj, 2, ht, wd, tiled=tiled)
    blk = CorrBlock.from_pyramid(pyr, tiled=tiled, hw=(ht, wd))
    lv = [v.cpu().numpy() for v in (blk.untiled() if tiled else blk.corr_pyramid)]
    ref = oracle_mod.corr_pyramid(f[[0, 2]], f[[1, 2]])
    d = np.abs(lv[0].astype(np.float32) - ref[0].astype(np.float32))
    ulp = np.maximum(np.spacing(np.abs(ref[0]).astype(np.float16)).astype(np.float32), 2.0 ** -24)
    assert (d <= ulp + 2e-6).all(), f"max {float((d / ulp).max()):.2f} ulp"
    assert (d > 0).mean() < 0.05
    for l in range(3):
        h, w = ht >> l, wd >> l
        out = np.empty((2, ht, wd, h // 2, w // 2), np.uint16)
        oracle_mod.lib().orc_corr_pool_f16(lv[l].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                           C.c_long(2 * ht * wd), h, w)
        got = lv[l + 1].view(np.uint16)
        assert ((got == out) | ((lv[l + 1] == 0) & (out.view(np.float16) == 0))).all(), f"pool level {l + 1}"


def _rcm(oracle_mod, dev, p):
    import droid_backends
    ref = oracle_mod.reduced_camera_matrix(p["poses"], p["disps"], p["intr"], p["extr"], p["disps_sens"], p["targets"],
                                           p["weights"], p["eta"], p["ii"], p["jj"], p["kf0"], p["kf1"])
    d = {k: T(p[k], dev) for k in ("poses", "disps", "intr", "extr", "disps_sens", "targets", "weights", "eta", "ii", "jj")}
    got = droid_backends.reduced_camera_matrix(d["poses"], d["poses"], d["disps"], d["intr"], d["extr"], d["disps_sens"],
                                               d["targets"], d["weights"], d["eta"], d["ii"], d["jj"], p["kf0"], p["kf1"])
    return ref, got, d


def _close(got, ref, rel, what):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    err = np.abs(got.astype(np.float64) - np.asarray(ref, np.float64)).max()
    assert err <= rel * (np.abs(ref).max() + 1e-30), f"{what}: {err:.3e} vs {rel:.1e} * {np.abs(ref).max():.3e}"


def test_solve_depth_c640(oracle_mod, dev):
    import droid_backends
    p = synth.make_problem(**C640)
    ref, got, d = _rcm(oracle_mod, dev, p)
    P = p["kf1"] - p["kf0"]
    dx = (np.random.default_rng(9).standard_normal((P, 6)) * 1e-2).astype(np.float32)
    want = oracle_mod.solve_depth(dx, p["disps"], ref[2], ref[3], ref[4], p["ii"], p["jj"], p["kf0"], p["kf1"])
    disps = d["disps"].clone()
    droid_backends.solve_depth(T(dx, dev), disps, got[2], got[3], got[4], d["ii"], d["jj"], p["kf0"], p["kf1"])
    _close(disps - d["disps"], want - p["disps"], 1e-4, "dz")
    untouched = np.setdiff1d(np.arange(p["poses"].shape[0]), ref[5])
    assert torch.equal(disps[T(untouched, dev)], d["disps"][T(untouched, dev)])


@pytest.mark.parametrize("with_prior", [False, True])
def test_ba_solve_retract_and_covariances_c640(oracle_mod, dev, with_prior):
    """6P = 60 system of the C640 problem: device Cholesky + retraction vs float64 numpy; pose marginals and the
    depth covariances vs oracle.ba_covariances (pinned by tests/test_oracle_pins.py::test_covariance_block_*)."""
    from nerfslam import ba_plan
    p = synth.make_problem(**C640)
    ref, got, d = _rcm(oracle_mod, dev, p)
    kf0, kf1 = p["kf0"], p["kf1"]
    wTb = np.stack([oracle_mod.se3_inv64(q) for q in p["poses"]]).astype(np.float32)
    prior = wTb[kf0].copy() if with_prior else None
    if with_prior:
        prior[:3] += 1e-3
    H = got[0].clone()
    H += 1e-3 * torch.eye(H.shape[0], device=dev) * H.abs().max()
    delta, wTb_new, cTw_new, Hfull = oracle_mod.ba_solve_retract(H.cpu().numpy(), got[1].cpu().numpy(), wTb, p["extr"],
                                                                 kf0, kf1, prior_pose=prior)
    wd_, cd_ = T(wTb, dev), T(p["poses"], dev).clone()
    sol = ba_plan.ba_solve(H, got[1], kf0, kf1, wd_, cd_, T(p["extr"], dev),
                           prior_pose=None if prior is None else T(prior, dev), want_cov=True, want_hfull=True)
    assert sol["info"].item() == 0
    _close(sol["Hfull"], Hfull, 1e-12, "Hfull")
    _close(sol["dx"], delta.astype(np.float32), 1e-4, "dx")
    _close(wd_[kf0:kf1], wTb_new.astype(np.float32), 1e-5, "world_T_body")
    _close(cd_[kf0:kf1], cTw_new.astype(np.float32), 1e-5, "cam_T_world")
    assert torch.equal(wd_[:kf0].cpu(), torch.from_numpy(wTb[:kf0]))
    plan = ba_plan.BaPlan(p["ii"], p["jj"], kf0, kf1, dev)
    sig, zc, kx = oracle_mod.ba_covariances(Hfull, got[3].cpu().numpy(), got[2].cpu().numpy(), p["ii"], p["jj"], kf0, kf1,
                                            p["HW"])
    _close(sol["sigma_g"], sig.astype(np.float32), 2e-3, "sigma_g")
    z = ba_plan.depth_cov(plan, sol["Linv"], got[2], got[3], p["HW"])
    assert z.shape == zc.shape
    _close(z, zc.astype(np.float32), 2e-3, "z_cov")
    rel = np.abs(z.cpu().numpy() - zc) / np.abs(zc)
    assert np.median(rel) < 1e-4
