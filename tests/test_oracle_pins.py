"""Pins of the CPU oracle (CPU-only, no GPU, no /root/reference at run time).

The reference ships no tests and its CUDA extension cannot be built here, so the oracle is pinned by
  (a) golden vectors produced by RUNNING the reference's own Python modules (tools/gen_golden.py):
      networks/geom/projective_ops.py, networks/geom/chol.py, networks/modules/corr.py;
  (b) mathematical identities: bilinear lookup == F.grid_sample(align_corners=True, zeros);
      altcorr lookup == volume lookup of the f32 all-pairs volume; Schur-reduced solve == dense solve
      of the full (pose, depth) normal equations;
  (c) an independent numpy restatement of the half-precision accumulation order of K12.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------------------------
# (b) lookups
# ------------------------------------------------------------------------------------------------
def _grid_sample_lookup(vol, coords, r):
    B, h1, w1, h2, w2 = vol.shape
    v = torch.from_numpy(vol.astype(np.float32)).reshape(B * h1 * w1, 1, h2, w2)
    c = torch.from_numpy(coords).permute(0, 2, 3, 1).reshape(B * h1 * w1, 1, 1, 2)
    d = torch.arange(-r, r + 1).float()
    dx, dy = torch.meshgrid(d, d, indexing="ij")  # first index = x offset (channel-major in the kernel)
    pts = c + torch.stack([dx, dy], -1).reshape(1, -1, 1, 2)
    g = torch.stack([2 * pts[..., 0] / (w2 - 1) - 1, 2 * pts[..., 1] / (h2 - 1) - 1], -1)
    s = F.grid_sample(v, g, align_corners=True, padding_mode="zeros")
    rd = 2 * r + 1
    return s.reshape(B, h1, w1, rd, rd).permute(0, 3, 4, 1, 2).numpy()


@pytest.mark.parametrize("r", [1, 3])
def test_k12_f32_equals_grid_sample(oracle_mod, r):
    pyr, coords = synth.lookup_inputs(2, 7, 9, seed=3, dtype=np.float32, spread=6.0)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    got = oracle_mod.corr_index_forward(pyr[0], cf, r)
    ref = _grid_sample_lookup(pyr[0], cf, r)
    np.testing.assert_allclose(got, ref, atol=2e-6 * np.abs(ref).max())


def test_k12_f16_accumulation_order(oracle_mod):
    """numpy float16 restatement of correlation_kernels.cu:47-67: every product and every running sum
    rounded to half, loop order i (x) outer, j (y) inner, out-of-image taps skipped."""
    pyr, coords = synth.lookup_inputs(2, 6, 8, seed=4, spread=5.0)
    vol = pyr[0]
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    got = oracle_mod.corr_index_forward(vol, cf, 3)
    B, h1, w1, h2, w2 = vol.shape
    ref = np.zeros((B, 7, 7, h1, w1), np.float16)
    for n in range(B):
        for y in range(h1):
            for x in range(w1):
                x0, y0 = cf[n, 0, y, x], cf[n, 1, y, x]
                dx, dy = np.float32(x0 - np.floor(x0)), np.float32(y0 - np.floor(y0))
                w = {(1, 1): np.float16(dx * dy), (1, 0): np.float16(dx * (np.float32(1) - dy)),
                     (0, 1): np.float16((np.float32(1) - dx) * dy),
                     (0, 0): np.float16((np.float32(1) - dx) * (np.float32(1) - dy))}
                for i in range(8):
                    for j in range(8):
                        x1, y1 = int(np.floor(x0)) - 3 + i, int(np.floor(y0)) - 3 + j
                        if 0 <= x1 < w2 and 0 <= y1 < h2:
                            s = vol[n, y, x, y1, x1]
                            for (a, b), (oi, oj) in (((1, 1), (i - 1, j - 1)), ((1, 0), (i - 1, j)), ((0, 1), (i, j - 1)),
                                                     ((0, 0), (i, j))):
                                if 0 <= oi < 7 and 0 <= oj < 7:
                                    ref[n, oi, oj, y, x] = np.float16(ref[n, oi, oj, y, x] + np.float16(s * w[(a, b)]))
    assert (got.view(np.uint16) == ref.view(np.uint16)).all() or ((got == ref) | ((got == 0) & (ref == 0))).all()


def test_altcorr_equals_lookup_of_f32_volume(oracle_mod):
    rng = np.random.default_rng(5)
    B, H, W, Cc = 2, 6, 8, 64
    f1 = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    f2 = rng.standard_normal((B, H, W, Cc)).astype(np.float32)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    coords = (np.stack([gx, gy], -1)[None, None] + rng.uniform(-4, 4, (B, 1, H, W, 2))).astype(np.float32)
    got = oracle_mod.altcorr_forward(f1, f2, coords, 3)[:, 0]  # [B,49,H,W], channel = iy + 7*ix
    vol = np.einsum("bhwc,bxyc->bhwxy", f1, f2).astype(np.float32)
    ref = oracle_mod.corr_index_forward(vol, np.ascontiguousarray(coords[:, 0].transpose(0, 3, 1, 2)), 3)
    np.testing.assert_allclose(got, ref.reshape(B, 49, H, W), atol=2e-5 * np.abs(ref).max())


# ------------------------------------------------------------------------------------------------
# (a) golden vectors from the reference's Python
# ------------------------------------------------------------------------------------------------
def test_golden_projective_transform(oracle_mod):
    """oracle K1 reprojection / Jacobians == reference networks/geom/projective_ops.py:98-145
    (same sign flip and [w,t] reordering, :132-138) wherever both call the pixel valid."""
    g = np.load(os.path.join(GOLD, "projective_transform.npz"))
    ht, wd = g["disps"].shape[1:]
    for e in range(g["ii"].shape[0]):
        i, j = int(g["ii"][e]), int(g["jj"][e])
        coords, Ji, Jj, Jz = oracle_mod.edge_jacobians(g["poses"][i], g["poses"][j], g["disps"][i], g["intr"], g["extr"])
        ok = (g["valid"][e, ..., 0].reshape(-1) > 0)
        assert ok.mean() > 0.9
        for name, got, ref, tol in (("coords", coords, g["coords"][e].reshape(-1, 2), 2e-5),
                                    ("Ji", Ji, g["Ji"][e].reshape(-1, 2, 6), 2e-4),
                                    ("Jj", Jj, g["Jj"][e].reshape(-1, 2, 6), 2e-4),
                                    ("Jz", Jz, g["Jz"][e].reshape(-1, 2), 2e-4)):
            err = np.abs(got[ok] - ref[ok]).max()
            assert err <= tol * np.abs(ref[ok]).max(), f"edge {e} {name}: {err}"


def test_golden_schur_solve(oracle_mod):
    """oracle reduced camera matrix + dense solve + depth back-substitution == reference
    networks/geom/chol.py:46-73 (`schur_solve`, ep=0.1, lm=0) run on the same per-edge blocks."""
    g = np.load(os.path.join(GOLD, "schur_solve.npz"))
    kf0, kf1 = int(g["kf0"]), int(g["kf1"])
    H, v, Q, E, w, kx = oracle_mod.reduced_camera_matrix(g["poses"], g["disps"], g["intr"], g["extr"], g["disps_sens"],
                                                         g["targets"], g["weights"], g["eta"], g["ii"], g["jj"], kf0, kf1)
    np.testing.assert_array_equal(kx, g["kx"])
    n = H.shape[0]
    dx = np.linalg.solve(H.astype(np.float64) + 0.1 * np.eye(n), v.astype(np.float64)[:, 0]).reshape(-1, 6)
    assert np.abs(dx - g["dx"]).max() <= 2e-4 * np.abs(g["dx"]).max()
    new = oracle_mod.solve_depth(dx.astype(np.float32), g["disps"], Q, E, w, g["ii"], g["jj"], kf0, kf1)
    dz = (new - g["disps"])[kx].reshape(len(kx), -1)
    assert np.abs(dz - g["dz_masked"]).max() <= 5e-4 * np.abs(g["dz_masked"]).max()


def test_golden_corr_pyramid(oracle_mod):
    """CorrBlock.__init__ of the reference (corr.py:23-38, 63-72) run in half on the CPU."""
    g = np.load(os.path.join(GOLD, "corr_pyramid.npz"))
    f32 = np.load(os.path.join(GOLD, "corr_pyramid_f32_level0.npz"))["level0"]
    pyr = oracle_mod.corr_pyramid(g["fmap1"][0], g["fmap2"][0])
    # level 0: both are one half-rounding away from the exact product sum (f32 run of the same code)
    ulp = np.maximum(np.spacing(np.abs(f32).astype(np.float16)).astype(np.float32), 2.0 ** -24)
    assert (np.abs(pyr[0].astype(np.float32) - f32) <= 0.5 * ulp + 2e-6).all()
    assert (np.abs(g["level0"].astype(np.float32) - f32) <= 1.0 * ulp + 2e-6).all()
    # pooled levels: the oracle's pooling applied to the REFERENCE's level l reproduces the reference's
    # level l+1 bit for bit (f32 accumulate in window order, one rounding)
    import ctypes as C
    for l in range(3):
        src = np.ascontiguousarray(g[f"level{l}"])
        nsl = src.shape[0] * src.shape[1] * src.shape[2]
        h, w = src.shape[3:]
        out = np.empty(src.shape[:3] + (h // 2, w // 2), np.uint16)
        oracle_mod.lib().orc_corr_pool_f16(src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(nsl), h, w)
        assert (out == g[f"level{l + 1}"].view(np.uint16)).all(), f"level {l + 1}"


# ------------------------------------------------------------------------------------------------
# (b) Schur identity and SE3 algebra
# ------------------------------------------------------------------------------------------------
def test_schur_reduction_equals_full_system(oracle_mod):
    """(H, v) of the oracle == Schur complement of the full normal equations [[A, E],[E^T, C]] built from
    the oracle's K1 outputs in float64; dz of solve_depth == the depth part of the full solution."""
    p = synth.make_problem(ht=5, wd=6, P=3, M=6, seed=8)
    k1 = oracle_mod.projective_transform(p["targets"], p["weights"], p["poses"], p["disps"], p["intr"], p["extr"],
                                         p["ii"], p["jj"])
    H, v, Q, E, w, kx = oracle_mod.reduced_camera_matrix(p["poses"], p["disps"], p["intr"], p["extr"], p["disps_sens"],
                                                         p["targets"], p["weights"], p["eta"], p["ii"], p["jj"], 0, 3)
    P, HW, K = 3, p["HW"], len(kx)
    kk = np.searchsorted(kx, p["ii"])
    A = np.zeros((6 * P, 6 * P)); a = np.zeros(6 * P); Ef = np.zeros((6 * P, K * HW)); Cd = np.zeros(K * HW); wf = np.zeros(K * HW)
    for e in range(len(p["ii"])):
        i, j, k = p["ii"][e], p["jj"][e], kk[e]
        for (r, c, blk) in ((i, i, 0), (i, j, 1), (j, i, 2), (j, j, 3)):
            A[6 * r:6 * r + 6, 6 * c:6 * c + 6] += k1["Hs"][blk, e]
        a[6 * i:6 * i + 6] += k1["vs"][0, e]; a[6 * j:6 * j + 6] += k1["vs"][1, e]
        Ef[6 * i:6 * i + 6, k * HW:(k + 1) * HW] += k1["Eiz"][e]
        Ef[6 * j:6 * j + 6, k * HW:(k + 1) * HW] += k1["Ejz"][e]
        Cd[k * HW:(k + 1) * HW] += k1["Cii"][e]; wf[k * HW:(k + 1) * HW] += k1["bz"][e]
    Cd += p["eta"].reshape(-1)[:K * HW]
    S = A - (Ef / Cd) @ Ef.T
    s = a - Ef @ (wf / Cd)
    assert np.abs(H - S).max() <= 1e-4 * np.abs(S).max()
    assert np.abs(v[:, 0] - s).max() <= 1e-4 * np.abs(s).max()


def test_se3_algebra_consistency(oracle_mod):
    from nerfslam import se3
    rng = np.random.default_rng(9)
    for _ in range(10):
        pi = np.concatenate([rng.normal(0, 1, 3), synth.quat_exp(rng.normal(0, 1, 3))]).astype(np.float32)
        pj = np.concatenate([rng.normal(0, 1, 3), synth.quat_exp(rng.normal(0, 1, 3))]).astype(np.float32)
        rel = oracle_mod.se3_rel(pi, pj)  # Gj * Gi^-1 (droid_kernels.cu:107-120)
        ref = se3.mul(torch.from_numpy(pj).double(), se3.inv(torch.from_numpy(pi).double())).numpy()
        ref = ref * np.sign(ref[6] * rel[6])
        np.testing.assert_allclose(rel, ref, atol=2e-6)
        X = rng.normal(0, 1, 4).astype(np.float32)
        np.testing.assert_allclose(oracle_mod.se3_act(pi, X), se3.act(torch.from_numpy(pi).double(), torch.from_numpy(X).double()).numpy(), atol=3e-6)
        J = rng.normal(0, 1, 6).astype(np.float32)
        np.testing.assert_allclose(oracle_mod.se3_adj(pi, J), se3.adjT(torch.from_numpy(pi).double(), torch.from_numpy(J).double()).numpy(), atol=5e-6)
        xi = rng.normal(0, 0.3, 6)
        T = oracle_mod.se3_exp64(xi)
        np.testing.assert_allclose(oracle_mod.se3_log64(T), xi, atol=1e-10)
        np.testing.assert_allclose(se3.exp_wv(torch.from_numpy(xi)).numpy(), T, atol=1e-12)
        np.testing.assert_allclose(se3.log_wv(torch.from_numpy(T)).numpy(), xi, atol=1e-10)
        # droid's own exponential ([tau, phi] order, droid_kernels.cu:160-188) agrees with the [omega, v] one
        e2 = oracle_mod.se3_exp(np.concatenate([xi[3:], xi[:3]]).astype(np.float32))
        e2 = e2 * np.sign(e2[6] * T[6])
        np.testing.assert_allclose(e2, T, atol=3e-6)


def test_golden_reproject(oracle_mod):
    """oracle reprojection (the frontend's per-update reproject) == the reference's projective_transform
    output `x1` / `valid` (tests/golden/projective_transform.npz)."""
    g = np.load(os.path.join(GOLD, "projective_transform.npz"))
    coords, valid = oracle_mod.reproject(g["poses"], g["disps"], g["intr"], g["ii"], g["jj"])
    np.testing.assert_array_equal(valid, g["valid"][..., 0])
    assert np.abs(coords - g["coords"]).max() <= 2e-5 * np.abs(g["coords"]).max()


def test_cvx_upsample_matches_reference_function(oracle_mod):
    """utils/flow_viz.py:166-183 run on seeded inputs (tools/gen_golden.py section 5), incl. the pow variant"""
    z = np.load(os.path.join(GOLD, "cvx_upsample.npz"))
    up = oracle_mod.cvx_upsample(z["data"], z["mask"])
    assert np.abs(up - z["up"]).max() <= 2e-6
    up2 = oracle_mod.cvx_upsample(z["data"], z["mask"], 0.5)
    assert np.abs(up2 - z["up_pow05"]).max() <= 2e-6


# ------------------------------------------------------------------------------------------------
# (c) covariance block (visual_frontend.py:1164-1230): identity pin of oracle.ba_covariances
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [dict(P=3, M=6, kf0=0, extra_fixed=0), dict(P=3, M=20, kf0=2, extra_fixed=2)])
def test_covariance_block_equals_full_inverse(oracle_mod, cfg):
    """The full normal equations [[A, X],[X^T, C]] over (window poses, every depth map with outgoing edges) are
    assembled here from K1's raw per-edge blocks in float64 and inverted: the pose block of the inverse must equal
    sigma_g, the diagonal of its depth block must equal z_cov of oracle.ba_covariances(marginal_form=True).  The
    reference's own composition (:1215 multiplies by L^-1 where the marginal needs L^-T) is the same code with that
    one operand transposed; the two forms are checked to differ (so the switch is live) and to share the Q term."""
    p = synth.make_problem(ht=4, wd=5, seed=21, **cfg)
    kf0, kf1, HW, P = p["kf0"], p["kf1"], p["HW"], cfg["P"]
    k1 = oracle_mod.projective_transform(p["targets"], p["weights"], p["poses"], p["disps"], p["intr"], p["extr"],
                                         p["ii"], p["jj"])
    H, v, Q, E, w, kx = oracle_mod.reduced_camera_matrix(p["poses"], p["disps"], p["intr"], p["extr"], p["disps_sens"],
                                                         p["targets"], p["weights"], p["eta"], p["ii"], p["jj"], kf0, kf1)
    K = len(kx)
    kk = np.searchsorted(kx, p["ii"])
    n = 6 * P
    A = np.zeros((n, n)); X = np.zeros((n, K * HW)); Cd = np.zeros(K * HW)
    for e in range(len(p["ii"])):
        i, j, k = int(p["ii"][e]) - kf0, int(p["jj"][e]) - kf0, kk[e]
        for (r, c, blk) in ((i, i, 0), (i, j, 1), (j, i, 2), (j, j, 3)):
            if 0 <= r < P and 0 <= c < P:
                A[6 * r:6 * r + 6, 6 * c:6 * c + 6] += k1["Hs"][blk, e]
        if 0 <= i < P:
            X[6 * i:6 * i + 6, k * HW:(k + 1) * HW] += k1["Eiz"][e]
        if 0 <= j < P:
            X[6 * j:6 * j + 6, k * HW:(k + 1) * HW] += k1["Ejz"][e]
        Cd[k * HW:(k + 1) * HW] += k1["Cii"][e]
    Cd += p["eta"].reshape(-1)[:K * HW].astype(np.float64)
    lam = 1e-2 * np.abs(np.diag(A)).max()                    # gauge fixing (the frontend's prior plays this role)
    A += lam * np.eye(n)
    full = np.block([[A, X], [X.T, np.diag(Cd)]])
    inv = np.linalg.inv(full)
    Hred = H.astype(np.float64) + lam * np.eye(n)
    assert np.abs(Hred - (A - (X / Cd) @ X.T)).max() <= 2e-4 * np.abs(Hred).max()       # same system
    sig, z, kx2 = oracle_mod.ba_covariances(Hred, E, Q, p["ii"], p["jj"], kf0, kf1, HW, marginal_form=True)
    np.testing.assert_array_equal(kx2, kx)
    want_sig = np.stack([inv[6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(P)])
    assert np.abs(sig - want_sig).max() <= 2e-3 * np.abs(want_sig).max()
    want_z = np.diag(inv)[n:].reshape(K, HW)
    assert np.abs(z - want_z).max() <= 2e-3 * np.abs(want_z).max(), np.abs(z / want_z - 1).max()
    # the reference's form: same Q term, different (L^-1 instead of L^-T) correction
    sig_r, z_r, _ = oracle_mod.ba_covariances(Hred, E, Q, p["ii"], p["jj"], kf0, kf1, HW)
    np.testing.assert_array_equal(sig_r, sig)
    assert (z_r >= Q - 1e-12).all() and np.abs(z_r - z).max() > 1e-6 * np.abs(z).max()
    L = np.linalg.cholesky(Hred.astype(np.float32).astype(np.float64))
    Es = np.zeros((n, K * HW))                               # E of the reduced system laid out like X (independent scatter)
    for pi in range(P):
        Es[6 * pi:6 * pi + 6, np.searchsorted(kx, kf0 + pi) * HW:][:, :HW] += E[pi]
    for e in range(len(p["ii"])):
        j = int(p["jj"][e]) - kf0
        if 0 <= j < P:
            Es[6 * j:6 * j + 6, kk[e] * HW:(kk[e] + 1) * HW] += E[P + e]
    assert np.abs(Es - X).max() <= 1e-5 * np.abs(X).max()
    Fr = (Es.T * Q.reshape(-1, 1).astype(np.float64)) @ np.linalg.inv(L)
    assert np.abs(z_r.reshape(-1) - (Q.reshape(-1) + (Fr ** 2).sum(-1))).max() <= 1e-9 * np.abs(z_r).max()


def test_srgb_and_ingest_restatement_known_values(oracle_mod):
    """utils/utils.py:136-139 is the standard sRGB decoding: known values (IEC 61966-2-1), continuity at the knee, and the
    pose part of the ingest restatement: camera-to-world of the identity / of a pure translation"""
    v = oracle_mod.srgb_to_linear([0.0, 0.04045, 0.5, 1.0])
    assert abs(v[0]) == 0 and abs(v[1] - 0.04045 / 12.92) < 1e-15 and abs(v[2] - 0.21404114048223255) < 1e-12 and abs(v[3] - 1) < 1e-15
    assert abs(oracle_mod.srgb_to_linear(0.04045 + 1e-9) - oracle_mod.srgb_to_linear(0.04045)) < 1e-7
    pk = {"cam0_poses": np.array([[0, 0, 0, 0, 0, 0, 1.0], [0.3, -0.2, 0.1, 0, 0, 0, 1.0]]),
          "cam0_images": np.full((2, 3, 2, 2), 128, np.uint8), "cam0_idepths_up": np.full((2, 2, 2), 0.5),
          "cam0_depths_cov_up": np.full((2, 2, 2), 0.1)}
    out = oracle_mod.nerf_ingest(pk)
    assert np.allclose(out["poses"][0], np.eye(4)[:3]) and np.allclose(out["poses"][1, :, 3], [-0.3, 0.2, -0.1])
    assert np.allclose(out["depths"], 2.0) and np.allclose(out["images"][..., 3], 1.0)
    assert np.allclose(out["images"][..., :3], oracle_mod.srgb_to_linear(128 / 255.0))
