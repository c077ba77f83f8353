"""End-to-end check of the tracking hot path through the host mirror of RaftVisualFrontend:
a synthetic scene with known poses / depths, an update operator that returns the TRUE induced flow
(standing in for the ConvGRU, whose weights the reference tree lacks), and the device BA.  The
estimate must converge to the ground truth (up to the monocular gauge, which is anchored by the frame-0
prior and a sensed-depth prior on frame 0)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_tracking_converges_to_ground_truth(oracle_mod, dev):
    from nerfslam import se3
    from nerfslam.frontend import TrackingFrontend
    rng = np.random.default_rng(0)
    H, W, nkf = 96, 128, 6
    ht, wd = H // 8, W // 8
    intr = np.array([100.0, 100.0, W / 2, H / 2], np.float32)
    # ground truth: smooth depth maps, small motion
    gt_poses = np.zeros((nkf, 7), np.float32); gt_poses[:, 6] = 1
    for k in range(1, nkf):
        gt_poses[k, :3] = 0.04 * k * np.array([1.0, 0.2, 0.1]) + rng.normal(0, 0.005, 3)
        gt_poses[k, 3:] = synth.quat_exp(rng.normal(0, 0.01, 3))
    yy, xx = np.meshgrid(np.linspace(0, 1, ht), np.linspace(0, 1, wd), indexing="ij")
    gt_disp = np.stack([0.5 + 0.3 * np.sin(3 * xx + k) * np.cos(2 * yy) for k in range(nkf)]).astype(np.float32)
    gtP, gtD = torch.from_numpy(gt_poses).to(dev), torch.from_numpy(gt_disp).to(dev)

    def feature_fn(img):
        g = torch.Generator(device="cpu").manual_seed(int(img.sum().item()) % 1000)
        return torch.randn((128, ht, wd), generator=g)

    fe = None

    def true_flow_op(corr, motion, ii, jj):
        # the ConvGRU stand-in: flow correction = (true induced flow) - (current reprojection)
        E = ii.shape[0]
        true_c = torch.empty((E, ht, wd, 2), device=dev)
        from nerfslam._lib import check, lib, ptr, stream_ptr
        check(lib().ns_reproject(ptr(gtP), ptr(gtD), ptr(fe.intr8), ptr(ii), ptr(jj), ptr(true_c), None, E, ht, wd,
                                 stream_ptr()), "reproject")
        cur = fe.reproject(ii, jj)
        delta = (true_c - cur)[None]
        weight = torch.ones_like(delta)
        nk = len(np.unique(fe.graph.ii))
        return delta, weight, torch.full((nk, ht, wd), 1e-4, device=dev)

    fe = TrackingFrontend(8, H, W, intr, dev, feature_fn=feature_fn, update_op=true_flow_op)
    for k in range(nkf):
        fe.set_keyframe(k, torch.full((3, H, W), k, dtype=torch.uint8))
    fe.kf_idx = nkf - 1
    # initial state: every pose at frame 0's, constant depth; frame 0 anchored (prior + sensed depth)
    fe.prior_pose = fe.world_T_body[0].clone()
    fe.cam0_idepths[:] = 0.6
    fe.cam0_idepths_sensed[0] = gtD[0]
    fe.cam0_idepths[0] = gtD[0]
    fe.add_neighborhood_factors(0, nkf - 1, radius=3)
    assert fe.ii.shape[0] == len(fe.graph.ii) == 24 and len(fe.slots) == 24 and len(set(fe.slots.tolist())) == 24

    def err():
        c = fe.reproject(fe.ii, fe.jj)
        t = torch.empty_like(c)
        from nerfslam._lib import check, lib, ptr, stream_ptr
        check(lib().ns_reproject(ptr(gtP), ptr(gtD), ptr(fe.intr8), ptr(fe.ii), ptr(fe.jj), ptr(t), None, c.shape[0], ht,
                                 wd, stream_ptr()), "reproject")
        return (c - t).norm(dim=-1).mean().item()

    e0 = err()
    for _ in range(12):
        sol = fe.update(itrs=2)
        assert sol["info"].item() == 0
    e1 = err()
    assert e0 > 0.2 and e1 < 1e-3 * e0, (e0, e1)
    # poses: cam0_T_world vs ground truth
    dT = se3.log_wv(se3.mul(fe.cam0_T_world[:nkf].double(), se3.inv(gtP.double())))
    assert dT.abs().max().item() < 5e-3, dT
    assert (fe.cam0_idepths[:nkf] - gtD).abs().mean().item() < 5e-3
    assert torch.isfinite(fe.cam0_idepths_cov[:nkf]).all() and torch.isfinite(fe.world_T_body_cov[:nkf]).all()
    pkt = fe.get_viz_out()
    assert pkt["cam0_poses"].shape == (nkf, 7) and pkt["cam0_idepths_up"].shape == (nkf, H, W)
    # graph maintenance on the device payloads
    fe.add_proximity_factors(kf0=0, kf1=0, rad=2, nms=2, thresh=1e9)
    n = fe.ii.shape[0]
    assert fe.target.shape[0] == n == len(fe.slots) == fe.slots_dev.shape[0]
    fe.rm_factors(fe.graph.age > 5, store=True)
    assert fe.target_inactive.shape[0] == len(fe.graph.ii_inactive)


def test_cvx_upsample_kernel(oracle_mod, dev):
    """ns_cvx_upsample against the numpy restatement pinned to utils/flow_viz.py (f32 and f16 masks, pow, 1-pixel-wide maps)"""
    import ctypes as C
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(0)
    for (n, ht, wd, pw, dt) in ((3, 12, 16, 1.0, np.float32), (2, 43, 77, 1.0, np.float16), (2, 13, 22, 1.0, np.float16), (1, 60, 80, 0.5, np.float16), (1, 3, 2, 1.0, np.float16),
                               (1, 5, 1, 0.5, np.float32), (1, 1, 9, 1.0, np.float32)):
        data = rng.uniform(0.1, 2.0, (n, ht, wd)).astype(np.float32)
        mask = (rng.standard_normal((n, 576, ht, wd)) * 2).astype(dt)
        ref = oracle_mod.cvx_upsample(data, mask.astype(np.float32), pw)
        d, m = torch.from_numpy(data).to(dev), torch.from_numpy(mask).to(dev)
        out = torch.empty((n, 8 * ht, 8 * wd), device=dev)
        check(lib().ns_cvx_upsample(ptr(d), ptr(m), 1 if dt == np.float16 else 2, ptr(out), n, ht, wd, C.c_float(pw), stream_ptr()),
              "cvx_upsample")
        assert np.abs(out.cpu().numpy() - ref).max() <= 5e-6 * max(1.0, np.abs(ref).max()), (n, ht, wd)
        # a convex combination never leaves the range of the data
        assert out.min().item() >= data.min() - 1e-5 and (pw != 1.0 or out.max().item() <= data.max() + 1e-5)


def test_cvx_upsample_keyframes_kernel(oracle_mod, dev):
    """the frontend's one-launch form: two maps, one mask, gathered from / scattered into the keyframe buffers by index"""
    import ctypes as C
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(1)
    kx = np.array([5, 0, 3], np.int64)
    for (nbuf, ht, wd, dt) in ((7, 11, 19, np.float16), (7, 11, 19, np.float32), (6, 9, 20, np.float16), (6, 9, 20, np.float32)):
        a = rng.uniform(0.1, 2.0, (nbuf, ht, wd)).astype(np.float32)
        b = rng.uniform(0.0, 9.0, (nbuf, ht, wd)).astype(np.float32)
        mask = (rng.standard_normal((3, 576, ht, wd)) * 2).astype(dt)
        ta, tb, tm, tk = (torch.from_numpy(v).to(dev) for v in (a, b, mask, kx))
        oa = torch.full((nbuf, 8 * ht, 8 * wd), -7.0, device=dev)
        ob = torch.full((nbuf, 8 * ht, 8 * wd), -9.0, device=dev)
        check(lib().ns_cvx_upsample_keyframes(ptr(ta), ptr(tb), ptr(tk), ptr(tm), 1 if dt == np.float16 else 2, ptr(oa), ptr(ob),
                                              3, ht, wd, C.c_float(1.0), stream_ptr()), "cvx_upsample_keyframes")
        ra = oracle_mod.cvx_upsample(a[kx], mask.astype(np.float32))
        rb = oracle_mod.cvx_upsample(b[kx], mask.astype(np.float32))
        assert np.abs(oa[tk].cpu().numpy() - ra).max() <= 5e-6 * np.abs(ra).max()
        assert np.abs(ob[tk].cpu().numpy() - rb).max() <= 5e-6 * np.abs(rb).max()
        rest = [k for k in range(nbuf) if k not in kx]
        assert (oa[rest] == -7.0).all() and (ob[rest] == -9.0).all()       # other keyframes untouched
        # single-map form of the same entry point == ns_cvx_upsample bit for bit
        o1 = torch.zeros((nbuf, 8 * ht, 8 * wd), device=dev)
        check(lib().ns_cvx_upsample_keyframes(ptr(ta), None, ptr(tk), ptr(tm), 1 if dt == np.float16 else 2, ptr(o1), None,
                                              3, ht, wd, C.c_float(1.0), stream_ptr()), "cvx_upsample_keyframes")
        o2 = torch.empty((3, 8 * ht, 8 * wd), device=dev)
        check(lib().ns_cvx_upsample(ptr(ta[tk].contiguous()), ptr(tm), 1 if dt == np.float16 else 2, ptr(o2), 3, ht, wd,
                                    C.c_float(1.0), stream_ptr()), "cvx_upsample")
        assert torch.equal(o1[tk], o2) and torch.equal(o1[tk], oa[tk])
        if dt == np.float16:      # channels-last mask variant == plane-major one (same arithmetic, other addressing)
            oc, od = torch.zeros_like(oa), torch.zeros_like(ob)
            check(lib().ns_cvx_upsample_keyframes_nhwc(ptr(ta), ptr(tb), ptr(tk), ptr(tm.permute(0, 2, 3, 1).contiguous()), ptr(oc),
                                                       ptr(od), 3, ht, wd, C.c_float(1.0), stream_ptr()), "cvx_upsample_keyframes_nhwc")
            assert torch.equal(oc[tk], oa[tk]) and torch.equal(od[tk], ob[tk])


def test_motion_features_kernel(dev):
    """one launch == sub, sub, cat, permute, clamp of visual_frontend.py:379-386"""
    from nerfslam.frontend import TrackingFrontend
    fe = TrackingFrontend(4, 96, 128, np.array([100.0, 100.0, 64.0, 48.0], np.float32), dev)
    g = torch.Generator().manual_seed(0)
    c1 = (fe.coords0[None] + torch.randn((5, 12, 16, 2), generator=g).to(dev) * 50).contiguous()
    tg = (c1 + torch.randn((5, 12, 16, 2), generator=g).to(dev) * 50).contiguous()
    ref = torch.cat([c1 - fe.coords0, tg - c1], -1).permute(0, 3, 1, 2).clamp(-64.0, 64.0)
    assert torch.equal(fe.motion_features(c1, tg), ref.contiguous())
