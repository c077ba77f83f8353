"""GPU parity of the mapping-path kernels against oracle/ngp_oracle.c (parity UNPINNED by the reference:
instant-ngp is an un-vendored dependency, see the oracle header)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def _cfgs(oracle_mod):
    return [oracle_mod.ngp_cfg(), oracle_mod.ngp_cfg(n_levels=8, log2_hashmap=14, base_res=4, per_level_scale=2.0)]


@pytest.mark.parametrize("which", [0, 1])
def test_hash_encode_forward_backward(oracle_mod, dev, which):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    cfg = _cfgs(oracle_mod)[which]
    _, res, off = oracle_mod.ngp_grid_layout(cfg)
    n_par = int(off[-1]) * 2
    rng = np.random.default_rng(which)
    params = rng.uniform(-0.5, 0.5, n_par).astype(np.float16)
    N = 3001
    pos = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    pos[0] = [0, 0, 0]
    pos[1] = [1, 1, 1]
    pos[2] = [0.5, 1.0, 0.0]
    L = cfg.n_levels
    args = (L, 2, cfg.log2_hashmap, cfg.base_res, C.c_float(cfg.per_level_scale))
    # device layout agrees with the oracle's
    o2 = (C.c_uint32 * (L + 1))()
    r2 = (C.c_int * L)()
    check(lib().ns_ngp_grid_layout(*args, None, r2, o2), "layout")
    assert list(o2) == list(off) and list(r2) == list(res)
    out = torch.empty((N, 2 * L), dtype=torch.float16, device=dev)
    d_pos, d_par = T(pos, dev), T(params, dev)  # keep alive: ptr() of a temporary would dangle
    check(lib().ns_ngp_encode_forward(*args, ptr(d_pos), ptr(d_par), ptr(out), 0, C.c_long(N), stream_ptr()), "fwd")
    outT = torch.empty((2 * L, N), dtype=torch.float16, device=dev)
    check(lib().ns_ngp_encode_forward(*args, ptr(d_pos), ptr(d_par), ptr(outT), 1, C.c_long(N), stream_ptr()), "fwd")
    assert torch.equal(outT.t().contiguous(), out)   # unit-major variant: same values, transposed
    ref = oracle_mod.ngp_encode_fwd(cfg, pos, params)
    got = out.cpu().numpy()
    # f32 trilinear blend of 8 f16 values, one rounding: bit-exact up to FMA order -> at most one f16 ulp (2^-10 relative)
    assert np.abs(got.astype(np.float32) - ref.astype(np.float32)).max() <= 2.0 ** -10 * np.abs(ref).max() + 1e-7
    assert (got.view(np.uint16) == ref.view(np.uint16)).mean() > 0.995
    dL = (rng.standard_normal((N, 2 * L)) * 1e-2).astype(np.float16)
    dL[rng.uniform(size=N) < 0.3] = 0
    grad = torch.zeros(n_par, dtype=torch.float32, device=dev)
    d_dL = T(dL, dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dL), 0, ptr(grad), None, C.c_size_t(0), C.c_float(0), C.c_long(N), stream_ptr()), "bwd")
    # unit-major gradient rows (the trainer's layout), workspace argument accepted (unused by the owner-computes kernel)
    ws = torch.zeros(max(lib().ns_ngp_encode_backward_workspace_bytes(*args, C.c_long(N)) // 4, 1), device=dev)
    grad_ws = torch.zeros_like(grad)
    d_dLT = d_dL.t().contiguous()
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(grad_ws), ptr(ws), C.c_size_t(ws.numel() * ws.element_size()), C.c_float(0), C.c_long(N), stream_ptr()), "bwd")
    assert not ws.any()
    assert (grad_ws - grad).abs().max().item() <= 1e-5 * grad.abs().max().item()
    # += semantics: a second call on top of the first doubles the gradient
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(grad_ws), ptr(ws), C.c_size_t(ws.numel() * ws.element_size()), C.c_float(0), C.c_long(N), stream_ptr()), "bwd")
    assert (grad_ws - 2 * grad).abs().max().item() <= 2e-5 * grad.abs().max().item()
    # packed fixed-point accumulation (Q18 pairs in one 64-bit word): order-independent -> two runs agree bit for bit
    S = 262144.0
    packed = []
    for _ in range(2):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(gq), ptr(ws), C.c_size_t(ws.numel() * ws.element_size()), C.c_float(S), C.c_long(N),
                                           stream_ptr()), "bwd")
        packed.append(gq.cpu().numpy())
    w = packed[0]
    lo = (w & 0xffffffff).astype(np.uint32).view(np.int32).astype(np.int64)
    hi = (w - lo) >> 32
    gq = np.stack([lo, hi], 1).reshape(-1).astype(np.float64) / S
    gmax = grad.abs().max().item()
    # each contribution is rounded to 2^-18: error <= (#contributions per entry) * 2^-19
    assert np.abs(gq - grad.cpu().numpy()).max() <= 64 * 2.0 ** -19 + 1e-5 * gmax
    assert np.array_equal(packed[0], packed[1])   # integer accumulation on every level: order-independent, bit-reproducible
    gref = oracle_mod.ngp_encode_bwd(cfg, pos, dL, n_par)
    # f32 atomics in arbitrary order: 1e-5 of max
    assert np.abs(grad.cpu().numpy() - gref).max() <= 1e-5 * np.abs(gref).max()


def _weights(rng):
    from oracle import MLP_SHAPES
    return [(rng.uniform(-1, 1, s) * np.sqrt(6.0 / sum(s))).astype(np.float16) for s in MLP_SHAPES]


def test_mlp_forward_backward(oracle_mod, dev):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(2)
    N = 1000  # multiple of 8, not of 256
    Ws = _weights(rng)
    feat = (rng.standard_normal((N, 32)) * 0.5).astype(np.float16)
    dirs = rng.standard_normal((N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    act = oracle_mod.ngp_mlp_fwd(Ws, feat, dirs)
    Wd = T(np.concatenate([w.reshape(-1) for w in Ws]), dev)
    out = torch.empty((N, 4), dtype=torch.float16, device=dev)
    d_featT, d_dirs = T(np.ascontiguousarray(feat.T), dev), T(dirs, dev)
    bufs = [d_featT] + [torch.empty((u, N), dtype=torch.float16, device=dev) for u in (64, 32, 64, 64)]
    check(lib().ns_ngp_mlp_forward(ptr(Wd), ptr(d_featT), ptr(d_dirs), ptr(out), *[ptr(b) for b in bufs[1:]],
                                   C.c_long(N), stream_ptr()), "mlp fwd")
    out_inf = torch.empty_like(out)   # inference mode (no activation buffers): same outputs
    nul = C.c_void_p(0)
    check(lib().ns_ngp_mlp_forward(ptr(Wd), ptr(d_featT), ptr(d_dirs), ptr(out_inf), nul, nul, nul, nul, C.c_long(N),
                                   stream_ptr()), "mlp fwd")
    assert torch.equal(out_inf, out)
    o = out.cpu().numpy().astype(np.float32)
    ref = np.concatenate([act["rgb"][:, :3], act["dens"][:, :1]], 1).astype(np.float32)
    # same f16 storage / f32 accumulation as the oracle, different accumulation order: a few half-ulps of the
    # layer scale
    assert np.abs(o - ref).max() <= 4e-3 * np.abs(ref).max()
    for b, k in zip(bufs[1:], ("h1", "cin", "h3", "h4")):
        r = act[k].astype(np.float32).T
        assert np.abs(b.cpu().numpy().astype(np.float32) - r).max() <= 4e-3 * np.abs(r).max(), k
    # backward
    dout = (rng.standard_normal((N, 4)) * 1e-2).astype(np.float16)
    dLdrgb = np.zeros((N, 16), np.float16)
    dLdrgb[:, :3] = dout[:, :3]
    dLddens = np.zeros((N, 16), np.float16)
    dLddens[:, 0] = dout[:, 3]
    dfeat_ref, dW_ref = oracle_mod.ngp_mlp_bwd(Ws, feat, act, dLdrgb, dLddens)
    dfeat = torch.empty((32, N), dtype=torch.float16, device=dev)
    dbufs = [torch.empty((u, N), dtype=torch.float16, device=dev) for u in (16, 64, 64, 16, 64)]
    ks = 7
    partial = torch.empty((ks, 10240), dtype=torch.float32, device=dev)
    gw = torch.zeros(10240, dtype=torch.float32, device=dev)
    d_dout = T(dout, dev)
    check(lib().ns_ngp_mlp_backward(ptr(Wd), ptr(d_dout), *[ptr(b) for b in bufs], ptr(dfeat),
                                    *[ptr(b) for b in dbufs], ptr(partial), ks, ptr(gw), C.c_long(N), stream_ptr()),
          "mlp bwd")
    d = dfeat.t().cpu().numpy().astype(np.float32)
    assert np.abs(d - dfeat_ref.astype(np.float32)).max() <= 1e-2 * np.abs(dfeat_ref.astype(np.float32)).max()
    g = gw.cpu().numpy()
    off = 0
    for w, r in zip(Ws, dW_ref):
        gg = g[off:off + w.size].reshape(w.shape)
        assert np.abs(gg - r).max() <= 1e-2 * np.abs(r).max(), w.shape
        off += w.size
    # ReLU bit masks (round 3): the forward pass writes one bit per hidden unit, the activation backward takes the ReLU
    # derivative from them instead of re-reading the activations -- same outputs bit for bit, also for a device sample count
    masks = torch.zeros(6 * N, dtype=torch.int32, device=dev)
    out_m = torch.empty_like(out)
    bufs_m = [torch.empty_like(b) for b in bufs[1:]]
    check(lib().ns_ngp_mlp_forward_m_n(ptr(Wd), ptr(d_featT), ptr(d_dirs), ptr(out_m), *[ptr(b) for b in bufs_m], ptr(masks),
                                       C.c_long(N), None, stream_ptr()), "mlp fwd masks")
    assert torch.equal(out_m, out) and all(torch.equal(a, b) for a, b in zip(bufs_m, bufs[1:]))
    # (with a device-side count, whatever dL/dout holds beyond it is never used: the loss-gradient buffer is not cleared per step)
    poisoned = d_dout.clone()
    poisoned[500:] = float("nan")
    for n_dev in (None, torch.tensor([500], dtype=torch.int32, device=dev)):
        dd_in = d_dout if n_dev is None else poisoned
        ref_d = [torch.zeros_like(b) for b in [dfeat] + dbufs]
        got_d = [torch.zeros_like(b) for b in [dfeat] + dbufs]
        check(lib().ns_ngp_mlp_dgrad_n(ptr(Wd), ptr(dd_in), ptr(bufs[1]), ptr(bufs[3]), ptr(bufs[4]), *[ptr(b) for b in ref_d],
                                       C.c_long(N), ptr(n_dev), stream_ptr()), "dgrad")
        check(lib().ns_ngp_mlp_dgrad_m_n(ptr(Wd), ptr(dd_in), ptr(masks), *[ptr(b) for b in got_d], C.c_long(N), ptr(n_dev),
                                         stream_ptr()), "dgrad masks")
        for a, b in zip(ref_d, got_d):
            assert torch.equal(a, b)
        assert n_dev is not None or torch.equal(ref_d[0], dfeat)
    # fused backward pass (round 3): forward recomputed on chip, weight gradients contracted on chip -- dL/dfeature bit for bit
    # (same MFMA sequence), weight gradients equal up to summation order; also with a device sample count and several workgroup
    # counts (the partial slabs are summed in slab order)
    for n_dev, n_valid in ((None, N), (torch.tensor([500], dtype=torch.int32, device=dev), 504)):
        dd_in = d_dout if n_dev is None else poisoned
        ref_feat = torch.zeros_like(dfeat)
        ref_db = [torch.zeros_like(b) for b in dbufs]
        gw_ref = torch.zeros(10240, dtype=torch.float32, device=dev)
        part = torch.empty((7, 10240), dtype=torch.float32, device=dev)
        check(lib().ns_ngp_mlp_backward_n(ptr(Wd), ptr(dd_in), *[ptr(b) for b in bufs], ptr(ref_feat), *[ptr(b) for b in ref_db],
                                          ptr(part), 7, ptr(gw_ref), C.c_long(N), ptr(n_dev), stream_ptr()), "mlp bwd")
        for wgs in (1, 3, 64):
            got_feat = torch.zeros_like(dfeat)
            gw_got = torch.zeros(10240, dtype=torch.float32, device=dev)
            part_f = torch.full((wgs, 10240), float("nan"), dtype=torch.float32, device=dev)
            check(lib().ns_ngp_mlp_backward_fused_n(ptr(Wd), ptr(d_featT), ptr(d_dirs), ptr(dd_in), ptr(got_feat), ptr(part_f), wgs,
                                                    ptr(gw_got), C.c_long(N), ptr(n_dev), stream_ptr()), "mlp bwd fused")
            assert torch.equal(got_feat[:, :n_valid], ref_feat[:, :n_valid]), wgs
            assert torch.isfinite(gw_got).all()
            err = (gw_got - gw_ref).abs().max().item()
            assert err <= 2e-3 * gw_ref.abs().max().item(), (wgs, err, gw_ref.abs().max().item())
        # split form (the trainer's default): activation gradients without their five stores + weight gradients recomputed on
        # chip from a packed fragment table (46-KB-LDS kernel for a side stream); masks-only forward (no activation buffers)
        masks2 = torch.zeros(6 * N, dtype=torch.int32, device=dev)
        out2 = torch.empty_like(out)
        check(lib().ns_ngp_mlp_forward_m_n(ptr(Wd), ptr(d_featT), ptr(d_dirs), ptr(out2), None, None, None, None, ptr(masks2),
                                           C.c_long(N), ptr(n_dev), stream_ptr()), "mlp fwd masks only")
        assert torch.equal(out2[:n_valid], out[:n_valid])
        lean = torch.zeros_like(dfeat)
        check(lib().ns_ngp_mlp_dgrad_m_n(ptr(Wd), ptr(dd_in), ptr(masks2), ptr(lean), None, None, None, None, None, C.c_long(N),
                                         ptr(n_dev), stream_ptr()), "lean dgrad")
        assert torch.equal(lean[:, :n_valid], ref_feat[:, :n_valid]) and torch.isfinite(lean[:, :n_valid]).all()
        assert n_dev is None or (lean[:, 500:504] == 0).all()
        frags = torch.zeros(int(lib().ns_ngp_mlp_fragment_table_bytes()) // 2, dtype=torch.float16, device=dev)
        check(lib().ns_ngp_mlp_pack_fragments(ptr(Wd), ptr(frags), stream_ptr()), "pack")
        # ... and the same two kernels with their weights taken from that table (what the trainer launches): same bits
        masks3 = torch.zeros(6 * N, dtype=torch.int32, device=dev)
        out3 = torch.empty_like(out)
        check(lib().ns_ngp_mlp_forward_f_n(ptr(frags), ptr(d_featT), ptr(d_dirs), ptr(out3), ptr(masks3), C.c_long(N), ptr(n_dev),
                                           stream_ptr()), "mlp fwd frags")
        assert torch.equal(out3[:n_valid], out[:n_valid]) and torch.equal(masks3.view(6, N)[:, :n_valid], masks2.view(6, N)[:, :n_valid])
        lean3 = torch.zeros_like(dfeat)
        check(lib().ns_ngp_mlp_dgrad_f_n(ptr(frags), ptr(dd_in), ptr(masks3), ptr(lean3), C.c_long(N), ptr(n_dev), stream_ptr()),
              "lean dgrad frags")
        assert torch.equal(lean3[:, :n_valid], ref_feat[:, :n_valid])
        for wgs in (1, 5, 64):
            gw_got = torch.zeros(10240, dtype=torch.float32, device=dev)
            part_f = torch.full((wgs, 10240), float("nan"), dtype=torch.float32, device=dev)
            check(lib().ns_ngp_mlp_wgrad_recompute_n(ptr(frags), ptr(d_featT), ptr(d_dirs), ptr(dd_in), ptr(part_f), wgs, ptr(gw_got),
                                                     C.c_long(N), ptr(n_dev), stream_ptr()), "wgrad recompute")
            assert torch.isfinite(gw_got).all()
            err = (gw_got - gw_ref).abs().max().item()
            assert err <= 2e-3 * gw_ref.abs().max().item(), ("recompute", wgs, err, gw_ref.abs().max().item())


def test_composite_loss_and_adam(oracle_mod, dev):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(3)
    R = 300
    ray_n = rng.integers(0, 40, R).astype(np.int32)
    ray_n[[10, 20, 30, 40]] = [100, 256, 300, 700]     # several 64-sample rounds; beyond the kernel's register-cached rounds
    ray_start = np.concatenate([[0], np.cumsum(ray_n)[:-1]]).astype(np.int32)
    ray_n[[5, 77, 200]] = -1          # rays refused by the marcher: no samples, no loss
    S = int(np.maximum(ray_n, 0).sum()) + 3 * 40
    net = np.zeros((S, 4), np.float16)
    net[:, :3] = rng.standard_normal((S, 3))
    net[:, 3] = rng.uniform(-2, 3, S)
    dt = rng.uniform(0.002, 0.02, S).astype(np.float32)
    tm = rng.uniform(0.1, 4, S).astype(np.float32)
    for s0, n in zip(ray_start, ray_n):
        tm[s0:s0 + max(n, 0)].sort()
    gt_rgb = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    gt_d = rng.uniform(0.5, 3, R).astype(np.float32)
    gt_d[::3] = -1
    gt_c = rng.uniform(0.01, 1, R).astype(np.float32)
    rgb16 = np.zeros((S, 16), np.float16); rgb16[:, :3] = net[:, :3]
    den16 = np.zeros((S, 16), np.float16); den16[:, 0] = net[:, 3]
    orgb, odep, loss, dr, dd = oracle_mod.ngp_composite_loss(rgb16, den16, dt, tm, ray_start, ray_n, gt_rgb, gt_d, gt_c,
                                                            1.0, 128.0)
    out_rgb = torch.empty((R, 3), device=dev); out_d = torch.empty(R, device=dev)
    l = torch.zeros(1, device=dev); dout = torch.zeros((S, 4), dtype=torch.float16, device=dev)
    keep = [T(x, dev) for x in (net, dt, tm, ray_start, ray_n, gt_rgb, gt_d, gt_c)]
    check(lib().ns_ngp_composite(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]),
                                 ptr(keep[4]), R, ptr(keep[5]), ptr(keep[6]), ptr(keep[7]),
                                 C.c_float(1.0), C.c_float(128.0), ptr(out_rgb), ptr(out_d), ptr(l), ptr(dout),
                                 stream_ptr()), "composite")
    np.testing.assert_allclose(out_rgb.cpu().numpy(), orgb, atol=2e-5)
    np.testing.assert_allclose(out_d.cpu().numpy(), odep, rtol=2e-5, atol=2e-5)
    assert abs(l.item() / R - loss) <= 1e-4 * abs(loss)
    ref = np.concatenate([dr[:, :3], dd[:, :1]], 1).astype(np.float32)
    got = dout.cpu().numpy().astype(np.float32)
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max()  # fast exp + f16 output
    # per-ray loss form (the trainer's): same outputs, losses per ray instead of ~R atomics on one address; with a device-side
    # ray count the rays beyond it get a zero and nothing else
    out_rgb2 = torch.empty((R, 3), device=dev); out_d2 = torch.empty(R, device=dev)
    rl = torch.full((R,), float("nan"), device=dev); dout2 = torch.zeros_like(dout)
    check(lib().ns_ngp_composite_rays(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), R, ptr(keep[5]),
                                      ptr(keep[6]), ptr(keep[7]), C.c_float(1.0), C.c_float(128.0), ptr(out_rgb2), ptr(out_d2),
                                      None, ptr(rl), ptr(dout2), None, stream_ptr()), "composite rays")
    assert torch.equal(out_rgb2, out_rgb) and torch.equal(out_d2, out_d) and torch.equal(dout2, dout)
    assert abs(rl.double().sum().item() / R - loss) <= 1e-5 * abs(loss) and (rl[[5, 77, 200]] == 0).all()
    ctl = torch.tensor([0, 250, 0, 1, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    rl2 = torch.full((R,), float("nan"), device=dev)
    check(lib().ns_ngp_composite_rays(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), R, ptr(keep[5]),
                                      ptr(keep[6]), ptr(keep[7]), C.c_float(1.0), C.c_float(128.0), ptr(out_rgb2), ptr(out_d2),
                                      None, ptr(rl2), ptr(dout2), ptr(ctl), stream_ptr()), "composite rays ctl")
    assert torch.equal(rl2[:250], rl[:250]) and (rl2[250:] == 0).all()
    # Adam
    n = 5000
    m = rng.standard_normal(n).astype(np.float32); g = (rng.standard_normal(n) * 128).astype(np.float32)
    g[::4] = 0
    m1 = (rng.standard_normal(n) * 0.1).astype(np.float32); m2 = rng.uniform(0, 0.1, n).astype(np.float32)
    rm, rh, r1, r2 = oracle_mod.ngp_adam(m, g, m1, m2, step=3, lr=1e-2, l2=0.0, grad_scale=128.0)
    dm, dg, d1, d2 = T(m, dev), T(g, dev), T(m1, dev), T(m2, dev)
    hp = torch.empty(n, dtype=torch.float16, device=dev)
    check(lib().ns_ngp_adam(ptr(dm), ptr(hp), ptr(dg), ptr(d1), ptr(d2), C.c_long(n), 3, C.c_float(1e-2), C.c_float(0.9),
                            C.c_float(0.99), C.c_float(1e-15), C.c_float(0.0), C.c_float(128.0), C.c_float(0.0), stream_ptr()), "adam")
    # packed fixed-point gradient input: same update as the float gradient rounded to 2^-18
    S = 262144.0
    gr = np.round(g.astype(np.float64) * S).astype(np.int64)
    word = torch.from_numpy(gr[0::2] + (gr[1::2] << 32)).to(dev)
    rm2, rh2, _, _ = oracle_mod.ngp_adam(m, (gr / S).astype(np.float32), m1, m2, step=3, lr=1e-2, l2=0.0, grad_scale=128.0)
    dm2, d12, d22 = T(m, dev), T(m1, dev), T(m2, dev)
    check(lib().ns_ngp_adam(ptr(dm2), ptr(hp), ptr(word), ptr(d12), ptr(d22), C.c_long(n), 3, C.c_float(1e-2), C.c_float(0.9),
                            C.c_float(0.99), C.c_float(1e-15), C.c_float(0.0), C.c_float(128.0), C.c_float(S), stream_ptr()),
          "adam")
    np.testing.assert_allclose(dm2.cpu().numpy(), rm2, rtol=2e-6, atol=1e-7)
    assert not word.any()
    np.testing.assert_allclose(dm.cpu().numpy(), rm, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(d1.cpu().numpy(), r1, rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(d2.cpu().numpy(), r2, rtol=2e-6, atol=1e-8)
    assert dg.abs().max().item() == 0 and (hp.cpu().numpy() == rh).mean() > 0.999


def test_ray_marching(oracle_mod, dev):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(4)
    G, nc = 32, 3
    bits = rng.integers(0, 256, nc * G ** 3 // 8).astype(np.uint8)
    bits[rng.uniform(size=bits.shape) < 0.5] = 0
    R = 200
    o = rng.uniform(0.2, 0.8, (R, 3)).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tr = np.stack([np.full(R, 0.05), rng.uniform(0.5, 3.0, R)], 1).astype(np.float32)
    cone, mn, mx = 1 / 256.0, np.sqrt(3) / 1024, np.sqrt(3) / 1024 * 32
    S = 1 << 17
    cnt = torch.zeros(3, dtype=torch.int32, device=dev)
    rs = torch.empty(R, dtype=torch.int32, device=dev); rn = torch.empty(R, dtype=torch.int32, device=dev)
    pos = torch.empty((S, 3), device=dev); dirs = torch.empty((S, 3), device=dev)
    dt = torch.empty(S, device=dev); tm = torch.empty(S, device=dev)
    keep = [T(x, dev) for x in (bits, o, d, tr)] + [torch.empty_like(pos), torch.empty_like(dirs), torch.empty_like(dt), torch.empty_like(tm)]
    check(lib().ns_ngp_march(ptr(keep[0]), G, nc, ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), R,
                             C.c_float(cone), C.c_float(mn), C.c_float(mx), C.c_float(0.0), C.c_float(1.0), 1024, C.c_long(S), ptr(cnt), ptr(rs),
                             ptr(rn), ptr(pos), ptr(dirs), ptr(dt), ptr(tm), stream_ptr()), "march")
    rs, rn, pos, dt, tm = (x.cpu().numpy() for x in (rs, rn, pos, dt, tm))
    total = 0
    for r in range(R):
        p, dts, ts = oracle_mod.ngp_march_ray(bits, G, nc, o[r], d[r], cone, mn, mx, tr[r, 0], tr[r, 1], 1024)
        # the cell test is a float comparison on positions computed with FMA contraction on the device:
        # a sample exactly on a cell face may flip; allow one per ray
        assert abs(int(rn[r]) - len(ts)) <= 1, r
        if rn[r] == len(ts) and len(ts):
            np.testing.assert_allclose(tm[rs[r]:rs[r] + rn[r]], ts, rtol=1e-5)
            np.testing.assert_allclose(pos[rs[r]:rs[r] + rn[r]], p, atol=1e-5)
        total += int(rn[r])
    assert cnt.tolist() == [total, int((rn > 0).sum()), total]
    # batch smaller than the demand: refused rays get nothing, the accepted ranges tile [0, counter[2]) without holes
    S2 = total // 3
    cnt2 = torch.zeros(3, dtype=torch.int32, device=dev)
    rs2 = torch.empty(R, dtype=torch.int32, device=dev); rn2 = torch.empty(R, dtype=torch.int32, device=dev)
    check(lib().ns_ngp_march(ptr(keep[0]), G, nc, ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), R,
                             C.c_float(cone), C.c_float(mn), C.c_float(mx), C.c_float(-1.5), C.c_float(0.25), 1024, C.c_long(S2), ptr(cnt2), ptr(rs2),
                             ptr(rn2), ptr(keep[4]), ptr(keep[5]), ptr(keep[6]), ptr(keep[7]), stream_ptr()), "march")
    rs2, rn2, c2 = rs2.cpu().numpy(), rn2.cpu().numpy(), cnt2.tolist()
    acc = rn2 > 0
    assert (rn2 == -1).any() and (rn2 >= -1).all()
    assert c2[0] == total and c2[1] == acc.sum() and 0 < c2[2] <= S2
    order = np.argsort(rs2[acc])
    st, ln = rs2[acc][order], rn2[acc][order]
    assert st[0] == 0 and (st[1:] == st[:-1] + ln[:-1]).all() and st[-1] + ln[-1] == c2[2]
    pos2 = keep[4].cpu().numpy()     # this call asked for positions mapped by (p + 1.5) * 0.25
    checked = 0
    for r in np.nonzero(acc)[0][:20]:
        if rn2[r] == rn[r]:
            np.testing.assert_allclose(pos2[rs2[r]:rs2[r] + rn2[r]], (pos[rs[r]:rs[r] + rn[r]] + 1.5) * 0.25, atol=1e-6)
            checked += 1
    assert checked > 0


def test_sample_rays(oracle_mod, dev):
    """training-ray sampler: same picks (integer hash) and rays / supervision as the C restatement"""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(5)
    n, H, W, R = 5, 37, 53, 3000
    images = rng.uniform(0, 1, (n, H, W, 4)).astype(np.float32)
    depths = rng.uniform(-0.5, 4, (n, H, W)).astype(np.float32)
    covs = rng.uniform(0, 1, (n, H, W)).astype(np.float32); covs[covs < 0.1] = 0
    c2w = np.zeros((n, 3, 4), np.float32)
    for k in range(n):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[k, :, :3], c2w[k, :, 3] = q, rng.uniform(-0.5, 1.5, 3)
    intr = (60.0, 55.0, 26.0, 18.0)
    ref = oracle_mod.ngp_sample_rays(images, depths, covs, c2w, intr, -1.5, 2.5, 0.05, 12345, R)
    keep = [T(x, dev) for x in (images, depths, covs, c2w)]
    f = dict(dtype=torch.float32, device=dev)
    out = [torch.empty((R, 3), **f), torch.empty((R, 3), **f), torch.empty((R, 2), **f), torch.empty((R, 3), **f),
           torch.empty(R, **f), torch.empty(R, **f)]
    img_idx = torch.empty(R, dtype=torch.int32, device=dev)
    check(lib().ns_ngp_sample_rays(*[ptr(k) for k in keep], n, H, W, *[C.c_float(v) for v in intr], C.c_float(-1.5),
                                   C.c_float(2.5), C.c_float(0.05), C.c_uint32(12345), R, *[ptr(t) for t in out],
                                   ptr(img_idx), stream_ptr()), "sample_rays")
    assert np.array_equal(img_idx.cpu().numpy(), ref["picks"][:, 0])
    for t, k in zip(out, ("rays_o", "rays_d", "t_range", "gt_rgb", "gt_depth", "gt_cov")):
        tol = dict(rtol=0, atol=0) if k.startswith("gt") or k == "rays_o" else dict(rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(t.cpu().numpy(), ref[k], err_msg=k, **tol)
    picks = ref["picks"]
    assert len(np.unique(picks[:, 0])) == n and picks[:, 1].max() == W - 1 and picks[:, 2].max() == H - 1


def test_training_converges_on_a_synthetic_scene(dev):
    """End-to-end training steps through every kernel: a coloured sphere in front of 6 cameras; loss must drop."""
    from nerfslam.ngp import NgpConfig, NgpNerf
    cfg = NgpConfig(n_rays=2048, max_samples=1 << 17)
    net = NgpNerf(cfg, dev, seed=0)
    H, W, f = 48, 64, 60.0
    imgs, deps, covs, poses = [], [], [], []
    centre = np.array([0.5, 0.5, 0.5])
    for k in range(6):
        ang = 2 * np.pi * k / 6
        eye = centre + 0.9 * np.array([np.cos(ang), 0.2, np.sin(ang)])
        fwd = (centre - eye) / np.linalg.norm(centre - eye)
        right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w = np.stack([right, -up, fwd, eye], 1)  # camera looks along +z, y down
        vv, uu = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        d = np.stack([(uu + 0.5 - W / 2) / f, (vv + 0.5 - H / 2) / f, np.ones_like(uu, float)], -1) @ c2w[:, :3].T
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        oc = eye - centre
        b = (d * oc).sum(-1); cc = (oc * oc).sum() - 0.25 ** 2
        disc = b * b - cc
        hit = disc > 0
        t = np.where(hit, -b - np.sqrt(np.maximum(disc, 0)), -1.0)
        pts = eye + t[..., None] * d
        col = np.where(hit[..., None], 0.5 + 0.5 * (pts - centre) / 0.25, 0.0)
        imgs.append(np.concatenate([col, hit[..., None].astype(float)], -1)); deps.append(t); covs.append(np.full((H, W), 0.05))
        poses.append(c2w)
    net.set_images(torch.tensor(np.array(imgs)), torch.tensor(np.array(deps)), torch.tensor(np.array(covs)),
                   torch.tensor(np.array(poses)), (f, f, W / 2, H / 2))
    losses = [float(net.train_step()) for _ in range(200)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    rgb, dep = net.render(torch.tensor(np.array(poses[0])), H, W)
    assert torch.isfinite(rgb).all() and rgb.shape == (H, W, 3)
    # the rays of a step are marched during the previous one; an occupancy update in between would invalidate them (training on
    # the stale march made the result a coin toss).  Round 6 (cfg.refresh_in_step): the refresh RIDES on the step before the
    # update, ahead of the next step's rays on the same stream -- they are marched on the refreshed grid and stay valid ...
    while net.step % cfg.grid_update_every != cfg.grid_update_every - 1:
        net.train_step(return_loss=False)
    assert net._primed and cfg.refresh_in_step
    bits0 = net.bits.clone()
    net.train_step(return_loss=False)                       # this step carries the occupancy update
    assert net.step % cfg.grid_update_every == 0 and net._primed and net._refresh_seen
    assert not torch.equal(bits0, net.bits)                 # (the grid did change)
    # ... and with the refresh AFTER the step (rounds 2-5's form) the next step has to march again
    old = NgpNerf(NgpConfig(n_rays=2048, max_samples=1 << 17, refresh_in_step=False), dev, seed=0)
    old.set_images(torch.tensor(np.array(imgs)), torch.tensor(np.array(deps)), torch.tensor(np.array(covs)),
                   torch.tensor(np.array(poses)), (f, f, W / 2, H / 2))
    lo = [float(old.train_step()) for _ in range(cfg.grid_update_every)]
    assert old.step % cfg.grid_update_every == 0 and not old._primed and np.isfinite(lo).all()
    old.bits.zero_()                                        # an (artificial) update that empties the grid ...
    old.train_step(return_loss=False)
    torch.cuda.synchronize()
    assert old.last_samples == 0 and old._primed            # ... is what the next step marches on: no samples at all


def test_pyngp_surface_as_nerf_fusion_drives_it(dev):
    """The call sequence of fusion/nerf_fusion.py:57-101 (init), :285-289 (send_data), :299 (frame),
    :388-424 (render) against the shim."""
    import pyngp as ngp
    tb = ngp.Testbed(ngp.TestbedMode.Nerf, 0)
    tb.create_empty_nerf_dataset(8, 1.0, np.array([np.inf] * 3), 4, ngp.BoundingBox(np.array([-np.inf] * 3), np.array([np.inf] * 3)))
    tb.nerf.training.n_images_for_training = 0
    tb.reload_network_from_file("base.json")
    tb.shall_train = True
    tb.dynamic_res = True
    tb.dynamic_res_target_fps = 15
    tb.camera_smoothing = True
    tb.nerf.training.optimize_extrinsics = True
    tb.nerf.training.depth_supervision_lambda = 1.0
    tb.nerf.training.depth_loss_type = ngp.LossType.L2
    assert tb.frame()  # nothing to train on yet
    H, W, n = 24, 32, 3
    rng = np.random.default_rng(0)
    poses = np.tile(np.eye(4, dtype=np.float32)[None, :3], (n, 1, 1))
    poses[:, :, 3] = rng.uniform(-0.1, 0.1, (n, 3))
    images = rng.uniform(0, 1, (n, H, W, 4)).astype(np.float32)
    images[..., 3] = 1
    depths = rng.uniform(0.5, 2.0, (n, H, W, 1)).astype(np.float32)
    cov = np.ones((n, H, W, 1), np.float32)
    tb.nerf.training.update_training_images([0, 1, 2], list(poses), list(images), list(depths), list(cov), [W, H],
                                            [0.5, 0.5], [30.0, 30.0], 1.0, 1.0)
    assert tb.nerf.training.n_images_for_training == 3
    tb.steps_per_frame = 4
    assert tb.frame() and np.isfinite(tb.loss) and tb.training_step == 4 and tb.elapsed_training_time > 0
    tb.set_camera_to_training_view(1)
    tb.render_mode = ngp.Shade
    img = tb.render(W, H, 1, True)
    tb.render_mode = ngp.Depth
    dep = tb.render(W, H, 1, True)
    assert img.shape == (H, W, 4) and dep.shape == (H, W, 4) and np.isfinite(img).all()


@pytest.mark.gpu
def test_nerf_fusion_consumes_slam_packet(dev):
    """SLAM -> mapper packet (visual_frontend.py:1364-1382) through NerfFusion.fuse (nerf_fusion.py:238-262):
    the first spin ingests (no training, like the reference), later spins train; the loss decreases."""
    import argparse
    from nerfslam.nerf_fusion import NerfFusion
    g = torch.Generator().manual_seed(0)
    n, H, W = 3, 32, 48
    poses = torch.tensor([[0.0, 0, 0, 0, 0, 0, 1], [0.05, 0, 0, 0, 0, 0, 1], [-0.05, 0.02, 0, 0, 0, 0, 1]])
    img = torch.randint(0, 255, (1, 3, 4, 6), generator=g).float()
    images = torch.nn.functional.interpolate(img, size=(H, W), mode="bilinear").repeat(n, 1, 1, 1).to(torch.uint8)
    pkt = {"cam0_poses": poses, "cam0_images": images, "cam0_idepths_up": torch.full((n, H, W), 1.0),
           "cam0_depths_cov_up": torch.full((n, H, W), 0.01), "cam0_intrinsics": torch.tensor([[40.0, 40.0, 24.0, 16.0]] * n),
           "viz_idx": torch.tensor([0, 1, 2]), "kf_idx": 2, "is_last_frame": False}
    fusion = NerfFusion("nerf", argparse.Namespace(buffer=8, mask_type="ours"), dev)
    assert fusion.fuse({"slam": [None, pkt]}) is True
    assert fusion.ngp.nerf.training.n_images_for_training == 3 and fusion.total_iters == 0
    fusion.fuse(False)
    l0, s0 = fusion.ngp.loss, fusion.total_iters
    for _ in range(8):
        fusion.fuse(False)
    assert s0 > 0 and fusion.total_iters > s0 and np.isfinite(fusion.ngp.loss) and fusion.ngp.loss < l0
    m = fusion.evaluate(stride=2)
    assert m["views"] == 2 and np.isfinite(m["psnr"]) and m["psnr"] > 5.0 and np.isfinite(m["depth_l1_cm"])


@pytest.mark.parametrize("mask_type", ["ours", "raw", "ours_w_thresh", "no_depth"])
def test_ingest_arithmetic_matches_the_reference_restatement(oracle_mod, dev, mask_type):
    """SLAM packet -> training images (fusion/nerf_fusion.py:158-226): what the device-side
    `NerfFusion.process_slam` leaves in the trainer's slot arrays vs `oracle.nerf_ingest` -- camera-to-world from cam_T_world
    (random rotations), sRGB -> linear, alpha, depth = 1 / idepth, covariance pass-through, the four mask policies, slots
    selected by viz_idx.  The dataset offset 0.5 is this project's pyngp convention (the fork's is not in the tree)."""
    import argparse
    from nerfslam.nerf_fusion import NerfFusion
    rng = np.random.default_rng(3)
    n, H, W = 3, 24, 40
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.normal(0, 0.3, (n, 3)), q], 1).astype(np.float32)
    images = rng.integers(0, 256, (n, 3, H, W), dtype=np.uint8)
    idepth = rng.uniform(0.2, 2.0, (n, H, W)).astype(np.float32)
    cov = rng.uniform(0.001, 0.5, (n, H, W)).astype(np.float32)
    viz = np.array([4, 1, 6])
    pkt = {"cam0_poses": torch.from_numpy(poses), "cam0_images": torch.from_numpy(images), "cam0_idepths_up": torch.from_numpy(idepth),
           "cam0_depths_cov_up": torch.from_numpy(cov), "cam0_intrinsics": torch.tensor([[30.0, 31.0, 20.0, 12.0]] * n),
           "viz_idx": torch.from_numpy(viz), "kf_idx": 6, "is_last_frame": False}
    fusion = NerfFusion("nerf", argparse.Namespace(buffer=8, mask_type=mask_type), dev)
    assert fusion.process_slam(pkt) is False
    tb = fusion.ngp
    ref = oracle_mod.nerf_ingest({k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in pkt.items()}, mask_type=mask_type,
                                 scale=1.0, offset=(0.5, 0.5, 0.5))
    assert tb.nerf.training.n_images_for_training == 7 and tb._intr == (30.0, 31.0, 20.0, 12.0)
    got_c2w = tb._c2w[torch.from_numpy(viz).to(dev)].cpu().numpy()
    assert np.abs(got_c2w - ref["poses"]).max() <= 2e-6
    got_img = tb._imgs[torch.from_numpy(viz).to(dev)].cpu().numpy()
    assert np.abs(got_img - ref["images"]).max() <= 2e-6          # pow(x, 2.4) in f32 vs f64
    got_dep = tb._deps[torch.from_numpy(viz).to(dev)].cpu().numpy()
    assert np.array_equal(got_dep < 0, ref["depths"][..., 0] < 0)                      # masked pixels
    assert np.abs(got_dep - ref["depths"][..., 0]).max() <= 1e-6 * np.abs(ref["depths"]).max()
    got_cov = tb._covs[torch.from_numpy(viz).to(dev)].cpu().numpy()
    assert np.abs(got_cov - ref["depths_cov"][..., 0]).max() <= 1e-7 * np.abs(ref["depths_cov"]).max()
    if mask_type == "ours_w_thresh":   # the reference compares sqrt(cov) with the median of cov (:177-179): for cov < 1 most pixels go
        assert 0.7 < (got_dep < 0).mean() < 0.95
    untouched = [i for i in range(8) if i not in viz.tolist()]
    assert not tb._imgs[torch.tensor(untouched, device=dev)].any()


def test_camera_refinement_kernels(oracle_mod, dev):
    """optimize_extrinsics path: encoding input gradient, per-image 6-dof gradient, Adam + retraction -- each against the C
    restatement, and the input gradient against central differences of a float64 trilinear interpolation"""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    cfg = oracle_mod.ngp_cfg(n_levels=6, log2_hashmap=12, base_res=4, per_level_scale=1.6)
    _, res, off = oracle_mod.ngp_grid_layout(cfg)
    L, n_par = cfg.n_levels, int(off[-1]) * 2
    rng = np.random.default_rng(11)
    params = rng.uniform(-0.5, 0.5, n_par).astype(np.float16)
    N = 2000
    pos = rng.uniform(0.02, 0.98, (N, 3)).astype(np.float32)
    dL = (rng.standard_normal((N, 2 * L)) * 1e-1).astype(np.float16)
    dL[rng.uniform(size=N) < 0.2] = 0
    ref = oracle_mod.ngp_encode_bwd_input(cfg, pos, params, dL)
    args = (L, 2, cfg.log2_hashmap, cfg.base_res, C.c_float(cfg.per_level_scale))
    d_pos, d_par, d_dLT = T(pos, dev), T(params, dev), T(np.ascontiguousarray(dL.T), dev)
    out = torch.empty((N, 3), dtype=torch.float32, device=dev)
    check(lib().ns_ngp_encode_backward_input(*args, ptr(d_pos), ptr(d_par), ptr(d_dLT), ptr(out), C.c_long(N), stream_ptr()), "bwd_in")
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    # finite differences (float64) of  sum_lf dL[l,f] * feat_lf(pos)  on the dense levels (0, 1: res^3 <= table)
    scale = [float(np.exp2(l * np.log2(cfg.per_level_scale)) * cfg.base_res - 1.0) for l in range(L)]
    dense = [l for l in range(L) if res[l] ** 3 <= (1 << cfg.log2_hashmap)]
    assert len(dense) >= 2
    P64 = params.astype(np.float64).reshape(-1, 2)

    def energy(p, i):
        e = 0.0
        for l in dense:
            q = scale[l] * p + 0.5
            g = np.floor(q).astype(int); w = q - g
            for corner in range(8):
                c = [(corner >> d) & 1 for d in range(3)]
                wt = np.prod([w[d] if c[d] else 1 - w[d] for d in range(3)])
                idx = ((g[0] + c[0]) + (g[1] + c[1]) * int(res[l]) + (g[2] + c[2]) * int(res[l]) ** 2) % (int(off[l + 1]) - int(off[l]))
                v = P64[int(off[l]) + idx]
                e += wt * (float(dL[i, 2 * l]) * v[0] + float(dL[i, 2 * l + 1]) * v[1])
        return e
    dl_dense = dL.copy()
    for l in range(L):
        if l not in dense:
            dl_dense[:, 2 * l:2 * l + 2] = 0
    ref_dense = oracle_mod.ngp_encode_bwd_input(cfg, pos, params, dl_dense)
    h = 1e-5
    for i in range(0, 40):
        qs = [scale[l] * pos[i].astype(np.float64) + 0.5 for l in dense]
        if min(np.min(np.minimum(q - np.floor(q), np.ceil(q) - q)) for q in qs) < 1e-3:
            continue     # too close to a cell face (of some level) for a central difference
        fd = [(energy(pos[i].astype(np.float64) + h * np.eye(3)[d], i) - energy(pos[i].astype(np.float64) - h * np.eye(3)[d], i)) / (2 * h)
              for d in range(3)]
        assert np.abs(np.array(fd) - ref_dense[i]).max() <= 2e-3 * max(1.0, np.abs(ref_dense[i]).max()), i
    # per-image gradient
    R, n_img = 300, 7
    ray_n = rng.integers(-1, 30, R).astype(np.int32)
    ray_start = np.concatenate([[0], np.cumsum(np.maximum(ray_n, 0))[:-1]]).astype(np.int32)
    S = int(np.maximum(ray_n, 0).sum())
    dpos = rng.standard_normal((S, 3)).astype(np.float32)
    tm = rng.uniform(0.1, 4, S).astype(np.float32)
    rd = rng.standard_normal((R, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ri = rng.integers(0, n_img, R).astype(np.int32)
    gref = oracle_mod.ngp_camera_gradient(dpos, tm, rd, ray_start, ray_n, ri, 0.25, n_img)
    keep = [T(x, dev) for x in (dpos, tm, rd, ray_start, ray_n, ri)]
    cg = torch.zeros((n_img, 6), dtype=torch.float32, device=dev)
    check(lib().ns_ngp_camera_gradient(*[ptr(k) for k in keep], C.c_float(0.25), ptr(cg), R, stream_ptr()), "cam_grad")
    assert np.abs(cg.cpu().numpy() - gref).max() <= 1e-4 * np.abs(gref).max()
    # Adam + retraction
    c2w = np.zeros((n_img, 3, 4), np.float32)
    for k in range(n_img):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[k, :, :3], c2w[k, :, 3] = q * np.sign(np.linalg.det(q)), rng.uniform(-1, 1, 3)
    g6 = gref.astype(np.float32) * 128
    g6[2] = 0                                                   # an image no ray hit: untouched
    m1 = (rng.standard_normal((n_img, 6)) * 0.1).astype(np.float32); m2 = rng.uniform(0, 0.1, (n_img, 6)).astype(np.float32)
    rc, r1, r2 = oracle_mod.ngp_camera_step(c2w, g6, m1, m2, step=5, lr_pos=1e-2, lr_rot=2e-2, grad_scale=128.0)
    dc, dg, d1, d2 = T(c2w, dev), T(g6, dev), T(m1, dev), T(m2, dev)
    check(lib().ns_ngp_camera_step(ptr(dc), ptr(dg), ptr(d1), ptr(d2), n_img, 5, C.c_float(1e-2), C.c_float(2e-2), C.c_float(0.9),
                                   C.c_float(0.99), C.c_float(1e-15), C.c_float(128.0), stream_ptr()), "cam_step")
    np.testing.assert_allclose(dc.cpu().numpy(), rc, atol=2e-6)
    np.testing.assert_allclose(d1.cpu().numpy(), r1, rtol=1e-5, atol=1e-8)
    assert not dg.any() and np.array_equal(dc.cpu().numpy()[2], c2w[2])
    Rn = dc.cpu().numpy()[:, :, :3]
    assert np.abs(Rn @ Rn.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5        # still rotations


def test_pose_refinement_pulls_a_perturbed_camera_back(dev):
    """end to end: train on the true poses, freeze the field, move one view 3.7 cm off and let optimize_extrinsics
    (translation only) pull it back through the rendering loss"""
    import importlib.util
    import os
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    imgs, deps, covs, poses, intr = sc.sphere_scene(n=8, H=60, W=80, f=75.0)
    net = NgpNerf(NgpConfig(), dev, seed=0)
    net.set_images(imgs, deps, covs, poses.clone(), intr)
    for _ in range(500):
        net.train_step()
    delta = torch.tensor([0.03, -0.02, 0.01])
    net.c2w[0, :, 3] += delta.to(dev)
    net.cfg.lr, net.cfg.optimize_extrinsics, net.cfg.extrinsic_lr_pos, net.cfg.extrinsic_lr_rot = 0.0, True, 3e-4, 0.0
    errs = []
    for k in range(500):
        net.train_step()
        if k % 100 == 99:
            errs.append((net.c2w[0, :, 3].cpu() - poses[0, :, 3]).norm().item())
    others = (net.c2w[1:, :, 3].cpu() - poses[1:, :, 3]).norm(dim=-1).max().item()
    assert errs[-1] < 0.5 * float(delta.norm()) and others < 0.01, (errs, others)


def test_trained_field_renders_the_scene(dev):
    """quality, not just a falling loss: after 600 steps on the synthetic sphere the rendered training views reach > 30 dB
    PSNR and the rendered depth is within 1.5 cm on the object (the depth-covariance-weighted term at work)"""
    import importlib.util
    import os
    from nerfslam import eval as ev
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    imgs, deps, covs, poses, intr = sc.sphere_scene(n=8, H=60, W=80, f=75.0)
    net = NgpNerf(NgpConfig(), dev, seed=0)
    net.set_images(imgs, deps, covs, poses, intr)
    for _ in range(600):
        net.train_step()
    ps, de = [], []
    for k in (1, 5):
        rgb, dep = net.render(poses[k], 60, 80)
        ps.append(ev.psnr(rgb.cpu(), imgs[k, ..., :3]))
        m = deps[k] > 0
        de.append(float((dep.cpu()[m] - deps[k][m]).abs().mean()))
    assert min(ps) > 30.0 and max(de) < 0.015, (ps, de)


@pytest.mark.parametrize("steps", [600, 3000])
def test_fixed_point_table_gradient_trains_like_the_f32_gradient(dev, steps):
    """the product accumulates the table gradient as packed fixed point of the 128x loss-scaled gradient where
    the published algorithm adds every contribution (tiny-cuda-nn: loss-scaled f16 / f32 atomics) under an Adam whose eps = 1e-15
    acts on arbitrarily small gradients.  The f32 path still exists (grad_fixed_scale = 0: f32 atomics, streaming Adam, twice the
    step time).  Same scene and step count through both, several ray seeds each: one seed moves the PSNR by +-1.3 dB (the f32 arm is
    not even reproducible for one seed -- its atomics order the sums), so the PSNR criterion is statistical -- the deficit of the
    fixed-point arm's mean, less two standard errors of the difference, must not exceed 1.0 dB -- while the object's depth L1, which
    is stable to ~0.1 mm, must be within 0.5 mm; early (600 steps, 5 seeds) and deep into convergence (3000 steps, 3 seeds).
    Measured (profiles/r06_ab_records.json): the default 2^22 is 0.4-1.5 dB and 0.02-0.25 mm behind f32; rounds 2-5's 2^18 was
    1.3-2.7 dB and 0.25-0.7 mm behind -- that arm is run too and must show the larger depth deficit, so that this test keeps
    measuring what it claims to."""
    import importlib.util
    import json
    import os
    from nerfslam import eval as ev
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    imgs, deps, covs, poses, intr = sc.sphere_scene(n=8, H=60, W=80, f=75.0)
    assert NgpConfig().grad_fixed_scale == 2.0 ** 22
    seeds = range(5) if steps <= 600 else range(3)
    out = {}
    for name, scale in (("default_q22", NgpConfig().grad_fixed_scale), ("f32", 0.0), ("q18", 262144.0)):
        ps, de = [], []
        for seed in seeds:
            net = NgpNerf(NgpConfig(grad_fixed_scale=scale), dev, seed=seed)
            net.set_images(imgs, deps, covs, poses, intr)
            for _ in range(steps):
                net.train_step()
            p1, d1 = [], []
            for k in range(8):
                rgb, dep = net.render(poses[k], 60, 80)
                p1.append(ev.psnr(rgb.cpu(), imgs[k, ..., :3]))
                m = deps[k] > 0
                d1.append(float((dep.cpu()[m] - deps[k][m]).abs().mean()))
            ps.append(float(np.mean(p1))); de.append(1e3 * float(np.mean(d1)))
            del net
        out[name] = {"psnr_db_mean": float(np.mean(ps)), "psnr_db_by_seed": ps, "psnr_db_stderr": float(np.std(ps, ddof=1) / np.sqrt(len(ps))),
                     "depth_l1_mm_mean": float(np.mean(de)), "depth_l1_mm_by_seed": de}
    print("FIXED_POINT_VS_F32 " + json.dumps({"steps": steps, **out}))
    q, f, o = out["default_q22"], out["f32"], out["q18"]
    se = float(np.hypot(q["psnr_db_stderr"], f["psnr_db_stderr"]))
    assert (f["psnr_db_mean"] - q["psnr_db_mean"]) - 2.0 * se <= 1.0, (out, se)
    assert q["depth_l1_mm_mean"] <= f["depth_l1_mm_mean"] + 0.5, out
    assert f["psnr_db_mean"] > 28.0, out                 # the f32 arm itself trains (the comparison is not between two failures)
    assert o["depth_l1_mm_mean"] > q["depth_l1_mm_mean"], out               # the coarser scale's deficit is visible at this sample size


def test_hash_encode_backward_full_budget_properties(oracle_mod, dev):
    """the owner-computes encode backward at the trainer's full sample budget (2^18 samples, default 16-level grid), samples
    clustered along rays like the marcher's: (1) partition of unity -- per level and feature the table gradient sums to the
    sum of the incoming gradients; (2) packed fixed-point runs are bit-identical; (3) a 20k-sample prefix equals the oracle."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpConfig, unpack_fixed
    c = NgpConfig()
    cfg = oracle_mod.ngp_cfg(n_levels=c.n_levels, log2_hashmap=c.log2_hashmap, base_res=c.base_res, per_level_scale=c.per_level_scale)
    _, _, off = oracle_mod.ngp_grid_layout(cfg)
    n_par = int(off[-1]) * 2
    L = c.n_levels
    args = (L, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
    rng = np.random.default_rng(7)
    N, R = 1 << 18, 2048
    o = rng.uniform(0.3, 0.7, (R, 1, 3))
    d = rng.standard_normal((R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (0.02 + 0.0017 * np.arange(N // R))[None, :, None]
    pos = np.clip(o + t * d, 0.0, 1.0).reshape(N, 3).astype(np.float32)
    dLT = (rng.standard_normal((2 * L, N)) * 1e-2).astype(np.float16)
    dLT[:, rng.uniform(size=N) < 0.2] = 0
    d_pos, d_dLT = T(pos, dev), T(dLT, dev)
    S = 262144.0
    runs = []
    for _ in range(2):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(gq), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
        runs.append(gq.cpu().numpy())
    assert np.array_equal(runs[0], runs[1])
    # the binned path (workspace given) == the owner-computes path (no workspace), bit for bit, and reproducible
    ws = torch.zeros(lib().ns_ngp_encode_backward_workspace_bytes(*args, C.c_long(N)) // 4 + 1, device=dev)
    for _ in range(2):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(gq), ptr(ws), C.c_size_t(ws.numel() * ws.element_size()), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
        assert np.array_equal(gq.cpu().numpy(), runs[0])
    # a workspace that is too small for this N is never written: the call takes the owner-computes kernels
    before = ws.clone()
    gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(gq), ptr(ws), C.c_size_t(ws.numel() * ws.element_size() // 3),
                                       C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
    assert np.array_equal(gq.cpu().numpy(), runs[0]) and torch.equal(ws.view(torch.int32), before.view(torch.int32))
    # ... also for a sample count that is not a multiple of the 1024-sample tiles, and for the non-unit-major layout
    n_odd = 100003
    sub_T = T(np.ascontiguousarray(dLT[:, :n_odd]), dev)
    sub_N = T(np.ascontiguousarray(dLT[:, :n_odd].T), dev)
    outs = []
    for (dl, um, w) in ((sub_T, 1, None), (sub_T, 1, ws), (sub_N, 0, ws)):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(dl), um, ptr(gq), ptr(w), C.c_size_t(0 if w is None else w.numel() * w.element_size()), C.c_float(S), C.c_long(n_odd), stream_ptr()), "bwd")
        outs.append(gq.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    # ... and for the training step's form: the full-budget buffers with the sample count in device memory (the tail of the
    # gradient buffer then holds stale values that must not be read: poison it)
    n_dev = torch.tensor([n_odd], dtype=torch.int32, device=dev)
    poisoned = dLT.copy()
    poisoned[:, n_odd:] = np.float16(3.0)
    d_poison = T(poisoned, dev)
    outs_n = []
    for w in (None, ws):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward_n(*args, ptr(d_pos), ptr(d_poison), 1, ptr(gq), ptr(w), C.c_size_t(0 if w is None else w.numel() * w.element_size()), C.c_float(S), C.c_long(N), ptr(n_dev),
                                             stream_ptr()), "bwd_n")
        outs_n.append(gq.cpu().numpy())
    assert np.array_equal(outs_n[0], outs[0]) and np.array_equal(outs_n[1], outs[0])
    g0, g1 = unpack_fixed(runs[0], S)
    want = dLT.astype(np.float64).reshape(L, 2, N).sum(-1)
    for l in range(L):
        a, b = int(off[l]), int(off[l + 1])
        # every contribution is rounded to 2^-18 once: |error| <= 8 N 2^-19 in the worst case, ~sqrt of that in practice
        assert abs(g0[a:b].sum() - want[l, 0]) <= 4e-3 * max(1.0, abs(want[l, 0])) + 0.05, (l, g0[a:b].sum(), want[l, 0])
        assert abs(g1[a:b].sum() - want[l, 1]) <= 4e-3 * max(1.0, abs(want[l, 1])) + 0.05, l
    # float accumulation path on a prefix, against the oracle
    n = 20000
    gf = torch.zeros(n_par, dtype=torch.float32, device=dev)
    sub = T(np.ascontiguousarray(dLT[:, :n]), dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(sub), 1, ptr(gf), None, C.c_size_t(0), C.c_float(0), C.c_long(n), stream_ptr()), "bwd")
    gref = oracle_mod.ngp_encode_bwd(cfg, pos[:n], np.ascontiguousarray(dLT[:, :n].T), n_par)
    assert np.abs(gf.cpu().numpy() - gref).max() <= 2e-5 * np.abs(gref).max()


def test_occupancy_refresh_kernels_against_torch(dev):
    """ns_ngp_grid_cells + ns_ngp_grid_update (the subset form of update_density_grid on HIP kernels) against the torch
    statement of the same rule: cells uniform over all cascades, a jittered point INSIDE each drawn cell, grid = max(decay *
    grid, exp(log-density) * min_step), occupied = grid > min(mean, threshold), bit i of byte j = cell 8 j + i."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpConfig, NgpNerf
    cfg = NgpConfig()
    net = NgpNerf(cfg, dev, seed=3)
    g = torch.Generator(device=dev).manual_seed(0)
    net.grid_half.copy_((torch.rand(net.grid_half.shape, device=dev, generator=g) - 0.5).half())     # a field with structure
    net.density_grid.copy_(torch.rand(net.density_grid.shape, device=dev, generator=g) * 0.02)
    G, nc = cfg.grid_size, cfg.n_cascades
    total, n = nc * G ** 3, 1 << 16
    s = float(cfg.aabb_scale)
    cells = torch.empty(n, dtype=torch.int32, device=dev)
    pos = torch.empty((n, 3), dtype=torch.float32, device=dev)
    check(lib().ns_ngp_grid_cells(G, nc, C.c_uint32(12345), n, C.c_float(0.5 - 0.5 * s), C.c_float(0.5 + 0.5 * s), ptr(cells), ptr(pos),
                                  stream_ptr()), "grid_cells")
    cl = cells.long()
    assert int(cl.min()) >= 0 and int(cl.max()) < total
    counts = torch.bincount(cl // (G ** 3), minlength=nc).float() / n
    assert (counts - 1.0 / nc).abs().max().item() < 0.02                       # uniform over the cascades
    scene = pos * s + (0.5 - 0.5 * s)
    mip, r = cl // (G ** 3), cl % (G ** 3)
    xyz = torch.stack([r % G, (r // G) % G, r // (G * G)], -1)
    inside = ((scene - 0.5) / (2.0 ** mip.float())[:, None] + 0.5) * G
    assert (inside.floor().long() == xyz).float().mean().item() > 0.999       # (a point exactly on a cell face may round out)
    # expected update, in torch
    before = net.density_grid.clone()
    dens = net.density_at(scene.contiguous()) * cfg.min_step
    want = before * cfg.grid_decay
    want.scatter_reduce_(0, cl, dens, "amax", include_self=True)
    thr = min(float(want.double().mean()), cfg.min_optical_thickness)
    feat = net.encode(pos)
    out = torch.empty((n, 4), dtype=torch.float16, device=dev)
    dirs = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    nul = C.c_void_p(0)
    check(lib().ns_ngp_mlp_forward(ptr(net.mlp_half), ptr(feat), ptr(dirs), ptr(out), nul, nul, nul, nul, C.c_long(n), stream_ptr()), "mlp")
    part = torch.zeros(256, dtype=torch.float64, device=dev)
    check(lib().ns_ngp_grid_update(ptr(out), ptr(cells), n, C.c_float(cfg.min_step), C.c_float(cfg.grid_decay),
                                   C.c_float(cfg.min_optical_thickness), ptr(net.density_grid), C.c_long(total), ptr(part), ptr(net.bits),
                                   stream_ptr()), "grid_update")
    assert torch.allclose(net.density_grid, want, rtol=2e-3, atol=1e-9)
    occ = ((net.bits[:, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(-1).bool()
    sure = (want - thr).abs() > 1e-3 * thr                                    # cells a rounding of the mean could flip are not judged
    assert torch.equal(occ[sure], (want > thr)[sure]) and sure.float().mean().item() > 0.99
    # the trainer's own update path runs the same kernels and is reproducible (replicas must stay in lockstep)
    a, b = NgpNerf(cfg, dev, seed=5), NgpNerf(cfg, dev, seed=5)
    for m in (a, b):
        m.update_density_grid(); m.update_density_grid()
    assert torch.equal(a.bits, b.bits) and torch.equal(a.density_grid, b.density_grid) and a._grid_updates == 2


def test_hash_encode_survives_positions_outside_the_unit_cube(dev):
    """Positions outside [0,1]^3 have no meaning for the encoding, but they reach the kernels: the 8-rounded sample count covers
    up to 7 stale slots of the sample array, which after render() hold SCENE coordinates (e.g. -1.2).  On the dense levels such
    a position used to index gigabytes outside the table (no modulo in the fast index): an intermittent memory fault in the
    training steps that follow a render().  Forward, input gradient and table gradient must stay inside the tables."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpConfig, NgpNerf
    net = NgpNerf(NgpConfig(), dev, seed=2)
    args = net._grid_args()
    N = 4096
    g = torch.Generator(device=dev).manual_seed(4)
    pos = torch.rand((N, 3), device=dev, generator=g)
    wild = torch.tensor([-1.2, 2.7, -1e-7, 1.0000001, -3.0e9, 4.0e9, 1e30, -1e30], device=dev)
    pos[::5] = wild[torch.randint(0, 8, (pos[::5].shape[0], 3), device=dev, generator=g)]
    pos[7] = float("nan")
    feat = torch.empty((32, N), dtype=torch.float16, device=dev)
    dfe = (torch.randn((32, N), device=dev, generator=g) * 1e-2).half()
    dpos = torch.empty((N, 3), dtype=torch.float32, device=dev)
    gq = torch.zeros(net.n_grid // 2, dtype=torch.int64, device=dev)
    ws = torch.zeros(lib().ns_ngp_encode_backward_workspace_bytes(*args, C.c_long(N)) // 4 + 1, device=dev)
    for _ in range(3):
        check(lib().ns_ngp_encode_forward(*args, ptr(pos), ptr(net.grid_half), ptr(feat), 1, C.c_long(N), stream_ptr()), "fwd")
        check(lib().ns_ngp_encode_backward_input_n(*args, ptr(pos), ptr(net.grid_half), ptr(dfe), ptr(dpos), C.c_long(N), None,
                                                   stream_ptr()), "bwd_input")
        for w in (None, ws):
            check(lib().ns_ngp_encode_backward(*args, ptr(pos), ptr(dfe), 1, ptr(gq), ptr(w), C.c_size_t(0 if w is None else w.numel() * w.element_size()), C.c_float(262144.0), C.c_long(N),
                                               stream_ptr()), "bwd")
        torch.cuda.synchronize()
    ok = torch.isfinite(pos).all(-1) & (pos.abs() < 1e6).all(-1)
    assert torch.isfinite(feat[:, ok].float()).all()
    inside = ((pos >= 0) & (pos <= 1)).all(-1)
    ref = torch.empty((32, int(inside.sum())), dtype=torch.float16, device=dev)   # the in-cube samples are unaffected by their neighbours
    check(lib().ns_ngp_encode_forward(*args, ptr(pos[inside].contiguous()), ptr(net.grid_half), ptr(ref), 1, C.c_long(ref.shape[1]),
                                      stream_ptr()), "fwd")
    assert torch.equal(feat[:, inside], ref)


def _fused_setup(oracle_mod, dev, N, seed, clustered=False):
    from nerfslam._lib import lib
    from nerfslam.ngp import NgpConfig
    c = NgpConfig()
    cfg = oracle_mod.ngp_cfg(n_levels=c.n_levels, log2_hashmap=c.log2_hashmap, base_res=c.base_res, per_level_scale=c.per_level_scale)
    _, _, off = oracle_mod.ngp_grid_layout(cfg)
    L = c.n_levels
    args = (L, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
    rng = np.random.default_rng(seed)
    R = 2048
    if clustered:      # every sample in one of two x-adjacent finest-level cells, alternating (neighbours never merge there):
        pos = (0.5 + rng.uniform(0, 1e-6, (N, 3))).astype(np.float32)   # a dozen indices per level carry all of its records
        pos[1::2, 0] += np.float32(1.3e-4)
    else:
        o = rng.uniform(0.3, 0.7, (R, 1, 3))
        d = rng.standard_normal((R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        t = (0.02 + 0.0017 * np.arange(N // R))[None, :, None]
        pos = np.clip(o + t * d, 0.0, 1.0).reshape(N, 3).astype(np.float32)
    dLT = (rng.standard_normal((2 * L, N)) * 1e-2).astype(np.float16)
    dLT[:, rng.uniform(size=N) < 0.2] = 0
    wsb = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(N)))
    assert wsb > 0
    ws = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)
    return c, args, int(off[-1]) * 2, pos, dLT, ws, wsb


@pytest.mark.parametrize("clustered", [False, True], ids=["rays", "one_cell"])
def test_fused_table_gradient_matches_the_other_paths(oracle_mod, dev, clustered):
    """round-3 binned path (no count pass, 64 bins, fixed regions + overflow list): the packed sums equal the owner-computes
    path bit for bit -- also when a handful of bins receive every record (regions overflow into the list), with the sample
    count in device memory, and when called repeatedly on the same workspace (it leaves its counters cleared)."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    N = 1 << 18 if not clustered else 1 << 16
    c, args, n_par, pos, dLT, ws, wsb = _fused_setup(oracle_mod, dev, N, 11, clustered)
    d_pos, d_dLT = T(pos, dev), T(dLT, dev)
    S = 262144.0
    nul = C.c_void_p(0)
    ref = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dLT), 1, ptr(ref), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")

    def fused(n_dev=None, dl=d_dLT):
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(dl), ptr(gq), ptr(ws), C.c_size_t(wsb), C.c_float(S),
                                                   C.c_long(N), ptr(n_dev), nul, nul, nul, nul, 1, C.c_float(0), C.c_float(0),
                                                   C.c_float(0), C.c_float(0), C.c_float(1), nul, 15, stream_ptr()), "fused")
        return gq
    for _ in range(3):
        assert torch.equal(fused(), ref)
    w32 = ws.view(torch.int32)
    assert int(w32[0]) == 0 and int(w32[1]) == 0, "overflow counter not left cleared / error flag set"
    ntiles = (N + 1023) // 1024
    run_lengths = w32[64:64 + 12 * 64 * ntiles]
    if clustered:   # the overflow list was really used: runs longer than their 512-record slots
        assert int((run_lengths == 512).sum()) > 0
    else:           # no slot is full: 512 records on the hashed levels, up to 8192 on the binned dense ones (fewer bins share a tile's records)
        assert 0 < int(run_lengths.max()) < 8192
    # sample count in device memory; the tail holds poison that must not be read
    n_odd = 100003 if not clustered else 33331
    ref_n = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    sub = T(np.ascontiguousarray(dLT[:, :n_odd]), dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(sub), 1, ptr(ref_n), None, C.c_size_t(0), C.c_float(S), C.c_long(n_odd), stream_ptr()), "bwd")
    poisoned = dLT.copy(); poisoned[:, n_odd:] = np.float16(3.0)
    nd = torch.tensor([n_odd], dtype=torch.int32, device=dev)
    assert torch.equal(fused(nd, T(poisoned, dev)), ref_n)
    # workspace size is checked
    with pytest.raises(Exception):
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dLT), ptr(ref), ptr(ws), C.c_size_t(wsb // 2), C.c_float(S),
                                                   C.c_long(N), nul, nul, nul, nul, nul, 1, C.c_float(0), C.c_float(0), C.c_float(0),
                                                   C.c_float(0), C.c_float(1), nul, 15, stream_ptr()), "fused")


def test_fused_table_gradient_adam_is_bit_identical(oracle_mod, dev):
    """Adam applied in the flush of the accumulation == table gradient into the buffer, then ns_ngp_adam over the whole table:
    master, both moments and the f16 working copy bit for bit, over three consecutive steps (moments carried)"""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    N = 1 << 17
    c, args, n_par, pos, dLT, ws, wsb = _fused_setup(oracle_mod, dev, N, 12)
    S, lr, b1, b2, eps, gs = 262144.0, 1e-2, 0.9, 0.99, 1e-15, 128.0
    g = torch.Generator().manual_seed(3)
    m0 = (torch.rand(n_par, generator=g) * 2e-4 - 1e-4).to(dev)
    st = {}
    for name in ("two_pass", "fused"):
        st[name] = dict(master=m0.clone(), hp=m0.half(), m1=torch.zeros(n_par, device=dev), m2=torch.zeros(n_par, device=dev))
    rng = np.random.default_rng(5)
    nul = C.c_void_p(0)
    d_pos = T(pos, dev)
    for step in (1, 2, 3):
        dl = dLT.copy()
        dl[:, rng.uniform(size=N) < 0.5] = 0          # different entries touched each step
        d_dl = T(dl, dev)
        a = st["two_pass"]
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dl), 1, ptr(gq), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
        touched = int((gq != 0).sum())
        check(lib().ns_ngp_adam(ptr(a["master"]), ptr(a["hp"]), ptr(gq), ptr(a["m1"]), ptr(a["m2"]), C.c_long(n_par), step, C.c_float(lr),
                                C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(0.0), C.c_float(gs), C.c_float(S), stream_ptr()),
              "adam")
        f = st["fused"]
        check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dl), nul, ptr(ws), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                                   nul, ptr(f["master"]), ptr(f["hp"]), ptr(f["m1"]), ptr(f["m2"]), step, C.c_float(lr),
                                                   C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(gs), nul, 15, stream_ptr()), "fused")
        assert 0.02 * n_par / 2 < touched < 0.9 * n_par / 2
        for k in ("master", "m1", "m2", "hp"):
            assert torch.equal(a[k], f[k]), (step, k, int((a[k] != f[k]).sum()))
    assert not torch.equal(st["fused"]["master"], m0)


def test_interleaved_optimiser_records_are_bit_identical(oracle_mod, dev):
    """The table's optimiser state as ONE 32-byte record per entry ([master.xy | m1.xy | m2.xy | unused], what the trainer keeps:
    nerfslam.ngp.NgpNerf.new_grid_state; the library tells it from m1 == master + 2, m2 == master + 4) against three dense
    arrays: the fused flush (ns_ngp_encode_backward_fused_n) and the streaming pass (ns_ngp_adam, which then leaves untouched
    records alone) give the same master / moments / f16 copy bit for bit over three steps, and the unused 8 bytes stay zero."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpNerf
    N = 1 << 17
    c, args, n_par, pos, dLT, ws, wsb = _fused_setup(oracle_mod, dev, N, 12)
    S, lr, b1, b2, eps, gs = 262144.0, 1e-2, 0.9, 0.99, 1e-15, 128.0
    g = torch.Generator().manual_seed(3)
    m0 = (torch.rand(n_par, generator=g) * 2e-4 - 1e-4).to(dev)
    st = {}
    # (round 6: "rec6_*" = 24-byte records, the trainer's layout since then, TOLD to the library through the `_rec` entry points;
    #  "rec_*" = 32-byte records recognised from the pointers by the older entry points)
    for name in ("dense_fused", "rec_fused", "rec_stream", "rec6_fused", "rec6_stream"):
        if name.startswith("rec"):
            rf = 6 if name.startswith("rec6") else 8
            rec, ma, m1, m2 = NgpNerf.new_grid_state(n_par // 2, dev, rf)
            ma.copy_(m0.view(-1, 2))
            assert m1.data_ptr() == ma.data_ptr() + 8 and m2.data_ptr() == ma.data_ptr() + 16 and rec.stride(0) == rf
            st[name] = dict(rec=rec, master=ma, m1=m1, m2=m2, hp=m0.half())
        else:
            st[name] = dict(master=m0.clone(), hp=m0.half(), m1=torch.zeros(n_par, device=dev), m2=torch.zeros(n_par, device=dev))
    rng = np.random.default_rng(5)
    nul = C.c_void_p(0)
    d_pos = T(pos, dev)
    for step in (1, 2, 3):
        dl = dLT.copy()
        dl[:, rng.uniform(size=N) < 0.5] = 0
        d_dl = T(dl, dev)
        for name in ("dense_fused", "rec_fused"):
            f = st[name]
            check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dl), nul, ptr(ws), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                                       nul, ptr(f["master"]), ptr(f["hp"]), ptr(f["m1"]), ptr(f["m2"]), step, C.c_float(lr),
                                                       C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(gs), nul, 15, stream_ptr()), "fused")
        f = st["rec6_fused"]
        check(lib().ns_ngp_encode_backward_fused_rec_n(*args, ptr(d_pos), ptr(d_dl), nul, ptr(ws), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                                       nul, ptr(f["master"]), ptr(f["hp"]), ptr(f["m1"]), ptr(f["m2"]), 6, step, C.c_float(lr),
                                                       C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(gs), nul, 15, stream_ptr()), "fused rec6")
        r = st["rec6_stream"]
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dl), 1, ptr(gq), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
        check(lib().ns_ngp_adam_rec_ctl(ptr(r["master"]), ptr(r["hp"]), ptr(gq), ptr(r["m1"]), ptr(r["m2"]), 6, C.c_long(n_par), step,
                                        C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(0.0), C.c_float(gs), C.c_float(S),
                                        nul, stream_ptr()), "adam rec6")
        r = st["rec_stream"]
        gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
        check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dl), 1, ptr(gq), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
        check(lib().ns_ngp_adam(ptr(r["master"]), ptr(r["hp"]), ptr(gq), ptr(r["m1"]), ptr(r["m2"]), C.c_long(n_par), step, C.c_float(lr),
                                C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(0.0), C.c_float(gs), C.c_float(S), stream_ptr()), "adam")
        assert int((gq != 0).sum()) == 0                      # the streaming pass clears the gradient behind itself
        a = st["dense_fused"]
        for name in ("rec_fused", "rec_stream", "rec6_fused", "rec6_stream"):
            f = st[name]
            for k in ("master", "m1", "m2"):
                assert torch.equal(a[k], f[k].reshape(-1)), (name, step, k, int((a[k] != f[k].reshape(-1)).sum()))
            assert torch.equal(a["hp"], f["hp"]), (name, step)
            assert f["rec"].shape[1] == 6 or float(f["rec"][:, 6:].abs().max()) == 0.0
    # a record size that does not describe the pointers is refused, not guessed
    f = st["rec_fused"]
    rc = lib().ns_ngp_adam_rec_ctl(ptr(f["master"]), ptr(f["hp"]), ptr(gq), ptr(f["m1"]), ptr(f["m2"]), 2, C.c_long(n_par), 1, C.c_float(lr),
                                   C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(0.0), C.c_float(gs), C.c_float(S), nul, stream_ptr())
    assert rc != 0
    assert not torch.equal(st["rec_fused"]["master"].reshape(-1), m0)


@pytest.mark.parametrize("slabs", [64, 7, 300])
def test_mlp_optimiser_step_in_one_launch_is_bit_identical(dev, slabs):
    """ns_ngp_mlp_step_fused (slab reduce + Adam + f16 copy + both fragment tables, one launch) == ns_ngp_mlp_reduce +
    ns_ngp_adam + ns_ngp_mlp_pack_fragments: master, moments, f16 weights, the cleared gradient buffer and every byte of the
    fragment tables, over three steps (moments carried); weight decay on (every parameter moves) and off (zero gradients skip)."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    n = 10240
    g = torch.Generator().manual_seed(11 + slabs)
    W0 = (torch.rand(n, generator=g) - 0.5).to(dev) * 0.4
    nb = int(lib().ns_ngp_mlp_fragment_table_bytes()) // 2
    lr, b1, b2, eps, gs = 1e-2, 0.9, 0.99, 1e-15, 128.0
    st = {}
    for name in ("three", "one"):
        d = dict(master=W0.clone(), hp=W0.half(), m1=torch.zeros(n, device=dev), m2=torch.zeros(n, device=dev),
                 grad=torch.zeros(n, device=dev), frags=torch.zeros(nb, dtype=torch.float16, device=dev))
        check(lib().ns_ngp_mlp_pack_fragments(ptr(d["hp"]), ptr(d["frags"]), stream_ptr()), "pack")
        st[name] = d
    for step, l2 in ((1, 0.0), (2, 1e-6), (3, 0.0)):
        part = (torch.randn((slabs, n), generator=g) * 3.0).to(dev)
        part[:, torch.rand(n, generator=g) < 0.3] = 0.0                       # parameters without a gradient this step
        pre = (torch.randn(n, generator=g) * (step == 2)).to(dev)             # something already in the gradient buffer
        a, f = st["three"], st["one"]
        a["grad"].copy_(pre)
        f["grad"].copy_(pre)
        check(lib().ns_ngp_mlp_reduce(ptr(part), slabs, ptr(a["grad"]), stream_ptr()), "reduce")
        check(lib().ns_ngp_adam(ptr(a["master"]), ptr(a["hp"]), ptr(a["grad"]), ptr(a["m1"]), ptr(a["m2"]), C.c_long(n), step, C.c_float(lr),
                                C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(l2), C.c_float(gs), C.c_float(0.0), stream_ptr()),
              "adam")
        check(lib().ns_ngp_mlp_pack_fragments(ptr(a["hp"]), ptr(a["frags"]), stream_ptr()), "pack")
        check(lib().ns_ngp_mlp_step_fused(ptr(part), slabs, ptr(f["grad"]), ptr(f["master"]), ptr(f["hp"]), ptr(f["m1"]), ptr(f["m2"]),
                                          ptr(f["frags"]), step, C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(l2),
                                          C.c_float(gs), None, stream_ptr()), "step_fused")
        for k in ("master", "m1", "m2", "hp", "grad", "frags"):
            assert torch.equal(a[k].view(torch.int16 if a[k].dtype == torch.float16 else torch.int32),
                               f[k].view(torch.int16 if f[k].dtype == torch.float16 else torch.int32)), (step, k)
        assert float(f["grad"].abs().max()) == 0.0
    assert not torch.equal(st["one"]["master"], W0)


def test_pose_gradient_from_forward_jacobian(oracle_mod, dev):
    """ns_ngp_encode_forward_j_n + ns_ngp_encode_jacobian_dot_n == ns_ngp_encode_backward_input_n (8 x 16 gathers per sample) up
    to the f16 rounding of the Jacobian rows; features unchanged by the extra output; device sample count honoured"""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpConfig, NgpNerf
    net = NgpNerf(NgpConfig(), dev, seed=5)
    g = torch.Generator(device=dev).manual_seed(6)
    net.grid_half.copy_(((torch.rand(net.n_grid, device=dev, generator=g) - 0.5) * 0.2).half())     # a "trained" table
    args = net._grid_args()
    N = 8192
    pos = torch.rand((N, 3), device=dev, generator=g).contiguous()
    dfe = (torch.randn((32, N), device=dev, generator=g) * 1e-2).half().contiguous()
    feat, feat_j = (torch.empty((32, N), dtype=torch.float16, device=dev) for _ in range(2))
    jac = torch.zeros((96, N), dtype=torch.float16, device=dev)
    nd = torch.tensor([5000], dtype=torch.int32, device=dev)
    for n_dev in (None, nd):
        ref, got = torch.zeros((N, 3), device=dev), torch.zeros((N, 3), device=dev)
        check(lib().ns_ngp_encode_forward_n(*args, ptr(pos), ptr(net.grid_half), ptr(feat), 1, C.c_long(N), ptr(n_dev), stream_ptr()), "fwd")
        check(lib().ns_ngp_encode_forward_j_n(*args, ptr(pos), ptr(net.grid_half), ptr(feat_j), 1, ptr(jac), C.c_long(N), ptr(n_dev),
                                              stream_ptr()), "fwd j")
        n = N if n_dev is None else 5000
        assert torch.equal(feat[:, :n], feat_j[:, :n])
        check(lib().ns_ngp_encode_backward_input_n(*args, ptr(pos), ptr(net.grid_half), ptr(dfe), ptr(ref), C.c_long(N), ptr(n_dev),
                                                   stream_ptr()), "bwd input")
        check(lib().ns_ngp_encode_jacobian_dot_n(*args, ptr(jac), ptr(dfe), ptr(got), C.c_long(N), ptr(n_dev), stream_ptr()), "jac dot")
        assert ref[:n].abs().max() > 0 and torch.equal(ref[n:], got[n:])
        err = (got[:n] - ref[:n]).abs().max().item()
        assert err <= 2e-3 * ref[:n].abs().max().item(), (err, ref[:n].abs().max().item())
        rel = ((got[:n] - ref[:n]).norm() / ref[:n].norm()).item()
        assert rel < 5e-4, rel


def test_graph_capture_while_another_thread_synchronises(dev):
    """--parallel_run on one GPU: the mapper thread captures its step graphs (again whenever an address in the step
    changes) while the tracker thread launches kernels and reads results back.  ROCm 7.2 answered a synchronising call made
    during another thread's capture with hipErrorIllegalState; captures and the tracker's read-backs therefore share
    nerfslam._lib.capture_lock.  Loop both for a while: no exception in either thread, training still converges."""
    import importlib.util
    import threading
    import time
    from nerfslam._lib import capture_lock
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    imgs, deps, covs, poses, intr = sc.sphere_scene(n=4, H=60, W=80, f=75.0)
    net = NgpNerf(NgpConfig(), dev, seed=0)
    net.set_images(imgs, deps, covs, poses, intr)
    errors, stop, captures = [], threading.Event(), [0]
    side = torch.cuda.Stream(device=dev)

    def mapper():
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                for k in range(240):
                    if k % 12 == 0:               # a changed constant of the captured step: both parities are captured again
                        net.cfg.depth_lambda = 1.0 + 1e-3 * (k // 12)
                        captures[0] += 1
                    net.train_step(return_loss=False)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:               # noqa: BLE001
            errors.append(("mapper", e))
        finally:
            stop.set()

    def tracker():
        try:
            torch.cuda.set_device(dev)
            x = torch.randn((256, 256), device=dev)
            n = 0
            while not stop.is_set():
                y = x @ x                          # launches outside the lock, like the tracker's kernels
                with capture_lock:                 # host read-backs under the lock, like TrackingSLAM's
                    float(y[0, 0])
                    torch.cuda.current_stream().synchronize()
                n += 1
            assert n > 20
        except BaseException as e:               # noqa: BLE001
            errors.append(("tracker", e))

    ta, tb = threading.Thread(target=mapper), threading.Thread(target=tracker)
    t0 = time.time()
    ta.start(); tb.start()
    ta.join(timeout=180); tb.join(timeout=30)
    assert not ta.is_alive() and not tb.is_alive(), "threads hung"
    assert not errors, errors
    assert captures[0] >= 20 and net.step == 240 and np.isfinite(float(net.loss_tensor)) and time.time() - t0 < 180


@pytest.mark.gpu
def test_paired_step_graph_trains_like_single_steps(dev):
    """train_steps(n) replays two steps from one graph where it can (no occupancy update in between, both single-step graphs
    there): same bookkeeping (step count, set parity, marched-ahead rays, occupancy updates on the same steps) and -- since the
    marcher hands out its sample ranges in workgroup order (round 5: ns_ngp_march_ordered) -- the SAME training as n calls of
    train_step(), bit for bit: hash grid, MLP, refined poses, occupancy bits, loss."""
    import importlib.util
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    scene = sc.sphere_scene(n=4, H=60, W=80, f=75.0)
    nets = []
    for mode in ("single", "single", "paired", "chained"):
        net = NgpNerf(NgpConfig(optimize_extrinsics=True, **({"steps_per_graph": 2} if mode == "paired" else {})), dev, seed=0)
        net.set_images(*scene)
        if mode == "single":
            for _ in range(23 + 64):
                net.train_step(return_loss=False)
        else:
            for _ in range(23):                       # an odd start: chains begin on an even step, never straddle an update
                net.train_step(return_loss=False)
            assert net._pair is None and net._pair_r is None and not net._chains
            net.train_steps(64, return_loss=False)
            if mode == "paired":
                assert net._pair is not None and net._pair_r is not None        # (the pair that ends on an update carries the refresh)
            else:                                     # round 6 default: 8 steps per graph launch, the refresh in the last of a chain
                assert net.cfg.steps_per_graph == 8 and (8, True) in net._chains and (8, False) in net._chains   # (+ the 6-step tail)
        torch.cuda.synchronize()
        assert net.step == 87 and net.cur == 1
        nets.append(net)
    a, b, p, ch = nets
    for other, what in ((b, "second run of the same path"), (p, "paired-step graphs"), (ch, "8-step chains")):
        for name in ("grid_master", "grid_half", "mlp_master", "c2w", "bits", "density_grid"):
            assert torch.equal(getattr(other, name), getattr(a, name)), (what, name)
        assert float(other.loss_tensor) == float(a.loss_tensor), what
        assert other.last_samples == a.last_samples


@pytest.mark.gpu
def test_training_is_bit_reproducible(dev):
    """The optimiser step has no order-dependent arithmetic left (round 5): integer table gradient, fixed-order slab reduce of
    the MLP gradients, per-ray loss, order-independent occupancy max -- and sample ranges in workgroup order.  Two trainers of one
    seed agree bit for bit after 40 steps (two occupancy updates, a full-budget first step that REFUSES rays), whether the steps
    are launched eagerly or replayed from HIP graphs; a different seed does not."""
    import importlib.util
    from nerfslam.ngp import NgpConfig, NgpNerf
    spec = importlib.util.spec_from_file_location("ngp_scene", os.path.join(os.path.dirname(__file__), "..", "tools", "ngp_scene.py"))
    sc = importlib.util.module_from_spec(spec); spec.loader.exec_module(sc)
    scene = sc.sphere_scene()

    def run(seed, use_graph):
        net = NgpNerf(NgpConfig(optimize_extrinsics=True, use_graph=use_graph), dev, seed=seed)
        net.set_images(*scene)
        first = None
        for s in range(40):
            net.train_step(return_loss=False)
            if s == 0:
                torch.cuda.synchronize()
                X = net.sets[1 - net.cur]
                first = (X["counter"].clone(), X["ray_start"].clone(), X["ray_n"].clone(), X["s_pos"].clone())
        torch.cuda.synchronize()
        return net, first
    (a, fa), (b, fb), (e, fe_), (c, _) = run(0, True), run(0, True), run(0, False), run(1, True)
    req, _, end = fa[0].tolist()
    assert req > a.cfg.max_samples >= end and int((fa[2] < 0).sum()) > 0       # the first batch overflowed: some rays were refused
    for other, fo, what in ((b, fb, "second run"), (e, fe_, "eager launches")):
        for x, y in zip(fa, fo):
            assert torch.equal(x, y), (what, "first batch")
        for name in ("grid_master", "mlp_master", "c2w", "bits"):
            assert torch.equal(getattr(other, name), getattr(a, name)), (what, name)
    assert not torch.equal(c.grid_master, a.grid_master)


@pytest.mark.gpu
def test_fused_table_gradient_with_binned_dense_levels(dev):
    """NS_ENC_DENSE_BINNED=1 (off by default, read once per process): the multi-slice dense levels through the bins -- per-level
    slot sizes, dense indices in the scatter -- give the owner-computes path's sums bit for bit (tools/check_dense_binned.py)."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "check_dense_binned.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bit-identical" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("regime", ["converged", "mixed"])
@pytest.mark.parametrize("aabb", [1, 4])
def test_fused_table_gradient_in_the_converged_scene_regime(oracle_mod, dev, regime, aabb):
    """Round 4's paths for a converged scene's gradients (most contributions round to zero in Q18 fixed point): samples below
    half a unit are skipped before any weight is formed, sparse bins are read record by record through the prefix sums of their
    run lengths, every dense level goes through the bins (wave-level merge, small bins) -- the packed sums must equal the round-2
    path bit for bit, with 5 dense levels (aabb 1) and 4 (aabb 4), on tiny gradients and on a mix of tiny, ordinary and
    near-saturating ones; and Adam in the sparse flush must equal the streaming pass."""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.ngp import NgpConfig
    c = NgpConfig(aabb_scale=aabb)
    L = c.n_levels
    args = (L, 2, c.log2_hashmap, c.base_res, C.c_float(c.per_level_scale))
    off = (C.c_uint32 * (L + 1))()
    check(lib().ns_ngp_grid_layout(*args, None, None, off), "layout")
    n_par = int(off[L]) * 2
    N, R = 1 << 17, 1024
    rng = np.random.default_rng(40 + aabb)
    o = rng.uniform(0.3, 0.7, (R, 1, 3))
    d = rng.standard_normal((R, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (0.02 + 0.0017 * np.arange(N // R))[None, :, None]
    pos = np.clip(o + t * d, 0.0, 1.0).reshape(N, 3).astype(np.float32)
    sigma = np.full(N, 2e-6)
    if regime == "mixed":
        sigma[rng.uniform(size=N) < 0.2] = 1e-3
        sigma[rng.uniform(size=N) < 0.01] = 8.0                        # large contributions (> 2^21 units: never merged), 5 sigma
                                                                       # still inside the records' 25-bit fields (64 gradient units)
    dLT = (rng.standard_normal((2 * L, N)) * sigma[None, :]).astype(np.float16)
    dLT[:, (np.arange(N) % (N // R)) > 100] = 0                        # the tail of every ray carries nothing
    wsb = int(lib().ns_ngp_encode_backward_fused_workspace_bytes(*args, C.c_long(N)))
    ws = torch.zeros(wsb // 8 + 1, dtype=torch.int64, device=dev)
    d_pos, d_dl = T(pos, dev), T(dLT, dev)
    S = 262144.0
    nul = C.c_void_p(0)
    ref = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward(*args, ptr(d_pos), ptr(d_dl), 1, ptr(ref), None, C.c_size_t(0), C.c_float(S), C.c_long(N), stream_ptr()), "bwd")
    gq = torch.zeros(n_par // 2, dtype=torch.int64, device=dev)
    check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dl), ptr(gq), ptr(ws), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                               nul, nul, nul, nul, nul, 1, C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0),
                                               C.c_float(1), nul, 15, stream_ptr()), "fused")
    assert torch.equal(gq, ref)
    touched = int((ref != 0).sum())
    ntiles = (N + 1023) // 1024
    records = int(ws.view(torch.int32)[64:64 + L * 64 * ntiles].sum())
    assert 0 < touched <= records
    if regime == "converged":          # a few records per slot: the sparse path of the accumulate pass is what ran
        assert records < 4 * L * 64 * ntiles, records
    # Adam in the flush == gradient buffer + streaming Adam, bit for bit
    lr, b1, b2, eps, gs = 1e-2, 0.9, 0.99, 1e-15, 128.0
    g = torch.Generator().manual_seed(9)
    m0 = (torch.rand(n_par, generator=g) * 2e-4 - 1e-4).to(dev)
    a = dict(master=m0.clone(), hp=m0.half(), m1=torch.zeros(n_par, device=dev), m2=torch.zeros(n_par, device=dev))
    f = dict(master=m0.clone(), hp=m0.half(), m1=torch.zeros(n_par, device=dev), m2=torch.zeros(n_par, device=dev))
    check(lib().ns_ngp_adam(ptr(a["master"]), ptr(a["hp"]), ptr(ref), ptr(a["m1"]), ptr(a["m2"]), C.c_long(n_par), 1, C.c_float(lr),
                            C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(0.0), C.c_float(gs), C.c_float(S), stream_ptr()), "adam")
    check(lib().ns_ngp_encode_backward_fused_n(*args, ptr(d_pos), ptr(d_dl), nul, ptr(ws), C.c_size_t(wsb), C.c_float(S), C.c_long(N),
                                               nul, ptr(f["master"]), ptr(f["hp"]), ptr(f["m1"]), ptr(f["m2"]), 1, C.c_float(lr),
                                               C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(gs), nul, 15, stream_ptr()), "fused adam")
    for k in ("master", "m1", "m2", "hp"):
        assert torch.equal(a[k], f[k]), (k, int((a[k] != f[k]).sum()))


@pytest.mark.gpu
def test_training_matches_golden_checksums(dev):
    """tests/golden/ngp_training.json (tools/gen_golden_ngp.py, generated on an MI355X): sha256 of the f16 table, the MLP master
    weights, the refined poses and the occupancy bits after 48 optimiser steps (three occupancy updates, pair graphs) on the sphere
    scene.  A REGRESSION pin of this repository's own arithmetic -- possible since the step is bit-reproducible -- not a parity pin
    against the instant-ngp fork (absent: SURVEY 8c).  A kernel change that alters the arithmetic on purpose regenerates it."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("gen_golden_ngp", os.path.join(os.path.dirname(__file__), "..", "tools", "gen_golden_ngp.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ngp_training.json")))
    got = gen.state(dev, want["steps"])
    assert got["samples_of_last_step"] == want["samples_of_last_step"] and got["loss"] == want["loss"]
    assert got["sha256"] == want["sha256"]
