"""Derivation-independent pins of oracle/ngp_oracle.c (VERDICT r05 "what's missing" 5): every hand-derived BACKWARD of the mapping
path's CPU oracle -- the compositing suffix sums and the depth-covariance-weighted loss, the two MLPs, the hash-grid table and
input gradients -- against float64 torch AUTOGRAD of an independently written forward, and the oracle's Adam against
torch.optim.Adam.  The HIP kernels are compared with this oracle in tests/test_ngp_gpu.py; with these pins a shared
misreading of the published algorithm (Mueller et al. 2022; boundary: fusion/nerf_fusion.py:93-101, 285-299) in oracle AND
kernel would have to survive autograd as well.  The forward here is written from the paper's formulas in torch, not from
ngp_oracle.c's loops.  (The fork itself is absent: rows B1-B7 stay "parity unpinned"; this pins the derivatives, not the
fork's hyper-parameters.)  CPU only."""
import numpy as np
import pytest
import torch

import oracle

F16_EPS = 2.0 ** -11          # half a unit in the last place of an f16 in [1, 2): one rounding of a gradient


def _rays(rng, R, lo=3, hi=40):
    n = rng.integers(lo, hi, R).astype(np.int32)
    n[rng.integers(0, R)] = 0                                   # an empty ray
    start = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int32)
    return start, n, int(n.sum())


def test_composite_loss_backward_equals_autograd():
    """alpha compositing + RGB/3 + lambda (D - d)^2 / cov: loss and both gradients.  The oracle returns its gradients as f16
    (one rounding, <= 2^-11 relative per element), scaled by loss_scale / n_rays."""
    rng = np.random.default_rng(7)
    R = 48
    start, n, S = _rays(rng, R)
    rgb_raw = np.zeros((S, 16), np.float16)
    dens_raw = np.zeros((S, 16), np.float16)
    rgb_raw[:, :3] = rng.normal(0, 1.5, (S, 3))
    dens_raw[:, 0] = rng.normal(0.5, 1.5, S)
    dt = rng.uniform(0.004, 0.03, S).astype(np.float32)
    tmid = np.concatenate([np.cumsum(rng.uniform(0.01, 0.05, k)) + 0.1 for k in n]).astype(np.float32) if S else np.zeros(0, np.float32)
    gt_rgb = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    gt_depth = rng.uniform(0.3, 1.5, R).astype(np.float32)
    gt_depth[::5] = -1.0                                       # invalid depth: no depth term (nerf_fusion.py:179-181)
    cov = rng.uniform(0.01, 0.5, R).astype(np.float32)
    lam, loss_scale = 0.7, 128.0
    o_rgb, o_depth, o_loss, dLdrgb, dLddens = oracle.ngp_composite_loss(rgb_raw, dens_raw, dt, tmid, start, n, gt_rgb, gt_depth,
                                                                        cov, lam, loss_scale)
    # --- independent forward, float64, autograd
    raw = torch.tensor(rgb_raw[:, :3].astype(np.float64), requires_grad=True)
    dens = torch.tensor(dens_raw[:, 0].astype(np.float64), requires_grad=True)
    tdt, tt = torch.tensor(dt.astype(np.float64)), torch.tensor(tmid.astype(np.float64))
    total = torch.zeros((), dtype=torch.float64)
    cols, deps = [], []
    for r in range(R):
        sl = slice(int(start[r]), int(start[r] + n[r]))
        sigma = torch.exp(dens[sl])
        alpha = 1.0 - torch.exp(-sigma * tdt[sl])
        trans = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1.0 - alpha]), 0)[:-1]
        wgt = alpha * trans
        col = (wgt[:, None] * torch.sigmoid(raw[sl])).sum(0)
        dep = (wgt * tt[sl]).sum()
        ell = ((col - torch.tensor(gt_rgb[r].astype(np.float64))) ** 2).sum() / 3.0
        if gt_depth[r] > 0:
            ell = ell + lam * (dep - float(gt_depth[r])) ** 2 / float(cov[r])
        total = total + ell
        cols.append(col.detach().numpy())
        deps.append(float(dep.detach()))
    loss = total / R
    (loss * loss_scale).backward()
    loss = float(loss.detach())
    assert abs(o_loss - loss) <= 2e-6 * abs(loss)
    assert np.abs(o_rgb - np.stack(cols)).max() <= 2e-6 and np.abs(o_depth - np.array(deps)).max() <= 2e-6
    g_rgb, g_dens = raw.grad.numpy(), dens.grad.numpy()
    for name, got, want in (("dL/drgb", dLdrgb[:, :3].astype(np.float64), g_rgb), ("dL/ddensity", dLddens[:, 0].astype(np.float64), g_dens)):
        err = np.abs(got - want)
        assert (err <= 1.5 * F16_EPS * np.abs(want) + 2.0 ** -24 + 1e-5 * np.abs(want).max()).all(), (name, float(err.max()), float(np.abs(want).max()))
    assert (dLdrgb[:, 3:] == 0).all() and (dLddens[:, 1:] == 0).all()
    assert np.abs(g_dens).max() > 1e-3 and np.abs(g_rgb).max() > 1e-3        # not a vacuous comparison


def _ste_round_f16(x):
    """value: x rounded to f16; gradient: identity (what a fully-fused f16 network's backward differentiates)"""
    return x + (x.detach().to(torch.float16).to(torch.float64) - x.detach())


def _sh16(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
        0.54627421529603959 * (x2 - y2), 0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], 1)


def test_mlp_backward_equals_autograd():
    """density net 32->64->16 and colour net (16 + SH16)->64->64->16: dL/dfeatures and the five weight gradients against
    float64 autograd of the same network with f16-rounded activations (straight-through).  The oracle additionally rounds
    the activation gradients to f16 between layers (as the fused f16 kernels do): a few 2^-11 per layer."""
    rng = np.random.default_rng(11)
    N = 96
    Ws = [(rng.normal(0, 1.0 / np.sqrt(s[1]), s) * 1.4).astype(np.float16) for s in oracle.MLP_SHAPES]
    feat = rng.normal(0, 0.5, (N, 32)).astype(np.float16)
    dirs = rng.normal(0, 1, (N, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    act = oracle.ngp_mlp_fwd(Ws, feat, dirs)
    dLdrgb = np.zeros((N, 16), np.float16)
    dLddens = np.zeros((N, 16), np.float16)
    dLdrgb[:, :3] = rng.normal(0, 1, (N, 3))
    dLddens[:, 0] = rng.normal(0, 1, N)
    dfeat, dW = oracle.ngp_mlp_bwd(Ws, feat, act, dLdrgb, dLddens)
    # --- independent forward
    tW = [torch.tensor(w.astype(np.float64), requires_grad=True) for w in Ws]
    x = torch.tensor(feat.astype(np.float64), requires_grad=True)
    h1 = _ste_round_f16(torch.relu(x @ tW[0].T))
    dens = _ste_round_f16(h1 @ tW[1].T)
    sh = _sh16(torch.tensor(dirs.astype(np.float64))).to(torch.float16).to(torch.float64)
    cin = torch.cat([dens, sh], 1)
    h3 = _ste_round_f16(torch.relu(cin @ tW[2].T))
    h4 = _ste_round_f16(torch.relu(h3 @ tW[3].T))
    rgb = _ste_round_f16(h4 @ tW[4].T)
    for name, a, b in (("h1", act["h1"], h1), ("dens", act["dens"], dens), ("h3", act["h3"], h3), ("h4", act["h4"], h4), ("rgb", act["rgb"], rgb)):
        # f32 fmaf chain vs float64 sum before the same f16 rounding: equal except where a sum sits on a rounding boundary;
        # one such ulp upstream moves the sums of the later layers by a few ulps of THEIR inputs
        d = np.abs(a.astype(np.float64) - b.detach().numpy())
        print(name, "max forward difference / max|activation|", d.max() / np.abs(b.detach().numpy()).max())
        assert d.max() <= 4 * F16_EPS * np.abs(b.detach().numpy()).max(), name
    obj = (rgb * torch.tensor(dLdrgb.astype(np.float64))).sum() + (dens * torch.tensor(dLddens.astype(np.float64))).sum()
    obj.backward()
    want = x.grad.numpy()
    err = np.abs(dfeat.astype(np.float64) - want).max()
    print("dfeat", err / np.abs(want).max())
    assert err <= 2 * F16_EPS * np.abs(want).max(), (err, np.abs(want).max())      # measured 4.2e-4 of max
    for k in range(5):
        w = tW[k].grad.numpy()
        e = np.abs(dW[k] - w).max()
        print("dW", k, e / np.abs(w).max())
        assert e <= 2 * F16_EPS * np.abs(w).max(), (k, e, np.abs(w).max())               # measured <= 4.1e-4 of max
        assert np.abs(w).max() > 1e-2


def _torch_encode(cfg, pos, params):
    """multiresolution hash encoding written from the paper (Mueller et al. 2022, eq. 2-4 and tiny-cuda-nn's published scale /
    resolution / index rules), float64, differentiable in `params` and `pos`"""
    scale, res, off = oracle.ngp_grid_layout(cfg)
    outs = []
    primes = (1, 2654435761, 805459861)
    for l in range(cfg.n_levels):
        hs = int(off[l + 1]) - int(off[l])
        p = pos * float(scale[l]) + 0.5
        # the published encoding positions a sample in f32 (one fused multiply-add): on the finest levels (scale ~ 8000) an f32
        # ulp of p is 1e-3 of a cell.  Same value here -- the f64 expression rounded once -- with the f64 derivative.
        p = p + (p.detach().to(torch.float32).to(torch.float64) - p.detach())
        g = torch.floor(p.detach()).to(torch.int64)
        w = p - g.to(torch.float64)
        acc = torch.zeros((pos.shape[0], 2), dtype=torch.float64)
        r = int(res[l])
        dense = r ** 3 <= hs or (r ** 3 + 7) // 8 * 8 <= hs
        for corner in range(8):
            q = [g[:, d] + ((corner >> d) & 1) for d in range(3)]
            wt = torch.ones(pos.shape[0], dtype=torch.float64)
            for d in range(3):
                wt = wt * (w[:, d] if (corner >> d) & 1 else 1.0 - w[:, d])
            if dense:
                idx = (q[0] + q[1] * r + q[2] * r * r) % hs
            else:
                idx = (((q[0] * primes[0]) & 0xFFFFFFFF) ^ ((q[1] * primes[1]) & 0xFFFFFFFF) ^ ((q[2] * primes[2]) & 0xFFFFFFFF)) % hs
            acc = acc + wt[:, None] * params[(int(off[l]) + idx)]
        outs.append(acc)
    return torch.cat(outs, 1)


@pytest.mark.parametrize("grid", [dict(n_levels=6, log2_hashmap=12, base_res=4, per_level_scale=1.7),
                                  dict(n_levels=16, log2_hashmap=19, base_res=16, per_level_scale=1.5157165665)],
                         ids=["small_grid", "default_grid"])
def test_hash_encoding_forward_and_both_gradients_equal_autograd(grid):
    rng = np.random.default_rng(13)
    cfg = oracle.ngp_cfg(**grid)
    _, _, off = oracle.ngp_grid_layout(cfg)
    n_entries = int(off[-1])
    N = 200
    pos = rng.uniform(0.02, 0.98, (N, 3)).astype(np.float32)
    params = rng.uniform(-0.5, 0.5, (n_entries, 2)).astype(np.float16)
    dLdout = rng.normal(0, 1, (N, cfg.n_levels * 2)).astype(np.float16)
    fwd = oracle.ngp_encode_fwd(cfg, pos, params.reshape(-1))
    g_tab = oracle.ngp_encode_bwd(cfg, pos, dLdout, n_entries * 2).reshape(n_entries, 2)
    g_pos = oracle.ngp_encode_bwd_input(cfg, pos, params.reshape(-1), dLdout)
    tp = torch.tensor(params.astype(np.float64), requires_grad=True)
    tx = torch.tensor(pos.astype(np.float64), requires_grad=True)
    out = _torch_encode(cfg, tx, tp)
    ref = out.detach().numpy()
    assert np.abs(fwd.astype(np.float64) - ref).max() <= 2 * F16_EPS * np.abs(ref).max() + 1e-6      # one f16 rounding of the output (+ f32 position arithmetic)
    (out * torch.tensor(dLdout.astype(np.float64))).sum().backward()
    wt, wp = tp.grad.numpy(), tx.grad.numpy()
    assert np.abs(g_tab - wt).max() <= 2e-5 * np.abs(wt).max(), np.abs(g_tab - wt).max()      # f32 sums, f32 cell coordinates
    assert np.abs(g_pos - wp).max() <= 2e-4 * np.abs(wp).max(), (np.abs(g_pos - wp).max(), np.abs(wp).max())
    assert (wt != 0).sum() > 4 * N


def test_adam_equals_torch_optim_adam():
    """tiny-cuda-nn's Adam (beta 0.9 / 0.99, eps 1e-15, no weight decay on the table; L2 on the MLP weights) == torch.optim.Adam
    with the same constants, over several steps, for touched entries; untouched entries (zero gradient) keep their moments."""
    rng = np.random.default_rng(17)
    n = 512
    p0 = rng.normal(0, 0.3, n).astype(np.float32)
    for l2 in (0.0, 1e-6):
        master, m1, m2 = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        tp = torch.tensor(p0.astype(np.float64), requires_grad=True)
        opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=l2)
        for step in range(1, 6):
            g = (rng.normal(0, 1, n) * 128.0).astype(np.float32)          # loss-scaled gradient, as the trainer passes it
            master, hp, m1, m2 = oracle.ngp_adam(master, g, m1, m2, step, 1e-2, l2=l2, grad_scale=128.0)
            tp.grad = torch.tensor(g.astype(np.float64) / 128.0)
            opt.step()
            assert np.abs(master - tp.detach().numpy()).max() <= 5e-6, step
            assert (hp == master.astype(np.float16)).all()
    # an entry whose gradient is zero is left alone (hash entries no sample touched): moments AND parameter
    master, m1, m2 = p0.copy(), np.full(n, 0.25, np.float32), np.full(n, 0.5, np.float32)
    g = np.zeros(n, np.float32)
    g[::2] = 3.0
    a, _, b, c = oracle.ngp_adam(master, g, m1, m2, 3, 1e-2)
    assert (a[1::2] == p0[1::2]).all() and (b[1::2] == 0.25).all() and (c[1::2] == 0.5).all() and (a[::2] != p0[::2]).all()
