"""HipUpdateOperator (MFMA convolutions, channels-last) against the torch UpdateModule it is built from
(networks/droid_net.py:78-150 restated in nerfslam/droid_nets.py, itself pinned to the reference modules by
tests/golden/droid_nets_forward.npz): same weights, same inputs, f16 autocast on the torch side."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("E,ht,wd", [(12, 60, 80), (5, 43, 77)])
def test_update_operator_matches_torch_module(dev, E, ht, wd):
    from nerfslam.droid_nets import UpdateModule
    from nerfslam.update_op import HipUpdateOperator
    torch.manual_seed(0)
    um = UpdateModule().to(dev).eval()
    op = HipUpdateOperator(um)
    g = torch.Generator().manual_seed(1)
    net = torch.tanh(torch.randn((E, 128, ht, wd), generator=g)).to(dev)
    inp = torch.relu(torch.randn((E, 128, ht, wd), generator=g)).to(dev)
    corr = (torch.randn((E, 196, ht, wd), generator=g) * 2).half().to(dev)
    flow = (torch.randn((E, 4, ht, wd), generator=g) * 4).to(dev)
    ii = torch.tensor([3, 3, 4, 9, 4, 3, 7, 7, 9, 9, 9, 4][:E], device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        h_ref, d_ref, w_ref, eta_ref, up_ref = um(net[None].half(), inp[None].half(), corr[None], flow[None], ii, ii)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().half()
    h, d, w, eta, up = op(cl(net), cl(inp), corr, flow, ii.tolist())
    assert h.shape == (E, ht, wd, 128) and d.shape == (E, ht, wd, 2) and up.shape == (up_ref.shape[1], ht, wd, 576)
    # both sides round to f16 between layers, at different points: agreement to a few 1e-3 of the signal
    assert _rel(h.permute(0, 3, 1, 2), h_ref[0]) < 4e-3
    assert _rel(d, d_ref[0]) < 1e-2 and (d - d_ref[0].float()).abs().max().item() < 2e-2 * d_ref.abs().max().item() + 1e-3
    assert _rel(w, w_ref[0]) < 4e-3
    assert _rel(eta, eta_ref[0]) < 1e-2
    assert _rel(up.permute(0, 3, 1, 2), up_ref[0]) < 1e-2


def test_droid_networks_adapter_uses_the_hip_operator(dev):
    """DroidNetworks.update through the HIP operator == through torch, incl. the hidden-state carry-over between calls"""
    from nerfslam.droid_nets import DroidNetworks
    ht, wd = 24, 32
    a = DroidNetworks(dev, seed=3, hip_update=True)
    b = DroidNetworks(dev, seed=3, hip_update=False)
    assert a.update_op is not None and b.update_op is None
    g = torch.Generator().manual_seed(0)
    for k in range(3):
        img = torch.randint(0, 255, (3, 8 * ht, 8 * wd), generator=g, dtype=torch.uint8)
        for n in (a, b):
            n.features(img); n.begin_keyframe(k, img)
    ii = torch.tensor([0, 1, 1, 2], device=dev); jj = torch.tensor([1, 0, 2, 1], device=dev)
    for it in range(2):
        corr = torch.randn((1, 4, 196, ht, wd), generator=g).half().to(dev)
        motion = torch.randn((4, 4, ht, wd), generator=g).to(dev)
        ra, rb = a.update(corr, motion, ii, jj), b.update(corr, motion, ii, jj)
        ra = ra[:3] + (ra[3].permute(0, 3, 1, 2),)      # the HIP operator's mask is channels-last
        for x, y, tol in zip(ra, rb, (2e-2, 1e-2, 2e-2, 2e-2)):
            assert x.shape == y.shape and _rel(x, y) < tol, (it, x.shape, _rel(x, y))


def test_motion_filter_delta_only_equals_full_operator(dev):
    """DroidNetworks.motion (one edge, zero motion features, delta head only) == the delta of the full operator"""
    from nerfslam.droid_nets import DroidNetworks
    ht, wd = 24, 32
    n = DroidNetworks(dev, seed=1, hip_update=True)
    g = torch.Generator().manual_seed(0)
    for k in range(2):
        img = torch.randint(0, 255, (3, 8 * ht, 8 * wd), generator=g, dtype=torch.uint8)
        n.features(img); n.begin_keyframe(k, img)
    for it in range(3):
        corr = torch.randn((1, 1, 196, ht, wd), generator=g).half().to(dev)
        kf = it % 2
        d = n.motion(corr, kf)
        _, ref, _, _, _ = n.update_op(n.ctx_cl[kf][None], n.inp_cl[kf][None], corr[0], torch.zeros((1, 4, ht, wd), device=dev), [0])
        assert d.shape == (1, 1, ht, wd, 2) and torch.equal(d[0], ref)       # same weights, same arithmetic: bit-identical


@pytest.mark.parametrize("E,ht,wd", [(48, 60, 80), (3, 43, 77), (1, 12, 16)])
def test_gru_global_context_bias(dev, E, ht, wd):
    """ns_gru_glo_bias (csrc/conv.hip) = mean_p(sigmoid(w(net)) * net) @ [convz_glo | convr_glo | convq_glo] + biases
    (networks/modules/gru.py:25-33) against a float64 evaluation on the same f16 inputs"""
    from nerfslam.conv import gru_glo_bias
    g = torch.Generator().manual_seed(E + wd)
    wg = torch.sigmoid(torch.randn((E, ht, wd, 128), generator=g)).half().to(dev)
    net = torch.tanh(torch.randn((E, ht, wd, 128), generator=g) + 0.2).half().to(dev)
    W = (torch.randn((128, 384), generator=g) / 11.0).to(dev)
    b = torch.randn(384, generator=g).to(dev)
    got = gru_glo_bias(wg, net, W, b)
    ref = (wg.double() * net.double()).mean((1, 2)) @ W.double() + b.double()
    assert got.shape == (E, 384) and got.dtype == torch.float32
    assert (got.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_hidden_state_cache_across_changing_edge_lists(dev):
    """DroidNetworks.update keeps the stacked hidden / context tensors of an unchanged edge list (nerfslam/droid_nets.py): the same
    call sequence -- same list twice, a different list, a keyframe removed, the first list again -- through the HIP operator with
    the cache and through the torch module without it must agree, i.e. a stale stack is never reused."""
    from nerfslam.droid_nets import DroidNetworks
    ht, wd = 24, 32
    a = DroidNetworks(dev, seed=5, hip_update=True)
    b = DroidNetworks(dev, seed=5, hip_update=False)
    g = torch.Generator().manual_seed(2)
    for k in range(4):
        img = torch.randint(0, 255, (3, 8 * ht, 8 * wd), generator=g, dtype=torch.uint8)
        for n in (a, b):
            n.features(img); n.begin_keyframe(k, img)
    L1 = ([0, 1, 1, 2, 3], [1, 0, 2, 1, 2])
    L2 = ([1, 2, 3, 0], [2, 1, 2, 1])            # overlaps L1 (edges (1,2), (2,1), (3,2), (0,1)) in another order
    seq = [L1, L1, L2, L1, "rm", L1, L1]
    hits = 0
    for it, item in enumerate(seq):
        if item == "rm":
            for n in (a, b):
                n.remove_keyframe(2)             # (the hook's contract: keyframe 3 slides onto 2) every cached stack is stale
            L1 = ([0, 1, 1, 2, 2], [1, 0, 2, 1, 0])
            seq[it + 1:] = [L1, L1]
            continue
        ii, jj = torch.tensor(item[0], device=dev), torch.tensor(item[1], device=dev)
        E = len(item[0])
        corr = torch.randn((1, E, 196, ht, wd), generator=g).half().to(dev)
        motion = torch.randn((E, 4, ht, wd), generator=g).to(dev)
        before = a._stacked
        ra, rb = a.update(corr, motion, ii, jj), b.update(corr, motion, ii, jj)
        hits += int(before is not None and before[0] == a._stacked[0])
        ra = ra[:3] + (ra[3].permute(0, 3, 1, 2),)
        # (f16 rounding differences of the two operators travel through the chained hidden states: looser than the two-call test
        #  above; a stale stack shows up as an O(1) error)
        for x, y, tol in zip(ra, rb, (5e-2, 3e-2, 5e-2, 5e-2)):
            assert x.shape == y.shape and _rel(x, y) < tol, (it, x.shape, _rel(x, y))
    assert hits == 2                              # L1 -> L1 before and after the removal; every other call re-stacks


def test_hidden_state_written_between_updates_is_not_served_from_the_stack(dev):
    """the stacked hidden / context tensors of an unchanged edge list are a cache; a state written between two
    updates (DroidNetworks.set_hidden / set_context -- the one place such writes go through) must reach the next update, and a
    list that names an edge twice must behave like the re-stack path.  Reference: the same sequence with the cache dropped before
    every call (what rounds 1-4 did), bit for bit."""
    from nerfslam.droid_nets import DroidNetworks
    ht, wd = 24, 32
    a = DroidNetworks(dev, seed=7, hip_update=True)
    b = DroidNetworks(dev, seed=7, hip_update=True)
    g = torch.Generator().manual_seed(3)
    for k in range(3):
        img = torch.randint(0, 255, (3, 8 * ht, 8 * wd), generator=g, dtype=torch.uint8)
        for n in (a, b):
            n.features(img); n.begin_keyframe(k, img)
    lists = [([0, 1, 1, 2], [1, 0, 2, 1]), ([0, 1, 1, 2], [1, 0, 2, 1]), ([0, 1, 1, 2], [1, 0, 2, 1]), ([0, 1, 0, 2], [1, 0, 1, 1]),
             ([0, 1, 0, 2], [1, 0, 1, 1])]
    for it, (il, jl) in enumerate(lists):
        ii, jj = torch.tensor(il, device=dev), torch.tensor(jl, device=dev)
        corr = torch.randn((1, len(il), 196, ht, wd), generator=g).half().to(dev)
        motion = torch.randn((len(il), 4, ht, wd), generator=g).to(dev)
        if it == 1:                              # a hidden state replaced from outside between two updates of the SAME list
            h = torch.tanh(torch.randn((ht, wd, 128), generator=g)).half().to(dev)
            a.set_hidden(1, 2, h); b.set_hidden(1, 2, h.clone())
        if it == 2:                              # ... and a keyframe's context features
            c = torch.tanh(torch.randn((128, ht, wd), generator=g)).to(dev)
            r = torch.relu(torch.randn((128, ht, wd), generator=g)).to(dev)
            a.set_context(1, c, r); b.set_context(1, c.clone(), r.clone())
        b._stacked = None                        # the reference arm re-stacks on every call
        ra, rb = a.update(corr, motion, ii, jj), b.update(corr, motion, ii, jj)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), it
        assert all(torch.equal(a.hidden[e], b.hidden[e]) for e in a.hidden)
