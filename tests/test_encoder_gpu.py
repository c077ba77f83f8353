"""HipEncoder (feature / context encoders on the MFMA convolution, nerfslam/encoder_op.py + csrc/encoder.hip) against
 (a) the outputs of the REFERENCE's own BasicEncoder modules (tests/golden/droid_nets_forward.npz, written by
     tools/gen_golden.py section 4 from /root/reference/networks/modules/extractor.py:118-198), and
 (b) the torch modules of nerfslam/droid_nets.py (themselves pinned to that fixture on the CPU) in f32 at 640x480 and at an
     odd size.
Both sides of (b) see the same weights and the same image; ours rounds activations to f16 between layers (like the f16 MIOpen
path it replaces), so agreement is to a few 1e-3 of the signal, stated per assertion."""
import os
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _seeded(net):
    sd = net.state_dict()
    for k in sd:
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
        sd[k] = torch.randn(sd[k].shape, generator=g) * (0.3 / max(1.0, float(np.sqrt(sd[k][0].numel()))))
    net.load_state_dict(sd)
    return net.eval()


def test_encoders_match_the_reference_modules_outputs(dev):
    from nerfslam.droid_nets import DroidNet
    from nerfslam.encoder_op import HipEncoder
    z = np.load(os.path.join(G, "droid_nets_forward.npz"))
    net = _seeded(DroidNet()).to(dev)
    img = torch.from_numpy(z["img"]).to(dev)[0]                     # [2,3,32,48], already normalised: mean 0, std 1/255 pass it through
    for enc, norm, name in ((net.feature_net, True, "fmap"), (net.context_net, False, "cmap")):
        op = HipEncoder(enc, norm, (0.0, 0.0, 0.0), (1 / 255.0,) * 3, use_graph=False)
        got = op(img.float().contiguous()).permute(0, 3, 1, 2).float().cpu().numpy()
        ref = z[name][0]
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max(), (name, np.abs(got - ref).max(), np.abs(ref).max())
        assert np.linalg.norm(got - ref) <= 4e-3 * np.linalg.norm(ref), name


@pytest.mark.parametrize("H,W,N", [(480, 640, 1), (47, 61, 2), (96, 128, 3)])
def test_encoders_match_torch_f32_modules(dev, H, W, N):
    from nerfslam.droid_nets import DroidNet
    from nerfslam.encoder_op import HipEncoder
    torch.manual_seed(3)
    net = DroidNet().to(dev).eval()
    g = torch.Generator().manual_seed(H)
    img = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8).to(dev)
    # smooth it a little: white noise through a stride-8 encoder is all cancellation
    img = torch.nn.functional.avg_pool2d(img.float(), 5, stride=1, padding=2).round().clamp(0, 255).to(torch.uint8)
    m = torch.tensor(MEAN, device=dev)[:, None, None]
    s = torch.tensor(STD, device=dev)[:, None, None]
    x = ((img.float() / 255.0 - m) / s)[None]
    for enc, norm in ((net.feature_net, True), (net.context_net, False)):
        with torch.no_grad():
            ref = enc(x)[0]
        op = HipEncoder(enc, norm, MEAN, STD, use_graph=False)
        got = op(img).permute(0, 3, 1, 2)
        assert got.shape == ref.shape and got.dtype == torch.float16
        assert _rel(got, ref) < 4e-3, (norm, _rel(got, ref))
        assert (got.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
        got_f = op(img.float())                                      # float image holding 0..255: same bits
        assert torch.equal(got_f.permute(0, 3, 1, 2), got)


def test_graph_replay_is_the_eager_result(dev):
    from nerfslam.droid_nets import DroidNet
    from nerfslam.encoder_op import HipEncoder
    torch.manual_seed(4)
    net = DroidNet().to(dev).eval()
    g = torch.Generator().manual_seed(9)
    imgs = [torch.randint(0, 256, (1, 3, 120, 160), generator=g, dtype=torch.uint8).to(dev) for _ in range(3)]
    eager = HipEncoder(net.feature_net, True, MEAN, STD, use_graph=False)
    graph = HipEncoder(net.feature_net, True, MEAN, STD, use_graph=True)
    for im in imgs + imgs[:1]:
        a, b = eager(im), graph(im)
        assert torch.equal(a, b)
    assert len(graph._graphs) == 1
    other = torch.randint(0, 256, (1, 3, 64, 96), generator=g, dtype=torch.uint8).to(dev)      # a second shape: its own graph
    assert torch.equal(eager(other), graph(other)) and len(graph._graphs) == 2


def test_instance_norm_pieces_against_torch(dev):
    """ns_enc_in_stats + ns_enc_in_apply alone: relu(IN(y)), relu(x + relu(IN(y))), relu(IN(d) + relu(IN(y))), relu(x + relu(y))"""
    from nerfslam._lib import check, lib, ptr, stream_ptr
    import ctypes as C
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for (N, H, W, Cc) in ((2, 37, 53, 32), (1, 60, 80, 128), (3, 9, 7, 64)):
        y = (torch.randn((N, H, W, Cc), generator=g) * 1.7 + 0.4).half().to(dev)
        x = torch.randn((N, H, W, Cc), generator=g).half().to(dev)
        P = int(lib().ns_enc_in_parts(H * W))
        tickets = torch.zeros((N,), dtype=torch.int32, device=dev)       # caller-owned arrival counters: zero before, zero after
        def stats(t):
            p = torch.empty((N, P, 2, Cc), dtype=torch.float32, device=dev)
            check(lib().ns_enc_in_stats(ptr(t), ptr(p), ptr(tickets), N, H * W, Cc, stream_ptr()), "stats")
            assert int(tickets.abs().sum()) == 0
            return p
        def apply(y, ys, x, xs):
            o = torch.empty_like(y)
            check(lib().ns_enc_in_apply(ptr(y), ptr(ys), ptr(x), ptr(xs), ptr(o), N, H * W, Cc, C.c_float(1e-5), stream_ptr()), "apply")
            return o.float()
        ys, xs = stats(y), stats(x)
        ref_s = y.float().sum((1, 2))
        # (row 0 of an image holds the TOTALS: the last workgroup to arrive adds the partial rows up)
        assert torch.allclose(ys[:, 0, 0], ref_s, rtol=1e-4, atol=1e-2)
        assert torch.allclose(ys[:, 0, 1], y.float().pow(2).sum((1, 2)), rtol=1e-4, atol=1e-2)
        assert torch.equal(stats(y)[:, 0], ys[:, 0])          # fixed summation order: whichever workgroup arrives last
        inn = lambda t: F.instance_norm(t.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        r1 = F.relu(inn(y))
        for got, ref in ((apply(y, ys, None, None), r1), (apply(y, ys, x, None), F.relu(x.float() + r1)),
                         (apply(y, ys, x, xs), F.relu(inn(x) + r1)), (apply(y, None, x, None), F.relu(x.float() + F.relu(y.float())))):
            assert (got - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


def test_bad_arguments_fail_loudly(dev):
    from nerfslam._lib import NerfSlamHipError
    from nerfslam.droid_nets import DroidNet
    from nerfslam.encoder_op import HipEncoder
    op = HipEncoder(DroidNet().to(dev).eval().feature_net, True, MEAN, STD, use_graph=False)
    with pytest.raises(NerfSlamHipError):
        op(torch.zeros((1, 4, 32, 32), dtype=torch.uint8, device=dev))
    with pytest.raises(NerfSlamHipError):
        op(torch.zeros((1, 3, 32, 32), dtype=torch.float16, device=dev))
