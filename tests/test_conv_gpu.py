"""ns_conv_nhwc_f16 (csrc/conv.hip) against torch's fp32 conv2d on the same f16-rounded inputs: the convolutions of the
tracker's update operator (networks/droid_net.py:78-150, networks/modules/gru.py:5-34) at its shapes and at awkward ones."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {None: lambda x: x, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}


def _case(dev, N, H, W, chans, cout, k, act, per_image_bias=False, out_stride=None, out_offset=0, seed=0):
    from nerfslam.conv import PackedConv, conv_nhwc
    g = torch.Generator().manual_seed(seed)
    cin = sum(chans)
    srcs = [torch.randn((N, H, W, c), generator=g).half().to(dev) for c in chans]
    w = (torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)).half().float().to(dev)
    b = torch.randn((cout,), generator=g).to(dev)
    layer = PackedConv(w, b)
    bias = torch.randn((N, cout), generator=g).to(dev) if per_image_bias else None
    out = None
    if out_stride is not None:
        out = torch.full((N, H, W, out_stride), 7.0, dtype=torch.float16, device=dev)
    res = conv_nhwc(srcs, layer, act=act, out=out, out_offset=out_offset, bias=bias)
    x = torch.cat(srcs, -1).float().permute(0, 3, 1, 2)
    ref = F.conv2d(x, w, None, padding=k // 2)
    ref = ref + (bias[:, :, None, None] if per_image_bias else b[None, :, None, None])
    ref = ACTS[act](ref).permute(0, 2, 3, 1)
    got = res[..., out_offset:out_offset + cout].float()
    tol = 2.5e-3 * max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= tol, (N, H, W, chans, cout, k, act, err, tol)
    if out_stride is not None:   # the rest of the output row is untouched
        mask = torch.ones(out_stride, dtype=torch.bool, device=dev)
        mask[out_offset:out_offset + cout] = False
        assert (res[..., mask] == 7.0).all()
    return err


@pytest.fixture(params=[2, 4])
def ut(request):
    """both pixel-tile variants of the 128-cout kernels (16- and 32-row tiles; NS_CONV_UT is read at every launch)"""
    os.environ["NS_VARIANTS"] = "1"            # A/B switches count only together with the master switch (csrc/common.h)
    os.environ["NS_CONV_UT"] = str(request.param)
    yield request.param
    os.environ.pop("NS_CONV_UT", None)
    os.environ.pop("NS_VARIANTS", None)


@pytest.fixture(params=[None, 1, 2])
def mt(request):
    """the cout tile: None = the launch heuristic (128-cout tiles unless the launch would have < 200 workgroups), 1 | 2 = forced
    32- / 64-cout tiles (what a single-edge launch gets)"""
    if request.param is not None:
        os.environ["NS_VARIANTS"] = "1"
        os.environ["NS_CONV_MT"] = str(request.param)
    yield request.param
    os.environ.pop("NS_CONV_MT", None)
    os.environ.pop("NS_VARIANTS", None)


def test_single_image_launches(dev, mt):
    # the motion filter's one edge and the encoders' N = 1: small tiles by the heuristic, same arithmetic
    _case(dev, 1, 60, 80, (128, 128, 128, 64), 256, 3, "sigmoid", per_image_bias=True)
    _case(dev, 1, 60, 80, (128,), 128, 3, "relu", seed=1)
    _case(dev, 1, 120, 160, (64,), 64, 3, None, seed=2)
    _case(dev, 1, 60, 80, (576,), 128, 1, None, seed=3)
    _case(dev, 1, 60, 80, (128,), 576, 1, None, seed=4)


def test_gru_gate_shapes(dev, ut, mt):
    # convz|convr fused: [h, inp, corr, flow] -> 256, sigmoid, per-edge global-context bias (gru.py:28-29)
    _case(dev, 3, 60, 80, (128, 128, 128, 64), 256, 3, "sigmoid", per_image_bias=True)
    # convq: -> 128, tanh
    _case(dev, 2, 60, 80, (128, 128, 128, 64), 128, 3, "tanh", per_image_bias=True, seed=1)


def test_encoders_heads_and_slices(dev, ut):
    # corr encoder: 1x1 over 196 channels padded to 208, relu; then 3x3 into a slice of the GRU input buffer
    _case(dev, 2, 60, 80, (208,), 128, 1, "relu")
    _case(dev, 2, 60, 80, (128,), 128, 3, "relu", out_stride=448, out_offset=256, seed=2)
    # flow encoder's second conv (-> 64), heads (-> 2, ragged cout), eta (-> 1), upmask (1x1 -> 576)
    _case(dev, 2, 60, 80, (128,), 64, 3, "relu", out_stride=448, out_offset=384, seed=3)
    _case(dev, 2, 60, 80, (128,), 2, 3, None, seed=4)
    _case(dev, 2, 60, 80, (128,), 1, 3, None, out_stride=4, out_offset=0, seed=5)
    _case(dev, 2, 60, 80, (128,), 576, 1, None, seed=6)
    _case(dev, 1, 60, 80, (128,), 384, 3, "relu", seed=7)


def test_odd_and_tiny_images(dev, ut):
    _case(dev, 2, 43, 77, (32, 16), 40, 3, "relu")              # the real Replica grid, ragged everything
    _case(dev, 1, 5, 3, (16,), 6, 3, None, seed=1)
    _case(dev, 2, 9, 11, (16,), 7, 3, "relu", out_stride=13, out_offset=3, seed=5)   # unaligned slice: scalar stores
    _case(dev, 1, 1, 1, (16,), 130, 3, "tanh", seed=2)           # only the centre tap sees data
    _case(dev, 3, 17, 33, (48,), 128, 1, "sigmoid", per_image_bias=True, seed=3)
    _case(dev, 1, 33, 16, (16, 16, 16, 16), 96, 3, None, seed=4)


def test_channel_slices_as_sources(dev):
    """a source may be a channel slice of a wider channels-last tensor (the heads read slices of one fused 384-channel conv)"""
    from nerfslam.conv import PackedConv, conv_nhwc
    g = torch.Generator().manual_seed(0)
    wide = torch.randn((2, 21, 37, 384), generator=g).half().to(dev)
    other = torch.randn((2, 21, 37, 32), generator=g).half().to(dev)
    w = (torch.randn((48, 160, 3, 3), generator=g) / 38.0).half().float().to(dev)
    b = torch.randn((48,), generator=g).to(dev)
    got = conv_nhwc([wide[..., 128:256], other], PackedConv(w, b), act="relu").float()
    x = torch.cat([wide[..., 128:256], other], -1).float().permute(0, 3, 1, 2)
    ref = torch.relu(F.conv2d(x, w, b, padding=1)).permute(0, 2, 3, 1)
    assert (got - ref).abs().max().item() <= 2.5e-3 * max(1.0, ref.abs().max().item())
    with pytest.raises(RuntimeError):
        conv_nhwc([wide.permute(0, 2, 1, 3)[..., :160]], PackedConv(w, b))      # not channels-last


def test_fused_gru_epilogues(dev, ut, mt):
    """the two ConvGRU steps folded into the epilogue == conv followed by the torch elementwise ops (gru.py:28-33)"""
    from nerfslam.conv import PackedConv, conv_nhwc
    g = torch.Generator().manual_seed(0)
    N, H, W = 2, 23, 37
    h = torch.tanh(torch.randn((N, H, W, 128), generator=g)).half().to(dev)
    x = torch.randn((N, H, W, 64), generator=g).half().to(dev)
    wzr = (torch.randn((256, 192, 3, 3), generator=g) / 41.0).to(dev)
    wq = (torch.randn((128, 192, 3, 3), generator=g) / 41.0).to(dev)
    bzr, bq = torch.randn((N, 256), generator=g).to(dev), torch.randn((N, 128), generator=g).to(dev)
    Lzr, Lq = PackedConv(wzr), PackedConv(wq)
    zr = conv_nhwc([h, x], Lzr, act="sigmoid", bias=bzr)
    zrh = conv_nhwc([h, x], Lzr, act="sigmoid", bias=bzr, fuse=("mul_hi", h))
    assert torch.equal(zrh[..., :128], zr[..., :128])
    assert (zrh[..., 128:].float() - zr[..., 128:].float() * h.float()).abs().max().item() <= 1e-3
    q = conv_nhwc([zrh[..., 128:], x], Lq, act="tanh", bias=bq)
    h2 = conv_nhwc([zrh[..., 128:], x], Lq, act="tanh", bias=bq, fuse=("gru", zrh[..., :128], h))
    ref = h.float() + zrh[..., :128].float() * (q.float() - h.float())
    assert (h2.float() - ref).abs().max().item() <= 2e-3
    with pytest.raises(RuntimeError):
        conv_nhwc([h, x], PackedConv(wzr[:100]), fuse=("mul_hi", h))          # ragged cout tile: not on the fused path


def test_flow_im2col_makes_the_7x7_conv_a_1x1(dev):
    from nerfslam.conv import PackedConv, conv_nhwc, flow_im2col
    g = torch.Generator().manual_seed(0)
    for (E, ht, wd) in ((3, 60, 80), (2, 5, 3), (1, 1, 1)):
        flow = (torch.randn((E, 4, ht, wd), generator=g) * 3).to(dev)
        col = flow_im2col(flow)
        ref = F.unfold(flow.half().float(), 7, padding=3).reshape(E, 196, ht, wd).permute(0, 2, 3, 1)
        assert torch.equal(col[..., :196].float(), ref) and (col[..., 196:] == 0).all()
        w = (torch.randn((128, 4, 7, 7), generator=g) / 14.0).half().float().to(dev)
        b = torch.randn((128,), generator=g).to(dev)
        got = conv_nhwc([col], PackedConv(w.reshape(128, 196, 1, 1), b, pad_cin_to=208), act="relu").float()
        want = torch.relu(F.conv2d(flow.half().float(), w, b, padding=3)).permute(0, 2, 3, 1)
        assert (got - want).abs().max().item() <= 2.5e-3 * max(1.0, want.abs().max().item())


def test_layout_glue_kernels(dev):
    """planes -> channels-last with zero pad channels (bit-exact), and GraphAgg's scatter-mean"""
    from nerfslam.conv import group_mean, planes_to_nhwc
    g = torch.Generator().manual_seed(0)
    for (E, Cc, ht, wd) in ((3, 196, 60, 80), (2, 196, 43, 77), (1, 5, 3, 3)):
        x = torch.randn((E, Cc, ht, wd), generator=g).half().to(dev)
        cp = (Cc + 15) // 16 * 16
        y = planes_to_nhwc(x, cp)
        assert torch.equal(y[..., :Cc], x.permute(0, 2, 3, 1)) and (y[..., Cc:] == 0).all()
    wide = torch.randn((7, 9, 11, 384), generator=g).half().to(dev)
    groups = [5, 2, 5, 9, 2, 5, 9]
    mean, k = group_mean(wide[..., 256:], groups)
    assert k == 3 and mean.shape == (3, 9, 11, 128)
    for row, gid in enumerate((2, 5, 9)):
        ref = torch.stack([wide[e, ..., 256:].float() for e in range(7) if groups[e] == gid]).mean(0)
        assert (mean[row].float() - ref).abs().max().item() <= 2e-3


def test_rejects_bad_arguments(dev):
    from nerfslam._lib import NerfSlamHipError
    from nerfslam.conv import PackedConv, conv_nhwc
    layer = PackedConv(torch.zeros((8, 32, 3, 3), device=dev))
    x = torch.zeros((1, 4, 4, 32), dtype=torch.float16, device=dev)
    with pytest.raises(RuntimeError):
        conv_nhwc([x[..., :16].contiguous()], layer)                    # channel count does not match the packing
    with pytest.raises(RuntimeError):
        conv_nhwc([x.float()], layer)
    out = torch.zeros((1, 4, 4, 10), dtype=torch.float16, device=dev)
    with pytest.raises(NerfSlamHipError):
        conv_nhwc([x], layer, out=out, out_offset=4)                    # slice [4, 12) leaves the 10-channel row
    assert conv_nhwc([x[:0]], layer).shape == (0, 4, 4, 8)             # empty batch is a no-op
