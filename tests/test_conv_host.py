"""Host side of the MFMA convolution (nerfslam/conv.py): the fragment packing of the weights is what the kernel's A operand
reads, so it is checked here (CPU, no device work) against the layout documented in include/nerfslam_hip.h."""
import numpy as np
import pytest
import torch


def test_packed_cout_rule():
    from nerfslam.conv import packed_cout
    assert [packed_cout(c) for c in (1, 2, 32, 33, 64, 65, 128, 129, 256, 384, 576)] == [32, 32, 32, 64, 64, 128, 128, 256, 256, 384, 640]


@pytest.mark.parametrize("co,ci,k", [(40, 48, 3), (128, 196, 1), (2, 128, 3), (130, 16, 1)])
def test_pack_weights_layout(co, ci, k):
    """packed[c][t][ct][h][i][e] == w[32 ct + i][16 c + 8 h + e][tap t], zero in the padding"""
    from nerfslam.conv import PackedConv, packed_cout
    g = torch.Generator().manual_seed(co + ci)
    w = torch.randn((co, ci, k, k), generator=g).half().float()
    layer = PackedConv(w, torch.zeros(co), pad_cin_to=208 if ci == 196 else None)
    cip, cop = layer.cin_padded, packed_cout(co)
    assert cip % 16 == 0 and cip >= ci and tuple(layer.w.shape) == (cip // 16, k * k, cop // 32, 2, 32, 8)
    full = np.zeros((cop, cip, k * k), np.float32)
    full[:co, :ci] = w.reshape(co, ci, k * k).numpy()
    p = layer.w.float().numpy()
    rng = np.random.default_rng(0)
    for _ in range(300):
        c, t, ct, h, i, e = (int(rng.integers(0, n)) for n in p.shape)
        assert p[c, t, ct, h, i, e] == full[32 * ct + i, 16 * c + 8 * h + e, t]
    assert np.count_nonzero(p) == np.count_nonzero(full)


def test_fused_layers_and_cpu_tensors_are_refused():
    from nerfslam._lib import NerfSlamHipError
    from nerfslam.conv import PackedConv, conv_nhwc
    a, b = torch.nn.Conv2d(32, 24, 3, padding=1), torch.nn.Conv2d(32, 40, 3, padding=1)
    layer = PackedConv.from_modules(a, b)
    assert layer.cout == 64 and layer.cin == 32 and layer.ksize == 3
    assert torch.equal(layer.bias, torch.cat([a.bias, b.bias]).detach().float())
    with pytest.raises(NerfSlamHipError):      # no CPU fallback
        conv_nhwc([torch.zeros((1, 4, 4, 32), dtype=torch.float16)], layer)


def test_update_operator_packs_the_torch_modules_weights():
    """HipUpdateOperator construction is host work: fused layers carry the right weights, biases and channel padding"""
    from nerfslam.droid_nets import UpdateModule
    from nerfslam.update_op import HipUpdateOperator
    torch.manual_seed(0)
    um = UpdateModule().eval()
    op = HipUpdateOperator(um)
    g = um.gru
    assert (op.zr.cout, op.zr.cin_padded, op.zr.ksize) == (256, 448, 3) and (op.q.cout, op.q.cin_padded) == (128, 448)
    assert (op.corr1.cin, op.corr1.cin_padded, op.corr1.ksize) == (196, 208, 1)
    assert (op.flow1.cin, op.flow1.cin_padded, op.flow1.cout, op.flow1.ksize) == (196, 208, 128, 1)      # 7x7x4 as im2col
    assert (op.heads.cout, op.delta2.cout, op.weight2.cout, op.eta.cout, op.upmask.cout) == (384, 2, 2, 1, 576)
    # convz | convr fused along the couts: fragment (c=0, tap 4 = centre, ct, h, i, e) of the second half is convr's weight
    w = op.zr.w.float()
    assert w[0, 4, 4, 1, 3, 5].item() == pytest.approx(g.convr.weight[3, 13, 1, 1].half().float().item())
    assert w[0, 4, 0, 1, 3, 5].item() == pytest.approx(g.convz.weight[3, 13, 1, 1].half().float().item())
    # the three global-context 1x1 convolutions as one [128, 384] matrix with the gate biases folded in
    glo = torch.randn((5, 128))
    ref = torch.cat([g.convz_glo(glo[:, :, None, None])[:, :, 0, 0] + g.convz.bias,
                     g.convr_glo(glo[:, :, None, None])[:, :, 0, 0] + g.convr.bias,
                     g.convq_glo(glo[:, :, None, None])[:, :, 0, 0] + g.convq.bias], 1)
    assert torch.allclose(torch.addmm(op.glo_b, glo, op.glo_w), ref.detach(), atol=1e-5)


def _unpack(layer):
    """PackedConv fragments -> dense [cout, cin_padded, taps] f32 (inverse of pack_weights)"""
    p = layer.w.float()                                  # c, t, ct, h, i, e
    nc, nt, nct = p.shape[:3]
    return p.permute(2, 4, 0, 3, 5, 1).reshape(nct * 32, nc * 16, nt)[:layer.cout]


def test_encoder_layers_as_1x1_over_patches():
    """nerfslam/encoder_op.py host logic (CPU): the 7x7 / stride-2 stem and the stride-2 3x3 layers become 1x1 convolutions
    over tap-major patches, and a block's 1x1 / stride-2 shortcut reads the centre-tap slice of the same patches -- the
    re-ordered weights times a numpy statement of the patch kernels' layout (csrc/encoder.hip) must equal F.conv2d."""
    import torch.nn.functional as F
    from nerfslam.droid_nets import BasicEncoder
    from nerfslam.encoder_op import HipEncoder, _half_out

    def patches(x, k, pad):            # [N,C,H,W] -> [N,Ho,Wo,k*k*C], out[..., (ky*k+kx)*C + c] = x[n, c, 2oy+ky-pad, 2ox+kx-pad]
        N, Cc, H, W = x.shape
        Ho, Wo = _half_out(H), _half_out(W)
        xp = F.pad(x, (pad, pad + 2, pad, pad + 2))
        out = torch.zeros((N, Ho, Wo, k * k * Cc))
        for ky in range(k):
            for kx in range(k):
                out[..., (ky * k + kx) * Cc:(ky * k + kx + 1) * Cc] = xp[:, :, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2].permute(0, 2, 3, 1)
        return out

    torch.manual_seed(1)
    enc = BasicEncoder(128, "instance").eval()
    op = HipEncoder(enc, True, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    assert len(op.blocks) == 6 and [b.stride2 for b in op.blocks] == [False, False, True, False, True, False]
    h = lambda t: t.half().float()                       # the packed weights are f16
    with torch.no_grad():
        x = torch.randn(2, 3, 33, 47)
        w = _unpack(op.stem)[:, :, 0]                    # [32, 160]: 147 taps + zero padding
        assert op.stem.cin_padded == 160 and not w[:, 147:].any()
        got = patches(x, 7, 3) @ w[:, :147].t() + enc.conv1.bias
        assert torch.allclose(got.permute(0, 3, 1, 2), F.conv2d(x, h(enc.conv1.weight), enc.conv1.bias, stride=2, padding=3), atol=2e-5)
        for blk, rb, cin in ((op.blocks[2], enc.layer2[0], 32), (op.blocks[4], enc.layer3[0], 64)):
            y = torch.randn(1, cin, 17, 24)
            P = patches(y, 3, 1)
            got = P @ _unpack(blk.conv1)[:, :, 0].t() + rb.conv1.bias
            assert torch.allclose(got.permute(0, 3, 1, 2), F.conv2d(y, h(rb.conv1.weight), rb.conv1.bias, stride=2, padding=1), atol=1e-4)
            got = P[..., 4 * cin:5 * cin] @ _unpack(blk.down)[:, :, 0].t() + rb.downsample[0].bias
            assert torch.allclose(got.permute(0, 3, 1, 2), F.conv2d(y, h(rb.downsample[0].weight), rb.downsample[0].bias, stride=2), atol=1e-4)
    assert _half_out(480) == 240 and _half_out(47) == 24 and _half_out(1) == 1
