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
