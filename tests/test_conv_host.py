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
