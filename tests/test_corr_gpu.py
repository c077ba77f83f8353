"""GPU parity of the correlation kernels against the CPU oracle (through the C ABI shim)."""
import ctypes as C

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    """bit-exact, except that +0 and -0 compare equal."""
    return ((a.view(np.uint16) == b.view(np.uint16)) | ((a == 0) & (b == 0))).all()


@pytest.mark.parametrize("shape", [(3, 12, 16), (2, 30, 40), (1, 43, 77)])
def test_corr_index_forward_f16_bitexact(oracle_mod, dev, shape):
    import droid_backends
    E, ht, wd = shape
    pyr, coords = synth.lookup_inputs(E, ht, wd, seed=1)
    # edge cases: first / last slice runs poking outside the tensor, NaN / inf / huge coordinates
    coords[0, 0, 0] = [-2.5, -2.25]
    coords[-1, -1, -1] = [wd + 1.5, ht + 0.75]
    coords[0, 1, 1] = [1e9, 3.0]
    coords[0, 1, 2] = [3.0, -np.inf]
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    for l in range(4):
        c = (cf / np.float32(2 ** l)).astype(np.float32)
        got, = droid_backends.corr_index_forward(torch.from_numpy(pyr[l]).to(dev), torch.from_numpy(c).to(dev), 3)
        assert got.shape == (E, 7, 7, ht, wd) and got.dtype == torch.float16
        c_or = np.where(np.isfinite(c), c, np.float32(-1e5))
        ref = oracle_mod.corr_index_forward(pyr[l], c_or, 3)
        assert _bits_equal(got.cpu().numpy(), ref), f"level {l}"


def test_corr_lookup_pyramid_matches_cat_of_levels(oracle_mod, dev):
    from nerfslam.corr import CorrBlock
    E, ht, wd = 3, 16, 24
    pyr, coords = synth.lookup_inputs(E, ht, wd, seed=2)
    blk = CorrBlock.from_pyramid([torch.from_numpy(p).to(dev) for p in pyr])
    out = blk(torch.from_numpy(coords).to(dev)[None])  # [1,E,196,ht,wd]
    assert out.shape == (1, E, 196, ht, wd)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    ref = np.concatenate([oracle_mod.corr_index_forward(pyr[l], cf / np.float32(2 ** l), 3).reshape(E, 49, ht, wd)
                          for l in range(4)], 1)
    assert _bits_equal(out[0].cpu().numpy(), ref)


@pytest.mark.parametrize("radius", [1, 3, 4])
def test_corr_index_forward_f32_and_generic_radius(oracle_mod, dev, radius):
    import droid_backends
    E, ht, wd = 2, 10, 14
    pyr, coords = synth.lookup_inputs(E, ht, wd, seed=3, dtype=np.float32)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    got, = droid_backends.corr_index_forward(torch.from_numpy(pyr[0]).to(dev), torch.from_numpy(cf).to(dev), radius)
    ref = oracle_mod.corr_index_forward(pyr[0], cf, radius)
    # f32: the reference's `+=` contracts to FMA on nvcc; tolerance 1e-6 of max|ref|
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-6 * np.abs(ref).max())
    if radius != 3:
        h = pyr[0].astype(np.float16)
        got16, = droid_backends.corr_index_forward(torch.from_numpy(h).to(dev), torch.from_numpy(cf).to(dev), radius)
        assert _bits_equal(got16.cpu().numpy(), oracle_mod.corr_index_forward(h, cf, radius))


def test_corr_index_backward(oracle_mod, dev):
    import droid_backends
    E, ht, wd = 2, 8, 10
    pyr, coords = synth.lookup_inputs(E, ht, wd, seed=4, dtype=np.float32)
    cf = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    g = np.random.default_rng(0).standard_normal((E, 7, 7, ht, wd)).astype(np.float32)
    got, = droid_backends.corr_index_backward(torch.from_numpy(pyr[0]).to(dev), torch.from_numpy(cf).to(dev),
                                              torch.from_numpy(g).to(dev), 3)
    ref = oracle_mod.corr_index_backward(cf, g, ht, wd, 3)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())


def test_contiguity_error_like_reference(dev):
    import droid_backends
    v = torch.zeros((1, 4, 4, 4, 8), dtype=torch.float16, device=dev)[..., ::2]
    c = torch.zeros((1, 2, 4, 4), device=dev)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        droid_backends.corr_index_forward(v, c, 3)


def test_cpu_tensors_fail_loudly():
    import droid_backends
    from nerfslam._lib import NerfSlamHipError
    with pytest.raises(NerfSlamHipError):
        droid_backends.corr_index_forward(torch.zeros((1, 4, 4, 4, 4), dtype=torch.float16), torch.zeros((1, 2, 4, 4)), 3)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("hw", [(12, 16), (15, 21), (16, 24), (60, 80)])
def test_corr_pyramid_build(oracle_mod, dev, hw, fused):
    """CorrBlock(fmap1, fmap2): all-pairs volume + 3 pooled levels (corr.py:23-38, 63-72).
    Level 0 is a half-rounded f32-accumulated GEMM: the BLAS summation order is unspecified, so the
    bar is <= 1 half-ulp of the exactly-accumulated oracle; pooled levels are compared to the oracle's
    pooling of the *device's* level below them, bit-exactly."""
    from nerfslam.corr import CorrBlock
    ht, wd = hw
    n, Cc = (2, 128) if ht < 60 else (1, 128)
    rng = np.random.default_rng(5)
    f1 = rng.standard_normal((1, n, Cc, ht, wd)).astype(np.float16)
    f2 = rng.standard_normal((1, n, Cc, ht, wd)).astype(np.float16)
    blk = CorrBlock(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), fused=fused)
    lv = [p.cpu().numpy() for p in blk.corr_pyramid]
    assert [p.shape for p in lv] == [(n, ht, wd, ht >> l, wd >> l) for l in range(4)]
    if ht < 60:
        ref = oracle_mod.corr_pyramid(f1[0], f2[0])
        d = np.abs(lv[0].astype(np.float32) - ref[0].astype(np.float32))
        ulp = np.maximum(np.spacing(np.abs(ref[0]).astype(np.float16)).astype(np.float32), 2.0 ** -24)
        # torch.matmul(half) may reduce in f16 inside the BLAS (allow_fp16_reduced_precision_reduction, the
        # default on CUDA and ROCm alike): a few half-ulps, exactly like the reference's cuBLAS call
        if fused:  # f32 MFMA accumulation (error <= ~2e-6 absolute here) then ONE rounding to half
            assert (d <= ulp + 2e-6).all(), f"max {float((d / ulp).max()):.2f} ulp"
            assert (d > 0).mean() < 0.05
        else:
            assert (d <= 4 * ulp).all(), f"max {float((d / ulp).max()):.2f} ulp"
            assert (d > ulp).mean() < 0.02
    for l in range(3):
        h, w = ht >> l, wd >> l
        out = np.empty((n, ht, wd, h // 2, w // 2), np.uint16)
        oracle_mod.lib().orc_corr_pool_f16(lv[l].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                           C.c_long(n * ht * wd), h, w)
        assert _bits_equal(lv[l + 1], out.view(np.float16)), f"pool level {l + 1}"


@pytest.mark.parametrize("cfg", [(2, 1, 12, 16, 12, 16, 128), (1, 2, 9, 11, 4, 5, 64), (1, 1, 16, 20, 8, 10, 256)])
def test_altcorr_forward(oracle_mod, dev, cfg):
    import droid_backends
    B, N, H1, W1, H2, W2, Cc = cfg
    rng = np.random.default_rng(6)
    f1 = rng.standard_normal((B, H1, W1, Cc)).astype(np.float32)
    f2 = rng.standard_normal((B, H2, W2, Cc)).astype(np.float32)
    gy, gx = np.meshgrid(np.arange(H1), np.arange(W1), indexing="ij")
    coords = (np.stack([gx, gy], -1)[None, None] * (W2 / W1) + rng.uniform(-6, 6, (B, N, H1, W1, 2))).astype(np.float32)
    coords[0, 0, 0, 0] = [-20.0, 3.0]
    got, = droid_backends.altcorr_forward(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev),
                                          torch.from_numpy(coords).to(dev), 3)
    ref = oracle_mod.altcorr_forward(f1, f2, coords, 3)
    assert got.shape == ref.shape
    # f32, different summation order than the reference's 32-channel slabs: 1e-5 of max|ref|
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-5 * np.abs(ref).max())


def _ref_levels(fm, half):
    """reference pyramid of AltCorrBlock (corr.py:96-105) as channels-last float32 arrays: `/ 4` and avg_pool2d in the dtype of
    the features -- half features give a pyramid ROUNDED TO HALF after every pooling (the float cast happens at the call, :121)"""
    lv = torch.from_numpy(fm[0])
    lv = lv / 4.0 if half else lv.float() / 4.0
    out = []
    for _ in range(4):
        out.append(lv.float().permute(0, 2, 3, 1).contiguous().numpy())
        lv = torch.nn.functional.avg_pool2d(lv, 2, stride=2)
    return out


@pytest.mark.parametrize("half", [True, False], ids=["f16_mfma", "f32"])
def test_altcorr_block_fused_pyramid(oracle_mod, dev, half):
    """AltCorrBlock (fused, frame-indexed) == per-level altcorr_forward of the oracle on gathered maps."""
    from nerfslam.corr import AltCorrBlock
    rng = np.random.default_rng(7)
    nfr, Cc, H, W, E = 5, 128, 16, 24, 6
    fm = rng.standard_normal((1, nfr, Cc, H, W)).astype(np.float16)
    ii = rng.integers(0, nfr, E).astype(np.int64)
    jj = rng.integers(0, nfr, E).astype(np.int64)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    coords = (np.stack([gx, gy], -1)[None, None] + rng.uniform(-7, 7, (1, E, H, W, 2))).astype(np.float32)
    fmt = torch.from_numpy(fm).to(dev)
    blk = AltCorrBlock(fmt if half else fmt.float())
    assert blk.half == half
    got = blk(torch.from_numpy(coords).to(dev), torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev))
    assert got.shape == (1, E, 196, H, W)
    lvs = _ref_levels(fm, half)
    for l in range(4):
        ref = oracle_mod.altcorr_forward(lvs[0][ii], lvs[l][jj], coords[0][:, None] / np.float32(2 ** l), 3)[:, 0]
        g = got[0, :, 49 * l:49 * (l + 1)].cpu().numpy()
        np.testing.assert_allclose(g, ref, rtol=0, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("half", [True, False], ids=["f16_mfma", "f32"])
def test_altcorr_block_smooth_flow_and_tiles_that_leave_the_image(oracle_mod, dev, half):
    """The staged (LDS) path of the tile kernel: a smooth flow keeps a tile's windows inside a small box.  Edge 1 is shifted
    so far that whole 8x8 tiles look outside the image (their box stays empty -> exact zeros, altcorr_kernel.cu:63-66 `within`),
    edge 2 leaves only on the right-hand side; H is not a multiple of 8 (the 1280x720 grid is 90 rows)."""
    from nerfslam.corr import AltCorrBlock
    rng = np.random.default_rng(17)
    nfr, Cc, H, W, E = 3, 128, 26, 40, 3
    fm = rng.standard_normal((1, nfr, Cc, H, W)).astype(np.float16)
    ii, jj = np.array([0, 1, 2], np.int64), np.array([1, 2, 0], np.int64)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([gx, gy], -1).astype(np.float32)
    coords = np.stack([base + [1.3, -0.6], base + [3.0 * W, 0.4], base + [W - 20.25, 2.5]])[None].astype(np.float32)
    fmt = torch.from_numpy(fm).to(dev)
    blk = AltCorrBlock(fmt if half else fmt.float())
    got = blk(torch.from_numpy(coords).to(dev), torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev))[0].cpu().numpy()
    assert np.all(got[1] == 0.0), "windows entirely outside the image correlate to exact zeros"
    lvs = _ref_levels(fm, half)
    for l in range(4):
        ref = oracle_mod.altcorr_forward(lvs[0][ii], lvs[l][jj], coords[0][:, None] / np.float32(2 ** l), 3)[:, 0]
        np.testing.assert_allclose(got[:, 49 * l:49 * (l + 1)], ref, rtol=0, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("half", [True, False], ids=["f16_mfma", "f32"])
def test_altcorr_block_vs_oracle_at_c1280(oracle_mod, dev, half):
    """altcorr_tile_kernel at config #5's grid (1280x720 -> 90x160; 90 is not a multiple of 8) against the oracle, every
    level: a smooth flow (LDS-staged path), a flow that leaves the image on the right and at the bottom, a per-pixel random
    flow (windows of one tile far apart: the box does not fit the staging buffer), and an edge entirely outside."""
    from nerfslam.corr import AltCorrBlock
    rng = np.random.default_rng(1280)
    nfr, Cc, H, W = 4, 128, 90, 160
    fm = rng.standard_normal((1, nfr, Cc, H, W)).astype(np.float16)
    ii, jj = np.array([0, 1, 2, 3], np.int64), np.array([1, 2, 3, 0], np.int64)
    gy, gx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([gx, gy], -1).astype(np.float32)
    swirl = np.stack([3.5 * np.sin(gy / 17.0) + 0.013 * gx, 2.5 * np.cos(gx / 23.0) - 0.02 * gy], -1).astype(np.float32)
    coords = np.stack([base + swirl,
                       base + [W - 30.25, H - 20.5] + 0.5 * swirl,
                       base + rng.uniform(-12, 12, base.shape).astype(np.float32),
                       base + [-2.0 * W, 3.0 * H]])[None].astype(np.float32)
    fmt = torch.from_numpy(fm).to(dev)
    blk = AltCorrBlock(fmt if half else fmt.float())
    got = blk(torch.from_numpy(coords).to(dev), torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev))[0].cpu().numpy()
    assert got.shape == (4, 196, H, W)
    assert np.all(got[3] == 0.0)
    lvs = _ref_levels(fm, half)
    for l in range(4):
        ref = oracle_mod.altcorr_forward(lvs[0][ii], lvs[l][jj], coords[0][:, None] / np.float32(2 ** l), 3)[:, 0]
        g = got[:, 49 * l:49 * (l + 1)]
        assert np.abs(ref[:3]).max() > 0.2 and (ref[1] == 0).mean() > 0.5 and (ref[1] != 0).mean() > 0.005
        np.testing.assert_allclose(g, ref, rtol=0, atol=1e-5 * np.abs(ref).max(), err_msg=f"level {l}")


def test_altcorr_backward_against_autograd(dev):
    """droid_backends.altcorr_backward (dead in the reference's inference path, src/droid.cpp:315-327) against torch autograd
    through a float64 restatement of the forward pass (raw 8x8 taps, bilinear blend, channel = iy + 7 ix)."""
    import droid_backends
    g = torch.Generator().manual_seed(3)
    B, N, H1, W1, H2, W2, Cc = 2, 2, 7, 9, 6, 8, 72
    f1 = torch.randn((B, H1, W1, Cc), generator=g, dtype=torch.float64, requires_grad=True)
    f2 = torch.randn((B, H2, W2, Cc), generator=g, dtype=torch.float64, requires_grad=True)
    gy, gx = torch.meshgrid(torch.arange(H1), torch.arange(W1), indexing="ij")
    coords = torch.stack([gx, gy], -1)[None, None].double() * (W2 / W1) + (torch.rand((B, N, H1, W1, 2), generator=g, dtype=torch.float64) * 8 - 4)
    coords = coords.float().double()
    up = torch.randn((B, N, 49, H1, W1), generator=g, dtype=torch.float64).float().double()
    x0, y0 = coords[..., 0].floor(), coords[..., 1].floor()
    dx, dy = coords[..., 0] - x0, coords[..., 1] - y0
    f2p = torch.nn.functional.pad(f2, (0, 0, 12, 12, 12, 12))                   # zero border: out-of-bounds taps contribute 0
    bidx = torch.arange(B)[:, None, None, None].expand(B, N, H1, W1)
    raw = {}
    for ty in range(8):
        for tx in range(8):
            hy = (y0 - 3 + ty).long().clamp(-12, H2 + 11) + 12
            hx = (x0 - 3 + tx).long().clamp(-12, W2 + 11) + 12
            raw[ty, tx] = (f1[:, None] * f2p[bidx, hy, hx]).sum(-1)             # [B,N,H1,W1]
    out = torch.zeros((B, N, 49, H1, W1), dtype=torch.float64)
    chans = []
    for ix in range(7):
        for iy in range(7):
            chans.append(raw[iy, ix] * (1 - dy) * (1 - dx) + raw[iy, ix + 1] * (1 - dy) * dx + raw[iy + 1, ix] * dy * (1 - dx)
                         + raw[iy + 1, ix + 1] * dy * dx)                       # channel iy + 7 ix
    out = torch.stack(chans, 2)
    fwd, = droid_backends.altcorr_forward(f1.detach().float().to(dev), f2.detach().float().to(dev), coords.float().to(dev), 3)
    assert (fwd.cpu().double() - out.detach()).abs().max() <= 1e-5 * out.detach().abs().max()      # the restatement is the op
    (out * up).sum().backward()
    g1, g2, gc = droid_backends.altcorr_backward(f1.detach().float().to(dev), f2.detach().float().to(dev), coords.float().to(dev),
                                                 up.float().to(dev), 3)
    assert g1.shape == f1.shape and g2.shape == f2.shape and gc.shape == coords.shape and not gc.any()
    assert (g1.cpu().double() - f1.grad).abs().max() <= 2e-5 * f1.grad.abs().max()
    assert (g2.cpu().double() - f2.grad).abs().max() <= 2e-5 * f2.grad.abs().max()


@pytest.mark.parametrize("shape", [(16, 24), (43, 77), (60, 80)])
def test_tiled_layout_same_volume_and_same_lookup_bits(dev, shape):
    """the private 8x8-tiled slice layout (levels 0 / 1): the volume values are those of the row-major build, and the
    lookup through it is bit-identical, including windows that straddle tiles, borders, and odd image sizes"""
    from nerfslam.corr import CorrBlock
    ht, wd = shape
    g = torch.Generator().manual_seed(ht * 100 + wd)
    E = 3
    f1 = torch.randn((1, E, 128, ht, wd), generator=g).half().to(dev)
    f2 = torch.randn((1, E, 128, ht, wd), generator=g).half().to(dev)
    a = CorrBlock(f1, f2, fused=True, tiled=False)
    b = CorrBlock(f1, f2, fused=True, tiled=True)
    for l, (x, y) in enumerate(zip(a.corr_pyramid, b.untiled())):
        assert x.shape == y.shape and torch.equal(x, y), l
    gy, gx = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    base = torch.stack([gx, gy], -1).float()
    for scale in (1.5, 12.0, 200.0):
        coords = (base[None, None] + torch.randn((1, E, ht, wd, 2), generator=g) * scale).to(dev)
        coords[0, 0, 0, 0] = float("nan")
        coords[0, 1, 1, 1, 0] = -3.25
        coords[0, 2, 2, 2] = torch.tensor([wd + 2.5, ht + 1.75])
        ra, rb = a(coords), b(coords)
        assert torch.equal(ra.view(torch.int16), rb.view(torch.int16)), scale
    # payload operations of the frontend keep working on the tiled block
    c = CorrBlock(f1, f2, fused=True, tiled=True)
    c = c[torch.tensor([True, False, True], device=dev)]
    assert c.corr_pyramid[0].shape[0] == 2 and torch.equal(c(coords[:, [0, 2]]), rb[:, [0, 2]])


def test_full_size_c1280_properties(dev):
    """BASELINE config #5 size (1280x720 -> 90x160 grid, one edge = 415 MB of volume): size-independent properties --
    tiled build == row-major build bit for bit, lookups through both agree bit for bit, the identity lookup returns the
    volume's own entries, and the on-the-fly (altcorr) path agrees with the volume lookup to f16 resolution"""
    from nerfslam.corr import AltCorrBlock, CorrBlock
    ht, wd = 90, 160
    g = torch.Generator().manual_seed(7)
    f1 = (torch.randn((1, 1, 128, ht, wd), generator=g) * 0.5).half().to(dev)
    f2 = (torch.randn((1, 1, 128, ht, wd), generator=g) * 0.5).half().to(dev)
    a = CorrBlock(f1, f2, fused=True, tiled=False)
    b = CorrBlock(f1, f2, fused=True, tiled=True)
    for l, (x, y) in enumerate(zip(a.corr_pyramid, b.untiled())):
        assert torch.equal(x, y), l
    gy, gx = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    base = torch.stack([gx, gy], -1).float()[None, None].to(dev)
    coords = base + torch.randn((1, 1, ht, wd, 2), generator=g).to(dev) * 20.0
    ra, rb = a(coords), b(coords)
    assert torch.equal(ra.view(torch.int16), rb.view(torch.int16))
    # integer coordinates: the centre tap (channel 3*7+3 of level 0) is volume[p, y, x] itself
    centre = a(base)[0, 0, 24]
    vol = a.corr_pyramid[0][0]
    own = vol.reshape(ht * wd, ht * wd).diagonal().view(ht, wd)
    assert torch.equal(centre, own)
    scale = ra.float().abs().max().item()
    outs = []
    for feats in (torch.cat([f1, f2], 1).float(), torch.cat([f1, f2], 1)):       # f32 kernels, then the half pyramid on the matrix cores
        alt = AltCorrBlock(feats)
        rc = alt(coords, torch.tensor([0], device=dev), torch.tensor([1], device=dev))
        assert (rc[0, 0] - ra[0, 0].float()).abs().max().item() <= 4e-3 * scale     # f16 volume vs f32 on-the-fly
        outs.append(rc)
    # level 0 of both on-the-fly paths sees the same f16 values: equal up to the summation order
    assert (outs[0][0, 0, :49] - outs[1][0, 0, :49]).abs().max().item() <= 1e-5 * scale


def test_empty_edge_sets(dev):
    """E = 0 everywhere an edge count appears: the entry points accept it and return empty results"""
    from nerfslam.corr import AltCorrBlock, CorrBlock
    import droid_backends
    ht, wd = 12, 16
    bank = torch.randn((4, ht * wd, 128), device=dev).half()
    empty = torch.zeros(0, dtype=torch.long, device=dev)
    for tiled in (False, True):
        pyr = CorrBlock.build_pyramid(bank, bank, empty, empty, 0, ht, wd, tiled=tiled)
        blk = CorrBlock.from_pyramid(pyr, tiled=tiled, hw=(ht, wd))
        out = blk(torch.zeros((1, 0, ht, wd, 2), device=dev))
        assert out.shape == (1, 0, 196, ht, wd)
    d = droid_backends.frame_distance(torch.zeros((4, 7), device=dev), torch.ones((4, ht, wd), device=dev),
                                      torch.tensor([10.0, 10.0, 8.0, 6.0], device=dev), empty, empty, 0.3)
    assert d.shape == (0,)
    alt = AltCorrBlock(torch.randn((1, 4, 128, ht, wd), device=dev))
    assert alt(torch.zeros((1, 0, ht, wd, 2), device=dev), empty, empty).shape == (1, 0, 196, ht, wd)


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 3, 5), (3, 8, 8), (2, 9, 17), (1, 24, 8), (5, 31, 33)])
def test_lookup_small_and_odd_shapes_against_oracle(oracle_mod, dev, shape):
    """the cooperative fused lookup on degenerate sizes (levels that shrink to 0-1 pixels are clamped by the reference's
    own floor division), row-major and tiled volumes, against the oracle's per-level lookups bit for bit"""
    from nerfslam.corr import CorrBlock
    E, ht, wd = shape
    g = torch.Generator().manual_seed(E * 1000 + ht * 10 + wd)
    nl = 1
    while nl < 4 and (ht >> nl) > 0 and (wd >> nl) > 0:
        nl += 1
    f1 = torch.randn((1, E, 128, ht, wd), generator=g).half().to(dev)
    f2 = torch.randn((1, E, 128, ht, wd), generator=g).half().to(dev)
    gy, gx = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    coords = (torch.stack([gx, gy], -1).float()[None, None] + torch.randn((1, E, ht, wd, 2), generator=g) * 3.0).to(dev)
    blocks = [CorrBlock(f1, f2, num_levels=nl, fused=True, tiled=t) for t in (False, True)]
    outs = [b(coords) for b in blocks]
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    pyr = [p.cpu().numpy() for p in blocks[0].corr_pyramid]
    cf = np.ascontiguousarray(coords[0].cpu().numpy().transpose(0, 3, 1, 2))
    ref = np.concatenate([oracle_mod.corr_index_forward(pyr[l], cf / np.float32(2 ** l), 3).reshape(E, 49, ht, wd)
                          for l in range(nl)], 1)
    got = outs[0][0].cpu().numpy()
    assert ((got.view(np.uint16) == ref.view(np.uint16)) | ((got == 0) & (ref == 0))).all()


@pytest.mark.parametrize("shape", [(60, 80), (13, 21)])
def test_lookup_fused_with_the_correlation_encoder(dev, shape):
    """CorrPool.lookup_encoded (csrc/corr_lookup.hip: corr_lookup_enc_kernel) = relu(Conv2d(196,128,1)(lookup)) of the reference
    chain corr.py:40-50 -> droid_net.py:83-87,133: against (a) the product's own unfused launches (lookup, transposition, the
    MFMA 1x1 convolution) and (b) a float64 evaluation of the same layer on the bit-exact lookup -- equal up to the f16 rounding
    of the output (f32 accumulation in both kernels, different summation order).  Slots permuted, pixels that leave the image,
    a pixel count that is not a multiple of the 256-pixel workgroup tile."""
    from nerfslam.conv import PackedConv, planes_to_nhwc
    from nerfslam.corr import CorrPool
    from nerfslam.update_op import CorrEncoderWeights
    ht, wd = shape
    g = torch.Generator().manual_seed(7 * ht + wd)
    nf, E = 6, 5
    bank = (torch.randn((nf, ht * wd, 128), generator=g) / 4.0).half().to(dev)
    ii = torch.tensor([0, 1, 2, 3, 4], device=dev)
    jj = torch.tensor([1, 2, 3, 4, 5], device=dev)
    pool = CorrPool(ht, wd, 9, dev)
    slots = torch.tensor([7, 2, 5, 0, 3], dtype=torch.int32, device=dev)
    pool.build(bank, bank, ii, jj, slots)
    gy, gx = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    coords = (torch.stack([gx, gy], -1).float()[None, None] + 4.0 * torch.randn((1, E, ht, wd, 2), generator=g)).to(dev)
    coords[0, 0, 0, 0] = float("nan")
    coords[0, 1, 1, 1] = torch.tensor([-30.0, -30.0])
    W = (torch.randn((128, 196, 1, 1), generator=g) / 14.0).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    enc = CorrEncoderWeights(W, b)
    fused = pool.lookup_encoded(coords, slots, enc).c1
    look = pool.lookup(coords, slots)                                         # [1,E,196,ht,wd] f16, bit-exact (tests above)
    unfused = PackedConv(W, b, pad_cin_to=208)([planes_to_nhwc(look[0].contiguous(), 208)], act="relu")
    assert fused.shape == unfused.shape == (E, ht, wd, 128) and fused.dtype == torch.float16
    x = look[0].double().permute(0, 2, 3, 1)                                  # [E,ht,wd,196]
    ref = torch.relu(x @ W.half().double().reshape(128, 196).t() + b.double())
    scale = float(ref.abs().max())
    for name, got in (("fused", fused), ("unfused", unfused)):
        err = float((got.double() - ref).abs().max())
        assert err <= 1.5e-3 * scale + 1e-3, (name, err, scale)                # one f16 ulp at the top of the range
    assert float((fused.double() - unfused.double()).abs().max()) <= 2e-3 * scale + 1e-3
    assert torch.isfinite(fused).all()


@pytest.mark.parametrize("shape", [(90, 160), (21, 35)])
def test_altcorr_fused_with_the_correlation_encoder(dev, shape):
    """AltCorrBlock.encoded (csrc/altcorr.hip: altcorr_tile_enc_kernel) = relu(Conv2d(196,128,1)(half(AltCorrBlock(...)))) -- the
    reference chain corr.py:107-131 -> droid_net.py:83-87,133 under autocast -- against the product's unfused launches and a
    float64 evaluation of the layer on the unfused correlation: equal up to the f16 rounding of the output.  Smooth flow, flow
    that leaves the image, wild flow (the wave-per-pixel fallback), image sizes that are not multiples of the 8x8 tile."""
    from nerfslam.conv import PackedConv, planes_to_nhwc
    from nerfslam.corr import AltCorrBlock
    from nerfslam.update_op import CorrEncoderWeights
    H, W = shape
    g = torch.Generator().manual_seed(H * 31 + W)
    nf, E = 5, 4
    fm = torch.randn((1, nf, 128, H, W), generator=g).half().to(dev)
    alt = AltCorrBlock(fm)
    gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    base = torch.stack([gx, gy], -1).float()
    coords = base[None, None].repeat(1, E, 1, 1, 1)
    coords[0, 0] += 1.5 * torch.randn((H, W, 2), generator=g)                   # smooth-ish
    coords[0, 1] += torch.tensor([W * 0.9, -H * 0.7])                           # leaves the image
    coords[0, 2] += 40.0 * torch.randn((H, W, 2), generator=g)                  # wild: region > 1024 pixels
    coords[0, 3, 0, 0] = float("nan")
    coords = coords.to(dev)
    ii = torch.tensor([0, 1, 2, 3], device=dev)
    jj = torch.tensor([1, 2, 3, 4], device=dev)
    Wt = (torch.randn((128, 196, 1, 1), generator=g) / 14.0).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    fused = alt.encoded(coords, ii, jj, CorrEncoderWeights(Wt, b)).c1
    look = alt(coords, ii, jj)[0].half()                                        # [E,196,H,W]: autocast's cast of the f32 result
    unfused = PackedConv(Wt, b, pad_cin_to=208)([planes_to_nhwc(look.contiguous(), 208)], act="relu")
    ref = torch.relu(look.double().permute(0, 2, 3, 1) @ Wt.half().double().reshape(128, 196).t() + b.double())
    scale = float(ref.abs().max())
    assert fused.shape == (E, H, W, 128) and torch.isfinite(fused).all()
    for name, got in (("fused", fused), ("unfused", unfused)):
        err = float((got.double() - ref).abs().max())
        assert err <= 1.5e-3 * scale + 1e-3, (name, err, scale)
