"""Host side of the MFMA convolution (`csrc/conv.hip`, `ns_conv_nhwc_f16`): weight packing and the launch wrapper.

Everything here is channels-last: an activation is a contiguous f16 tensor [N, H, W, C].  A convolution reads the channel
concatenation of up to 4 such tensors (the `torch.cat` of networks/modules/gru.py:24-25 never materialises) and writes a
channel slice of its output tensor, so the encoders can write straight into the buffers the ConvGRU reads.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib, ptr, require_cuda, stream_ptr

ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


def packed_cout(cout):
    L = lib()
    return int(L.ns_conv_packed_cout(int(cout)))


def pack_weights(weight, pad_cin_to=None):
    """nn.Conv2d weight [CO, CI, k, k] (k = 1 or 3) -> f16 fragments [CI/16][k*k][COP/32][2][32][8] holding
    w[co = 32 ct + i][ci = 16 c + 8 h + e][tap] (include/nerfslam_hip.h); CI is zero-padded to a multiple of 16 (or to
    `pad_cin_to`), CO to ns_conv_packed_cout(CO)."""
    co, ci, kh, kw = weight.shape
    assert kh == kw and kh in (1, 3)
    cip = pad_cin_to if pad_cin_to is not None else (ci + 15) // 16 * 16
    assert cip % 16 == 0 and cip >= ci
    cop = packed_cout(co)
    w = torch.zeros((cop, cip, kh * kw), dtype=torch.float32, device=weight.device)
    w[:co, :ci] = weight.detach().float().reshape(co, ci, kh * kw)
    w = w.reshape(cop // 32, 32, cip // 16, 2, 8, kh * kw)          # ct, i, c, h, e, t
    return w.permute(2, 5, 0, 3, 1, 4).contiguous().half()          # c, t, ct, h, i, e


class PackedConv:
    """one convolution layer: packed weights + f32 bias, ready to launch"""

    def __init__(self, weight, bias=None, pad_cin_to=None):
        self.cout, self.cin = int(weight.shape[0]), int(weight.shape[1])
        self.ksize = int(weight.shape[2])
        self.w = pack_weights(weight, pad_cin_to)
        self.cin_padded = self.w.shape[0] * 16
        self.bias = None if bias is None else bias.detach().float().contiguous()

    @classmethod
    def from_modules(cls, *convs, pad_cin_to=None):
        """several nn.Conv2d with the same input, fused along the output channels (e.g. convz | convr)"""
        w = torch.cat([c.weight for c in convs], 0)
        b = torch.cat([c.bias for c in convs], 0) if convs[0].bias is not None else None
        return cls(w, b, pad_cin_to)

    def __call__(self, srcs, act=None, out=None, out_offset=0, bias=None, fuse=None):
        return conv_nhwc(srcs, self, act=act, out=out, out_offset=out_offset, bias=bias, fuse=fuse)


def _slice_operand(t, N, H, W):
    """(tensor, pixel stride) of a channels-last f16 operand that may be a channel slice of a wider tensor"""
    if t.dtype != torch.float16 or t.dim() != 4 or tuple(t.shape[:3]) != (N, H, W) or t.stride(3) != 1 or \
            t.stride(1) != W * t.stride(2) or t.stride(0) != H * W * t.stride(2):
        raise RuntimeError("conv_nhwc: fused operands must be channels-last f16 [N,H,W,C] (dense or a channel slice)")
    return t, int(t.stride(2))


def conv_nhwc(srcs, layer, act=None, out=None, out_offset=0, bias=None, fuse=None):
    """srcs: list of channels-last f16 tensors [N,H,W,C_s] whose channel counts add up to layer.cin_padded.
    bias: None (the layer's own), or a per-image f32 tensor [N, cout] (the layer's bias must then be folded in by the caller).
    out: None (a fresh [N,H,W,cout]) or a channels-last f16 tensor whose channels [out_offset, out_offset+cout) are written.
    fuse: None, ("mul_hi", h): the upper half of the couts is multiplied by h after the activation (z | r*h of the ConvGRU),
          or ("gru", z, h): the output is h + z * (act(conv) - h)."""
    if not isinstance(srcs, (list, tuple)):
        srcs = [srcs]
    require_cuda(*srcs)
    N, H, W = srcs[0].shape[:3]
    chans, strides = [], []
    for s in srcs:
        if s.dtype != torch.float16 or s.dim() != 4 or tuple(s.shape[:3]) != (N, H, W):
            raise RuntimeError("conv_nhwc: sources must be f16 [N,H,W,C] tensors of one spatial shape")
        ps = s.stride(2)                                 # pixel stride: C, or more for a channel slice of a wider tensor
        if N * H * W > 0 and (s.stride(3) != 1 or s.stride(1) != W * ps or s.stride(0) != H * W * ps):
            raise RuntimeError("conv_nhwc: sources must be channels-last (dense, or a channel slice of a dense tensor)")
        chans.append(int(s.shape[3]))
        strides.append(int(ps) if N * H * W > 0 else int(s.shape[3]))
    if sum(chans) != layer.cin_padded:
        raise RuntimeError(f"conv_nhwc: sources carry {sum(chans)} channels, the layer was packed for {layer.cin_padded}")
    if out is None:
        out = torch.empty((N, H, W, layer.cout), dtype=torch.float16, device=srcs[0].device)
    elif out.dtype != torch.float16 or not out.is_contiguous() or tuple(out.shape[:3]) != (N, H, W):
        raise RuntimeError("conv_nhwc: out must be a contiguous f16 [N,H,W,C] tensor")
    b, bstride = layer.bias, 0
    if bias is not None:
        if bias.dtype != torch.float32 or not bias.is_contiguous() or tuple(bias.shape) != (N, layer.cout):
            raise RuntimeError("conv_nhwc: a per-image bias is a contiguous f32 [N, cout] tensor")
        b, bstride = bias, layer.cout
    n = len(srcs)
    src_arr = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    ch_arr = (C.c_int * n)(*chans)
    st_arr = (C.c_int * n)(*strides)
    mode, e0, e0s, e1, e1s = 0, None, 0, None, 0
    if fuse is not None and N * H * W > 0:
        if fuse[0] == "mul_hi":
            mode, (e0, e0s) = 1, _slice_operand(fuse[1], N, H, W)
        elif fuse[0] == "gru":
            mode, (e0, e0s), (e1, e1s) = 2, _slice_operand(fuse[1], N, H, W), _slice_operand(fuse[2], N, H, W)
        else:
            raise RuntimeError(f"conv_nhwc: unknown fusion {fuse[0]!r}")
    with torch.cuda.device(out.device):
        check(lib().ns_conv_nhwc_f16_fused(src_arr, ch_arr, st_arr, n, N, H, W, ptr(layer.w), layer.ksize, layer.cout, ptr(b),
                                           C.c_long(bstride), ACT[act], ptr(out), int(out.shape[3]), int(out_offset), mode,
                                           ptr(e0), e0s, ptr(e1), e1s, stream_ptr()), "conv_nhwc_f16")
    return out


def flow_im2col(flow):
    """[E,4,ht,wd] f32 motion features -> [E,ht,wd,208] f16 patches of the flow encoder's 7x7 convolution (ns_flow_im2col)"""
    require_cuda(flow)
    if flow.dtype != torch.float32 or flow.dim() != 4 or flow.shape[1] != 4 or not flow.is_contiguous():
        raise RuntimeError("flow_im2col: expects a contiguous f32 [E,4,ht,wd] tensor")
    E, _, ht, wd = flow.shape
    out = torch.empty((E, ht, wd, 208), dtype=torch.float16, device=flow.device)
    with torch.cuda.device(flow.device):
        check(lib().ns_flow_im2col(ptr(flow), ptr(out), E, ht, wd, stream_ptr()), "flow_im2col")
    return out


def planes_to_nhwc(x, cp):
    """[E,C,ht,wd] f16 planes (the lookup's layout) -> channels-last [E,ht,wd,cp], channels C..cp-1 zero (ns_planes_to_nhwc_f16)"""
    require_cuda(x)
    if x.dtype != torch.float16 or x.dim() != 4 or not x.is_contiguous():
        raise RuntimeError("planes_to_nhwc: expects a contiguous f16 [E,C,ht,wd] tensor")
    E, Cc, ht, wd = x.shape
    out = torch.empty((E, ht, wd, cp), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().ns_planes_to_nhwc_f16(ptr(x), ptr(out), E, Cc, cp, ht * wd, stream_ptr()), "planes_to_nhwc_f16")
    return out


def group_mean(src, groups_host):
    """src: channels-last f16 [E,ht,wd,C] (dense or a channel slice); groups_host: the group id of every row of src (host
    ints, e.g. the source keyframe of every edge) -> ([K,ht,wd,C] f16 means in the order of np.unique(groups), K)"""
    src, stride = _slice_operand(src, *src.shape[:3])
    E, ht, wd, Cc = src.shape
    uniq, ix = np.unique(np.asarray(groups_host), return_inverse=True)
    K = len(uniq)
    order = np.argsort(ix, kind="stable").astype(np.int32)
    starts = np.concatenate([[0], np.cumsum(np.bincount(ix, minlength=K))]).astype(np.int32)
    csr = torch.from_numpy(np.concatenate([starts, order])).to(src.device)
    out = torch.empty((K, ht, wd, Cc), dtype=torch.float16, device=src.device)
    with torch.cuda.device(src.device):
        check(lib().ns_group_mean_nhwc_f16(ptr(src), stride, ptr(csr), C.c_void_p(csr.data_ptr() + 4 * (K + 1)), ptr(out), K,
                                           ht * wd, Cc, stream_ptr()), "group_mean_nhwc_f16")
    return out, K


def gru_glo_bias(wg, net, glo_w, glo_b):
    """ConvGRU global context -> per-edge biases of the gate convolutions (csrc/conv.hip: ns_gru_glo_bias; networks/modules/
    gru.py:25-33): wg = sigmoid(w(net)), net: dense channels-last f16 [E,ht,wd,128]; glo_w f32 [128,nout]; glo_b f32 [nout]
    -> f32 [E,nout] = mean_p(wg * net) @ glo_w + glo_b."""
    require_cuda(wg, net, glo_w, glo_b)
    E, ht, wd, Cc = net.shape
    if Cc != 128 or wg.shape != net.shape or wg.dtype != torch.float16 or net.dtype != torch.float16 or \
            not wg.is_contiguous() or not net.is_contiguous():
        raise RuntimeError("gru_glo_bias: wg / net must be contiguous channels-last f16 [E,ht,wd,128] tensors of one shape")
    if glo_w.dtype != torch.float32 or glo_w.dim() != 2 or glo_w.shape[0] != 128 or not glo_w.is_contiguous() or \
            glo_b.dtype != torch.float32 or tuple(glo_b.shape) != (glo_w.shape[1],) or not glo_b.is_contiguous():
        raise RuntimeError("gru_glo_bias: glo_w must be a contiguous f32 [128,nout] tensor, glo_b f32 [nout]")
    nout = int(glo_w.shape[1])
    P = int(lib().ns_gru_glo_parts(ht * wd))
    partial = torch.empty((E, P, 128), dtype=torch.float32, device=net.device)
    out = torch.empty((E, nout), dtype=torch.float32, device=net.device)
    with torch.cuda.device(net.device):
        check(lib().ns_gru_glo_bias(ptr(wg), ptr(net), ptr(glo_w), ptr(glo_b), ptr(partial), ptr(out), E, ht * wd, nout, stream_ptr()),
              "gru_glo_bias")
    return out
