"""The tracker's feature / context encoders (networks/modules/extractor.py:118-198 `BasicEncoder`, built by
networks/droid_net.py:157-158 with norm_fn "instance" / "none") on the MFMA convolution of `csrc/conv.hip`.

Same arithmetic as `nerfslam.droid_nets.BasicEncoder` (whose weights it is built from; f16 activations, f32 accumulation,
instance-norm statistics in f32/f64), different plumbing -- see csrc/encoder.hip:
  * the image normalisation, the 7x7 / stride-2 stem and both stride-2 3x3 convolutions are im2col + a 1x1 MFMA launch;
    a block's stride-2 1x1 shortcut reads the centre-tap slice of the same patch buffer;
  * InstanceNorm2d + relu (+ the residual add + relu of a block's tail) is a statistics pass and ONE apply pass instead of
    three batch-norm kernels, a clamp and an add; the context encoder (no norm) has bias + relu in the convolution's epilogue;
  * a call is ~50 launches of fixed shape.  `use_graph=True` (what nerfslam.droid_nets.DroidNetworks asks for since round 5)
    captures them once per image size in a HIP graph and replays it: alone on the device a replay takes the same 0.68 ms as the
    eager launches (the kernels, not the launches, are the time), but in the pipeline -- the tracker thread issuing ~130
    launches per frame next to the mapper thread -- it shortens the tracking leg by 0.2 ms per frame (6 of 6 paired runs).
The output is channels-last [N, H/8, W/8, C] f16.
"""
import ctypes as C

import torch

from ._lib import NerfSlamHipError, capture_lock, check, graph_capture, lib, ptr, require_cuda, stream_ptr
from .conv import PackedConv

EPS = 1e-5   # nn.InstanceNorm2d default
MAX_STAT_LAUNCHES, MAX_IMAGES = 32, 4096    # rows / columns of an encoder's arrival counters (BasicEncoder: 17 launches; csrc ENC_IN_MAXN)


def _half_out(n):
    return (n - 1) // 2 + 1


def _as_1x1(conv, pad_to=None):
    """a k x k convolution as the 1x1 over tap-major patches [(ky, kx), cin]"""
    co, ci, kh, kw = conv.weight.shape
    w = conv.weight.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci, 1, 1)
    return PackedConv(w, conv.bias, pad_cin_to=pad_to)


class _Block:
    def __init__(self, rb):
        self.stride2 = rb.downsample is not None
        self.conv1 = _as_1x1(rb.conv1) if self.stride2 else PackedConv(rb.conv1.weight, rb.conv1.bias)
        self.conv2 = PackedConv(rb.conv2.weight, rb.conv2.bias)
        self.down = PackedConv(rb.downsample[0].weight, rb.downsample[0].bias) if self.stride2 else None


class HipEncoder:
    """enc: a `BasicEncoder`; norm: True for instance norm (feature net), False for none (context net)."""

    def __init__(self, enc, norm, mean, std, use_graph=False):
        self.norm = bool(norm)
        self.stem = _as_1x1(enc.conv1, pad_to=160)
        self.blocks = [_Block(rb) for layer in (enc.layer1, enc.layer2, enc.layer3) for rb in layer]
        self.head = PackedConv(enc.conv2.weight, enc.conv2.bias)
        self.cout = self.head.cout
        self.mean = (C.c_float * 3)(*[float(v) for v in mean])
        self.std = (C.c_float * 3)(*[float(v) for v in std])
        self.use_graph = bool(use_graph)
        self._graphs = {}            # (N, H, W, dtype) -> (graph, static image, static output)
        self._tickets = None         # [statistics launches per call, N] int32 arrival counters of enc_in_stats (zeroed once)
        self._stat_calls = 0

    # ---- the pieces ----
    def _stats(self, y):
        N, H, W, Cc = y.shape
        part = torch.empty((N, int(lib().ns_enc_in_parts(H * W)), 2, Cc), dtype=torch.float32, device=y.device)
        # arrival counters: one row per statistics launch of a call (csrc/encoder.hip) -- rows are handed out in call order, so
        # a row is only ever reused by the SAME layer of this encoder, which stream order keeps apart (also under graph replay)
        k = self._stat_calls
        self._stat_calls += 1
        if self._tickets is None or self._tickets.device != y.device:
            self._tickets = torch.zeros((MAX_STAT_LAUNCHES, MAX_IMAGES), dtype=torch.int32, device=y.device)   # 512 KB, once
        if k >= MAX_STAT_LAUNCHES or N > MAX_IMAGES:
            raise NerfSlamHipError(f"HipEncoder: {k + 1} statistics launches / {N} images in one call (limits {MAX_STAT_LAUNCHES} / {MAX_IMAGES})")
        check(lib().ns_enc_in_stats(ptr(y), ptr(part), ptr(self._tickets[k]), N, H * W, Cc, stream_ptr()), "enc_in_stats")
        return part

    def _apply(self, y, ystats, x=None, xstats=None):
        N, H, W, Cc = y.shape
        out = torch.empty_like(y)
        check(lib().ns_enc_in_apply(ptr(y), ptr(ystats), ptr(x), ptr(xstats), ptr(out), N, H * W, Cc, C.c_float(EPS), stream_ptr()),
              "enc_in_apply")
        return out

    def _norm_relu(self, layer, srcs):
        """relu(norm(conv(srcs)))"""
        if not self.norm:
            return layer(srcs, act="relu")
        y = layer(srcs)
        return self._apply(y, self._stats(y))

    def _block(self, b, x):
        if b.stride2:
            N, H, W, Cc = x.shape
            patches = torch.empty((N, _half_out(H), _half_out(W), 9 * Cc), dtype=torch.float16, device=x.device)
            check(lib().ns_enc_im2col_3x3s2(ptr(x), ptr(patches), N, H, W, Cc, stream_ptr()), "enc_im2col_3x3s2")
            y = self._norm_relu(b.conv1, [patches])
            d = b.down([patches[..., 4 * Cc:5 * Cc]])              # the 1x1 / stride-2 shortcut sees the centre taps
        else:
            y = self._norm_relu(b.conv1, [x])
            d = x
        if not self.norm:
            y = b.conv2([y], act="relu")
            return self._apply(y, None, d, None)                   # relu(d + y)
        y = b.conv2([y])
        return self._apply(y, self._stats(y), d, self._stats(d) if b.stride2 else None)

    def _forward(self, img):
        N, _, H, W = img.shape
        self._stat_calls = 0
        patches = torch.empty((N, _half_out(H), _half_out(W), 160), dtype=torch.float16, device=img.device)
        check(lib().ns_enc_stem_im2col(ptr(img), 1 if img.dtype == torch.uint8 else 0, ptr(patches), N, H, W, self.mean, self.std,
                                       stream_ptr()), "enc_stem_im2col")
        x = self._norm_relu(self.stem, [patches])
        for b in self.blocks:
            x = self._block(b, x)
        return self.head([x])

    @torch.no_grad()
    def __call__(self, img):
        """img [N,3,H,W] uint8, or float32 holding 0..255 -> [N, H/8, W/8, cout] f16 channels-last"""
        require_cuda(img)
        if img.dim() != 4 or img.shape[1] != 3 or img.dtype not in (torch.uint8, torch.float32):
            raise NerfSlamHipError("HipEncoder: expects a uint8 or float32 [N,3,H,W] image tensor")
        img = img.contiguous()
        with torch.cuda.device(img.device), torch.autocast("cuda", enabled=False):
            if not self.use_graph:
                return self._forward(img)
            key = (tuple(img.shape), img.dtype)
            entry = self._graphs.get(key)
            if entry is None:
                static_in = img.clone()
                self._forward(static_in)                                   # warm-up outside the capture (lazy initialisation)
                # (the lock the mapper thread holds around ITS captures and the tracker around its host read-backs: a
                #  synchronising call of one thread while another captures is an illegal-state error on ROCm 7.2, _lib.py)
                with capture_lock:
                    torch.cuda.current_stream().synchronize()
                    g = torch.cuda.CUDAGraph()
                    with graph_capture(g, capture_error_mode="thread_local"):
                        static_out = self._forward(static_in)
                entry = self._graphs[key] = (g, static_in, static_out)
            g, static_in, static_out = entry
            static_in.copy_(img)
            g.replay()
            return static_out.clone()
