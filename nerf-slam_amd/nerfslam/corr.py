"""CorrBlock / AltCorrBlock -- host mirror of networks/modules/corr.py (reference lines cited inline).

Same class names, constructor arguments, call signatures and output layouts as the reference, so
`RaftVisualFrontend`-style callers work unchanged; the arithmetic runs in the HIP kernels of
libnerfslam_hip.so:
  * pyramid construction  -> torch.matmul for the plain GEMM (hipBLASLt) + ns_corr_pool2x2,
    or the fused ns_corr_volume_pyramid kernel when available;
  * lookup                -> ns_corr_lookup_pyramid: all four levels in one launch, reading the
    frontend's native [.., ht, wd, 2] coordinate layout and writing the concatenated
    [.., 196, ht, wd] tensor directly (the reference does 4 launches + permute + cat).
"""
import ctypes as C
import os as _os

import torch

from ._lib import NerfSlamHipError, check, lib, ptr, require_cuda, stream_ptr, variant_env


class CorrBlock:
    """networks/modules/corr.py:23-60."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, fused=None, tiled=False):
        """fmap1, fmap2 [batch, num, 128, ht, wd] (reference layout, corr.py:23-38).

        tiled=True (fused path only) keeps levels 0 and 1 in the private 8x8-tiled slice layout of
        ns_corr_volume_pyramid: same values and the same lookups bit for bit, ~1.4x less lookup traffic, but
        `corr_pyramid[0/1]` then have shape [E, HW, tiles, 64] instead of the reference's [E, ht, wd, h, w]
        (`untiled()` converts).

        fused=None picks the single-launch MFMA kernel (ns_corr_volume_pyramid) whenever it applies
        (f16, 128 channels, <= 4 levels); fused=False forces the reference's own decomposition
        (library GEMM + one pooling launch per level)."""
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = []
        self.tiled = False
        self.hw = None
        if fmap1 is None:
            return
        require_cuda(fmap1, fmap2)
        batch, num, dim, ht, wd = fmap1.shape
        can_fuse = fmap1.dtype == torch.float16 and dim == 128 and 1 <= num_levels <= 4
        if fused is None:
            fused = can_fuse
        if fused:
            if not can_fuse:
                raise NerfSlamHipError("CorrBlock(fused=True) needs float16 features with 128 channels")
            # channels-last, pre-divided by 4 exactly as corr.py:67-68 does (in half)
            f1 = (fmap1.reshape(batch * num, dim, ht * wd) / 4.0).transpose(1, 2).contiguous()
            f2 = (fmap2.reshape(batch * num, dim, ht * wd) / 4.0).transpose(1, 2).contiguous()
            self.corr_pyramid = CorrBlock.build_pyramid(f1, f2, None, None, batch * num, ht, wd, num_levels, tiled=tiled)
            self.tiled, self.hw = bool(tiled), (ht, wd)
            return
        # all pairs correlation (corr.py:63-72): both operands / 4, matmul in the features' dtype
        corr = CorrBlock.corr(fmap1, fmap2)
        corr = corr.reshape(batch * num, ht, wd, ht, wd)
        self.corr_pyramid.append(corr)
        h, w = ht, wd
        for _ in range(1, num_levels):  # corr.py:35-38
            src = self.corr_pyramid[-1]
            dst = torch.empty((batch * num, ht, wd, h // 2, w // 2), dtype=src.dtype, device=src.device)
            if src.dtype != torch.float16:
                raise NerfSlamHipError("CorrBlock: the pyramid kernels are built for float16 volumes "
                                       "(the reference builds them under autocast)")
            with torch.cuda.device(src.device):
                check(lib().ns_corr_pool2x2(ptr(src), ptr(dst), C.c_long(batch * num * ht * wd), h, w, stream_ptr()),
                      "corr_pool2x2")
            self.corr_pyramid.append(dst)
            h, w = h // 2, w // 2

    @staticmethod
    def build_pyramid(f1, f2, ii, jj, E, ht, wd, num_levels=4, tiled=False):
        """f1, f2: channels-last feature banks [n, ht*wd, 128] f16 already divided by 4; ii, jj: optional
        int64 frame ids (None: edge e uses row e).  One launch, every output byte written once."""
        dev = f1.device
        def level(l):
            h, w = ht >> l, wd >> l
            if tiled and l < 2:
                return torch.empty((E, ht * wd, ((h + 7) // 8) * ((w + 7) // 8), 64), dtype=torch.float16, device=dev)
            return torch.empty((E, ht, wd, h, w), dtype=torch.float16, device=dev)
        pyr = [level(l) for l in range(num_levels)]
        arr = (C.c_void_p * 4)(*[pyr[min(l, num_levels - 1)].data_ptr() for l in range(4)])
        with torch.cuda.device(dev):
            check(lib().ns_corr_volume_pyramid(ptr(f1), ptr(f2), ptr(ii), ptr(jj), arr, num_levels, E, f1.shape[-1], ht,
                                               wd, 1 if tiled else 0, stream_ptr()), "corr_volume_pyramid")
        return pyr

    @classmethod
    def from_pyramid(cls, pyramid, radius=3, tiled=False, hw=None):
        blk = cls(None, None, num_levels=len(pyramid), radius=radius)
        blk.corr_pyramid = list(pyramid)
        blk.tiled, blk.hw = bool(tiled), hw
        return blk

    def untiled(self):
        """the pyramid in the reference's layout [E, ht, wd, h, w] (copies levels 0 / 1 when they are tiled)"""
        if not self.tiled:
            return list(self.corr_pyramid)
        ht, wd = self.hw
        out = []
        for l, p in enumerate(self.corr_pyramid):
            h, w = ht >> l, wd >> l
            if l < 2:
                nty, ntx = (h + 7) // 8, (w + 7) // 8
                p = p.view(-1, ht, wd, nty, ntx, 8, 8).permute(0, 1, 2, 3, 5, 4, 6).reshape(-1, ht, wd, nty * 8, ntx * 8)
                p = p[..., :h, :w].contiguous()
            out.append(p)
        return out

    def __call__(self, coords):
        """coords [batch, num, ht, wd, 2] float -> [batch, num, num_levels*49, ht, wd] (corr.py:40-50)."""
        batch, num, ht, wd, _ = coords.shape
        E = batch * num
        p0 = self.corr_pyramid[0]
        require_cuda(coords)
        if self.radius != 3 or p0.dtype != torch.float16 or self.num_levels > 4:
            if self.tiled:
                raise NerfSlamHipError("CorrBlock: the tiled layout is read by the fused radius-3 lookup only")
            return self._call_per_level(coords)
        coords = coords.contiguous().float()
        out = torch.empty((batch, num, self.num_levels * 49, ht, wd), dtype=torch.float16, device=coords.device)
        for p in self.corr_pyramid:
            if not p.is_contiguous():
                raise RuntimeError("volume must be contiguous")
        arr = (C.c_void_p * 4)(*[self.corr_pyramid[min(l, self.num_levels - 1)].data_ptr() for l in range(4)])
        with torch.cuda.device(coords.device):
            check(lib().ns_corr_lookup_pyramid(arr, self.num_levels, ptr(coords), 1, ptr(out), E, ht, wd,
                                               1 if self.tiled else 0, stream_ptr()), "corr_lookup_pyramid")
        return out

    def _call_per_level(self, coords):
        import droid_backends
        batch, num, ht, wd, _ = coords.shape
        c = coords.permute(0, 1, 4, 2, 3).contiguous().view(batch * num, 2, ht, wd)
        outs = []
        for i in range(self.num_levels):
            corr, = droid_backends.corr_index_forward(self.corr_pyramid[i], c / 2 ** i, self.radius)
            outs.append(corr.view(batch, num, -1, ht, wd))
        return torch.cat(outs, dim=2)

    def cat(self, other):  # corr.py:52-55
        if self.tiled != other.tiled:
            raise NerfSlamHipError("CorrBlock.cat: both blocks must use the same volume layout")
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], 0)
        return self

    def __getitem__(self, index):  # corr.py:57-60
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index].contiguous()
        return self

    @staticmethod
    def corr(fmap1, fmap2):
        """all-pairs correlation (corr.py:63-72); plain library GEMM."""
        batch, num, dim, ht, wd = fmap1.shape
        fmap1 = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        fmap2 = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        corr = torch.matmul(fmap1.transpose(1, 2), fmap2)
        return corr.view(batch, num, ht, wd, ht, wd)


class EncodedCorr:
    """what CorrPool.lookup_encoded returns in place of the [1,E,196,ht,wd] lookup: the correlation encoder's first activation,
    relu(conv1x1(lookup)), channels-last f16 [E,ht,wd,128] in `.c1`.  An update operator that advertises `corr_encoder`
    (nerfslam.droid_nets.DroidNetworks on the HIP update operator) is handed this instead of the lookup."""

    def __init__(self, c1):
        self.c1 = c1


class CorrPool:
    """Slot-addressed store of per-edge correlation pyramids (8x8-tiled levels 0 / 1): what TrackingFrontend keeps instead of
    one growing / shrinking CorrBlock.

    The reference appends the volumes of new edges with `torch.cat` and drops those of removed edges with a boolean-mask
    copy (networks/modules/corr.py:52-60 called from visual_frontend.py:838-844, 868-892): at 640x480 every such call
    copies the whole active set (48 edges x 61 MB = 2.9 GB).  Here a volume is written once, into a free slot of a
    preallocated pool, and never moves: edges are (re)ordered by an int32 slot table that the build / lookup kernels read
    (ns_corr_volume_pyramid_slots, ns_corr_lookup_pyramid_slots).  Same values, same lookups bit for bit."""

    def __init__(self, ht, wd, capacity, device, num_levels=4):
        self.ht, self.wd, self.num_levels, self.device = ht, wd, num_levels, torch.device(device)
        self.capacity = 0
        self.levels = None
        self.grow(capacity)

    def _alloc(self, cap):
        out = []
        for l in range(self.num_levels):
            h, w = self.ht >> l, self.wd >> l
            shape = (cap, self.ht * self.wd, ((h + 7) // 8) * ((w + 7) // 8), 64) if l < 2 else (cap, self.ht, self.wd, h, w)
            out.append(torch.empty(shape, dtype=torch.float16, device=self.device))
        return out

    def grow(self, capacity):
        if capacity <= self.capacity:
            return
        new = self._alloc(capacity)
        if self.levels is not None:
            for a, b in zip(new, self.levels):
                a[:self.capacity] = b
        self.levels, self.capacity = new, capacity

    def _ptrs(self):
        return (C.c_void_p * 4)(*[self.levels[min(l, self.num_levels - 1)].data_ptr() for l in range(4)])

    def build(self, bank1, bank2, ii, jj, slots):
        """volumes of edges (ii[e], jj[e]) from the channels-last f16 feature banks (already / 4) into slots[e] (int32, device)"""
        E = int(slots.shape[0])
        if E == 0:
            return
        with torch.cuda.device(self.device):
            check(lib().ns_corr_volume_pyramid_slots(ptr(bank1), ptr(bank2), ptr(ii), ptr(jj), self._ptrs(), self.num_levels, E,
                                                     bank1.shape[-1], self.ht, self.wd, 1, ptr(slots), stream_ptr()),
                  "corr_volume_pyramid_slots")

    def lookup(self, coords, slots, out=None):
        """coords [1, E, ht, wd, 2] f32, slots [E] int32 (device) -> [1, E, 196, ht, wd] f16 (corr.py:40-50 on the pooled volumes)"""
        batch, E, ht, wd, _ = coords.shape
        coords = coords.contiguous().float()
        if out is None:
            out = torch.empty((batch, E, self.num_levels * 49, ht, wd), dtype=torch.float16, device=coords.device)
        with torch.cuda.device(self.device):
            check(lib().ns_corr_lookup_pyramid_slots(self._ptrs(), self.num_levels, ptr(coords), 1, ptr(out), batch * E, ht, wd, 1,
                                                     ptr(slots), self.capacity, stream_ptr()), "corr_lookup_pyramid_slots")
        return out

    def lookup_encoded(self, coords, slots, enc, out=None):
        """lookup + the update operator's correlation encoder (Conv2d(196,128,1) + ReLU, networks/droid_net.py:83-87) in one
        launch (csrc/corr_lookup.hip: corr_lookup_enc_kernel): coords [1, E, ht, wd, 2] f32, slots [E] int32, enc = the
        `CorrEncoderWeights` of nerfslam.update_op -> `EncodedCorr` around [E, ht, wd, 128] f16 channels-last.  The
        [1, E, 196, ht, wd] tensor of lookup() is never written."""
        if self.num_levels != 4:
            raise NerfSlamHipError("CorrPool.lookup_encoded: the encoder reads the 196 channels of a four-level pyramid")
        batch, E, ht, wd, _ = coords.shape
        coords = coords.contiguous().float()
        if out is None:
            out = torch.empty((batch * E, ht, wd, 128), dtype=torch.float16, device=coords.device)
        with torch.cuda.device(self.device):
            check(lib().ns_corr_lookup_encode_slots(self._ptrs(), ptr(coords), 1, ptr(enc.frags), ptr(enc.bias), ptr(out), batch * E,
                                                    ht, wd, 1, ptr(slots), self.capacity, stream_ptr()), "corr_lookup_encode_slots")
        return EncodedCorr(out)

    def block(self, slots):
        """the volumes of `slots` as a CorrBlock in the reference's layout (copies; tests / debugging)"""
        idx = slots.long()
        blk = CorrBlock.from_pyramid([lv[idx] for lv in self.levels], tiled=True, hw=(self.ht, self.wd))
        return CorrBlock.from_pyramid(blk.untiled())


class AltCorrBlock:
    """On-the-fly correlation for the global BA (reference: networks/modules/corr.py:92-140).

    Same constructor and call signature as the reference class.  Differences in mechanism only:
    the feature pyramid is kept as frame-indexed channels-last f32 tensors, and one call runs every
    level for every edge in a single launch that reads frames through (ii, jj) -- the reference
    gathers and `.float()`-copies both feature maps per edge and per level (corr.py:114-121) and
    concatenates four outputs."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        if radius != 3 or not 1 <= num_levels <= 4:
            raise NerfSlamHipError("AltCorrBlock: built for radius 3 and at most 4 levels (the reference's setting)")
        require_cuda(fmaps)
        self.num_levels, self.radius = num_levels, radius
        B, N, Cc, H, W = fmaps.shape
        if B != 1:
            raise NerfSlamHipError("AltCorrBlock: batch size 1 only (visual_frontend.py:479 always passes 1)")
        self.shape = (N, Cc, H, W)
        # The reference keeps the pyramid in the dtype of the features it is given (corr.py:96-105: `/ 4.0` and avg_pool2d stay
        # in that dtype; only the kernel call casts to float, :121).  Half features (what RaftVisualFrontend holds,
        # visual_frontend.py:209) therefore give a HALF pyramid, rounded to half after every pooling -- kept that way here and
        # correlated on the matrix cores (csrc/altcorr.hip: altcorr_tile_mfma_kernel); anything else takes the f32 kernels.
        self.half = fmaps.dtype == torch.float16 and Cc == 128 and not variant_env("NS_ALTCORR_F32")
        level = fmaps.reshape(N, Cc, H, W)
        level = (level / 4.0) if self.half else (level.float() / 4.0)  # corr.py:98
        self.pyramid = []
        for l in range(num_levels):
            self.pyramid.append(level.permute(0, 2, 3, 1).contiguous())  # [N, h, w, C]
            if l + 1 < num_levels:
                level = torch.nn.functional.avg_pool2d(level, 2, stride=2)   # corr.py:105

    def __call__(self, coords, ii, jj):
        """coords [1, E, H, W, 2] (or [1, E, H, W, S, 2]); ii, jj [E] frame ids
        -> [1, E, 196, H, W] (or [1, E, 196, H, W, S])."""
        N, Cc, H, W = self.shape
        squeeze = coords.dim() == 5
        if squeeze:
            coords = coords.unsqueeze(-2)
        _, E, _, _, S, _ = coords.shape
        ii = torch.as_tensor(ii, dtype=torch.long, device=coords.device).contiguous()
        jj = torch.as_tensor(jj, dtype=torch.long, device=coords.device).contiguous()
        outs = []
        arr = (C.c_void_p * 4)(*[self.pyramid[min(l, self.num_levels - 1)].data_ptr() for l in range(4)])
        for s in range(S):
            cs = coords[0, :, :, :, s].contiguous().float()
            out = torch.empty((E, self.num_levels * 49, H, W), dtype=torch.float32, device=coords.device)
            with torch.cuda.device(coords.device):
                fn = lib().ns_altcorr_pyramid_f16 if self.half else lib().ns_altcorr_pyramid
                check(fn(arr, self.num_levels, ptr(ii), ptr(jj), ptr(cs), ptr(out), E, H, W, Cc, stream_ptr()), "altcorr_pyramid")
            outs.append(out)
        out = outs[0][None] if squeeze else torch.stack(outs, dim=-1)[None]
        return out.contiguous()

    def encoded(self, coords, ii, jj, enc):
        """on-the-fly correlation + the update operator's correlation encoder (Conv2d(196,128,1) + ReLU, networks/droid_net.py:
        83-87) in one launch (csrc/altcorr.hip: altcorr_tile_enc_kernel): coords [1, E, H, W, 2], enc = `CorrEncoderWeights` of
        nerfslam.update_op -> `EncodedCorr` around [E, H, W, 128] f16 channels-last; the [1, E, 196, H, W] f32 tensor of
        __call__ is never written.  Half pyramids of four levels only (what RaftVisualFrontend holds)."""
        N, Cc, H, W = self.shape
        if not self.half or self.num_levels != 4 or coords.dim() != 5:
            raise NerfSlamHipError("AltCorrBlock.encoded: needs the half pyramid of four levels and coords [1,E,H,W,2]")
        E = coords.shape[1]
        ii = torch.as_tensor(ii, dtype=torch.long, device=coords.device).contiguous()
        jj = torch.as_tensor(jj, dtype=torch.long, device=coords.device).contiguous()
        cs = coords[0].contiguous().float()
        out = torch.empty((E, H, W, 128), dtype=torch.float16, device=coords.device)
        arr = (C.c_void_p * 4)(*[self.pyramid[l].data_ptr() for l in range(4)])
        with torch.cuda.device(coords.device):
            check(lib().ns_altcorr_pyramid_encode_f16(arr, ptr(ii), ptr(jj), ptr(cs), ptr(enc.frags), ptr(enc.bias), ptr(out), E, H, W,
                                                      Cc, stream_ptr()), "altcorr_pyramid_encode_f16")
        return EncodedCorr(out)
