"""The tracker's update operator (networks/droid_net.py:78-150 `UpdateModule`, networks/modules/gru.py:5-34 `ConvGRU`,
`GraphAgg`) on the MFMA convolution of `csrc/conv.hip`, channels-last end to end.

Same arithmetic as `nerfslam.droid_nets.UpdateModule` (whose weights it is built from), different plumbing:
  * every 3x3 / 1x1 convolution is one `ns_conv_nhwc_f16` launch with bias and activation fused;
  * no `torch.cat`: the ConvGRU reads [net | inp | corr features | flow features] as a list of tensors, and the encoders
    write their halves of one [E,ht,wd,192] buffer;
  * convz | convr are one 448 -> 256 convolution whose epilogue also forms r * net, and the convq launch's epilogue is
    the GRU blend (1 - z) net + z q: no elementwise kernels between the gates; the global-context terms conv*_glo(glo) are 1x1 convolutions of a
    per-edge vector, i.e. a per-image bias of that launch (`gru_glo_bias`: pixel mean of sigmoid(w(net)) * net and the [E,128] x
    [128,384] product for all three in two small launches);
  * the first convolutions of the delta head, the weight head and GraphAgg share their input: one 128 -> 384 launch,
    whose channel slices feed the second convolutions directly.
The 7x7 convolution of the flow encoder (4 input channels) is an im2col kernel + a 1x1 launch over 208 channels.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .conv import PackedConv, flow_im2col, group_mean, gru_glo_bias, planes_to_nhwc


class CorrEncoderWeights:
    """Conv2d(196,128,1) of the correlation encoder as the MFMA fragments of the lookup-fused kernel (include/nerfslam_hip.h:
    ns_corr_lookup_encode_slots): frags f16 [4][13][64][8], element q of lane l of fragment (nt, c) =
    W[32 nt + (l & 31)][16 c + 8 (l >> 5) + q], inputs 196..207 zero; bias f32 [128]."""

    def __init__(self, weight, bias):
        w = torch.zeros((128, 208), dtype=torch.float32, device=weight.device)
        w[:, :196] = weight.detach().float().reshape(128, 196)
        self.frags = w.reshape(4, 32, 13, 2, 8).permute(0, 2, 3, 1, 4).contiguous().half()      # nt, c, h, i, q  (lane = 32 h + i)
        self.bias = bias.detach().float().contiguous()


class HipUpdateOperator:
    def __init__(self, um):
        P = PackedConv
        ce, fe, g, a = um.corr_encoder, um.flow_encoder, um.gru, um.agg
        self.corr1 = P(ce[0].weight, ce[0].bias, pad_cin_to=208)      # 196 lookup channels, padded to 13 chunks of 16
        self.corr_enc = CorrEncoderWeights(ce[0].weight, ce[0].bias)   # the same layer for the lookup-fused kernel
        self.corr2 = P(ce[2].weight, ce[2].bias)
        self.flow1 = P(fe[0].weight.reshape(128, 196, 1, 1), fe[0].bias, pad_cin_to=208)   # 7x7 as a 1x1 over im2col patches
        self.flow2 = P(fe[2].weight, fe[2].bias)
        self.gw = P(g.w.weight, g.w.bias)
        self.zr = P.from_modules(g.convz, g.convr)
        self.q = P(g.convq.weight, g.convq.bias)
        # conv*_glo on the [E,128,1,1] context vector = one [E,128] x [128,384] product; the convolutions' own biases folded in
        self.glo_w = torch.cat([m.weight.detach().float().reshape(128, 128) for m in (g.convz_glo, g.convr_glo, g.convq_glo)], 0).t().contiguous()
        self.glo_b = torch.cat([g.convz_glo.bias + g.convz.bias, g.convr_glo.bias + g.convr.bias,
                                g.convq_glo.bias + g.convq.bias]).detach().float()
        self.heads = P.from_modules(um.delta[0], um.weight[0], a.conv1)
        self.delta1 = P(um.delta[0].weight, um.delta[0].bias)            # delta head alone (motion filter: delta_only)
        self.delta2 = P(um.delta[2].weight, um.delta[2].bias)
        self.weight2 = P(um.weight[2].weight, um.weight[2].bias)
        self.agg2 = P(a.conv2.weight, a.conv2.bias)
        self.eta = P(a.eta[0].weight, a.eta[0].bias)
        self.upmask = P(a.upmask[0].weight, a.upmask[0].bias)

    @torch.no_grad()
    def __call__(self, net, inp, corr, flow, ii_host):
        """net, inp [E,ht,wd,128] f16 channels-last; corr [E,196,ht,wd] f16 (the lookup's layout) or an `EncodedCorr` (lookup and
        first encoder convolution already fused: nerfslam.corr.CorrPool.lookup_encoded); flow [E,4,ht,wd] f32;
        ii_host: source keyframe of every edge (host ints).
        -> net' [E,ht,wd,128] f16, delta [E,ht,wd,2] f32, weight [E,ht,wd,2] f32, eta [k,ht,wd] f32, upmask [k,ht,wd,576] f16
        (channels-last; k = number of distinct source keyframes, in sorted order)"""
        with torch.autocast("cuda", enabled=False):     # dtypes are explicit here; an enclosing autocast would only add casts
            return self._forward(net, inp, corr, flow, ii_host)

    @torch.no_grad()
    def delta_only(self, net, inp, corr, flow=None):
        """The motion filter's use of the operator (visual_frontend.py:976-1007: only the flow correction of one edge is
        looked at): encoders + ConvGRU + delta head, no weight head / GraphAgg, no host work.  flow None = zero motion features."""
        with torch.autocast("cuda", enabled=False):
            if flow is None:
                flow = torch.zeros((net.shape[0], 4, net.shape[1], net.shape[2]), dtype=torch.float32, device=net.device)
            net2 = self._gru(net, inp, corr, flow)
            return self.delta2([self.delta1([net2], act="relu")]).float()

    def _forward(self, net, inp, corr, flow, ii_host):
        net2 = self._gru(net, inp, corr, flow)
        # ---- heads ----
        hd = self.heads([net2], act="relu")                                   # [delta | weight | agg] x 128
        delta = self.delta2([hd[..., :128]]).float()
        weight = self.weight2([hd[..., 128:256]], act="sigmoid").float()
        # ---- GraphAgg: mean over the edges of each source keyframe, conv, eta + upsampling mask ----
        mean, k = group_mean(hd[..., 256:], ii_host)
        x2 = self.agg2([mean], act="relu")
        eta = 0.01 * F.softplus(self.eta([x2]).float())[..., 0]
        upmask = self.upmask([x2])                                            # channels-last: ns_cvx_upsample_keyframes_nhwc reads it as is
        return net2, delta, weight, eta, upmask

    def _gru(self, net, inp, corr, flow):
        E, ht, wd, _ = net.shape
        dev = net.device
        # ---- encoders: X = [corr features 128 | flow features 64] ----
        if hasattr(corr, "c1"):      # nerfslam.corr.EncodedCorr: the lookup kernel already applied corr1 + ReLU
            c1 = corr.c1
        else:
            c1 = self.corr1([planes_to_nhwc(corr.contiguous(), 208)], act="relu")
        X = torch.empty((E, ht, wd, 192), dtype=torch.float16, device=dev)
        self.corr2([c1], act="relu", out=X, out_offset=0)
        f1 = self.flow1([flow_im2col(flow.float().contiguous())], act="relu")
        self.flow2([f1], act="relu", out=X, out_offset=128)
        # ---- ConvGRU ----
        wg = self.gw([net], act="sigmoid")
        # [E,384] per-edge biases of z | r | q = mean_p(wg * net) @ glo_w + glo_b: two small launches (csrc/conv.hip) instead of an
        # elementwise product (a third [E,ht,wd,128] tensor), a reduction and a library GEMM
        gb = gru_glo_bias(wg, net.contiguous(), self.glo_w, self.glo_b.contiguous())
        zrh = self.zr([net, inp, X], act="sigmoid", bias=gb[:, :256].contiguous(), fuse=("mul_hi", net))   # [z | r * net]
        net2 = self.q([zrh[..., 128:], inp, X], act="tanh", bias=gb[:, 256:].contiguous(),
                      fuse=("gru", zrh[..., :128], net))                      # (1 - z) net + z q
        return net2
