"""The learned components of the tracker -- feature / context encoders and the ConvGRU update operator
 -- as plain torch modules running on MIOpen / hipBLASLt under autocast.

These are NOT hot-path kernels of this project: they exist so that `TrackingSLAM` runs end to end and so that a
published DROID-SLAM checkpoint (`droid.pth`) drops in.  The layer layout and the PARAMETER NAMES follow the
reference (/root/reference/networks/droid_net.py:44-160, networks/modules/extractor.py:5-60,118-185,
networks/modules/gru.py:5-34), because the checkpoint is keyed by them; `load_weights` applies the reference's key
remapping (visual_frontend.py:1051-1068).  tests/test_droid_nets.py pins the forward passes to outputs generated
from the reference's own modules (tools/gen_golden.py, section 4).

`DroidNetworks` adapts the modules to the callable interface of nerfslam.slam.TrackingSLAM and owns the per-edge
ConvGRU hidden states (the reference keeps them in the frontend, visual_frontend.py:838-862,868-892).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _norm(kind, planes):
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)  # the tracker only instantiates these two (droid_net.py:157-158)


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.norm1, self.norm2 = _norm(norm_fn, planes), _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class BasicEncoder(nn.Module):
    """stride-8 residual encoder: 7x7/2 stem (32) -> 2 blocks @32 -> 2 @64 (/2) -> 2 @128 (/2) -> 1x1 to output_dim"""

    def __init__(self, output_dim=128, norm_fn="instance"):
        super().__init__()
        dim = 32
        self.norm1 = _norm(norm_fn, dim)
        self.conv1 = nn.Conv2d(3, dim, 7, stride=2, padding=3)
        stages, cin = [], dim
        for cout, stride in ((dim, 1), (2 * dim, 2), (4 * dim, 2)):
            stages.append(nn.Sequential(ResidualBlock(cin, cout, norm_fn, stride), ResidualBlock(cout, cout, norm_fn, 1)))
            cin = cout
        self.layer1, self.layer2, self.layer3 = stages
        self.conv2 = nn.Conv2d(4 * dim, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        b, n, c, h, w = x.shape
        x = F.relu(self.norm1(self.conv1(x.reshape(b * n, c, h, w))))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return x.view(b, n, *x.shape[1:])


class ConvGRU(nn.Module):
    """3x3 conv GRU with a global-context gate term (gru.py:5-34)"""

    def __init__(self, h_planes=128, i_planes=128):
        super().__init__()
        for name in ("convz", "convr", "convq"):
            setattr(self, name, nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1))
        self.w = nn.Conv2d(h_planes, h_planes, 1)
        for name in ("convz_glo", "convr_glo", "convq_glo"):
            setattr(self, name, nn.Conv2d(h_planes, h_planes, 1))

    def forward(self, net, *inputs):
        inp = torch.cat(inputs, 1)
        hx = torch.cat([net, inp], 1)
        glo = (torch.sigmoid(self.w(net)) * net).mean((2, 3), keepdim=True)
        z = torch.sigmoid(self.convz(hx) + self.convz_glo(glo))
        r = torch.sigmoid(self.convr(hx) + self.convr_glo(glo))
        q = torch.tanh(self.convq(torch.cat([r * net, inp], 1)) + self.convq_glo(glo))
        return (1 - z) * net + z * q


class GraphAgg(nn.Module):
    """per-source-frame aggregation of the hidden states -> damping eta and the 8x convex-upsampling mask"""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(128, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 128, 3, padding=1)
        self.eta = nn.Sequential(nn.Conv2d(128, 1, 3, padding=1), nn.Identity(), nn.Softplus())
        self.upmask = nn.Sequential(nn.Conv2d(128, 8 * 8 * 9, 1))

    def forward(self, net, ii):
        b, num, ch, ht, wd = net.shape
        _, ix = torch.unique(ii, return_inverse=True)
        x = F.relu(self.conv1(net.reshape(b * num, ch, ht, wd))).view(b, num, 128, ht, wd)
        k = int(ix.max().item()) + 1
        s = torch.zeros((b, k, 128, ht, wd), dtype=x.dtype, device=x.device).index_add_(1, ix, x)
        cnt = torch.zeros(k, dtype=x.dtype, device=x.device).index_add_(0, ix, torch.ones_like(ix, dtype=x.dtype))
        x = F.relu(self.conv2((s / cnt.view(1, k, 1, 1, 1)).reshape(b * k, 128, ht, wd)))
        return 0.01 * self.eta(x).view(b, -1, ht, wd), self.upmask(x).view(b, -1, 8 * 8 * 9, ht, wd)


def _head(out):
    return [nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(128, out, 3, padding=1), nn.Identity()]


class UpdateModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.corr_encoder = nn.Sequential(nn.Conv2d(4 * 49, 128, 1), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True))
        self.flow_encoder = nn.Sequential(nn.Conv2d(4, 128, 7, padding=3), nn.ReLU(inplace=True),
                                          nn.Conv2d(128, 64, 3, padding=1), nn.ReLU(inplace=True))
        self.weight = nn.Sequential(*_head(2), nn.Sigmoid())
        self.delta = nn.Sequential(*_head(2))
        self.gru = ConvGRU(128, 128 + 128 + 64)
        self.agg = GraphAgg()

    def forward(self, net, inp, corr, flow=None, ii=None, jj=None):
        b, num, ch, ht, wd = net.shape
        if flow is None:
            flow = torch.zeros((b, num, 4, ht, wd), device=net.device, dtype=net.dtype)
        f = lambda t: t.reshape(b * num, -1, ht, wd)
        h = self.gru(f(net), f(inp), self.corr_encoder(f(corr)), self.flow_encoder(f(flow)))
        delta = self.delta(h).view(b, num, 2, ht, wd).permute(0, 1, 3, 4, 2).contiguous()
        weight = self.weight(h).view(b, num, 2, ht, wd).permute(0, 1, 3, 4, 2).contiguous()
        h = h.view(b, num, -1, ht, wd)
        if ii is None:
            return h, delta, weight
        eta, upmask = self.agg(h, ii.to(h.device))
        return h, delta, weight, eta, upmask


class DroidNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_net = BasicEncoder(128, "instance")
        self.context_net = BasicEncoder(256, "none")
        self.update_net = UpdateModule()

    def load_weights(self, path_or_state):
        """accepts a DROID-SLAM checkpoint (keys module.fnet / cnet / update, 3-channel heads) like
        visual_frontend.py:1051-1068 does"""
        sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
        out = OrderedDict()
        for k, v in sd.items():
            k = k.replace("module.", "")
            for old, new in (("fnet.", "feature_net."), ("cnet.", "context_net."), ("update.", "update_net.")):
                k = k.replace(old, new)
            out[k] = v
        for k in ("update_net.weight.2.weight", "update_net.weight.2.bias", "update_net.delta.2.weight",
                  "update_net.delta.2.bias"):
            out[k] = out[k][:2]
        self.load_state_dict(out)
        return self


class DroidNetworks:
    """Callable bundle for TrackingSLAM (`args.networks`): features / motion / update + keyframe hooks."""

    MEAN = (0.485, 0.456, 0.406)
    STD = (0.229, 0.224, 0.225)

    def __init__(self, device, weights=None, buffer=512, seed=0, hip_update=None, hip_encoders=None, encoder_graphs=True):
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.net = DroidNet()
        if weights:
            self.net.load_weights(weights)
        self.net = self.net.to(self.device).eval()
        # f16 copies of the two encoders: under autocast the f32 weights are cast to f16 on EVERY call (~60 tiny cast kernels
        # per encoder call); same arithmetic (f16 convolutions, instance-norm statistics in f32) without them
        self.fnet_h = self.cnet_h = None
        self.fnet_hip = self.cnet_hip = None
        if self.device.type == "cuda":
            import copy
            import os
            self.fnet_h = copy.deepcopy(self.net.feature_net).half()
            self.cnet_h = copy.deepcopy(self.net.context_net).half()
            # both encoders on the MFMA convolution (nerfslam/encoder_op.py); NS_TORCH_ENCODERS=1 keeps the MIOpen path (A/B runs)
            from ._lib import variant_env
            if (hip_encoders is None and not variant_env("NS_TORCH_ENCODERS")) or hip_encoders:
                from .encoder_op import HipEncoder
                # encoder_graphs: replay each encoder's ~50 fixed-shape launches from a HIP graph (captured at the first call per
                # image size; nerfslam/encoder_op.py).  Round 5, six bench runs per arm on one box: tracking leg 3.24 -> 3.04 ms
                # (6 of 6), pipeline 132.4 +- 2.4 -> 135.7 +- 1.5 frames/s (profiles/r05_ab_records.json): on by default.
                self.fnet_hip = HipEncoder(self.net.feature_net, True, self.MEAN, self.STD, use_graph=encoder_graphs)
                self.cnet_hip = HipEncoder(self.net.context_net, False, self.MEAN, self.STD, use_graph=encoder_graphs)
        self.ctx, self.inp = {}, {}          # per keyframe: tanh / relu halves of the context encoder
        self.hidden = {}                     # per edge (i, j): ConvGRU hidden state [128, ht, wd]
        self._stacked = None                 # ((ii, jj, keyframe epoch), stacked hidden states, stacked context) of the last update
        self._kf_epoch = 0                   # bumped whenever a keyframe's context features are (re)assigned or shifted
        self._pending = None
        # update operator on the MFMA convolution kernel (nerfslam/update_op.py); hidden / context states are then kept
        # channels-last [ht, wd, 128].  Default: on whenever the HIP device is there.
        self.hip_update = (self.device.type == "cuda") if hip_update is None else bool(hip_update)
        self.update_op = None
        if self.hip_update:
            from .update_op import HipUpdateOperator
            self.update_op = HipUpdateOperator(self.net.update_net)
            # advertised to the frontend: with it, TrackingFrontend.update() calls the lookup-fused kernel and hands update() an
            # EncodedCorr instead of the [1,E,196,ht,wd] lookup (NS_LOOKUP_UNFUSED=1: round 3's three launches)
            self.corr_encoder = self.update_op.corr_enc
            self.ctx_cl, self.inp_cl = {}, {}

    # ---- the ONE place that writes per-keyframe context features / per-edge hidden states from outside an update:
    #      the stacked tensors of the last update are a cache of exactly these, so every such write drops them
    def _invalidate_stacked(self):
        self._kf_epoch += 1
        self._stacked = None

    def set_context(self, k, ctx_chw, inp_chw):
        """context features of keyframe k: tanh / relu halves, [128, ht, wd]"""
        self._invalidate_stacked()
        self.ctx[k], self.inp[k] = ctx_chw, inp_chw
        if self.hip_update:
            self.ctx_cl[k] = ctx_chw.permute(1, 2, 0).contiguous().half()
            self.inp_cl[k] = inp_chw.permute(1, 2, 0).contiguous().half()

    def set_hidden(self, i, j, h):
        """ConvGRU hidden state of edge (i, j) -- channels-last [ht, wd, 128] f16 on the HIP operator, [128, ht, wd] otherwise;
        None forgets it (the next update starts the edge from its source frame's context features)"""
        self._invalidate_stacked()
        if h is None:
            self.hidden.pop((i, j), None)
        else:
            self.hidden[(i, j)] = h

    def _normalize(self, img_u8):
        x = img_u8.to(self.device).float()[:3] / 255.0
        m = torch.tensor(self.MEAN, device=self.device)[:, None, None]
        s = torch.tensor(self.STD, device=self.device)[:, None, None]
        return ((x - m) / s)[None, None]

    @torch.no_grad()
    def features(self, img_u8):
        if self.fnet_hip is not None:
            img = img_u8.to(self.device)[:3]
            img = (img if img.dtype == torch.uint8 else img.float())[None]
            self._pending = img                                            # (the context encoder normalises it again itself)
            return self.fnet_hip(img)[0].permute(2, 0, 1)                  # [128, ht, wd] view of the channels-last output
        x = self._normalize(img_u8)
        if self.fnet_h is not None:
            f = self.fnet_h(x.half())[0, 0]
        else:
            f = self.net.feature_net(x)[0, 0]
        self._pending = x
        return f

    @torch.no_grad()
    def begin_keyframe(self, k, img_u8):
        """hook of TrackingSLAM._store: context features of the frame that just became keyframe k"""
        self._invalidate_stacked()
        if self.cnet_hip is not None:
            img = self._pending
            if img is None:
                img = img_u8.to(self.device)[:3]
                img = (img if img.dtype == torch.uint8 else img.float())[None]
            c = self.cnet_hip(img)[0]                                      # [ht, wd, 256] channels-last f16
            ctx, inp = torch.tanh(c[..., :128]).contiguous(), torch.relu(c[..., 128:]).contiguous()
            self.ctx[k], self.inp[k] = ctx.permute(2, 0, 1), inp.permute(2, 0, 1)
            if self.hip_update:
                self.ctx_cl[k], self.inp_cl[k] = ctx, inp
            return
        x = self._pending if self._pending is not None else self._normalize(img_u8)
        c = self.cnet_h(x.half())[0, 0] if self.cnet_h is not None else self.net.context_net(x)[0, 0]
        self.ctx[k], self.inp[k] = torch.tanh(c[:128]), torch.relu(c[128:])
        if self.hip_update:
            self.ctx_cl[k] = self.ctx[k].permute(1, 2, 0).contiguous().half()
            self.inp_cl[k] = self.inp[k].permute(1, 2, 0).contiguous().half()

    def remove_keyframe(self, k):
        """hook of TrackingSLAM.rm_keyframe: keyframe k+1 slides onto k, edges touching k disappear"""
        self._invalidate_stacked()
        for d in (self.ctx, self.inp) + ((self.ctx_cl, self.inp_cl) if self.hip_update else ()):
            if k + 1 in d:
                d[k] = d.pop(k + 1)
        sh = lambda a: a - (a >= k)
        self.hidden = {(sh(i) if i != k else -1, sh(j) if j != k else -1): h for (i, j), h in self.hidden.items()}
        self.hidden = {e: h for e, h in self.hidden.items() if -1 not in e}

    @torch.no_grad()
    def motion(self, corr, last_kf):
        if self.hip_update:
            c = (corr[0] if corr.dim() == 5 else corr).half()
            # delta head only, no GraphAgg / weight head: 0.60 vs 0.74 ms for the full operator at 80x60 (a HIP-graph replay of
            # it measured 0.62 ms: one edge is 19 workgroups per launch, the kernels' own latency, not the launches, is the time)
            return self.update_op.delta_only(self.ctx_cl[last_kf][None], self.inp_cl[last_kf][None], c)[None]
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.device.type == "cuda"):
            _, delta, _ = self.net.update_net(self.ctx[last_kf][None, None], self.inp[last_kf][None, None], corr)
        return delta.float()

    @torch.no_grad()
    def update(self, corr, motion, ii, jj, ii_host=None, jj_host=None):
        """ii_host / jj_host: the edge lists as host ints (TrackingFrontend keeps them on the host and passes them when the
        callable advertises `host_indices`): without them ii.tolist() is a device read-back, i.e. a synchronisation per update"""
        ih, jh = (ii.tolist(), jj.tolist()) if ii_host is None else (list(ii_host), list(jj_host))
        if self.hip_update:
            # The hidden states of an UNCHANGED edge list are the previous call's output tensor, already stacked in edge order,
            # and its context features the previous call's stack: only a call that follows a change of the graph (or of the
            # keyframes behind it) gathers E x [ht,wd,128] tensors again.  (Rounds 1-4 re-stacked both on every update: two
            # copies of 59 MB at E = 48, and six updates per keyframe see the same list.)
            key = (tuple(ih), tuple(jh), self._kf_epoch)
            cache = self._stacked
            # (a list that names an edge twice is never served from the stack: the re-stack path gives BOTH copies the state the
            #  later one wrote, and the two paths must not differ)
            if cache is not None and cache[0] == key and len(set(zip(ih, jh))) == len(ih):
                net, inp = cache[1], cache[2]
            else:
                net = torch.stack([self.hidden.get((i, j), self.ctx_cl[i]) for i, j in zip(ih, jh)])
                inp = torch.stack([self.inp_cl[i] for i in ih])
            if hasattr(corr, "c1"):          # EncodedCorr (the frontend fused the lookup with the correlation encoder)
                c = corr
            else:
                c = (corr[0] if corr.dim() == 5 else corr).half()
            net, delta, weight, eta, upmask = self.update_op(net, inp, c, motion.reshape(-1, 4, *motion.shape[-2:]).float(), ih)
            for e, (i, j) in enumerate(zip(ih, jh)):
                self.hidden[(i, j)] = net[e]
            # (`net` is this call's OUTPUT and the next call's input: the operator never writes its inputs in place)
            self._stacked = (key, net, inp)
            live = set(zip(ih, jh))
            if len(self.hidden) > 4 * max(len(live), 64):
                self.hidden = {e: h for e, h in self.hidden.items() if e in live}
            return delta[None], weight[None], eta, upmask
        net = torch.stack([self.hidden.get((i, j), self.ctx[i]) for i, j in zip(ih, jh)])[None]
        inp = torch.stack([self.inp[i] for i in ih])[None]
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.device.type == "cuda"):
            net, delta, weight, eta, upmask = self.net.update_net(net, inp, corr, motion, ii, jj)
        for e, (i, j) in enumerate(zip(ih, jh)):
            self.hidden[(i, j)] = net[0, e]
        live = set(zip(ih, jh))
        if len(self.hidden) > 4 * max(len(live), 64):       # edges that left the graph
            self.hidden = {e: h for e, h in self.hidden.items() if e in live}
        return delta.float(), weight.float(), eta[0].float(), upmask[0]

    update.host_indices = True
