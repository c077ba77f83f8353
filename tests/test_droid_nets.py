"""nerfslam/droid_nets.py against the REFERENCE's own modules: parameter names / shapes (a DROID-SLAM checkpoint must
load) and seeded forward passes (fixtures written by tools/gen_golden.py section 4 from /root/reference/networks)."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _seeded(net):
    sd = net.state_dict()
    for k in sd:
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
        sd[k] = torch.randn(sd[k].shape, generator=g) * (0.3 / max(1.0, float(np.sqrt(sd[k][0].numel()))))
    net.load_state_dict(sd)
    return net.eval()


def test_state_dict_matches_the_reference_checkpoint_layout():
    from nerfslam.droid_nets import DroidNet
    want = json.load(open(os.path.join(G, "droid_state_dict_shapes.json")))
    got = {k: list(v.shape) for k, v in DroidNet().state_dict().items()}
    assert got == want


def test_forward_passes_match_the_reference_modules():
    from nerfslam.droid_nets import DroidNet
    z = np.load(os.path.join(G, "droid_nets_forward.npz"))
    net = _seeded(DroidNet())
    t = lambda k: torch.from_numpy(z[k])
    with torch.no_grad():
        fm, cm = net.feature_net(t("img")), net.context_net(t("img"))
        h, delta, weight, eta, upmask = net.update_net(t("net"), t("inp"), t("corr"), t("flow"), t("ii"), t("jj"))
        _, d0, w0 = net.update_net(t("net")[:, :1], t("inp")[:, :1], t("corr")[:, :1])
    for name, got in (("fmap", fm), ("cmap", cm), ("h", h), ("delta", delta), ("weight", weight), ("eta", eta),
                      ("upmask", upmask[:, :, ::16]), ("delta_noflow", d0), ("weight_noflow", w0)):
        ref = z[name]
        assert got.shape == ref.shape, name
        assert np.abs(got.numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name   # fp32 CPU, same op order


def test_checkpoint_key_remapping():
    """DROID-SLAM checkpoints: `module.fnet / cnet / update` prefixes and 3-channel heads (visual_frontend.py:1051-1068)"""
    from nerfslam.droid_nets import DroidNet
    src = _seeded(DroidNet())
    ckpt = {}
    for k, v in src.state_dict().items():
        k2 = "module." + k.replace("feature_net.", "fnet.").replace("context_net.", "cnet.").replace("update_net.", "update.")
        if k in ("update_net.weight.2.weight", "update_net.weight.2.bias", "update_net.delta.2.weight", "update_net.delta.2.bias"):
            v = torch.cat([v, torch.zeros_like(v[:1])], 0)
        ckpt[k2] = v
    dst = DroidNet().load_weights(ckpt)
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
