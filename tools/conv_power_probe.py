"""Is the gate convolution limited by the chip's power budget?  The same launch (3x3, 448 -> 256, E = 48 at 60x80) on random
inputs, on all-zero inputs (MI355X_MICROARCH.md "DVFS give-back": same instruction stream, far fewer toggling bits) and on
inputs with only a quarter / a sixteenth of the channels non-zero.  usage: python tools/conv_power_probe.py [reps]"""
import json
import sys

import torch

sys.path.insert(0, "nerf-slam_amd")
from nerfslam.conv import PackedConv, conv_nhwc

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
E, HT, WD = 48, 60, 80
g = torch.Generator(device="cpu").manual_seed(1)
w = (torch.randn((256, 448, 3, 3), generator=g) / 60).half().float().to(dev)
pcs = {"random weights": PackedConv(w, torch.zeros(256, device=dev)), "zero weights": PackedConv(torch.zeros_like(w), torch.zeros(256, device=dev))}
xs_r = [torch.randn((E, HT, WD, c), generator=g).half().to(dev) for c in (128, 128, 192)]
xs_z = [torch.zeros_like(x) for x in xs_r]
out = {}
flop = 2.0 * 9 * 448 * 256 * E * HT * WD
for rnd in range(2):
    for name, xs, pc in (("random inputs, random weights", xs_r, pcs["random weights"]), ("zero inputs, random weights", xs_z, pcs["random weights"]),
                         ("zero inputs, zero weights", xs_z, pcs["zero weights"]), ("random inputs, zero weights", xs_r, pcs["zero weights"])):
        for _ in range(5):
            conv_nhwc(xs, pc, act="sigmoid")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            conv_nhwc(xs, pc, act="sigmoid")
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / reps * 1e3
        out.setdefault(name, []).append({"us": round(us, 1), "TFLOP/s": round(flop / us / 1e6, 0)})
print(json.dumps(out))
