"""Volume build (corr_volume_tiled_kernel) at E = 10, 60x80 with 4 / 3 / 2 / 1 pyramid levels: what the row-major level-2 / level-3
stores cost (round 6: they are 8- and 4-byte pieces, each its own write request).  usage: python tools/vol_levels_bench.py [reps]"""
import json
import sys

import torch

sys.path.insert(0, "nerf-slam_amd")
from nerfslam.corr import CorrBlock

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
E, ht, wd = 10, 60, 80
g = torch.Generator(device="cpu").manual_seed(3)
f1 = (torch.randn((E, ht * wd, 128), generator=g) / 4).half().to(dev)
f2 = (torch.randn((E, ht * wd, 128), generator=g) / 4).half().to(dev)
out = {}
import os
for nl in [int(x) for x in os.environ.get("LEVELS", "4,3,2,1,4").split(",")]:
    for _ in range(3):
        CorrBlock.build_pyramid(f1, f2, None, None, E, ht, wd, nl, tiled=True)
    torch.cuda.synchronize()
    pyr = CorrBlock.build_pyramid(f1, f2, None, None, E, ht, wd, nl, tiled=True)
    import ctypes as C
    from nerfslam._lib import check, lib, ptr, stream_ptr
    arr = (C.c_void_p * 4)(*[pyr[min(l, nl - 1)].data_ptr() for l in range(4)])
    VL = C.CDLL(os.environ["VOL_LIB"]) if os.environ.get("VOL_LIB") else lib()    # (VOL_LIB: another build of csrc/corr_volume.hip, for same-box A/Bs)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        check(VL.ns_corr_volume_pyramid(ptr(f1), ptr(f2), None, None, arr, nl, E, 128, ht, wd, 1, stream_ptr()), "vol")
    ev[1].record()
    torch.cuda.synchronize()
    out.setdefault(f"levels={nl}", []).append(round(ev[0].elapsed_time(ev[1]) / reps * 1e3, 1))
    del pyr
print(json.dumps({"us_per_launch": out}))
