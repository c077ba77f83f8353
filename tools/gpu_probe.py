"""First-contact probe for a GPU box: does the C-ABI library load next to torch's HIP runtime,
launch on torch's stream, and match the oracle?  (scratch tool, not a test)"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "nerf-slam_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from nerfslam._lib import lib, ptr, stream_ptr, check
import oracle

print("torch", torch.__version__, "hip", torch.version.hip, "dev", torch.cuda.get_device_name(0))
L = lib()
print("arch", L.ns_arch().decode(), "version", L.ns_version())
rng = np.random.default_rng(0)
E, ht, wd = 4, 30, 40
HW = ht * wd
pyr = []
for l in range(4):
    pyr.append(torch.from_numpy(rng.standard_normal((E, ht, wd, ht >> l, wd >> l)).astype(np.float16)).cuda())
gy, gx = np.meshgrid(np.arange(ht), np.arange(wd), indexing="ij")
coords = (np.stack([gx, gy], -1)[None].repeat(E, 0) + rng.uniform(-12, 12, (E, ht, wd, 2))).astype(np.float32)
coords[0, 0, 0] = [-2.5, -2.25]; coords[-1, -1, -1] = [wd + 1.5, ht + 0.75]; coords[1, 3, 3] = [np.nan, 1.0]
cd = torch.from_numpy(coords).cuda()
out = torch.empty((E, 196, ht, wd), dtype=torch.float16, device="cuda")
arr = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
check(L.ns_corr_lookup_pyramid(arr, 4, ptr(cd), 1, ptr(out), E, ht, wd, 0, stream_ptr()), "lookup")
torch.cuda.synchronize()
got = out.cpu().numpy()
ok = True
for l in range(4):
    c = np.ascontiguousarray(coords.transpose(0, 3, 1, 2)) / np.float32(2 ** l)
    c = np.nan_to_num(c, nan=-1e5)
    ref = oracle.corr_index_forward(pyr[l].cpu().numpy(), c, 3).reshape(E, 49, ht, wd)
    g = got[:, 49 * l:49 * (l + 1)]
    same = (ref.view(np.uint16) == g.view(np.uint16)) | ((ref == 0) & (g == 0))
    print("level", l, "bit-exact:", bool(same.all()), "mismatch", int((~same).sum()))
    ok &= bool(same.all())
    bad = np.argwhere(~same)
    for b in bad[:12]:
        n, ch, y, x = b
        print("   bad", b, "a,b=", ch // 7, ch % 7, "coords", c[n, :, y, x], "ref", ref[n, ch, y, x], "got", g[n, ch, y, x])
print("PROBE", "PASS" if ok else "FAIL")

# timing at C640
E, ht, wd = 48, 60, 80
pyr = [torch.randn((E, ht, wd, ht >> l, wd >> l), device="cuda", dtype=torch.float16) for l in range(4)]
gy, gx = torch.meshgrid(torch.arange(ht, device="cuda"), torch.arange(wd, device="cuda"), indexing="ij")
cd = (torch.stack([gx, gy], -1)[None].float() + torch.empty(E, ht, wd, 2, device="cuda").uniform_(-8, 8)).contiguous()
out = torch.empty((E, 196, ht, wd), dtype=torch.float16, device="cuda")
arr = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
for _ in range(5):
    L.ns_corr_lookup_pyramid(arr, 4, ptr(cd), 1, ptr(out), E, ht, wd, 0, stream_ptr())
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    L.ns_corr_lookup_pyramid(arr, 4, ptr(cd), 1, ptr(out), E, ht, wd, 0, stream_ptr())
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 50
alg = E * 4 * ht * wd * 234
print(f"lookup C640 E=48: {ms*1e3:.1f} us  algorithmic {alg/1e6:.1f} MB -> {alg/ms/1e9:.2f} TB/s")
