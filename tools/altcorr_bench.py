"""AltCorrBlock micro-benchmark (48 edges, 80x60, 4 levels, smooth and rough flow).  NS_ALTCORR_PER_PIXEL=1 selects the
wave-per-pixel kernel for comparison.  usage: python tools/altcorr_bench.py"""
import os, sys, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[ROOT, ROOT+"/nerf-slam_amd"]
from nerfslam.corr import AltCorrBlock
dev=torch.device("cuda:0"); ht,wd=60,80; E=48
g=torch.Generator().manual_seed(0)
fm=torch.randn((1,16,128,ht,wd),generator=g).to(dev)
alt=AltCorrBlock(fm)
ii=torch.randint(0,16,(E,),generator=g).to(dev); jj=torch.randint(0,16,(E,),generator=g).to(dev)
gy,gx=torch.meshgrid(torch.arange(ht),torch.arange(wd),indexing="ij")
base=torch.stack([gx,gy],-1).float()[None,None]
for name, noise in (("smooth flow (affine + 0.3 px noise)", 0.3), ("rough flow (4 px noise)", 4.0)):
    flow = torch.stack([0.05*gx.float()+3.0, -0.03*gy.float()-2.0], -1)[None,None]
    coords=(base+flow+torch.randn((1,E,ht,wd,2),generator=g)*noise).to(dev)
    r=alt(coords,ii,jj); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): alt(coords,ii,jj)
    e.record(); torch.cuda.synchronize()
    print(name, "altcorr_pyramid E=48 80x60: %.1f us"%(1e3*s.elapsed_time(e)/10), "checksum", float(r.double().abs().sum()))
