"""droid_backends -- drop-in for the reference's torch extension of the same name.

Same 12 operator names, positional arguments, return arity and error behaviour as
/root/reference/src/droid.cpp:347-363; every op is a thin ctypes call into the hand-written
HIP kernels of libnerfslam_hip.so (include/nerfslam_hip.h).  Outputs are freshly allocated
tensors owned by the caller, inputs are borrowed, launches go to torch's current stream.

There is no CPU path: CPU tensors or a missing shared library raise.
"""
import ctypes as C

import numpy as np
import torch

from nerfslam._lib import (NerfSlamHipError, check, check_contiguous, lib, ptr, require_cuda,
                           stream_ptr)
from nerfslam import ba_plan as _plan

__all__ = ["ba", "reduced_camera_matrix", "solve_depth", "solve_poses", "frame_distance", "projmap",
           "depth_filter", "iproj", "altcorr_forward", "altcorr_backward", "corr_index_forward",
           "corr_index_backward"]

_DT = {torch.float16: 1, torch.float32: 2}


# ------------------------------------------------------------------------------------------------
# correlation volume ops
# ------------------------------------------------------------------------------------------------
def corr_index_forward(volume, coords, radius):
    """src/droid.cpp:280-288.  volume [B,h1,w1,h2,w2] (half|float), coords [B,2,h1,w1] float
    -> [corr [B,2r+1,2r+1,h1,w1]]."""
    check_contiguous(volume=volume, coords=coords)
    require_cuda(volume, coords)
    if volume.dtype not in _DT:
        raise NerfSlamHipError(f"corr_index_forward: dtype {volume.dtype} not supported (half, float)")
    B, h1, w1, h2, w2 = volume.shape
    rd = 2 * int(radius) + 1
    corr = torch.empty((B, rd, rd, h1, w1), dtype=volume.dtype, device=volume.device)
    with torch.cuda.device(volume.device):
        check(lib().ns_corr_index_forward(ptr(volume), ptr(coords.float()), ptr(corr), _DT[volume.dtype], B, h1, w1,
                                          h2, w2, int(radius), stream_ptr()), "corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """src/droid.cpp:290-301 -> [volume_grad]."""
    check_contiguous(volume=volume, coords=coords, corr_grad=corr_grad)
    require_cuda(volume, coords, corr_grad)
    B, h1, w1, h2, w2 = volume.shape
    vg = torch.zeros(volume.shape, dtype=torch.float32, device=volume.device)
    with torch.cuda.device(volume.device):
        check(lib().ns_corr_index_backward(ptr(coords.float()), ptr(corr_grad.float().contiguous()), ptr(vg), B, h1,
                                           w1, h2, w2, int(radius), stream_ptr()), "corr_index_backward")
    return [vg.to(volume.dtype)]


def altcorr_forward(fmap1, fmap2, coords, radius):
    """src/droid.cpp:303-313.  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2]
    -> [corr [B,N,(2r+1)^2,H1,W1]]."""
    check_contiguous(fmap1=fmap1, fmap2=fmap2, coords=coords)
    require_cuda(fmap1, fmap2, coords)
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * int(radius) + 1
    out_dtype = fmap1.dtype
    f1 = fmap1.float().contiguous()
    f2 = fmap2.float().contiguous()
    corr = torch.empty((B, N, rd * rd, H1, W1), dtype=torch.float32, device=fmap1.device)
    with torch.cuda.device(fmap1.device):
        check(lib().ns_altcorr_forward(ptr(f1), ptr(f2), ptr(coords.float()), ptr(corr), B, H1, W1, H2, W2, Cc, N,
                                       int(radius), stream_ptr()), "altcorr_forward")
    return [corr.to(out_dtype)]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """src/droid.cpp:315-327 -> [fmap1_grad, fmap2_grad, coords_grad] (the last one all zeros, as in the reference:
    altcorr_kernel.cu:340 allocates it and no kernel writes it).  Dead in the reference's live path (grad is disabled
    globally, examples/slam_demo.py:198); answered for API parity."""
    check_contiguous(fmap1=fmap1, fmap2=fmap2, coords=coords, corr_grad=corr_grad)
    require_cuda(fmap1, fmap2, coords, corr_grad)
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    out_dtype = fmap1.dtype
    g1 = torch.empty((B, H1, W1, Cc), dtype=torch.float32, device=fmap1.device)
    g2 = torch.zeros((B, H2, W2, Cc), dtype=torch.float32, device=fmap1.device)
    with torch.cuda.device(fmap1.device):
        check(lib().ns_altcorr_backward(ptr(fmap1.float().contiguous()), ptr(fmap2.float().contiguous()),
                                        ptr(coords.float().contiguous()), ptr(corr_grad.float().contiguous()), ptr(g1), ptr(g2),
                                        B, H1, W1, H2, W2, Cc, N, int(radius), stream_ptr()), "altcorr_backward")
    return [g1.to(out_dtype), g2.to(out_dtype), torch.zeros_like(coords)]


# ------------------------------------------------------------------------------------------------
# geometry ops
# ------------------------------------------------------------------------------------------------
def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """src/droid.cpp:230-246 -> dist [num]."""
    check_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ii=ii, jj=jj)
    require_cuda(poses, disps, intrinsics, ii, jj)
    num = ii.shape[0]
    _, ht, wd = disps.shape
    dist = torch.empty((num,), dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        check(lib().ns_frame_distance(ptr(poses), ptr(disps), ptr(intrinsics), ptr(ii), ptr(jj), ptr(dist), num, ht,
                                      wd, C.c_float(float(beta)), stream_ptr()), "frame_distance")
    return dist


def projmap(poses, disps, intrinsics, ii, jj):
    """src/droid.cpp:249-264 -> [coords [num,ht,wd,3], valid [num,ht,wd,1]]."""
    check_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ii=ii, jj=jj)
    require_cuda(poses, disps, intrinsics, ii, jj)
    num = ii.shape[0]
    _, ht, wd = disps.shape
    coords = torch.zeros((num, ht, wd, 3), dtype=torch.float32, device=poses.device)
    valid = torch.zeros((num, ht, wd, 1), dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        check(lib().ns_projmap(ptr(poses), ptr(disps), ptr(intrinsics), ptr(ii), ptr(jj), ptr(coords), ptr(valid),
                               num, ht, wd, stream_ptr()), "projmap")
    return [coords, valid]


def iproj(poses, disps, intrinsics):
    """src/droid.cpp:267-276 -> points [nm,ht,wd,3]."""
    check_contiguous(poses=poses, disps=disps, intrinsics=intrinsics)
    require_cuda(poses, disps, intrinsics)
    nm, ht, wd = disps.shape
    pts = torch.empty((nm, ht, wd, 3), dtype=torch.float32, device=disps.device)
    with torch.cuda.device(disps.device):
        check(lib().ns_iproj(ptr(poses), ptr(disps), ptr(intrinsics), ptr(pts), nm, ht, wd, stream_ptr()), "iproj")
    return pts


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """src/droid.cpp:330-344 -> counter [num,ht,wd]."""
    check_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, ix=ix, thresh=thresh)
    require_cuda(poses, disps, intrinsics, ix, thresh)
    num = ix.shape[0]
    nf, ht, wd = disps.shape
    counter = torch.zeros((num, ht, wd), dtype=torch.float32, device=disps.device)
    with torch.cuda.device(disps.device):
        check(lib().ns_depth_filter(ptr(poses), ptr(disps), ptr(intrinsics), ptr(ix), ptr(thresh), ptr(counter), num,
                                    nf, ht, wd, stream_ptr()), "depth_filter")
    return counter


def solve_poses(poses, dx, t0, t1):
    """src/droid.cpp:220-228: poses[k] <- Exp(dx[k-t0]) * poses[k] in place."""
    check_contiguous(poses=poses, dx=dx)
    require_cuda(poses, dx)
    with torch.cuda.device(poses.device):
        check(lib().ns_pose_retr(ptr(poses), ptr(dx), int(t0), int(t1), stream_ptr()), "solve_poses")


# ------------------------------------------------------------------------------------------------
# dense bundle adjustment
# ------------------------------------------------------------------------------------------------
def reduced_camera_matrix(poses, body_poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii,
                          jj, t0, t1, plan=None):
    """src/droid.cpp:167-196 -> [H [6P,6P], v [6P,1], Q [K',HW], E [P+M,6,HW], w [K',HW]].

    `plan` (optional, nerfslam.ba_plan.BaPlan) lets a caller that already knows the edge lists on
    the host skip the one device->host copy of ii/jj this drop-in otherwise needs to size Q/w
    (the reference syncs here too: torch::_unique + accum_cuda, droid_kernels.cu:1066-1067,1706)."""
    check_contiguous(poses=poses, disps=disps, intrinsics=intrinsics, extrinsics=extrinsics, disps_sens=disps_sens,
                     targets=targets, weights=weights, eta=eta, ii=ii, jj=jj)
    require_cuda(poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj)
    if plan is None:
        plan = _plan.BaPlan.from_tensors(ii, jj, int(t0), int(t1))
    return list(_plan.reduced_camera_matrix(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights,
                                            eta, ii, jj))


def solve_depth(dx, disps, Q, E, w, ii, jj, t0, t1, plan=None):
    """src/droid.cpp:198-218: disps updated in place, returns None."""
    check_contiguous(dx=dx, disps=disps, Q=Q, E=E, w=w, ii=ii, jj=jj)
    require_cuda(dx, disps, Q, E, w, ii, jj)
    if plan is None:
        plan = _plan.BaPlan.from_tensors(ii, jj, int(t0), int(t1))
    _plan.solve_depth(plan, dx, disps, Q, E, w, clamp_min=-1.0)


def ba(poses, body_poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1,
       iterations, lm, ep, motion_only):
    """src/droid.cpp:133-165 -> ba_cuda (droid_kernels.cu:1441-1568).  Dead in the reference's live
    path; implemented on the same kernels: per iteration linearise, Schur-reduce,
    solve (H + ep + lm*diag) dx = v on the device, retract poses with Exp(dx)*T (pose_retr_kernel),
    back-substitute depths."""
    check_contiguous(targets=targets, weights=weights, poses=poses, body_poses=body_poses, disps=disps,
                     intrinsics=intrinsics, disps_sens=disps_sens, ii=ii, jj=jj)
    require_cuda(poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj)
    plan = _plan.BaPlan.from_tensors(ii, jj, int(t0), int(t1))
    return list(_plan.ba_reference_loop(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta,
                                        ii, jj, int(iterations), float(lm), float(ep), bool(motion_only)))
