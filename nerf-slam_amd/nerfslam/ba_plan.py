"""Graph plan + launch helpers for the dense bundle adjustment kernels (csrc/ba*.hip).

`BaPlan` is the host-side index structure the reference rebuilds on every BA call
(src/droid_kernels.cu:1702-1710, 1065-1103, 1359-1402); here it is built once per factor-graph
change (host C++: ns_ba_plan_build), uploaded once and reused by every linearisation until the
edge lists change.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import NerfSlamHipError, check, lib, ptr, stream_ptr


class _CPlan(C.Structure):
    _fields_ = [("M", C.c_int), ("P", C.c_int), ("K", C.c_int), ("kf0", C.c_int), ("kf1", C.c_int),
                ("n_pairs", C.c_int), ("n_rows", C.c_int), ("n_jobs", C.c_int), ("max_src", C.c_int)]


PLAN_PARTS = 13   # include/nerfslam_hip.h: NS_BA_PLAN_PARTS


class BaPlan:
    """Index block of one (ii, jj, kf0, kf1) configuration, resident on `device`."""

    def __init__(self, ii_host, jj_host, kf0, kf1, device):
        ii_host = np.ascontiguousarray(ii_host, dtype=np.int64)
        jj_host = np.ascontiguousarray(jj_host, dtype=np.int64)
        if ii_host.shape != jj_host.shape or ii_host.ndim != 1:
            raise NerfSlamHipError("BaPlan: ii and jj must be 1-D and of equal length")
        L = lib()
        M = int(ii_host.shape[0])
        pi = ii_host.ctypes.data_as(C.c_void_p)
        pj = jj_host.ctypes.data_as(C.c_void_p)
        count = L.ns_ba_plan_index_count(pi, pj, M, int(kf0), int(kf1))
        self.c = _CPlan()
        idx = np.zeros((max(int(count), 1),), np.int32)
        self.offsets = (C.c_size_t * PLAN_PARTS)()
        check(L.ns_ba_plan_build(pi, pj, M, int(kf0), int(kf1), C.byref(self.c), idx.ctypes.data_as(C.c_void_p),
                                 self.offsets), "ns_ba_plan_build")
        self.index_host = idx
        self.index = torch.from_numpy(idx).to(device, non_blocking=False)
        self.device = torch.device(device)
        self.ii_host, self.jj_host = ii_host, jj_host
        self.kx_host = idx[self.offsets[0]:self.offsets[0] + self.c.K].astype(np.int64)
        self._ws = None
        self._ws_hw = -1

    @classmethod
    def from_tensors(cls, ii, jj, kf0, kf1):
        """One device->host copy of the edge lists (the reference does several per call)."""
        return cls(ii.detach().cpu().numpy(), jj.detach().cpu().numpy(), kf0, kf1, ii.device)

    M = property(lambda s: s.c.M)
    P = property(lambda s: s.c.P)
    K = property(lambda s: s.c.K)
    kf0 = property(lambda s: s.c.kf0)
    kf1 = property(lambda s: s.c.kf1)
    n_pairs = property(lambda s: s.c.n_pairs)

    def workspace(self, HW):
        if self._ws is None or self._ws_hw != HW:
            nbytes = lib().ns_ba_workspace_bytes(C.byref(self.c), int(HW))
            self._ws = torch.zeros((nbytes + 256,), dtype=torch.uint8, device=self.device)
            self._ws_hw = HW
        off = (-self._ws.data_ptr()) % 256
        return C.c_void_p(self._ws.data_ptr() + off)


def reduced_camera_matrix(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj):
    """-> (H [6P,6P], v [6P,1], Q [K,HW], E [P+M,6,HW], w [K,HW]) exactly like the reference op."""
    dev = poses.device
    _, ht, wd = disps.shape
    HW = ht * wd
    P, M, K = plan.P, plan.M, plan.K
    if targets.shape[0] != M:
        raise NerfSlamHipError(f"reduced_camera_matrix: plan has M={M} edges, targets has {targets.shape[0]}")
    if eta.numel() != K * HW:
        # the reference fails the same way on its broadcast (droid_kernels.cu:1752)
        raise RuntimeError(f"eta has {eta.numel() // HW} rows but the graph has K'={K} depth maps")
    H = torch.empty((6 * P, 6 * P), dtype=torch.float32, device=dev)
    v = torch.empty((6 * P, 1), dtype=torch.float32, device=dev)
    Q = torch.empty((K, HW), dtype=torch.float32, device=dev)
    w = torch.empty((K, HW), dtype=torch.float32, device=dev)
    E = torch.empty((P + M, 6, HW), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib().ns_reduced_camera_matrix(ptr(poses), ptr(disps), ptr(intrinsics), ptr(extrinsics),
                                             ptr(disps_sens), ptr(targets), ptr(weights), ptr(eta), ptr(ii), ptr(jj),
                                             C.byref(plan.c), ptr(plan.index), plan.offsets, ht, wd, ptr(H), ptr(v),
                                             ptr(Q), ptr(E), ptr(w), plan.workspace(HW), 1, stream_ptr()),
              "reduced_camera_matrix")
    return H, v, Q, E, w


def reduced_camera_matrix_timed(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj, reps=10):
    """-> ((H, v, Q, E, w), {kernel: mean microseconds}) -- the same call with HIP events between its five launches"""
    dev = poses.device
    _, ht, wd = disps.shape
    HW = ht * wd
    P, M, K = plan.P, plan.M, plan.K
    H = torch.empty((6 * P, 6 * P), dtype=torch.float32, device=dev)
    v = torch.empty((6 * P, 1), dtype=torch.float32, device=dev)
    Q = torch.empty((K, HW), dtype=torch.float32, device=dev)
    w = torch.empty((K, HW), dtype=torch.float32, device=dev)
    E = torch.empty((P + M, 6, HW), dtype=torch.float32, device=dev)
    us = (C.c_float * 5)()
    with torch.cuda.device(dev):
        check(lib().ns_reduced_camera_matrix_timed(ptr(poses), ptr(disps), ptr(intrinsics), ptr(extrinsics), ptr(disps_sens),
                                                   ptr(targets), ptr(weights), ptr(eta), ptr(ii), ptr(jj), C.byref(plan.c),
                                                   ptr(plan.index), plan.offsets, ht, wd, ptr(H), ptr(v), ptr(Q), ptr(E), ptr(w),
                                                   plan.workspace(HW), 1, stream_ptr(), int(reps), us), "reduced_camera_matrix_timed")
    names = ("ba_edge_table_kernel", "ba_linearize_slot_kernel", "ba_schur_gram_kernel", "ba_schur_reduce_kernel", "ba_finalize_kernel")
    return (H, v, Q, E, w), {n: float(us[i]) for i, n in enumerate(names)}


def solve_depth(plan, dx, disps, Q, E, w, clamp_min=-1.0):
    _, ht, wd = disps.shape
    with torch.cuda.device(disps.device):
        check(lib().ns_solve_depth(ptr(dx), ptr(disps), ptr(Q), ptr(E), ptr(w), C.byref(plan.c), ptr(plan.index),
                                   plan.offsets, ht, wd, C.c_float(clamp_min), stream_ptr()), "solve_depth")


def projective_transform(targets, weights, poses, disps, intrinsics, extrinsics, ii, jj):
    """K1 alone, with the reference kernel's own per-edge outputs (droid_kernels.cu:192-536):
    -> dict(Hs[4,M,6,6], vs[2,M,6], Eiz[M,6,HW], Ejz[M,6,HW], Cii[M,HW], bz[M,HW])."""
    dev = poses.device
    M = ii.shape[0]
    _, ht, wd = disps.shape
    HW = ht * wd
    f = dict(dtype=torch.float32, device=dev)
    o = dict(Hs=torch.zeros((4, M, 6, 6), **f), vs=torch.zeros((2, M, 6), **f), Eiz=torch.empty((M, 6, HW), **f),
             Ejz=torch.empty((M, 6, HW), **f), Cii=torch.empty((M, HW), **f), bz=torch.empty((M, HW), **f))
    etab = torch.empty((max(M, 1), 80), **f)
    with torch.cuda.device(dev):
        check(lib().ns_projective_transform(ptr(targets), ptr(weights), ptr(poses), ptr(disps), ptr(intrinsics),
                                            ptr(extrinsics), ptr(ii), ptr(jj), M, ht, wd, ptr(o["Hs"]), ptr(o["vs"]),
                                            ptr(o["Eiz"]), ptr(o["Ejz"]), ptr(o["Cii"]), ptr(o["bz"]), ptr(etab),
                                            stream_ptr()), "projective_transform")
    return o


MAX_SMALL_SYSTEM = 192  # 6P handled by the single-workgroup LDS Cholesky (csrc/ba_solve.hip)
MAX_SMALL_SYSTEM_COV = 108  # ... when the identity border rows (L^-1) must fit in LDS as well


def ba_solve(H, v, kf0, kf1, world_T_body=None, cam_T_world=None, cam_T_body=None, prior_pose=None,
             prior_sigma=1e-4, ep=0.0, lm=0.0, retract=True, want_cov=False, want_hfull=False):
    """Device-resident replacement of the GTSAM round trip (visual_frontend.py:1123-1158).

    Solves (H [+prior]) dx = v in f64, optionally retracts world_T_body / recomputes cam_T_world in place.
    Returns dict(dx [P,6], info [1] int32, Hfull [n,n] f64|None (the damped system, on request), Linv [n,n] f32|None,
    sigma_g [P,6,6]|None)."""
    dev = H.device
    P = int(kf1) - int(kf0)
    n = 6 * P
    dx = torch.empty((P, 6), dtype=torch.float32, device=dev)
    info = torch.empty((1,), dtype=torch.int32, device=dev)
    Hfull = torch.empty((n, n), dtype=torch.float64, device=dev) if want_hfull else None
    Linv = torch.empty((n, n), dtype=torch.float32, device=dev) if want_cov else None
    Lws = None
    sig = torch.empty((P, 6, 6), dtype=torch.float32, device=dev) if want_cov else None
    if n > MAX_SMALL_SYSTEM or (want_cov and n > MAX_SMALL_SYSTEM_COV):
        return _ba_solve_large(H, v, kf0, kf1, world_T_body, cam_T_world, cam_T_body, prior_pose, prior_sigma, ep, lm,
                               retract, want_cov, want_hfull)
    with torch.cuda.device(dev):
        check(lib().ns_ba_solve(ptr(H), ptr(v), ptr(world_T_body), ptr(cam_T_world), ptr(cam_T_body),
                                ptr(prior_pose), C.c_float(prior_sigma), C.c_float(ep), C.c_float(lm), int(kf0),
                                int(kf1), 0 if retract else 1, ptr(dx), ptr(Hfull), ptr(Linv), ptr(Lws), ptr(sig),
                                ptr(info), stream_ptr()), "ba_solve")
    return dict(dx=dx, info=info, Hfull=Hfull, Linv=Linv, sigma_g=sig)


_large_ws = {}   # (device, bytes) workspace of ns_ba_solve_large, grown on demand and reused (f64, up to 2 x (6P)^2 + 6P)


def _ba_solve_large(H, v, kf0, kf1, world_T_body, cam_T_world, cam_T_body, prior_pose, prior_sigma, ep, lm, retract,
                    want_cov, want_hfull=False):
    """6P beyond the single-workgroup LDS solve (global BA over the whole buffer, visual_frontend.py:1255-1295; windows
    with covariances above 18 poses): blocked f64 Cholesky through HBM, csrc/ba_solve_large.hip.  Same semantics and
    outputs as the LDS path; nothing leaves the device."""
    dev = H.device
    P = int(kf1) - int(kf0)
    n = 6 * P
    L = lib()
    L.ns_ba_solve_large_workspace_bytes.restype = C.c_size_t
    need = int(L.ns_ba_solve_large_workspace_bytes(n, 1 if want_cov else 0))
    key = (dev.type, dev.index)
    ws = _large_ws.get(key)
    if ws is None or ws.numel() * 8 < need:
        ws = _large_ws[key] = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
    dx = torch.empty((P, 6), dtype=torch.float32, device=dev)
    info = torch.empty((1,), dtype=torch.int32, device=dev)
    Hfull = torch.empty((n, n), dtype=torch.float64, device=dev) if want_hfull else None
    Linv = torch.empty((n, n), dtype=torch.float32, device=dev) if want_cov else None
    sig = torch.empty((P, 6, 6), dtype=torch.float32, device=dev) if want_cov else None
    with torch.cuda.device(dev):
        check(L.ns_ba_solve_large(ptr(H), ptr(v), ptr(world_T_body), ptr(cam_T_world), ptr(cam_T_body), ptr(prior_pose),
                                  C.c_float(prior_sigma), C.c_float(ep), C.c_float(lm), int(kf0), int(kf1),
                                  0 if retract else 1, ptr(dx), ptr(Hfull), ptr(Linv), ptr(sig), ptr(info), ptr(ws),
                                  C.c_size_t(ws.numel() * 8), stream_ptr()), "ba_solve_large")
    return dict(dx=dx, info=info, Hfull=Hfull, Linv=Linv, sigma_g=sig)


def depth_cov(plan, Linv, Q, E, HW):
    """z_cov [K,HW] (visual_frontend.py:1191-1219)."""
    z = torch.empty((plan.K, HW), dtype=torch.float32, device=Q.device)
    with torch.cuda.device(Q.device):
        check(lib().ns_ba_depth_cov(ptr(Linv), ptr(Q), ptr(E), C.byref(plan.c), ptr(plan.index), plan.offsets, int(HW),
                                    ptr(z), stream_ptr()), "ba_depth_cov")
    return z


def ba_reference_loop(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, ii, jj,
                      iterations, lm, ep, motion_only):
    """The reference's dead `ba` op (ba_cuda, droid_kernels.cu:1441-1568) on the live kernels:
    per iteration linearise, (Schur-)solve with (ep + lm*diag) damping, retract poses with
    Exp(dx)*T (pose_retr_kernel) and, unless motion_only, update the disparities.  -> (dx, dz)."""
    _, ht, wd = disps.shape
    HW = ht * wd
    dx = dz = None
    for _ in range(iterations):
        before = None if motion_only else disps.clone()
        if motion_only:
            # poses only (droid_kernels.cu:1505-1522): the pose x pose block A of K1's per-edge blocks -- no depth terms, no
            # Schur complement -- solved with the same damping; the disparities stay.  A dead branch of a dead op: assembled
            # with torch index_put, solved by the device Cholesky (which reads the upper triangle: A^T hands it the LOWER
            # triangle of A, the one Eigen's SimplicialLLT reads, :1323-1329).
            o = projective_transform(targets, weights, poses, disps, intrinsics, extrinsics, ii, jj)
            P, kf0 = plan.kf1 - plan.kf0, plan.kf0
            ri = torch.cat([ii, ii, jj, jj]) - kf0
            ci = torch.cat([ii, jj, ii, jj]) - kf0
            ok = (ri >= 0) & (ci >= 0) & (ri < P) & (ci < P)              # SparseBlock::update_lhs drops fixed poses (:1270)
            A = torch.zeros((P, P, 6, 6), dtype=torch.float64, device=poses.device)
            A.index_put_((ri[ok], ci[ok]), o["Hs"].reshape(-1, 6, 6)[ok].double(), accumulate=True)
            A = A.permute(0, 2, 1, 3).reshape(6 * P, 6 * P)
            bi = torch.cat([ii, jj]) - kf0
            okb = (bi >= 0) & (bi < P)
            b = torch.zeros((P, 6), dtype=torch.float64, device=poses.device)
            b.index_put_((bi[okb],), o["vs"].reshape(-1, 6)[okb].double(), accumulate=True)
            sol = ba_solve(A.t().float().contiguous(), b.reshape(-1).float().contiguous(), plan.kf0, plan.kf1, ep=ep, lm=lm,
                           retract=False)
            dx = sol["dx"]
            with torch.cuda.device(poses.device):
                check(lib().ns_pose_retr(ptr(poses), ptr(dx), plan.kf0, plan.kf1, stream_ptr()), "pose_retr")
            continue
        H, v, Q, E, w = reduced_camera_matrix(plan, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights,
                                              eta, ii, jj)
        # SparseBlock::solve adds the damping to A-S, get_dense() transposes: H is symmetric
        sol = ba_solve(H, v, plan.kf0, plan.kf1, ep=ep, lm=lm, retract=False)
        dx = sol["dx"]
        # ba_cuda's [tau,phi] ordering is whatever the system's ordering is; apply as is (:1554)
        solve_depth(plan, dx, disps, Q, E, w, clamp_min=-1.0)
        with torch.cuda.device(poses.device):
            check(lib().ns_pose_retr(ptr(poses), ptr(dx), plan.kf0, plan.kf1, stream_ptr()), "pose_retr")
        kx = torch.as_tensor(plan.kx_host, device=disps.device)
        dz = (disps - before)[kx].reshape(-1, HW)
    return dx, dz
