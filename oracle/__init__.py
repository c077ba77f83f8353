"""CPU ORACLE package -- test infrastructure, NOT product code.

Loads ``oracle/liboracle.so`` (built from droid_oracle.c / ngp_oracle.c by
``oracle/Makefile``) and exposes numpy-in / numpy-out wrappers.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; nothing under ``nerf-slam_amd/`` does.

The pure-numpy functions at the bottom (``ba_solve_retract``, ``ba_covariances``)
restate the Python half of the reference's ``RaftVisualFrontend.ba()``
(/root/reference/slam/visual_frontends/visual_frontend.py:1097-1230) whose solve
runs inside GTSAM [EXTERNAL, un-vendored, unpinned]: parity of that step is
"unpinned" (see DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("droid_oracle.c", "ngp_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_h2f.restype = C.c_float
        _LIB.orc_h2f.argtypes = [C.c_uint16]
        _LIB.orc_f2h.restype = C.c_uint16
        _LIB.orc_f2h.argtypes = [C.c_float]
        _LIB.orc_reduced_camera_matrix.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# --------------------------------------------------------------------------- lookups
def corr_index_forward(volume, coords, radius):
    """K12.  volume [B,h1,w1,h2,w2] float16|float32, coords [B,2,h1,w1] float32."""
    B, h1, w1, h2, w2 = volume.shape
    rd = 2 * radius + 1
    coords = _f32(coords)
    if volume.dtype == np.float16:
        vol = np.ascontiguousarray(volume).view(np.uint16)
        out = np.empty((B, rd, rd, h1, w1), np.uint16)
        lib().orc_corr_index_forward_f16(_p(vol), _p(coords), _p(out), B, h1, w1, h2, w2, radius)
        return out.view(np.float16)
    vol = _f32(volume)
    out = np.empty((B, rd, rd, h1, w1), np.float32)
    lib().orc_corr_index_forward_f32(_p(vol), _p(coords), _p(out), B, h1, w1, h2, w2, radius)
    return out


def corr_index_backward(coords, corr_grad, h2, w2, radius):
    B, _, h1, w1 = coords.shape
    coords, corr_grad = _f32(coords), _f32(corr_grad)
    out = np.empty((B, h1, w1, h2, w2), np.float32)
    lib().orc_corr_index_backward_f32(_p(coords), _p(corr_grad), _p(out), B, h1, w1, h2, w2, radius)
    return out


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """A1.  fmap1, fmap2 [n, C, ht, wd] float16 -> list of [n, ht, wd, ht>>l, wd>>l] float16."""
    n, Cc, ht, wd = fmap1.shape
    HW = ht * wd
    f1 = np.ascontiguousarray(fmap1).view(np.uint16)
    f2 = np.ascontiguousarray(fmap2).view(np.uint16)
    vol = np.empty((n, HW, HW), np.uint16)
    lib().orc_corr_volume_f16(_p(f1), _p(f2), _p(vol), n, Cc, HW)
    pyr = [vol.reshape(n, ht, wd, ht, wd)]
    h, w = ht, wd
    for _ in range(num_levels - 1):
        src = pyr[-1]
        dst = np.empty((n, ht, wd, h // 2, w // 2), np.uint16)
        lib().orc_corr_pool_f16(_p(src), _p(dst), C.c_long(n * HW), h, w)
        pyr.append(dst)
        h, w = h // 2, w // 2
    return [p.view(np.float16) for p in pyr]


def altcorr_forward(fmap1, fmap2, coords, radius):
    """K14.  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2], all float32."""
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * radius + 1
    fmap1, fmap2, coords = _f32(fmap1), _f32(fmap2), _f32(coords)
    out = np.empty((B, N, rd * rd, H1, W1), np.float32)
    lib().orc_altcorr_forward_f32(_p(fmap1), _p(fmap2), _p(coords), _p(out), B, H1, W1, H2, W2, Cc, N, radius)
    return out


# --------------------------------------------------------------------------- geometry
def frame_distance(poses, disps, intr, ii, jj, beta):
    poses, disps, intr, ii, jj = _f32(poses), _f32(disps), _f32(intr), _i64(ii), _i64(jj)
    _, ht, wd = disps.shape
    out = np.empty((ii.shape[0],), np.float32)
    lib().orc_frame_distance(_p(poses), _p(disps), _p(intr), _p(ii), _p(jj), _p(out), ii.shape[0], ht, wd,
                             C.c_float(beta))
    return out


def projective_transform(targets, weights, poses, disps, intr, extr, ii, jj):
    """K1 -> dict(Hs, vs, Eiz, Ejz, Cii, bz)."""
    targets, weights, poses, disps = _f32(targets), _f32(weights), _f32(poses), _f32(disps)
    intr, extr, ii, jj = _f32(intr), _f32(extr), _i64(ii), _i64(jj)
    M = ii.shape[0]
    _, ht, wd = disps.shape
    HW = ht * wd
    o = dict(Hs=np.zeros((4, M, 6, 6), np.float32), vs=np.zeros((2, M, 6), np.float32),
             Eiz=np.zeros((M, 6, HW), np.float32), Ejz=np.zeros((M, 6, HW), np.float32),
             Cii=np.zeros((M, HW), np.float32), bz=np.zeros((M, HW), np.float32))
    lib().orc_projective_transform(_p(targets), _p(weights), _p(poses), _p(disps), _p(intr), _p(extr), _p(ii),
                                   _p(jj), M, ht, wd, _p(o["Hs"]), _p(o["vs"]), _p(o["Eiz"]), _p(o["Ejz"]),
                                   _p(o["Cii"]), _p(o["bz"]))
    return o


def edge_jacobians(pose_i, pose_j, disp, intr, extr):
    """Per-pixel (coords[HW,2], Ji[HW,2,6], Jj[HW,2,6], Jz[HW,2]) of one edge, K1 conventions."""
    ht, wd = disp.shape
    HW = ht * wd
    coords = np.empty((HW, 2), np.float32)
    Ji = np.empty((HW, 2, 6), np.float32)
    Jj = np.empty((HW, 2, 6), np.float32)
    Jz = np.empty((HW, 2), np.float32)
    lib().orc_edge_jacobians(_p(_f32(pose_i)), _p(_f32(pose_j)), _p(_f32(disp)), _p(_f32(intr)), _p(_f32(extr)),
                             ht, wd, _p(coords), _p(Ji), _p(Jj), _p(Jz))
    return coords, Ji, Jj, Jz


def reduced_camera_matrix(poses, disps, intr, extr, disps_sens, targets, weights, eta, ii, jj, kf0, kf1):
    """A5 -> (H[6P,6P], v[6P,1], Q[K',HW], E[P+M,6,HW], w[K',HW], kx[K'])."""
    poses, disps, intr, extr = _f32(poses), _f32(disps), _f32(intr), _f32(extr)
    disps_sens, targets, weights, eta = _f32(disps_sens), _f32(targets), _f32(weights), _f32(eta)
    ii, jj = _i64(ii), _i64(jj)
    M = ii.shape[0]
    _, ht, wd = disps.shape
    HW = ht * wd
    P = kf1 - kf0
    H = np.zeros((6 * P, 6 * P), np.float32)
    v = np.zeros((6 * P, 1), np.float32)
    Q = np.zeros((P + M, HW), np.float32)
    w = np.zeros((P + M, HW), np.float32)
    E = np.zeros((P + M, 6, HW), np.float32)
    kx = np.zeros((P + M,), np.int64)
    K = lib().orc_reduced_camera_matrix(_p(poses), _p(disps), _p(intr), _p(extr), _p(disps_sens), _p(targets),
                                        _p(weights), _p(eta), _p(ii), _p(jj), M, ht, wd, kf0, kf1, _p(H), _p(v),
                                        _p(Q), _p(E), _p(w), _p(kx))
    return H, v, Q[:K].copy(), E, w[:K].copy(), kx[:K].copy()


def solve_depth(dx, disps, Q, E, w, ii, jj, kf0, kf1):
    """A11; returns the updated copy of disps."""
    disps = _f32(disps).copy()
    ii, jj = _i64(ii), _i64(jj)
    _, ht, wd = disps.shape
    lib().orc_solve_depth(_p(_f32(dx)), _p(disps), _p(_f32(Q)), _p(_f32(E)), _p(_f32(w)), _p(ii), _p(jj),
                          ii.shape[0], ht, wd, kf0, kf1)
    return disps


def accum(data, ix, jx):
    data, ix, jx = _f32(data), _i64(ix), _i64(jx)
    out = np.empty((jx.shape[0], data.shape[1]), np.float32)
    lib().orc_accum(_p(data), _p(ix), ix.shape[0], _p(jx), jx.shape[0], C.c_long(data.shape[1]), _p(out))
    return out


def pose_retr(poses, dx, kf0, kf1):
    poses = _f32(poses).copy()
    lib().orc_pose_retr(_p(poses), _p(_f32(dx)), kf0, kf1)
    return poses


def projmap(poses, disps, intr, ii, jj):
    poses, disps, intr, ii, jj = _f32(poses), _f32(disps), _f32(intr), _i64(ii), _i64(jj)
    _, ht, wd = disps.shape
    n = ii.shape[0]
    coords = np.zeros((n, ht, wd, 3), np.float32)
    valid = np.zeros((n, ht, wd, 1), np.float32)
    lib().orc_projmap(_p(poses), _p(disps), _p(intr), _p(ii), _p(jj), n, ht, wd, _p(coords), _p(valid))
    return coords, valid


def reproject(poses, disps, intr, ii, jj):
    """torch-path reprojection (projective_ops.py:98-145, no Jacobians) -> coords [n,ht,wd,2], valid [n,ht,wd]."""
    poses, disps, intr, ii, jj = _f32(poses), _f32(disps), _f32(intr), _i64(ii), _i64(jj)
    _, ht, wd = disps.shape
    n = ii.shape[0]
    coords = np.zeros((n, ht, wd, 2), np.float32)
    valid = np.zeros((n, ht, wd), np.float32)
    lib().orc_reproject(_p(poses), _p(disps), _p(intr), _p(ii), _p(jj), n, ht, wd, _p(coords), _p(valid))
    return coords, valid


def iproj(poses, disps, intr):
    poses, disps, intr = _f32(poses), _f32(disps), _f32(intr)
    nm, ht, wd = disps.shape
    pts = np.zeros((nm, ht, wd, 3), np.float32)
    lib().orc_iproj(_p(poses), _p(disps), _p(intr), nm, ht, wd, _p(pts))
    return pts


def depth_filter(poses, disps, intr, inds, thresh):
    poses, disps, intr, inds, thresh = _f32(poses), _f32(disps), _f32(intr), _i64(inds), _f32(thresh)
    nf, ht, wd = disps.shape
    out = np.zeros((inds.shape[0], ht, wd), np.float32)
    lib().orc_depth_filter(_p(poses), _p(disps), _p(intr), _p(inds), _p(thresh), inds.shape[0], nf, ht, wd, _p(out))
    return out


def se3_rel(pi, pj):
    out = np.empty(7, np.float32)
    lib().orc_relSE3(_p(_f32(pi)), _p(_f32(pj)), _p(out))
    return out


def se3_act(p, X):
    out = np.empty(4, np.float32)
    lib().orc_actSE3(_p(_f32(p)), _p(_f32(X)), _p(out))
    return out


def se3_adj(p, X):
    out = np.empty(6, np.float32)
    lib().orc_adjSE3(_p(_f32(p)), _p(_f32(X)), _p(out))
    return out


def se3_exp(xi):
    out = np.empty(7, np.float32)
    lib().orc_expSE3(_p(_f32(xi)), _p(out))
    return out


# --------------------------------------------------------------------------- mapping path (ngp_oracle.c)
class _GridCfg(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("n_features", C.c_int), ("log2_hashmap", C.c_int), ("base_res", C.c_int),
                ("per_level_scale", C.c_float)]


def ngp_cfg(n_levels=16, log2_hashmap=19, base_res=16, per_level_scale=1.5157165665):
    return _GridCfg(n_levels, 2, log2_hashmap, base_res, per_level_scale)


def ngp_grid_layout(cfg):
    scale = np.zeros(cfg.n_levels, np.float32)
    res = np.zeros(cfg.n_levels, np.int32)
    off = np.zeros(cfg.n_levels + 1, np.uint32)
    lib().orc_ngp_grid_layout(C.byref(cfg), _p(scale), _p(res), _p(off))
    return scale, res, off


def ngp_encode_fwd(cfg, pos, params):
    pos = _f32(pos)
    params = np.ascontiguousarray(params, np.float16)
    out = np.zeros((pos.shape[0], cfg.n_levels * 2), np.float16)
    lib().orc_ngp_encode_fwd(C.byref(cfg), _p(pos), _p(params), _p(out), C.c_long(pos.shape[0]))
    return out


def ngp_encode_bwd(cfg, pos, dLdout, n_params):
    pos = _f32(pos)
    dLdout = np.ascontiguousarray(dLdout, np.float16)
    grad = np.zeros((n_params,), np.float32)
    lib().orc_ngp_encode_bwd(C.byref(cfg), _p(pos), _p(dLdout), _p(grad), C.c_long(pos.shape[0]))
    return grad


class _Mlp(C.Structure):
    _fields_ = [("W1", C.c_void_p), ("W2", C.c_void_p), ("W3", C.c_void_p), ("W4", C.c_void_p), ("W5", C.c_void_p)]


MLP_SHAPES = [(64, 32), (16, 64), (64, 32), (64, 64), (16, 64)]


def _mlp_struct(Ws):
    Ws = [np.ascontiguousarray(w, np.float16) for w in Ws]
    return _Mlp(*[w.ctypes.data for w in Ws]), Ws


def ngp_mlp_fwd(Ws, feat, dirs):
    """-> dict(h1, dens, cin, h3, h4, rgb) f16 arrays."""
    m, keep = _mlp_struct(Ws)
    feat = np.ascontiguousarray(feat, np.float16)
    dirs = _f32(dirs)
    N = feat.shape[0]
    o = dict(h1=np.zeros((N, 64), np.float16), dens=np.zeros((N, 16), np.float16), cin=np.zeros((N, 32), np.float16),
             h3=np.zeros((N, 64), np.float16), h4=np.zeros((N, 64), np.float16), rgb=np.zeros((N, 16), np.float16))
    lib().orc_ngp_mlp_fwd(C.byref(m), _p(feat), _p(dirs), C.c_long(N), _p(o["h1"]), _p(o["dens"]), _p(o["cin"]),
                          _p(o["h3"]), _p(o["h4"]), _p(o["rgb"]))
    return o


def ngp_mlp_bwd(Ws, feat, act, dLdrgb, dLddens):
    """-> (dLdfeat [N,32] f16, [dW1..dW5] float64)."""
    m, keep = _mlp_struct(Ws)
    feat = np.ascontiguousarray(feat, np.float16)
    N = feat.shape[0]
    dLdrgb = np.ascontiguousarray(dLdrgb, np.float16)
    dLddens = np.ascontiguousarray(dLddens, np.float16)
    dfeat = np.zeros((N, 32), np.float16)
    dW = [np.zeros(s, np.float64) for s in MLP_SHAPES]
    lib().orc_ngp_mlp_bwd(C.byref(m), _p(feat), C.c_long(N), _p(act["h1"]), _p(act["dens"]), _p(act["cin"]),
                          _p(act["h3"]), _p(act["h4"]), _p(act["rgb"]), _p(dLdrgb), _p(dLddens), _p(dfeat),
                          *[_p(w) for w in dW])
    return dfeat, dW


def ngp_composite_loss(rgb_raw, dens_raw, dt, tmid, ray_start, ray_n, gt_rgb, gt_depth, gt_depth_cov, depth_lambda,
                       loss_scale):
    """rgb_raw, dens_raw [S,16] f16.  -> (out_rgb [R,3], out_depth [R], loss, dLdrgb [S,16], dLddens [S,16])."""
    rgb_raw = np.ascontiguousarray(rgb_raw, np.float16)
    dens_raw = np.ascontiguousarray(dens_raw, np.float16)
    dt, tmid, gt_rgb, gt_depth, gt_depth_cov = _f32(dt), _f32(tmid), _f32(gt_rgb), _f32(gt_depth), _f32(gt_depth_cov)
    ray_start = np.ascontiguousarray(ray_start, np.int32)
    ray_n = np.ascontiguousarray(ray_n, np.int32)
    R, S = ray_start.shape[0], rgb_raw.shape[0]
    out_rgb = np.zeros((R, 3), np.float32)
    out_depth = np.zeros((R,), np.float32)
    loss = np.zeros((1,), np.float32)
    dLdrgb = np.zeros((S, 16), np.float16)
    dLddens = np.zeros((S, 16), np.float16)
    lib().orc_ngp_composite_loss(_p(rgb_raw), _p(dens_raw), _p(dt), _p(tmid), _p(ray_start), _p(ray_n), R, _p(gt_rgb),
                                 _p(gt_depth), _p(gt_depth_cov), C.c_float(depth_lambda), C.c_float(loss_scale),
                                 _p(out_rgb), _p(out_depth), _p(loss), _p(dLdrgb), _p(dLddens))
    return out_rgb, out_depth, float(loss[0]), dLdrgb, dLddens


def ngp_adam(master, grad, m1, m2, step, lr, beta1=0.9, beta2=0.99, eps=1e-15, l2=0.0, grad_scale=1.0):
    master, grad, m1, m2 = _f32(master).copy(), _f32(grad), _f32(m1).copy(), _f32(m2).copy()
    hp = np.zeros(master.shape, np.float16)
    lib().orc_ngp_adam(_p(master), _p(hp), _p(grad), _p(m1), _p(m2), C.c_long(master.size), int(step), C.c_float(lr),
                       C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(l2), C.c_float(grad_scale))
    return master, hp, m1, m2


def ngp_march_ray(bits, G, ncasc, o, d, cone, min_step, max_step, t0, t1, max_n):
    bits = np.ascontiguousarray(bits, np.uint8)
    pos = np.zeros((max_n, 3), np.float32)
    dts = np.zeros((max_n,), np.float32)
    ts = np.zeros((max_n,), np.float32)
    lib().orc_ngp_march_ray.restype = C.c_int
    n = lib().orc_ngp_march_ray(_p(bits), G, ncasc, _p(_f32(o)), _p(_f32(d)), C.c_float(cone), C.c_float(min_step),
                                C.c_float(max_step), C.c_float(t0), C.c_float(t1), max_n, _p(pos), _p(dts), _p(ts))
    return pos[:n], dts[:n], ts[:n]


def ngp_encode_bwd_input(cfg, pos, params, dLdout):
    pos = _f32(pos)
    params = np.ascontiguousarray(params, np.float16)
    dLdout = np.ascontiguousarray(dLdout, np.float16)
    out = np.zeros((pos.shape[0], 3), np.float32)
    lib().orc_ngp_encode_bwd_input(C.byref(cfg), _p(pos), _p(params), _p(dLdout), _p(out), C.c_long(pos.shape[0]))
    return out


def ngp_camera_gradient(dLdpos, tmid, rays_d, ray_start, ray_n, ray_img, pos_inv, n_images):
    g = np.zeros((n_images, 6), np.float64)
    a = [np.ascontiguousarray(x, np.int32) for x in (ray_start, ray_n, ray_img)]
    lib().orc_ngp_camera_gradient(_p(_f32(dLdpos)), _p(_f32(tmid)), _p(_f32(rays_d)), _p(a[0]), _p(a[1]), _p(a[2]),
                                  C.c_float(pos_inv), _p(g), len(a[0]))
    return g


def ngp_camera_step(c2w, cam_grad, m1, m2, step, lr_pos, lr_rot, beta1=0.9, beta2=0.99, eps=1e-15, grad_scale=1.0):
    c2w, m1, m2 = _f32(c2w).copy(), _f32(m1).copy(), _f32(m2).copy()
    lib().orc_ngp_camera_step(_p(c2w), _p(_f32(cam_grad)), _p(m1), _p(m2), c2w.shape[0], step, C.c_float(lr_pos),
                              C.c_float(lr_rot), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(grad_scale))
    return c2w, m1, m2


def ngp_sample_rays(images, depths, covs, c2w, intr, box_lo, box_hi, near, seed, R):
    images, depths, covs, c2w = _f32(images), _f32(depths), _f32(covs), _f32(c2w)
    n, H, W = depths.shape
    out = dict(rays_o=np.zeros((R, 3), np.float32), rays_d=np.zeros((R, 3), np.float32), t_range=np.zeros((R, 2), np.float32),
               gt_rgb=np.zeros((R, 3), np.float32), gt_depth=np.zeros(R, np.float32), gt_cov=np.zeros(R, np.float32),
               picks=np.zeros((R, 3), np.int32))
    fx, fy, cx, cy = (C.c_float(float(v)) for v in intr)
    lib().orc_ngp_sample_rays(_p(images), _p(depths), _p(covs), _p(c2w), n, H, W, fx, fy, cx, cy, C.c_float(box_lo),
                              C.c_float(box_hi), C.c_float(near), C.c_uint32(seed), R, *[_p(out[k]) for k in
                              ("rays_o", "rays_d", "t_range", "gt_rgb", "gt_depth", "gt_cov", "picks")])
    return out


def cvx_upsample(data, mask, pow_=1.0):
    """utils/flow_viz.py:166-183 restated in numpy: data [n,ht,wd], mask [n,576,ht,wd] -> [n,8ht,8wd]"""
    data, mask = np.asarray(data, np.float64), np.asarray(mask, np.float64)
    n, ht, wd = data.shape
    m = mask.reshape(n, 9, 8, 8, ht, wd).copy()
    pad = np.zeros((n, ht + 2, wd + 2)); pad[:, 1:-1, 1:-1] = data
    nb = np.stack([pad[:, 1 + dy:1 + dy + ht, 1 + dx:1 + dx + wd] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], 1)  # [n,9,ht,wd]
    yy, xx = np.meshgrid(np.arange(ht), np.arange(wd), indexing="ij")
    for k, (dy, dx) in enumerate([(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]):
        bad = (yy + dy < 0) | (yy + dy >= ht) | (xx + dx < 0) | (xx + dx >= wd)
        m[:, k][:, :, :, bad] = -np.inf
    m = np.exp(m - m.max(1, keepdims=True)); m /= m.sum(1, keepdims=True)
    m = m ** pow_
    up = (m * nb[:, :, None, None]).sum(1)                      # [n,8,8,ht,wd]
    return up.transpose(0, 3, 1, 4, 2).reshape(n, 8 * ht, 8 * wd).astype(np.float32)


# --------------------------------------------------------------------------- float64 SE3 (numpy)
def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    qv = np.asarray(q[:3], np.float64)
    uv = 2.0 * np.cross(qv, v)
    return v + q[3] * uv + np.cross(qv, uv)


def se3_mul64(a, b):
    """(ta,qa)*(tb,qb): x -> Ra(Rb x + tb) + ta.  pose = [t(3), q(4) xyzw]."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.concatenate([_qrot(a[3:], b[:3]) + a[:3], _qmul(a[3:], b[3:])])


def se3_inv64(a):
    a = np.asarray(a, np.float64)
    qi = a[3:] * np.array([-1, -1, -1, 1.0])
    return np.concatenate([-_qrot(qi, a[:3]), qi])


def se3_exp64(xi_wv):
    """SE3 exponential, xi = [omega(3), v(3)] (GTSAM Pose3 tangent order). Returns [t, q]."""
    w = np.asarray(xi_wv[:3], np.float64)
    v = np.asarray(xi_wv[3:], np.float64)
    th = np.linalg.norm(w)
    if th < 1e-10:
        q = np.concatenate([0.5 * w, [1.0]])
        q /= np.linalg.norm(q)
        t = v + 0.5 * np.cross(w, v)
    else:
        q = np.concatenate([np.sin(0.5 * th) / th * w, [np.cos(0.5 * th)]])
        a = (1 - np.cos(th)) / th ** 2
        b = (th - np.sin(th)) / th ** 3
        wv = np.cross(w, v)
        t = v + a * wv + b * np.cross(w, wv)
    return np.concatenate([t, q])


def se3_log64(p):
    """inverse of se3_exp64 -> [omega, v]."""
    p = np.asarray(p, np.float64)
    q = p[3:] / np.linalg.norm(p[3:])
    if q[3] < 0:
        q = -q
    n = np.linalg.norm(q[:3])
    th = 2.0 * np.arctan2(n, q[3])
    w = q[:3] * (2.0 if n < 1e-12 else th / n)
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * W
    else:
        Vinv = np.eye(3) - 0.5 * W + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * W @ W
    return np.concatenate([w, Vinv @ p[:3]])


def ba_solve_retract(H, v, world_T_body, cam_T_body, kf0, kf1, prior_pose=None, prior_sigma=1e-4):
    """visual_frontend.py:1123-1158 restated in float64 numpy.

    H [6P,6P], v [6P,1] as returned by reduced_camera_matrix.  The reference splits H into
    HessianFactors whose sum is H again (:1123-1134), optionally adds a PriorFactorPose3 with
    sigma 1e-4 on the first window pose (:1089-1095,1137-1139), solves densely (:1144) and
    retracts world_T_body <- world_T_body (+) delta with delta=[omega,v] in the body frame (:1145),
    then cam_T_world = cam_T_body * world_T_body^-1 (:1158).
    Returns (delta[P,6], world_T_body_new[P,7], cam_T_world_new[P,7], Hfull[6P,6P]).
    GTSAM is external: Pose3.retract is taken to be the full Expmap and the prior Jacobian
    identity (exact at zero error) -- parity unpinned.
    """
    P = kf1 - kf0
    # GTSAM's HessianFactor keeps the upper triangle of what it is given (SymmetricBlockMatrix):
    # the system is triu(H) mirrored, element-wise (H as returned is symmetric only to f32 rounding)
    Hd = np.triu(np.asarray(H, np.float64))
    Hd = Hd + np.triu(Hd, 1).T
    vd = np.asarray(v, np.float64).reshape(-1).copy()
    if prior_pose is not None:
        e = se3_log64(se3_mul64(se3_inv64(prior_pose), world_T_body[kf0]))
        info = 1.0 / (float(np.float32(prior_sigma)) ** 2)  # sigma travels as f32 through the C ABI
        Hd[:6, :6] += np.eye(6) * info
        vd[:6] += -e * info
    delta = np.linalg.solve(Hd, vd).reshape(P, 6)
    wTb = np.asarray(world_T_body, np.float64)[kf0:kf1].copy()
    cTw = np.zeros_like(wTb)
    for i in range(P):
        wTb[i] = se3_mul64(wTb[i], se3_exp64(delta[i]))
        wTb[i, 3:] /= np.linalg.norm(wTb[i, 3:])
        cTw[i] = se3_mul64(np.asarray(cam_T_body, np.float64), se3_inv64(wTb[i]))
    return delta, wTb, cTw, Hd


def ba_covariances(Hfull, E, Q, ii, jj, kf0, kf1, HW, marginal_form=False):
    """visual_frontend.py:1164-1230 restated (float64): pose marginals = 6x6 diagonal blocks of
    H^-1; z_cov = Q + sum((Q*E^T) @ L^-1)^2 exactly as the reference composes it (:1215-1218, i.e.
    right-multiplying by L^-1, not L^-T).  Returns (sigma_g[P,6,6], z_cov[K,HW], kx[K]).

    PIN (tests/test_oracle_pins.py::test_covariance_block_*): with `marginal_form=True` the ONE operand
    `L^-1` of :1215 is replaced by `L^-T`; z_cov is then the diagonal of the depth block of the inverse
    of the full (pose + depth) normal equations -- an identity that is checked in float64 against a
    system assembled independently from K1's raw per-edge blocks and that pins everything else in this
    function (the E scatter with fixed frames :1204-1211, the Ei diagonal :1211, the Q scaling, the
    index bookkeeping).  The reference's own form (default) differs from it by exactly that transposition."""
    P = kf1 - kf0
    ii = np.asarray(ii)
    jj = np.asarray(jj)
    L = np.linalg.cholesky(np.asarray(Hfull, np.float32).astype(np.float64))
    Linv = np.linalg.inv(L)
    sig = Linv.T @ Linv
    sigma_g = np.stack([sig[6 * i:6 * i + 6, 6 * i:6 * i + 6] for i in range(P)])
    kx, kk = np.unique(ii, return_inverse=True)
    K = kx.shape[0]
    mn = min(ii.min(), jj.min())
    if K != np.asarray(Q).shape[0] or max(ii.max(), jj.max()) - mn >= K or kf1 - mn > K:
        # the reference's dense scatter (visual_frontend.py:1204-1214) only works when every frame of
        # [min(ii,jj), kf1) has an outgoing edge; it raises an index / shape error otherwise
        raise ValueError("covariance block not applicable to this graph (same failure as the reference)")
    Ej = np.zeros((K, K, 6, HW))
    Ej[jj - mn, ii - mn] = np.asarray(E[P:P + ii.shape[0]], np.float64)
    Ej = Ej[kf0 - mn:kf1 - mn].copy()
    for p in range(P):
        Ej[p, kf0 - mn + p] = E[p]
    Es = Ej.transpose(0, 2, 1, 3).reshape(P * 6, K * HW)
    Q_ = np.asarray(Q, np.float64).reshape(K * HW, 1)
    F = (Q_ * Es.T) @ (Linv.T if marginal_form else Linv)
    z = Q_[:, 0] + (F ** 2).sum(-1)
    return sigma_g, z.reshape(K, HW), kx


# --------------------------------------------------------------------------- SLAM packet -> NeRF training images
def srgb_to_linear(img):
    """utils/utils.py:136-139 (float64 numpy)"""
    img = np.asarray(img, np.float64)
    return np.where(img > 0.04045, np.power((img + 0.055) / 1.055, 2.4), img / 12.92)


def nerf_ingest(packet, mask_type="ours", scale=1.0, offset=(0.0, 0.0, 0.0)):
    """fusion/nerf_fusion.py:158-226 restated in float64 numpy: what `process_slam` hands to
    `update_training_images` for a SLAM packet (visual_frontend.py:1364-1382).
    packet: cam0_poses [n,7] (cam_T_world, [t, q xyzw]), cam0_images [n,3,H,W] uint8, cam0_idepths_up / cam0_depths_cov_up [n,H,W].
    -> dict(poses [n,3,4] camera-to-world with t * scale + offset (:198-203, utils.py:163-165), images [n,H,W,4] linear
    premultiplied RGBA with alpha 1 (:194-196,204,209-213), depths [n,H,W,1] = 1 / idepth (:205), depths_cov [n,H,W,1] (:206)).
    Mask policies (:173-183).  [The reference passes scale = 1, offset = 0 (:168-169); what the un-vendored fork then does
    with the dataset's own offset inside update_training_images is not in the tree.]"""
    poses = np.asarray(packet["cam0_poses"], np.float64)
    images = np.asarray(packet["cam0_images"])
    idepth = np.asarray(packet["cam0_idepths_up"], np.float64).copy()
    cov = np.asarray(packet["cam0_depths_cov_up"], np.float64).copy()
    if mask_type == "raw":
        cov[...] = 1.0
    elif mask_type == "ours_w_thresh":
        idepth[np.sqrt(cov) > np.quantile(cov, 0.5)] = -1.0
    elif mask_type == "no_depth":
        idepth[...] = -1.0
    elif mask_type != "ours":
        raise NotImplementedError(mask_type)
    n = poses.shape[0]
    c2w = np.zeros((n, 3, 4))
    for k in range(n):
        inv = se3_inv64(poses[k])                                   # world_T_cam = cam_T_world^-1
        x, y, z, w = inv[3:]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        c2w[k, :, :3] = R
        c2w[k, :, 3] = inv[:3] * scale + np.asarray(offset, np.float64)
    rgb = srgb_to_linear(images.transpose(0, 2, 3, 1).astype(np.float64) / 255.0)
    rgba = np.concatenate([rgb, np.ones(rgb.shape[:3] + (1,))], -1)          # alpha 255 / 255; premultiplied == rgb
    with np.errstate(divide="ignore"):
        depths = 1.0 / idepth[..., None]
    return dict(poses=c2w, images=rgba, depths=depths, depths_cov=cov[..., None])
