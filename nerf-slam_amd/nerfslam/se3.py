"""Minimal batched SE3 algebra on torch tensors (host-side plumbing, device agnostic).

Stands in for the handful of lietorch [EXTERNAL, un-vendored] calls the reference's hot-path
drivers make (networks/geom/projective_ops.py:108,123,128-130,142;
slam/visual_frontends/visual_frontend.py:912,1103,1158; fusion/nerf_fusion.py:202).
Pose layout as everywhere in the reference: [tx,ty,tz, qx,qy,qz,qw] (visual_frontend.py:46-49,
src/droid_kernels.cu:264-271).  All functions broadcast over leading dimensions.
"""
import torch


def _cross(a, b):
    return torch.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], dim=-1)


def qrot(q, v):
    """rotate v [...,3] by unit quaternion q [...,4] (xyzw)."""
    qv = q[..., :3]
    uv = 2.0 * _cross(qv, v)
    return v + q[..., 3:4] * uv + _cross(qv, uv)


def qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def mul(a, b):
    """a*b: x -> Ra (Rb x + tb) + ta."""
    return torch.cat([qrot(a[..., 3:], b[..., :3]) + a[..., :3], qmul(a[..., 3:], b[..., 3:])], dim=-1)


def inv(a):
    qi = a[..., 3:] * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=a.dtype, device=a.device)
    return torch.cat([-qrot(qi, a[..., :3]), qi], dim=-1)


def act(a, X):
    """act on homogeneous points X [...,4] = [X,Y,Z,h] -> [R X + h t, h]."""
    return torch.cat([qrot(a[..., 3:], X[..., :3]) + X[..., 3:4] * a[..., :3], X[..., 3:4]], dim=-1)


def matrix(a):
    """4x4 homogeneous matrix."""
    eye = torch.eye(3, dtype=a.dtype, device=a.device).expand(a.shape[:-1] + (3, 3))
    R = torch.stack([qrot(a[..., None, 3:], eye)[..., i, :] for i in range(3)], dim=-1)
    top = torch.cat([R, a[..., :3, None]], dim=-1)
    bot = torch.zeros(a.shape[:-1] + (1, 4), dtype=a.dtype, device=a.device)
    bot[..., 0, 3] = 1.0
    return torch.cat([top, bot], dim=-2)


def adjT(a, J):
    """row vector J [...,6] ([t,w] order) times the adjoint of a: what lietorch's `adjT` returns
    and what adjSE3 of src/droid_kernels.cu:88-105 computes (non-aliased form)."""
    q = a[..., 3:]
    qi = q * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=a.dtype, device=a.device)
    t = a[..., :3]
    Jt, Jw = J[..., :3], J[..., 3:]
    return torch.cat([qrot(qi, Jt), qrot(qi, Jw) + qrot(qi, _cross(Jt, t))], dim=-1)


def exp_wv(xi):
    """SE3 exponential, xi [...,6] = [omega, v] -> pose [...,7]."""
    w, v = xi[..., :3], xi[..., 3:]
    th2 = (w * w).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th < 1e-8
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 * torch.ones_like(th), torch.sin(0.5 * ths) / ths)
    real = torch.where(small, torch.ones_like(th), torch.cos(0.5 * ths))
    a = torch.where(small, 0.5 * torch.ones_like(th), (1 - torch.cos(ths)) / (ths * ths))
    b = torch.where(small, torch.ones_like(th) / 6.0, (ths - torch.sin(ths)) / (ths * ths * ths))
    q = torch.cat([imag * w, real], dim=-1)
    q = q / q.norm(dim=-1, keepdim=True)
    wv = _cross(w, v)
    t = v + a * wv + b * _cross(w, wv)
    return torch.cat([t, q], dim=-1)


def log_wv(p):
    """inverse of exp_wv: pose [...,7] -> [omega, v]."""
    q = p[..., 3:] / p[..., 3:].norm(dim=-1, keepdim=True)
    q = torch.where(q[..., 3:4] < 0, -q, q)
    n = q[..., :3].norm(dim=-1, keepdim=True)
    th = 2.0 * torch.atan2(n, q[..., 3:4])
    k = torch.where(n < 1e-12, 2.0 * torch.ones_like(n), th / torch.where(n < 1e-12, torch.ones_like(n), n))
    w = q[..., :3] * k
    t2 = (w * w).sum(-1, keepdim=True)
    t1 = t2.sqrt()
    small = t1 < 1e-8
    t1s = torch.where(small, torch.ones_like(t1), t1)
    c = torch.where(small, torch.ones_like(t1) / 12.0,
                    1.0 / (t1s * t1s) - (1.0 + torch.cos(t1s)) / (2.0 * t1s * torch.sin(t1s)))
    t = p[..., :3]
    wt = _cross(w, t)
    return torch.cat([w, t - 0.5 * wt + c * _cross(w, wt)], dim=-1)
