"""Edge-sharded bundle adjustment across GPUs (SURVEY.md 8(e), row "BA").

The dense BA shards by SOURCE frame: when all edges that share a depth map (same `ii`) live on one rank, the
per-edge Hessian blocks, the depth diagonal C (and its damping / sensed-depth prior), Q = 1/C, the coupling blocks
E and therefore every Schur contribution E Q E^T are local and ADDITIVE.  One all-reduce of the reduced system
(`(6P)^2 + 6P` floats: 14.6 KB at P = 10, 9.4 MB at P = 256) gives every rank H - S and v; the 6P x 6P solve and the
pose retraction are replicated; the depth back-substitution touches only the depth maps a rank owns.
Correlation volumes / lookups shard the same way (edges are independent; no collective).

The reference has no multi-GPU BA (its only split is tracker | mapper, examples/slam_demo.py:63-77).
`torch.distributed` backend: "nccl" (= RCCL) on the GPUs, "gloo" in tests/test_parallel_ba.py.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition_by_source(ii, world):
    """-> list of `world` sorted index arrays into the edge list; all edges with the same source frame land on the
    same rank; source frames are dealt largest-first to the least loaded rank (deterministic)."""
    ii = np.asarray(ii, np.int64)
    frames, counts = np.unique(ii, return_counts=True)
    order = np.lexsort((frames, -counts))
    load = np.zeros(world, np.int64)
    owner = {}
    for k in order:
        r = int(np.argmin(load))          # ties -> lowest rank
        owner[int(frames[k])] = r
        load[r] += counts[k]
    ranks = np.array([owner[int(f)] for f in ii], np.int64) if ii.size else np.zeros(0, np.int64)
    return [np.nonzero(ranks == r)[0] for r in range(world)]


def depth_rows(ii_shard, kf0, kf1):
    """sorted unique depth-map ids a BA over this shard carries: the window frames and the shard's sources
    (droid_kernels.cu:1702-1710 applied to the shard)."""
    return np.unique(np.concatenate([np.arange(kf0, kf1), np.asarray(ii_shard, np.int64)]))


def shard_eta(eta, kx_all, kx_shard):
    """rows of the global damping tensor (ordered like kx_all) for a shard's depth maps."""
    pos = np.searchsorted(kx_all, kx_shard)
    assert np.array_equal(np.asarray(kx_all)[pos], kx_shard)
    if isinstance(eta, torch.Tensor):
        return eta[torch.as_tensor(pos, device=eta.device)].contiguous()
    return np.ascontiguousarray(np.asarray(eta)[pos])


def allreduce_reduced_system(H, v, group=None):
    """the one exchange step of a sharded BA iteration: in-place sum of the reduced camera system over the ranks"""
    dist.all_reduce(H, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return H, v


class ShardedBA:
    """One rank's part of a sharded dense BA over the edge list (ii, jj) and the pose window [kf0, kf1)."""

    def __init__(self, ii_host, jj_host, kf0, kf1, device, rank=None, world=None, group=None):
        from . import ba_plan
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.kf0, self.kf1 = int(kf0), int(kf1)
        ii_host, jj_host = np.asarray(ii_host, np.int64), np.asarray(jj_host, np.int64)
        self.mine = partition_by_source(ii_host, self.world)[self.rank]
        self.kx_all = depth_rows(ii_host, kf0, kf1)
        self.kx = depth_rows(ii_host[self.mine], kf0, kf1)
        self.plan = ba_plan.BaPlan(ii_host[self.mine], jj_host[self.mine], kf0, kf1, device)
        self.ii = torch.from_numpy(ii_host[self.mine]).to(device)
        self.jj = torch.from_numpy(jj_host[self.mine]).to(device)
        self._sel = torch.from_numpy(self.mine).to(device)
        own = np.unique(ii_host[self.mine])
        if self.rank == 0:
            own = np.union1d(own, np.setdiff1d(np.arange(kf0, kf1), np.unique(ii_host)))
        self._owned = own
        self._kx_all_d = torch.from_numpy(self.kx_all).to(device)
        self._own_mask = torch.from_numpy(np.isin(self.kx_all, own).astype(np.float32)).to(device)[:, None, None]

    def iteration(self, poses, disps, intrinsics, extrinsics, disps_sens, targets, weights, eta, world_T_body,
                  prior_pose=None, clamp_min=0.001, reduce=True, sync_depths=True):
        """targets / weights [M,2,ht,wd] and eta [K',HW] are the GLOBAL tensors (replicated inputs); poses / disps are
        updated in place, identically on every rank (sync_depths=False: disps only for the maps this rank owns, the
        others keep their previous values)."""
        from . import ba_plan
        H, v, Q, E, w = ba_plan.reduced_camera_matrix(self.plan, poses, disps, intrinsics, extrinsics, disps_sens,
                                                      targets[self._sel].contiguous(), weights[self._sel].contiguous(),
                                                      shard_eta(eta, self.kx_all, self.kx), self.ii, self.jj)
        if reduce and self.world > 1:
            allreduce_reduced_system(H, v, self.group)
        sol = ba_plan.ba_solve(H, v, self.kf0, self.kf1, world_T_body, poses, extrinsics, prior_pose=prior_pose)
        # depth back-substitution: the plan carries every window frame, but only the maps this rank OWNS see all of their
        # edges here -- on the others Q / w hold the damping / sensed-depth prior alone and the update would be wrong
        # (ADVICE r01).  Update a copy, keep the owned rows' change, and sum the changes over the ranks: every map is
        # owned by exactly one rank, so the sum IS the exchange (one all-reduce of |kx_all| maps).
        before = disps[self._kx_all_d]
        ba_plan.solve_depth(self.plan, sol["dx"], disps, Q, E, w, clamp_min=clamp_min)
        dz = (disps[self._kx_all_d] - before) * self._own_mask
        if sync_depths and self.world > 1:
            dist.all_reduce(dz, op=dist.ReduceOp.SUM, group=self.group)
        disps[self._kx_all_d] = before + dz
        return sol

    def owned_depth_maps(self):
        """depth maps whose update on this rank is the real one: the sources of this rank's edges, plus (rank 0) the
        window frames that are nobody's source (their update is the prior-only one on every rank)"""
        return self._owned


# ---- replicated NeRF trainers: the table gradient lands SHARDED, the updated parameters are gathered ----
def shard_size(n_entries, world):
    """entries per trainer: the table split evenly, rounded up to 1024 (every trainer's shard then starts on a bin boundary)"""
    return ((n_entries + world - 1) // world + 1023) // 1024 * 1024 if world > 1 else n_entries


def _device_collectives(group=None):
    """True when the group's backend runs all-to-all / all-gather on device tensors (nccl = RCCL).  Decided from the backend
    name ('nccl' anywhere in it: a group created without an explicit backend reports 'cpu:gloo,cuda:nccl'), once per call and
    never from a caught exception; an unknown backend raises instead of silently staging through the host: a failing RCCL collective must surface as what it is."""
    name = str(dist.get_backend(group)).lower()
    if "nccl" in name:           # "nccl", or a per-device map such as "cpu:gloo,cuda:nccl": device tensors go to RCCL
        return True
    if name == "gloo":
        return False
    raise RuntimeError(f"nerfslam.parallel: process-group backend {name!r} is neither RCCL ('nccl') nor 'gloo'; refusing to "
                       "fall back to host staging silently")


def exchange_sharded(send, recv, out_shard, group=None):
    """send: [world * Ns] packed int64 gradient words of THIS trainer (shard r = entries [r Ns, (r + 1) Ns)); recv: scratch
    [world, Ns]; out_shard [Ns] <- the sum over the trainers of shard `rank`, added in rank order (integer: exact, and the same on
    every run).  One all-to-all: every trainer sends (world - 1) / world of its buffer once -- an all-reduce would move twice that
    and leave every trainer with sums it then does not need (Adam runs on the own shard only).  The gloo backend (CPU tests) has
    no device all-to-all and goes through host tensors; that branch is chosen by backend name, not by catching errors."""
    world, Ns = recv.shape
    assert send.numel() == world * Ns and out_shard.numel() == Ns
    if _device_collectives(group):
        dist.all_to_all_single(recv.view(-1), send, group=group)   # RCCL: device tensors straight onto xGMI; errors propagate
    else:                                                          # gloo (CPU tests): staged through host tensors
        h_in, h_out = send.cpu(), torch.empty((world * Ns,), dtype=send.dtype)
        dist.all_to_all_single(h_out, h_in, group=group)
        recv.view(-1).copy_(h_out)
    torch.sum(recv, dim=0, out=out_shard)
    return (world - 1) * Ns * send.element_size()                 # bytes this trainer put on the wire


def gather_shards(full, rank, group=None):
    """full: [world * n] tensor whose slice [rank n, (rank + 1) n) this trainer has just updated -> every trainer's slice, in
    place (the f16 working copy of the table after the sharded Adam step)."""
    world = dist.get_world_size(group)
    n = full.numel() // world
    mine = full[rank * n:(rank + 1) * n].clone()
    if _device_collectives(group):
        dist.all_gather_into_tensor(full, mine, group=group)
    else:
        h = torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(h, mine.cpu(), group=group)
        full.copy_(h)
    return (world - 1) * n * full.element_size()
