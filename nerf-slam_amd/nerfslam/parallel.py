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


# ---- replicated NeRF trainers: what a step touched is exchanged, not the table ----
LIST_CLASS = 1 << 16      # the lists travel in multiples of 65536 pairs (1 MiB): every trainer sends the same, agreed, length


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


def list_class(max_count):
    """length (in pairs) at which lists holding at most `max_count` valid pairs are exchanged"""
    return max(LIST_CLASS, -(-int(max_count) // LIST_CLASS) * LIST_CLASS)


def gather_counts_begin(count, group=None):
    """count: [1] int32 (device).  Starts the all-gather of the trainers' list lengths and their copy to pinned host memory;
    -> handle for gather_counts_end.  Whatever the caller enqueues between the two calls runs on the device while the host
    waits for the lengths: the host read is the ONE synchronisation point of a replicated step (the length of the list exchange
    has to be the same on every trainer and is not known before the step's table gradient has run), and it waits for an EVENT
    behind the tiny all-gather, not for the stream."""
    world = dist.get_world_size(group) if (group is not None or dist.is_initialized()) else 1
    allc = torch.empty((world,), dtype=torch.int32, device=count.device)
    if _device_collectives(group):
        dist.all_gather_into_tensor(allc, count, group=group)
        host = torch.empty((world,), dtype=torch.int32, pin_memory=True)
        host.copy_(allc, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return allc, host, ev
    h = torch.empty((world,), dtype=torch.int32)                 # gloo (CPU tests): staged through host tensors
    dist.all_gather_into_tensor(h, count.cpu(), group=group)
    allc.copy_(h)
    return allc, h, None


def gather_counts_end(handle):
    """-> ([world] int32 on the device, list of ints on the host)"""
    allc, host, ev = handle
    if ev is not None:
        ev.synchronize()
    return allc, [int(v) for v in host.tolist()]


def gather_counts(count, group=None):
    return gather_counts_end(gather_counts_begin(count, group))


def gather_lists(mine, recv, n_pairs, group=None):
    """mine: [cap, 2] int64 (entry, packed sum) pairs of THIS trainer, the first `n_pairs` of them travel; recv: [world, n_pairs, 2]
    <- every trainer's prefix.  One all-gather: on the xGMI mesh every pair of trainers has its own link, each carries
    n_pairs x 16 B per step (a step of the default grid touches ~0.9 M of 12.6 M entries: ~14 MB instead of the 100-MB dense
    packed gradient + the 25-MB table that rounds 3-5 moved).  -> bytes this trainer put on the wire."""
    world = recv.shape[0]
    src = mine[:n_pairs]
    if _device_collectives(group):
        dist.all_gather_into_tensor(recv.view(-1), src.reshape(-1), group=group)     # RCCL: device tensors straight onto xGMI
    else:                                                                          # gloo (CPU tests): staged through host tensors
        h = torch.empty((world * n_pairs * 2,), dtype=torch.int64)
        dist.all_gather_into_tensor(h, src.reshape(-1).cpu(), group=group)
        recv.view(-1).copy_(h)
    return (world - 1) * n_pairs * 16
