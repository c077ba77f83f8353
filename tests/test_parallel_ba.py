"""Sharded BA: the reduced camera system is additive over shards that keep every source frame's
edges together.  CPU: world_size-2 gloo processes compute their shard with the ORACLE (the HIP kernels need a GPU)
through the product partition / all-reduce code; GPU: the HIP kernels on shards, summed in one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth


def test_partition_keeps_sources_together():
    from nerfslam.parallel import partition_by_source
    rng = np.random.default_rng(0)
    ii = rng.integers(3, 14, 120)
    for world in (1, 2, 3, 8):
        parts = partition_by_source(ii, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(120))
        owners = {}
        for r, p in enumerate(parts):
            for f in np.unique(ii[p]):
                assert owners.setdefault(int(f), r) == r
        sizes = [len(p) for p in parts]
        if world in (2, 3):
            assert max(sizes) - min(sizes) <= np.bincount(ii).max()


def _problem():
    pr = synth.make_problem(ht=12, wd=16, P=6, M=30, seed=3, kf0=3, extra_fixed=1, sensed_frac=0.2)
    return pr, pr["ii"], pr["jj"]


def _inputs(pr, ii, jj, kf0, kf1):
    from nerfslam.parallel import depth_rows
    kx = depth_rows(ii, kf0, kf1)
    return pr["targets"], pr["weights"], pr["eta"].reshape(len(kx), -1), kx


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from nerfslam.parallel import allreduce_reduced_system, depth_rows, partition_by_source, shard_eta
    pr, ii, jj = _problem()
    kf0, kf1 = 3, 9
    targets, weights, eta, kx = _inputs(pr, ii, jj, kf0, kf1)
    mine = partition_by_source(ii, world)[rank]
    kx_r = depth_rows(ii[mine], kf0, kf1)
    H, v, *_ = oracle.reduced_camera_matrix(pr["poses"], pr["disps"], pr["intr"], pr["extr"], pr["disps_sens"],
                                            targets[mine], weights[mine], shard_eta(eta, kx, kx_r), ii[mine], jj[mine],
                                            kf0, kf1)
    Ht, vt = torch.from_numpy(H.copy()), torch.from_numpy(v.copy())
    allreduce_reduced_system(Ht, vt)
    Hf, vf, *_ = oracle.reduced_camera_matrix(pr["poses"], pr["disps"], pr["intr"], pr["extr"], pr["disps_sens"], targets,
                                              weights, eta, ii, jj, kf0, kf1)
    ok = np.abs(Ht.numpy() - Hf).max() <= 2e-4 * np.abs(Hf).max() and np.abs(vt.numpy() - vf).max() <= 2e-4 * np.abs(vf).max()
    q.put((rank, bool(ok), len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2_sharded_system_equals_full(oracle_mod):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res) and sum(n for _, _, n in res) == 30 and min(n for _, _, n in res) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_hip_shards_sum_to_the_full_system_and_update(dev, world):
    from nerfslam import ba_plan
    from nerfslam.parallel import ShardedBA
    pr, ii, jj = _problem()
    kf0, kf1 = 3, 9
    targets, weights, eta, kx = _inputs(pr, ii, jj, kf0, kf1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    poses, disps, intr, extr, sens = (t(pr[k]) for k in ("poses", "disps", "intr", "extr", "disps_sens"))
    T, W, ETA = t(targets), t(weights), t(eta)
    full = ba_plan.BaPlan(ii, jj, kf0, kf1, dev)
    Hf, vf, Qf, Ef, wf = ba_plan.reduced_camera_matrix(full, poses, disps, intr, extr, sens, T, W, ETA, t(ii), t(jj))
    shards = [ShardedBA(ii, jj, kf0, kf1, dev, rank=r, world=world) for r in range(world)]
    Hs, vs = torch.zeros_like(Hf), torch.zeros_like(vf)
    parts = []
    for sh in shards:
        from nerfslam.parallel import shard_eta
        H, v, Q, E, w = ba_plan.reduced_camera_matrix(sh.plan, poses, disps, intr, extr, sens, T[sh._sel].contiguous(),
                                                      W[sh._sel].contiguous(), shard_eta(ETA, sh.kx_all, sh.kx), sh.ii, sh.jj)
        Hs += H; vs += v
        parts.append((Q, E, w))
    assert (Hs - Hf).abs().max().item() <= 2e-4 * Hf.abs().max().item()
    assert (vs - vf).abs().max().item() <= 2e-4 * vf.abs().max().item()
    # replicated solve on the summed system, local depth updates: the union equals the unsharded update
    wTb = poses.clone()
    sol = ba_plan.ba_solve(Hf, vf, kf0, kf1, wTb.clone(), poses.clone(), extr, retract=False)
    d_full = disps.clone()
    ba_plan.solve_depth(full, sol["dx"], d_full, Qf, Ef, wf, clamp_min=0.001)
    d_sh = disps.clone()
    for sh, (Q, E, w) in zip(shards, parts):
        d_r = disps.clone()
        ba_plan.solve_depth(sh.plan, sol["dx"], d_r, Q, E, w, clamp_min=0.001)
        own = torch.from_numpy(sh.owned_depth_maps()).to(dev)
        d_sh[own] = d_r[own]
    touched = torch.from_numpy(np.unique(ii)).to(dev)
    assert (d_sh[touched] - d_full[touched]).abs().max().item() <= 1e-5 * d_full.abs().max().item()


@pytest.mark.gpu
def test_sharded_iteration_touches_only_owned_depth_maps(dev):
    """ShardedBA.iteration: a rank changes only the depth maps it owns; the ownership sets of the ranks
    partition the depth rows of the problem, so the summed change is the exchange."""
    from nerfslam.parallel import ShardedBA
    pr, ii, jj = _problem()
    kf0, kf1 = 3, 9
    targets, weights, eta, kx = _inputs(pr, ii, jj, kf0, kf1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    owned = []
    for r in range(2):
        poses, disps, intr, extr, sens = (t(pr[k]) for k in ("poses", "disps", "intr", "extr", "disps_sens"))
        sh = ShardedBA(ii, jj, kf0, kf1, dev, rank=r, world=2)
        before = disps.clone()
        sol = sh.iteration(poses, disps, intr, extr, sens, t(targets), t(weights), t(eta), poses.clone(), reduce=False,
                           sync_depths=False)
        assert sol["info"].item() == 0
        changed = torch.nonzero((disps != before).flatten(1).any(1)).flatten().cpu().numpy()
        assert set(changed.tolist()) <= set(sh.owned_depth_maps().tolist())
        assert len(changed) > 0
        owned.append(sh.owned_depth_maps())
    assert len(np.intersect1d(owned[0], owned[1])) == 0
    assert np.array_equal(np.union1d(owned[0], owned[1]), kx)


# ---- replicated NeRF trainers: exchange of the touched-entry lists (nerfslam.parallel.gather_counts / gather_lists) ----
def _exchange_worker(rank, world, port, n_entries, q):
    from nerfslam.parallel import LIST_CLASS, gather_counts, gather_lists, list_class
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        # (entry, packed word) pairs as a trainer's flush emits them: distinct entries, two signed 32-bit fixed-point fields per word
        n_mine = 700 + 300 * rank
        entries = torch.randperm(n_entries, generator=g)[:n_mine]
        lo = torch.randint(-2 ** 20, 2 ** 20, (n_mine,), generator=g, dtype=torch.int64)
        hi = torch.randint(-2 ** 20, 2 ** 20, (n_mine,), generator=g, dtype=torch.int64)
        mine = torch.zeros((list_class(n_entries), 2), dtype=torch.int64)   # capacity: every entry, in whole exchange units
        mine[:n_mine, 0], mine[:n_mine, 1] = entries, lo + (hi << 32)
        counts_dev, counts = gather_counts(torch.tensor([n_mine], dtype=torch.int32))
        n_pairs = list_class(max(counts))
        recv = torch.zeros((world, n_pairs, 2), dtype=torch.int64)
        wire = gather_lists(mine, recv, n_pairs)
        # what every trainer then does with the lists (ns_ngp_sparse_table_update on the device): integer sums per entry
        acc = torch.zeros(n_entries, dtype=torch.int64)
        for r in range(world):
            acc.index_add_(0, recv[r, :counts[r], 0], recv[r, :counts[r], 1])
        dense = torch.zeros(n_entries, dtype=torch.int64)                # what a dense all-reduce of packed words would have given
        dense[entries] = lo + (hi << 32)
        dist.all_reduce(dense)
        ok = counts == [700 + 300 * r for r in range(world)] and counts_dev.tolist() == counts and n_pairs == LIST_CLASS and \
            wire == (world - 1) * n_pairs * 16 and torch.equal(acc, dense)
        q.put((rank, bool(ok), bool(torch.equal(recv[rank, :n_mine], mine[:n_mine]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_touched_entry_list_exchange_equals_allreduce(world):
    """the all-gather of the trainers' (entry, packed sum) lists -- agreed length, a multiple of 65536 pairs -- followed by integer
    sums per entry gives every trainer exactly the dense all-reduce of the packed gradient; wire bytes per trainer are
    (world - 1) x length x 16 B"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_exchange_worker, args=(r, world, port, 5000, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    assert res == [(r, True, True) for r in range(world)]


def _state_worker(rank, world, port, q):
    """TrackingSLAM._replicate_source_frame_state on a stand-in frontend (CPU tensors, gloo): every source frame has one owner;
    the non-owner's rows hold STALE garbage -- inf / NaN included -- and its has_up flags are whatever they were."""
    import types
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerfslam.slam import TrackingSLAM
    nbuf, ht, wd = 8, 3, 4
    g = torch.Generator().manual_seed(7)
    truth = {k: torch.rand((nbuf,) + shp, generator=g) + 0.1 for k, shp in
             (("damping", (ht, wd)), ("cam0_idepths_up", (8 * ht, 8 * wd)), ("cam0_depths_cov_up", (8 * ht, 8 * wd)))}
    ii_h = np.array([1, 1, 2, 3, 3, 5, 5, 6])                      # sources 1, 2, 3, 5, 6
    own = np.array([0, 1, 2, 3, 4]) if rank == 0 else np.array([5, 6, 7])      # rank 0 owns sources 1, 2, 3; rank 1 owns 5, 6
    mine = np.unique(ii_h[own])
    upsampled = {1, 2, 5}                                          # frames whose owner ran the upsampling (3 and 6: no upmask)
    fe = types.SimpleNamespace(has_up=torch.zeros(nbuf, dtype=torch.bool))
    for k, t in truth.items():
        buf = torch.full_like(t, float("nan"))                     # stale everywhere ...
        buf[::2] = float("inf")
        buf[torch.from_numpy(mine)] = t[torch.from_numpy(mine)]    # ... except the rows this rank owns
        setattr(fe, k, buf)
    fe.has_up[:] = rank == 1                                       # stale flags on the non-owner: all set on rank 1, none on rank 0
    for f in mine:
        fe.has_up[f] = int(f) in upsampled
    me = types.SimpleNamespace(fe=fe, device=torch.device("cpu"))
    TrackingSLAM._replicate_source_frame_state(me, ii_h, own, None)
    src = torch.tensor([1, 2, 3, 5, 6])
    ok = all(torch.equal(getattr(fe, k)[src], t[src]) for k, t in truth.items())
    ok = ok and fe.has_up[src].tolist() == [True, True, False, True, False]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_source_frame_state_exchange_ignores_stale_rows():
    """the owner-masked all-reduce must SELECT the owner's rows (a stale inf / NaN times 0 is NaN and would poison
    every rank), and a frame counts as upsampled only if its owner upsampled it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_state_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
