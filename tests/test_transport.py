"""N>1 path on CPU: world_size-2 gloo processes exchange a SLAM->mapper packet and all-reduce gradient
buffers through nerfslam.transport (the GPU build uses the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _make_packet(n=3, H=16, W=24, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"cam0_poses": torch.randn((n, 7), generator=g), "cam0_intrinsics": torch.rand((n, 4), generator=g),
            "viz_idx": torch.tensor([2, 5, 7][:n]), "cam0_images": torch.randint(0, 255, (n, 3, H, W), generator=g, dtype=torch.uint8),
            "cam0_idepths_up": torch.rand((n, H, W), generator=g), "cam0_depths_cov_up": torch.rand((n, H, W), generator=g),
            "kf_idx": 7, "is_last_frame": False}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerfslam import transport
    ok = True
    pkt = _make_packet()
    if rank == 0:
        transport.send_packet(pkt, 1)
    else:
        got = transport.recv_packet(0, "cpu")
        for k, v in pkt.items():
            ok &= torch.equal(got[k], v) if isinstance(v, torch.Tensor) else got[k] == v
    got = transport.broadcast_packet(pkt if rank == 0 else None, 0, "cpu")
    ok &= torch.equal(got["cam0_images"], pkt["cam0_images"]) and got["kf_idx"] == 7
    g = [torch.full((1000,), float(rank + 1)), torch.full((10,), float(10 * (rank + 1)))]
    transport.allreduce_gradients(g)
    ok &= bool((g[0] == 1.5).all() and (g[1] == 15.0).all())
    # replicated NeRF trainers: the packed fixed-point hash-grid gradient (two signed Q18 fields per int64 word) is summed
    # as int64 and still decodes to the sum of the fields, negative values and cross-field borrows included
    from nerfslam.ngp import pack_fixed, unpack_fixed
    rng = np.random.default_rng(10 + rank)
    a, b = rng.normal(0, 5, 4096), rng.normal(0, 5, 4096)
    words = torch.from_numpy(pack_fixed(a, b, 262144.0))
    dist.all_reduce(words)
    tot = [np.zeros(4096), np.zeros(4096)]
    for r in range(world):
        rr = np.random.default_rng(10 + r)
        x, y = rr.normal(0, 5, 4096), rr.normal(0, 5, 4096)
        tot[0] += np.rint(x * 262144.0) / 262144.0
        tot[1] += np.rint(y * 262144.0) / 262144.0
    lo, hi = unpack_fixed(words.numpy(), 262144.0)
    ok &= bool(np.array_equal(lo, tot[0]) and np.array_equal(hi, tot[1]))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_packet_roundtrip_single_process():
    from nerfslam import transport
    pkt = _make_packet(2, 8, 8, seed=3)
    h, p = transport.pack(pkt)
    assert p.numel() == transport.packet_nbytes(2, 8, 8)
    got = transport.unpack(h, p)
    for k, v in pkt.items():
        assert torch.equal(got[k], v) if isinstance(v, torch.Tensor) else got[k] == v


def test_gloo_world_size_2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == {0: True, 1: True}


# ------------------------------------------------------------------------------------------------
# PacketChannel: 1 tracker + 2 free-running replicated trainers (world size 3, gloo)
# ------------------------------------------------------------------------------------------------
def _channel_worker(rank, world, port, q):
    import time
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerfslam import transport
    trainers = [1, 2]
    control = dist.new_group([0, 1, 2], backend="gloo")
    tc = dist.new_group(trainers, backend="gloo")
    td = dist.new_group(trainers, backend="gloo")
    chan = transport.PacketChannel("cpu", tracker=0, trainers=trainers, control_group=control, data_group=None,
                                   trainer_control_group=tc)
    log = []
    if rank == 0:
        chan.publish(kind=transport.KIND_BARRIER)
        for k in range(4):
            time.sleep(0.02 * (k % 3))                     # irregular arrival times
            chan.publish(_make_packet(n=1 + k % 3, seed=k))
        chan.publish(kind=transport.KIND_BARRIER)
        chan.close()
        q.put((0, chan.packets, chan.bytes_sent))
    else:
        steps = 0
        grad = torch.zeros(64)
        while True:
            msg = chan.poll()
            if msg is None:                                 # a "training step": a collective among the trainers only
                g = torch.full((64,), float(rank))
                dist.all_reduce(g, group=td)
                grad += g
                steps += 1
                if rank == 2:
                    time.sleep(0.001)                       # one trainer slower than the other
                continue
            kind, pkt = msg
            if kind == transport.KIND_PACKET:
                log.append(("pkt", steps, int(pkt["cam0_poses"].shape[0]), float(pkt["cam0_idepths_up"].sum())))
            elif kind == transport.KIND_BARRIER:
                log.append(("barrier", steps))
            else:
                break
        q.put((rank, log, float(grad.sum()), steps))
    dist.barrier(group=control)
    dist.destroy_process_group()


def test_packet_channel_one_tracker_two_free_running_trainers():
    """both trainers take the same decision at the same step (identical logs incl. the step index at which each packet /
    barrier was consumed), every packet arrives intact, training steps happen between packets, nobody deadlocks"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_channel_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(3):
        r = q.get(timeout=180)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
    assert res[0][0] == 4 and res[0][1] > 0
    log1, g1, s1 = res[1]
    log2, g2, s2 = res[2]
    assert log1 == log2 and g1 == g2 and s1 == s2
    pk = [e for e in log1 if e[0] == "pkt"]
    assert [e[2] for e in pk] == [1, 2, 3, 1]
    for k, e in enumerate(pk):
        assert abs(e[3] - float(_make_packet(n=1 + k % 3, seed=k)["cam0_idepths_up"].sum())) < 1e-3
    assert [e[0] for e in log1].count("barrier") == 2 and s1 > 0
