"""RCCL transport for the --multi_gpu split (tracker on one GPU, NeRF trainers on others).

The reference moves the SLAM -> fusion packet to the CPU, pickles it through a
torch.multiprocessing.Queue and re-uploads it (visual_frontend.py:1355-1360, examples/slam_demo.py:63-77;
~4.7 MB per dirty keyframe).  Here the dirty keyframes are packed into ONE contiguous device buffer
(pose 28 B + uint8 RGB + idepth_up f32 + depth_cov_up f32 per keyframe) and shipped with a
torch.distributed point-to-point send (RCCL over xGMI; one link per destination) or a broadcast to R
replicated trainers; replicated trainers all-reduce their hash-grid and MLP gradients.

Backend "nccl" on GPUs (= RCCL), "gloo" in the CPU tests (tests/test_transport.py, world_size 2).
"""
import torch
import torch.distributed as dist

HEADER = 8  # int64 words: magic, n, H, W, kf_idx, is_last, payload bytes, kind
MAGIC = 0x4E53
KIND_PACKET, KIND_BARRIER, KIND_STOP = 0, 1, 2


def packet_nbytes(n, H, W):
    return n * (7 * 4 + 4 * 4 + 8 + 3 * H * W + 2 * 4 * H * W)


def pack(packet):
    """packet of TrackingFrontend.get_viz_out() -> (header int64[8] on CPU, payload uint8 [nbytes] on the
    packet's device).  Only what the mapper consumes is shipped (nerf_fusion.py:140-235)."""
    poses = packet["cam0_poses"].float().contiguous()
    n = poses.shape[0]
    imgs = packet["cam0_images"].contiguous()
    H, W = imgs.shape[-2:]
    # widest element type first so that every section stays naturally aligned inside the buffer
    parts = [packet["viz_idx"].long().contiguous().view(torch.uint8).reshape(-1),
             poses.view(torch.uint8).reshape(-1),
             packet["cam0_intrinsics"].float().contiguous().view(torch.uint8).reshape(-1),
             packet["cam0_idepths_up"].float().contiguous().view(torch.uint8).reshape(-1),
             packet["cam0_depths_cov_up"].float().contiguous().view(torch.uint8).reshape(-1),
             imgs.view(torch.uint8).reshape(-1)]
    payload = torch.cat(parts)
    header = torch.tensor([MAGIC, n, H, W, int(packet.get("kf_idx", 0)), int(bool(packet.get("is_last_frame", False))),
                           payload.numel(), 0], dtype=torch.int64)
    return header, payload


def unpack(header, payload):
    magic, n, H, W, kf_idx, last, nbytes, _ = (int(v) for v in header.tolist())
    assert magic == MAGIC and payload.numel() == nbytes
    off = 0

    def take(count, dtype, shape):
        nonlocal off
        nb = count * torch.empty((), dtype=dtype).element_size()
        t = payload[off:off + nb].view(dtype).reshape(shape)
        off += nb
        return t
    out = {"viz_idx": take(n, torch.int64, (n,)), "cam0_poses": take(n * 7, torch.float32, (n, 7)),
           "cam0_intrinsics": take(n * 4, torch.float32, (n, 4)),
           "cam0_idepths_up": take(n * H * W, torch.float32, (n, H, W)),
           "cam0_depths_cov_up": take(n * H * W, torch.float32, (n, H, W)),
           "cam0_images": take(n * 3 * H * W, torch.uint8, (n, 3, H, W)),
           "kf_idx": kf_idx, "is_last_frame": bool(last)}
    return out


def send_packet(packet, dst, group=None):
    """tracker side: one small header message + one payload message (device tensor, no host bounce)."""
    header, payload = pack(packet)
    dev = payload.device
    dist.send(header.to(dev), dst, group=group)
    dist.send(payload, dst, group=group)


def recv_packet(src, device, group=None):
    header = torch.zeros(HEADER, dtype=torch.int64, device=device)
    dist.recv(header, src, group=group)
    payload = torch.empty(int(header[6].item()), dtype=torch.uint8, device=device)
    dist.recv(payload, src, group=group)
    return unpack(header.cpu(), payload)


def broadcast_packet(packet, src, device, group=None):
    """one tracker, R replicated trainers: every destination is reached over its own xGMI link."""
    rank = dist.get_rank(group)
    if rank == src:
        header, payload = pack(packet)
        header = header.to(device)
    else:
        header = torch.zeros(HEADER, dtype=torch.int64, device=device)
    dist.broadcast(header, src, group=group)
    if rank != src:
        payload = torch.empty(int(header[6].item()), dtype=torch.uint8, device=device)
    dist.broadcast(payload, src, group=group)
    return unpack(header.cpu(), payload)


def allreduce_gradients(tensors, group=None):
    """replicated NeRF trainers: sum the gradient buffers in place (one flat bucket per tensor: the hash-grid
    gradient is a single 52 MB f32 buffer, the MLP gradient 40 KB), then average."""
    world = dist.get_world_size(group)
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(world)


class PacketChannel:
    """One tracker -> R FREE-RUNNING replicated trainers (examples/slam_demo.py:63-77 + fusion/fusion_module.py:30-45: the
    reference's trainer never blocks on its input queue and trains on every spin without a packet, nerf_fusion.py:249-253).

    Two planes:
      * control (tiny, host side): the tracker drops the 64-byte header into a mailbox in the job's rendezvous store
        (TCPStore: `set`); the LEADER trainer looks into the mailbox without blocking (`check`) and tells the other
        trainers what to do next with one small `gloo` broadcast per poll.  All trainers therefore take the same decision
        at the same point of their step sequence, which keeps the collectives below matched (a trainer that saw the packet
        one step earlier than another would otherwise leave its peers alone in the gradient all-reduce).  (A posted
        `irecv` cannot be polled: gloo only marks it complete inside `wait()`.)
      * data (RCCL over xGMI on the GPUs): the packed keyframes are broadcast from the tracker's HBM to every trainer's
        (each destination over its own link), enqueued asynchronously on the tracker so tracking never waits for it; the
        trainers' gradient all-reduce runs in their own sub-group.
    Control kinds: PACKET (a payload follows on the data plane), BARRIER (every rank meets in a control-plane barrier:
    brackets timed regions), STOP."""

    def __init__(self, device, tracker=0, trainers=None, control_group=None, data_group=None, trainer_control_group=None):
        self.device = torch.device(device)
        self.rank = dist.get_rank()
        world = dist.get_world_size()
        self.tracker = tracker
        self.trainers = list(trainers) if trainers is not None else [r for r in range(world) if r != tracker]
        self.leader = self.trainers[0]
        self.control, self.data, self.trainer_control = control_group, data_group, trainer_control_group
        self._inflight = []            # (work, tensors kept alive) of the tracker's asynchronous broadcasts
        self._seq = 0                  # next mailbox slot (tracker: to write, leader: to read)
        from torch.distributed.distributed_c10d import _get_default_store
        self._store = dist.PrefixStore("nerfslam_packet_channel", _get_default_store())
        self.bytes_sent = 0
        self.packets = 0
        self.max_inflight = 4          # payload broadcasts the tracker keeps in flight before it waits for the oldest (ADVICE r02:
                                       # ~4.7 MB per keyframe stay alive while the trainers lag; unbounded before)
        self.tracker_lost = False      # leader: the rendezvous store (hosted by the tracker process) stopped answering

    # ---- tracker --------------------------------------------------------------------------------------------
    def _reap(self, block=False):
        keep = []
        for work, hold in self._inflight:
            if block:
                work.wait()
            elif not work.is_completed():
                keep.append((work, hold))
        self._inflight = keep

    def publish(self, packet=None, kind=KIND_PACKET):
        """tracker side; returns immediately (header: mailbox write, payload: asynchronous device broadcast)"""
        self._reap()
        if kind == KIND_PACKET:
            header, payload = pack(packet)
        else:
            header, payload = torch.tensor([MAGIC, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int64), None
        header[7] = kind
        if payload is not None:
            while len(self._inflight) >= self.max_inflight:      # trainers are behind: wait for the oldest broadcast
                work, _hold = self._inflight.pop(0)
                work.wait()
            self._inflight.append((dist.broadcast(payload, self.tracker, group=self.data, async_op=True), payload))
            self.bytes_sent += payload.numel() * len(self.trainers)
            self.packets += 1
        self._store.set(f"h{self._seq}", header.numpy().tobytes())   # after the broadcast is enqueued: a trainer that reads
        self._seq += 1                                                # the header finds its peer already in the collective
        if kind == KIND_BARRIER:
            self.barrier()

    def barrier(self):
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        dist.barrier(group=self.control)

    def close(self):
        self.publish(kind=KIND_STOP)
        self._reap(block=True)

    # ---- trainers -------------------------------------------------------------------------------------------
    def poll(self):
        """trainer side, called by EVERY trainer at the same point of its loop.  -> (kind, packet | None) or None when
        nothing arrived (then train).  Never blocks on the tracker."""
        header = torch.zeros(HEADER, dtype=torch.int64)
        if self.rank == self.leader:
            key = f"h{self._seq}"
            try:
                found = self._store.check([key])
                if found:
                    import numpy as np
                    header = torch.from_numpy(np.frombuffer(self._store.get(key), dtype=np.int64).copy())
                    self._store.delete_key(key)
                    self._seq += 1
            except Exception:          # the store lives in the tracker's process: it is gone (crashed before publishing STOP).
                self.tracker_lost = True   # Tell every trainer to stop instead of polling a dead mailbox forever
                header = torch.tensor([MAGIC, 0, 0, 0, 0, 0, 0, KIND_STOP], dtype=torch.int64)
        if len(self.trainers) > 1:
            dist.broadcast(header, self.leader, group=self.trainer_control)
        if int(header[0]) != MAGIC:
            return None
        kind = int(header[7])
        if kind == KIND_PACKET:
            payload = torch.empty(int(header[6]), dtype=torch.uint8, device=self.device)
            dist.broadcast(payload, self.tracker, group=self.data)
            return kind, unpack(header, payload)
        if kind == KIND_BARRIER:
            self.barrier()
        return kind, None
