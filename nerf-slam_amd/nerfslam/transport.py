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

HEADER = 8  # int64 words: magic, n, H, W, kf_idx, is_last, fx*1e3, fy*1e3 ... kept small and explicit
MAGIC = 0x4E53


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
