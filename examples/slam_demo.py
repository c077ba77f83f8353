#!/usr/bin/env python3
"""Driver with the reference's command line (/root/reference/examples/slam_demo.py:20-61) and wiring
(:62-190): DataModule -> SlamModule("VioSLAM") -> FusionModule("nerf").

--parallel_run --multi_gpu runs one process per GPU under torch.distributed (RCCL): rank 0 tracks on its GPU
and ships the dirty keyframes with nerfslam.transport.PacketChannel; ranks >= 1 are replicated NeRF trainers that never
block on the tracker (they train on every poll without a packet) and all-reduce their gradients.  Launch with
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/slam_demo.py --slam \
        --fusion nerf --parallel_run --multi_gpu ...
Datasets and the DROID weights are not part of this project: `--dataset_dir` takes a .npz sequence
(images [N,H,W,3] uint8, intrinsics [4], optional depths [N,H,W]) and `--weights` a DROID-SLAM checkpoint
(`droid.pth`; loaded into nerfslam.droid_nets with the reference's key remapping).  Without a checkpoint the
networks run with random weights (plumbing only).
"""
import argparse
import os
import sys
from queue import Queue

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nerf-slam_amd"))
from nerfslam.pipeline import DataModule, FusionModule, SlamModule, StreamQueue, spin_in_thread  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="NeRF-SLAM demo (MI355X build)")
    p.add_argument("--parallel_run", action="store_true")
    p.add_argument("--multi_gpu", action="store_true")
    p.add_argument("--initial_k", type=int, default=0)
    p.add_argument("--final_k", type=int, default=-1)
    p.add_argument("--img_stride", type=int, default=1)
    p.add_argument("--stereo", action="store_true")
    p.add_argument("--weights", default="droid.pth")
    p.add_argument("--buffer", type=int, default=512)
    p.add_argument("--dataset_dir", type=str, default="")
    p.add_argument("--dataset_name", type=str, default="npz")
    p.add_argument("--mask_type", type=str, default="ours", choices=["no_depth", "raw", "ours", "ours_w_thresh"])
    p.add_argument("--slam", action="store_true")
    p.add_argument("--fusion", type=str, default="", choices=["tsdf", "sigma", "nerf", ""])
    p.add_argument("--gui", action="store_true")
    p.add_argument("--width", "--screenshot_w", type=int, default=0)
    p.add_argument("--height", "--screenshot_h", type=int, default=0)
    p.add_argument("--network", default="")
    p.add_argument("--eval", action="store_true")
    p.add_argument("--stop_iters", type=int, default=25000, help="NeRF training iterations (nerf_fusion.py:54 hard-codes 25000)")
    p.add_argument("--force_keyframes", action="store_true", help=argparse.SUPPRESS)      # test aid (random-weight runs)
    return p.parse_args(argv)


def load_sequence(args):
    if not args.dataset_dir:
        raise SystemExit("slam_demo: --dataset_dir <sequence.npz> is required")
    z = np.load(args.dataset_dir)
    n = z["images"].shape[0] if args.final_k < 0 else min(args.final_k, z["images"].shape[0])
    ks = list(range(args.initial_k, n, args.img_stride))
    for k in ks:
        yield {"k": [k], "images": [z["images"][k]], "poses": [np.eye(4, dtype=np.float32)], "t_cams": [float(k)],
               "depths": [z["depths"][k] if "depths" in z else None], "calibs": [z["intrinsics"]],
               "is_last_frame": k == ks[-1]}


def run(args, return_modules=False, tweak=None):
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    split = args.parallel_run and args.multi_gpu and world > 1
    chan = None
    if split:
        # rank 0 tracks, ranks 1.. are replicated FREE-RUNNING NeRF trainers (fusion_module.py:30-45: the reference's mapper
        # never blocks on its input and trains on every spin without a packet): nerfslam.transport.PacketChannel -- headers
        # through the rendezvous store, keyframe payloads over RCCL, gradients all-reduced in the trainer sub-group.
        # NS_DEMO_DIST_BACKEND=gloo + NS_DEMO_ONE_DEVICE=1 run the same topology on a one-GPU box (tests).
        import torch.distributed as dist
        from nerfslam import transport
        backend = os.environ.get("NS_DEMO_DIST_BACKEND", "nccl")
        local = 0 if os.environ.get("NS_DEMO_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        trainers = list(range(1, world))
        control = dist.new_group(list(range(world)), backend="gloo")
        trainer_control = dist.new_group(trainers, backend="gloo")
        trainer_data = dist.new_group(trainers, backend=backend) if len(trainers) > 1 else None
        chan = transport.PacketChannel(torch.device("cuda", local), tracker=0, trainers=trainers, control_group=control,
                                       data_group=None, trainer_control_group=trainer_control)
    dev = f"cuda:{torch.cuda.current_device()}"
    args_seq = argparse.Namespace(**{**vars(args), "parallel_run": False})
    if split and rank > 0:                                   # NeRF trainer rank
        args_seq.trainer_group = trainer_data
        fusion = FusionModule("nerf", args_seq, device=dev)
        fusion.initialize_module()
        import time
        idle = 0
        while True:                                          # (a trainer that reached its stop condition keeps answering the
            msg = chan.poll()                                #  tracker's broadcasts until STOP: collectives must stay matched)
            if msg is None:
                if not fusion.shutdown:
                    fusion.spin_once(False)                  # nothing arrived: train (nerf_fusion.py:249-253)
                    idle = 0
                else:
                    idle = min(idle + 1, 50)                 # done training: back off (1 .. 50 ms) instead of hammering the
                    time.sleep(0.001 * idle)                 # tracker's rendezvous store with a poll per millisecond
                continue
            kind, pkt = msg
            if kind == transport.KIND_PACKET and not fusion.shutdown:
                fusion.spin_once({"slam": [None, pkt]})
            elif kind == transport.KIND_STOP:
                break
        ngp = fusion.fusion.ngp
        net = ngp._net
        print("slam_demo trainer %d: %d training views, %d iterations, %d optimiser steps, parameter checksum %.6f" % (
            rank, int(ngp.nerf.training.n_images_for_training), int(fusion.fusion.total_iters), int(ngp.training_step),
            float(net.grid_half[:net.n_grid].double().sum().item()) + float(net.mlp_master.double().sum().item())), flush=True)
        dist.barrier(group=control)
        if return_modules:
            return {"data": None, "slam": None, "fusion": fusion}
        dist.destroy_process_group()
        return
    data_q, slam_q = Queue(), Queue()
    data = DataModule(args.dataset_name, args_seq, dataset=load_sequence(args))
    data.register_output_queue(data_q)
    slam = fusion = None
    if args.slam:
        from nerfslam.droid_nets import DroidNetworks
        have = os.path.exists(args.weights)
        if not have:
            print(f"slam_demo: {args.weights} not found -- running the DROID architecture with RANDOM weights (no tracking accuracy)")
        args_seq.networks = DroidNetworks(dev, weights=args.weights if have else None, buffer=args.buffer)
        slam = SlamModule("VioSLAM", args_seq, device=dev)
        slam.register_input_queue("data", data_q)
        if split:
            slam.register_output_callback(lambda out: chan.publish(out[1]) if (out and out[1] and "cam0_poses" in out[1]) else None)
    threaded = bool(args.parallel_run and args.fusion and not split and slam is not None)
    if args.fusion and not split:
        if threaded:
            # --parallel_run on one GPU: the mapper spins in its own host thread on its own HIP stream and never blocks on
            # its input (it trains on every spin without a packet, fusion_module.py:30-45); the tracker stays on this thread
            args_par = argparse.Namespace(**{**vars(args_seq), "parallel_run": True})
            fusion = FusionModule(args.fusion, args_par, device=dev)
            slam_q = StreamQueue(maxsize=8)
        else:
            fusion = FusionModule(args.fusion, args_seq, device=dev)
        if slam:
            slam.register_output_queue(slam_q)
            fusion.register_input_queue("slam", slam_q)
    if slam is not None and (tweak is not None or args.force_keyframes):
        slam.initialize_module()
        if args.force_keyframes:                             # test aid: every frame passes the motion filter and stays a keyframe
            slam.slam.motion_filter_thresh = slam.slam.keyframe_thresh = -1.0
        if tweak is not None:
            tweak(slam)
    if threaded:
        fusion.initialize_module()
        worker = spin_in_thread(fusion, dev)
        slam_q.consumer_alive = lambda: worker.is_alive() and getattr(fusion, "error", None) is None
        # the tracker on a stream of its own, too: the legacy default stream synchronises implicitly with every blocking stream,
        # and the internal streams of the mapper's HIP graphs are blocking ones (bench.py: 97 -> 104 frames/s)
        # -- taken once the mapper has instantiated its training graphs: a stream gets its hardware queue when it is first used,
        # and one used before the graphs existed ended up sharing a queue with one of their branches (103 -> 93 frames/s)
        own = None
        try:
            while data.spin() and slam.spin() and not fusion.shutdown:
                if own is None and getattr(getattr(getattr(fusion.fusion, "ngp", None), "_net", None), "_pair", None) is not None:
                    torch.cuda.synchronize()
                    own = torch.cuda.stream(torch.cuda.Stream(device=dev))
                    own.__enter__()
            torch.cuda.current_stream().synchronize()
        finally:
            if own is not None:
                own.__exit__(None, None, None)
        while worker.is_alive() and not fusion.shutdown:      # the data ran out: let the mapper reach its stop condition
            worker.join(timeout=0.05)
        if getattr(fusion, "error", None) is not None:        # the mapper thread died: surface its exception here
            raise fusion.error
        if return_modules:
            return {"data": data, "slam": slam, "fusion": fusion}
        return
    while data.spin() and (slam is None or slam.spin()) and (fusion is None or fusion.spin()):
        pass
    if chan is not None:
        chan.close()                                         # STOP to the trainers, payload broadcasts drained
        dist.barrier(group=control)                          # (the rendezvous store lives in this process: leave last)
        if not return_modules:
            dist.destroy_process_group()
    while fusion is not None and not fusion.shutdown and fusion.spin():
        pass
    if return_modules:
        return {"data": data, "slam": slam, "fusion": fusion}


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    run(parse_args())
