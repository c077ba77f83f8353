"""TrackingSLAM state machine (visual_frontend.py:240-368) on a synthetic sequence with stand-in networks that
return the TRUE induced flow: warm-up -> initialize -> per-keyframe tracking -> global BA at the last frame."""
import argparse

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_state_machine_runs_and_tracks(dev):
    from nerfslam import se3
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.slam import TrackingSLAM
    rng = np.random.default_rng(1)
    H, W, nfr = 96, 128, 14
    ht, wd = H // 8, W // 8
    intr = np.array([100.0, 100.0, W / 2, H / 2], np.float32)
    gt_poses = np.zeros((nfr + 2, 7), np.float32); gt_poses[:, 6] = 1
    for k in range(1, nfr + 2):
        gt_poses[k, :3] = 0.06 * k * np.array([1.0, 0.2, 0.1]) + rng.normal(0, 0.004, 3)
        gt_poses[k, 3:] = synth.quat_exp(rng.normal(0, 0.01, 3))
    gt_poses[10] = gt_poses[9]                                           # a frame without parallax: must be rejected
    yy, xx = np.meshgrid(np.linspace(0, 1, ht), np.linspace(0, 1, wd), indexing="ij")
    gt_disp = np.stack([0.5 + 0.2 * np.sin(3 * xx + 0.3 * k) * np.cos(2 * yy) for k in range(nfr + 2)]).astype(np.float32)
    gt_disp[10] = gt_disp[9]
    gtP, gtD = torch.from_numpy(gt_poses).to(dev), torch.from_numpy(gt_disp).to(dev)
    slam = None

    class Nets:
        def features(self, img):
            g = torch.Generator(device="cpu").manual_seed(int(img[0, 0, 0].item()))
            return torch.randn((128, ht, wd), generator=g).to(dev)

        def motion(self, corr, last_kf):
            assert corr.shape == (1, 1, 196, ht, wd)
            return torch.full((1, 1, ht, wd, 2), 3.0, device=dev)        # always "enough motion"

        def update(self, corr, motion, ii, jj):
            fe = slam.fe
            E = ii.shape[0]
            assert corr.shape == (1, E, 196, ht, wd) and torch.isfinite(corr).all()
            true_c = torch.empty((E, ht, wd, 2), device=dev)
            fr = torch.tensor([slam.kf_to_frame.get(i, 0) for i in range(args.buffer)], device=dev)
            P, D = gtP[fr].contiguous(), gtD[fr].contiguous()
            check(lib().ns_reproject(ptr(P), ptr(D), ptr(fe.intr8), ptr(ii), ptr(jj), ptr(true_c), None, E, ht, wd,
                                     stream_ptr()), "reproject")
            delta = (true_c - fe.reproject(ii, jj))[None]
            nk = torch.unique(ii).numel()
            return delta, torch.ones_like(delta), torch.full((nk, ht, wd), 1e-4, device=dev)

    args = argparse.Namespace(buffer=16, networks=Nets(), slam=True, global_ba=True)
    slam = TrackingSLAM("VioSLAM", args, dev)
    packets = []
    for k in range(nfr):
        img = np.full((H, W, 4), k, np.uint8)
        out = slam({"data": {"k": [k], "images": [img], "calibs": [intr], "depths": [1.0 / np.kron(gt_disp[k], np.ones((8, 8), np.float32)) if k == 0 else None],
                             "is_last_frame": k == nfr - 1}})
        assert out is not False
        packets.append(out[1])
        if k == 0:
            slam.fe.keyframe_thresh = 0.25                                # px at 1/8 resolution; the scene moves ~0.4 px / frame
    fe = slam.fe
    fr = [slam.kf_to_frame[i] for i in range(fe.kf_idx + 1)]
    assert slam.is_initialized and slam.stop_condition() and fe.kf_idx == nfr - 2, fe.kf_idx
    assert fr == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13], fr            # frame 10 was dropped by the keyframe test
    assert packets[-1]["is_last_frame"] and packets[-1]["cam0_poses"].shape[0] >= nfr - 2
    assert "cam0_poses" not in packets[3]                                 # warm-up: nothing dirty
    assert len(fe.graph.ii) == 0                                        # global BA cleared the graph (:1297-1300)
    n = fe.kf_idx + 1
    gtP = gtP[torch.tensor(fr, device=dev)]
    assert torch.isfinite(fe.cam0_T_world[:n]).all() and torch.isfinite(fe.cam0_idepths[:n]).all()
    # sensed depth on frame 0 fixes the scale -> the trajectory must match the ground truth
    dT = se3.log_wv(se3.mul(fe.cam0_T_world[:n].double(), se3.inv(gtP[:n].double())))
    assert dT.abs().max().item() < 2e-2, dT.abs().max()


def test_end_to_end_with_the_droid_architecture(dev):
    """the same state machine driven by the real network architecture (random weights: plumbing and shapes, no accuracy
    claim): encoders + ConvGRU through MIOpen under autocast, hidden-state bookkeeping across edge add / remove"""
    from nerfslam.droid_nets import DroidNetworks
    from nerfslam.slam import TrackingSLAM
    H, W, nfr = 96, 128, 12
    g = torch.Generator().manual_seed(0)
    nets = DroidNetworks(dev, weights=None, seed=0)
    slam = TrackingSLAM("VioSLAM", argparse.Namespace(buffer=16, networks=nets, slam=True, global_ba=True), dev)
    slam.motion_filter_thresh = -1.0                     # untrained motion head: accept every frame
    intr = np.array([100.0, 100.0, W / 2, H / 2], np.float32)
    base = torch.randint(0, 255, (H + 40, W + 40, 3), generator=g, dtype=torch.uint8).numpy()
    n_packets = 0
    for k in range(nfr):
        img = np.concatenate([base[2 * k:2 * k + H, 3 * k:3 * k + W], np.full((H, W, 1), 255, np.uint8)], -1)
        out = slam({"data": {"k": [k], "images": [img], "calibs": [intr], "depths": [None], "is_last_frame": k == nfr - 1}})
        assert out is not False
        n_packets += out[1] is not None
        if k == 0:
            slam.fe.keyframe_thresh = -1.0               # ... and keep every keyframe
    fe = slam.fe
    assert slam.is_initialized and slam.stop_condition() and fe.kf_idx == nfr - 1 and n_packets >= nfr - 1
    assert torch.isfinite(fe.cam0_T_world[:nfr]).all() and torch.isfinite(fe.cam0_idepths[:nfr]).all()
    assert (fe.cam0_idepths[:nfr] >= 0.001 - 1e-6).all()


@pytest.mark.parametrize("parallel", [False, True])
def test_demo_driver_sequential_slam_plus_nerf(dev, tmp_path, parallel):
    """examples/slam_demo.py wiring (DataModule -> SlamModule -> FusionModule, sequential mode) on a tiny .npz sequence:
    the tracker's packets reach the NeRF trainer, which trains on them until its stop condition"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("slam_demo", os.path.join(os.path.dirname(__file__), "..", "examples", "slam_demo.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    H, W, n = 96, 128, 11
    g = np.random.default_rng(0)
    base = g.integers(0, 255, (H + 40, W + 40, 3), dtype=np.uint8)
    imgs = np.stack([base[2 * k:2 * k + H, 3 * k:3 * k + W] for k in range(n)])
    seq = tmp_path / "seq.npz"
    np.savez(seq, images=imgs, intrinsics=np.array([100.0, 100.0, W / 2, H / 2], np.float32))
    args = demo.parse_args(["--slam", "--fusion", "nerf", "--dataset_dir", str(seq), "--buffer", "16", "--weights", "/nonexistent.pth",
                            "--stop_iters", "400"] + (["--parallel_run"] if parallel else []))
    # (--parallel_run on one GPU: the mapper spins free-running in its own thread / HIP stream, nerfslam.pipeline.StreamQueue)
    mods = demo.run(args, return_modules=True, tweak=lambda m: (setattr(m.slam, "motion_filter_thresh", -1.0), setattr(m.slam, "keyframe_thresh", -1.0)))
    slam, fusion = mods["slam"], mods["fusion"]
    assert fusion.shutdown and (slam.shutdown or parallel)
    assert fusion.fusion.ngp.nerf.training.n_images_for_training >= (1 if parallel else 9) and fusion.fusion.total_iters > 400
    assert np.isfinite(fusion.fusion.ngp.loss)


@pytest.mark.parametrize("nproc", [2, 3])
def test_demo_driver_multi_gpu_split_on_one_device(tmp_path, nproc):
    """examples/slam_demo.py --parallel_run --multi_gpu under torch.distributed.run (examples/slam_demo.py:63-77 of the
    reference): rank 0 tracks, ranks 1.. are free-running replicated trainers behind nerfslam.transport.PacketChannel.  All
    ranks on the one GPU of the test box over gloo (NS_DEMO_ONE_DEVICE): the same code path as the RCCL run except the
    backend name.  The run must end by itself (STOP, final barrier) with every rank exiting 0."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W, n = 96, 128, 11
    g = np.random.default_rng(0)
    base = g.integers(0, 255, (H + 40, W + 40, 3), dtype=np.uint8)
    imgs = np.stack([base[2 * k:2 * k + H, 3 * k:3 * k + W] for k in range(n)])
    seq = tmp_path / "seq.npz"
    np.savez(seq, images=imgs, intrinsics=np.array([100.0, 100.0, W / 2, H / 2], np.float32))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, NS_DEMO_DIST_BACKEND="gloo", NS_DEMO_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "examples", "slam_demo.py"), "--slam", "--fusion", "nerf", "--parallel_run",
           "--multi_gpu", "--dataset_dir", str(seq), "--buffer", "16", "--weights", "/nonexistent.pth", "--stop_iters", "400",
           "--force_keyframes"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import re
    rows = re.findall(r"slam_demo trainer (\d+): (\d+) training views, (\d+) iterations, (\d+) optimiser steps, parameter checksum (\S+)", r.stdout)
    assert sorted(int(x[0]) for x in rows) == list(range(1, nproc)), r.stdout[-1500:]
    # packets arrived and the trainers trained; how many views they saw before their stop condition depends on the race between
    # the tracker's start-up and the free-running trainers (as in the one-GPU --parallel_run test above)
    assert all(int(x[1]) >= 1 and int(x[2]) > 0 and int(x[3]) > 0 for x in rows), rows
    if nproc > 2:
        assert rows[0][3] == rows[1][3] and rows[0][4] == rows[1][4], rows                     # replicas in lockstep
