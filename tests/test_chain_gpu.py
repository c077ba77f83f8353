"""Parity of the COMPOSED tracking step: TrackingFrontend.update() (reproject -> lookup -> injected update operator ->
2 x [linearise -> Schur -> device Cholesky + retraction -> depth back-substitution] -> covariances) against
oracle.chain_oracle.ChainOracle, the same sequence composed from the per-kernel CPU oracle (visual_frontend.py:370-470,
:1071-1232), over several keyframes at the BASELINE shape (640x480 -> 60x80):

  steps 0-1  window [0, 7): frame-0 prior active, every pose free
  steps 2-3  edges out of frames 0..2 moved to the inactive list: window shifts to [3, 7), poses 0..2 fixed, their depths
             still optimised through the inactive edges
  steps 4-5  keyframe 7 added (new edges, seeded pose / depth)
  steps 6-7  window shifts to [5, 8); inactive edges touching frames < 2 drop out of the BA (kf0 - 3 rule, :420)

The update operator is replaced on BOTH sides by the same function of the current reprojection: delta = 0.7 x (flow induced
by a ground-truth scene - current reprojection) + seeded noise, seeded weights and damping.  Tolerances (north star "within
a stated fp32 tolerance"): poses 1e-4 (translation, quaternion), inverse depths 1e-4 relative, covariances 2e-3 relative.
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _scene(nkf, ht, wd, rng):
    gt_poses = np.zeros((nkf, 7), np.float32); gt_poses[:, 6] = 1
    for k in range(1, nkf):
        gt_poses[k, :3] = 0.03 * k * np.array([1.0, 0.2, 0.1]) + rng.normal(0, 0.004, 3)
        gt_poses[k, 3:] = synth.quat_exp(rng.normal(0, 0.008, 3))
    yy, xx = np.meshgrid(np.linspace(0, 1, ht), np.linspace(0, 1, wd), indexing="ij")
    gt_disp = np.stack([0.5 + 0.3 * np.sin(3 * xx + k) * np.cos(2 * yy) for k in range(nkf)]).astype(np.float32)
    return gt_poses, gt_disp


def _neigh(lo, hi, r=3):
    ii, jj = [], []
    for i in range(lo, hi + 1):
        for j in range(lo, hi + 1):
            if 0 < abs(i - j) <= r:
                ii.append(i); jj.append(j)
    return np.array(ii, np.int64), np.array(jj, np.int64)


@pytest.mark.parametrize("shape", [(60, 80), (12, 16)], ids=["c640", "small"])
def test_update_ba_chain_vs_composed_oracle(oracle_mod, dev, shape):
    from nerfslam._lib import check, lib, ptr, stream_ptr
    from nerfslam.frontend import TrackingFrontend
    from oracle.chain_oracle import ChainOracle
    ht, wd = shape
    H, W, buf, nkf = ht * 8, wd * 8, 10, 8
    intr = np.array([0.5 * W, 0.5 * W, (W - 1) / 2, (H - 1) / 2], np.float32)
    rng = np.random.default_rng(5)
    gt_poses, gt_disp = _scene(buf, ht, wd, rng)
    gtP, gtD = torch.from_numpy(gt_poses).to(dev), torch.from_numpy(gt_disp).to(dev)
    n_steps, Emax = 8, 64
    noise = [rng.normal(0, 0.05, (Emax, ht, wd, 2)).astype(np.float32) for _ in range(n_steps)]
    wts = [rng.uniform(0.2, 1.0, (Emax, ht, wd, 2)).astype(np.float32) for _ in range(n_steps)]
    damp = [rng.uniform(1e-4, 2e-2, (buf, ht, wd)).astype(np.float32) for _ in range(n_steps)]
    step = [0]

    # ---- product --------------------------------------------------------------------------------------------------------
    fe = None

    def op_dev(corr, motion, ii, jj):
        assert corr.shape == (1, ii.shape[0], 196, ht, wd) and torch.isfinite(corr).all()
        E, s = ii.shape[0], step[0]
        true_c = torch.empty((E, ht, wd, 2), device=dev)
        check(lib().ns_reproject(ptr(gtP), ptr(gtD), ptr(fe.intr8), ptr(ii), ptr(jj), ptr(true_c), None, E, ht, wd,
                                 stream_ptr()), "reproject")
        cur = fe.reproject(ii, jj)
        delta = 0.7 * (true_c - cur) + torch.from_numpy(noise[s][:E]).to(dev)
        nk = len(np.unique(fe.graph.ii))
        return delta[None], torch.from_numpy(wts[s][:E]).to(dev)[None], torch.from_numpy(damp[s][:nk]).to(dev)

    g = torch.Generator().manual_seed(0)
    fe = TrackingFrontend(buf, H, W, intr, dev, feature_fn=None, update_op=op_dev)
    for k in range(nkf):
        fe.set_keyframe(k, torch.full((3, H, W), k, dtype=torch.uint8), fmap=torch.randn((128, ht, wd), generator=g))
    fe.prior_pose = fe.world_T_body[0].clone()
    fe.cam0_idepths[:] = 0.6
    fe.cam0_idepths_sensed[0] = gtD[0]
    fe.cam0_idepths[0] = gtD[0]

    # ---- oracle chain -----------------------------------------------------------------------------------------------------
    oc = ChainOracle(buf, ht, wd, intr / 8.0)
    oc.prior_pose = oc.world_T_body[0].astype(np.float64).copy()
    oc.disps[:] = 0.6
    oc.disps_sens[0] = gt_disp[0]
    oc.disps[0] = gt_disp[0]

    def op_host(coords1, ii, jj):
        E, s = ii.shape[0], step[0]
        true_c = oracle_mod.reproject(gt_poses, gt_disp, intr / 8.0, ii, jj)[0].astype(np.float32)
        delta = np.float32(0.7) * (true_c - coords1) + noise[s][:E]
        return delta, wts[s][:E], damp[s][:len(np.unique(ii))]

    def both(fn_fe, fn_oc):
        fn_fe(); fn_oc()
        assert fe.graph.ii.tolist() == oc.ii.tolist() and fe.graph.jj.tolist() == oc.jj.tolist()
        assert fe.graph.ii_inactive.tolist() == oc.ii_in.tolist() and fe.graph.jj_inactive.tolist() == oc.jj_in.tolist()

    def compare(tag, n):
        cp, od = fe.cam0_T_world[:n].cpu().numpy(), fe.cam0_idepths[:n].cpu().numpy()
        wb = fe.world_T_body[:n].cpu().numpy()
        assert np.abs(cp - oc.cam_T_world[:n]).max() <= 1e-4, (tag, "cam_T_world", np.abs(cp - oc.cam_T_world[:n]).max())
        assert np.abs(wb - oc.world_T_body[:n]).max() <= 1e-4, (tag, "world_T_body")
        rel = np.abs(od - oc.disps[:n]) / np.abs(oc.disps[:n])
        assert rel.max() <= 1e-4, (tag, "idepths", rel.max())
        for name, a, b in (("idepths_cov", fe.cam0_idepths_cov, oc.idepths_cov), ("depths_cov", fe.cam0_depths_cov, oc.depths_cov),
                           ("world_T_body_cov", fe.world_T_body_cov, oc.world_T_body_cov)):
            a = a[:n].cpu().numpy().astype(np.float64)
            r = np.abs(a - b[:n]) / np.maximum(np.abs(b[:n]), 1e-30)
            big = np.abs(b[:n]) > 1e-3 * np.abs(b[:n]).max()          # (entries far below the scale of the matrix: absolute)
            assert r[big].max() <= 2e-3, (tag, name, r[big].max())
            assert np.abs(a - b[:n])[~big].max(initial=0.0) <= 2e-3 * 1e-3 * np.abs(b[:n]).max() * 10, (tag, name, "small entries")

    def run(tag, n):
        sol = fe.update(itrs=2)
        assert sol["info"].item() == 0
        info = oc.update(op_host, itrs=2)
        step[0] += 1
        compare(f"{tag} (window [{info['kf0']}, {info['kf1']}), M = {info['M']})", n)
        return info

    ni, nj = _neigh(0, 6)
    fe.kf_idx = 6
    both(lambda: fe.add_factors(ni, nj), lambda: oc.add_edges(ni, nj))
    assert run("step 0", 7)["kf0"] == 0
    run("step 1", 7)
    both(lambda: fe.rm_factors(fe.graph.ii < 3, store=True), lambda: oc.rm_edges(oc.ii < 3, store=True))
    i2 = run("step 2", 7)
    assert i2["kf0"] == 3 and i2["M"] == ni.shape[0]                     # every inactive edge still inside kf0 - 3
    run("step 3", 7)
    # keyframe 7: seeded from 6 as TrackingSLAM._seed_next does (:621-631), edges to its 3 predecessors
    fe.kf_idx = 7
    for b in (fe.cam0_T_world, fe.world_T_body):
        b[7] = b[6]
    fe.cam0_idepths[7] = fe.cam0_idepths[6].mean()
    oc.cam_T_world[7], oc.world_T_body[7] = oc.cam_T_world[6], oc.world_T_body[6]
    oc.disps[7] = oc.disps[6].mean(dtype=np.float32)
    ei = np.array([7, 7, 7, 4, 5, 6], np.int64); ej = np.array([4, 5, 6, 7, 7, 7], np.int64)
    both(lambda: fe.add_factors(ei, ej), lambda: oc.add_edges(ei, ej))
    run("step 4", 8)
    run("step 5", 8)
    both(lambda: fe.rm_factors(fe.graph.ii < 5, store=True), lambda: oc.rm_edges(oc.ii < 5, store=True))
    i6 = run("step 6", 8)
    assert i6["kf0"] == 5 and i6["M"] < fe.graph.ii.shape[0] + fe.graph.ii_inactive.shape[0]   # some inactive edges dropped
    run("step 7", 8)
    # and the chain did what a tracker should: the estimate moved to the ground-truth scene
    err = np.abs(oc.cam_T_world[:8, :3] - gt_poses[:8, :3]).max()
    assert err < 0.05, err
