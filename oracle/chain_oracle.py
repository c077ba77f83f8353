"""CPU ORACLE of the COMPOSED tracking step (test infrastructure, NOT product code).

`ChainOracle.update()` restates RaftVisualFrontend.update() + ba() of /root/reference/slam/visual_frontends/
visual_frontend.py (:370-470, :1071-1232) on numpy arrays, composed from the per-kernel oracle functions

    reproject (projective_ops.py:98-145) -> target/weight/damping bookkeeping (:405-428) -> 2 x [ reduced_camera_matrix
    (droid_kernels.cu K1/K6/K9/K10) -> GTSAM dense solve + retraction (:1123-1158) -> solve_depth (:1161-1162) ]
    -> covariances (:1164-1230)

so that a test can run the PRODUCT's TrackingFrontend.update() and this chain on the same injected update-operator outputs
(delta, weight, damping) and compare poses / inverse depths / covariances after several steps, window shifts included.
The factor-graph lists are plain numpy here; their bookkeeping is pinned separately (tests/test_factor_graph_golden.py).
"""
import numpy as np

import oracle as O


class ChainOracle:
    def __init__(self, buffer, ht, wd, intr8, cam_T_body=None):
        self.buffer, self.ht, self.wd, self.HW = buffer, ht, wd, ht * wd
        ident = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
        self.cam_T_world = np.tile(ident, (buffer, 1))                       # :184
        self.world_T_body = np.tile(ident, (buffer, 1))
        self.cam_T_body = ident.copy() if cam_T_body is None else np.asarray(cam_T_body, np.float32)
        self.world_T_body_cov = np.tile(np.eye(6, dtype=np.float32) * 1e-4, (buffer, 1, 1))
        self.disps = np.ones((buffer, ht, wd), np.float32)                   # :187
        self.disps_sens = np.zeros((buffer, ht, wd), np.float32)
        self.idepths_cov = np.ones((buffer, ht, wd), np.float32)
        self.depths_cov = np.ones((buffer, ht, wd), np.float32)
        self.damping = 1e-6 * np.ones((buffer, ht, wd), np.float32)          # :222
        self.intr8 = np.asarray(intr8, np.float32)
        z = np.zeros(0, np.int64)
        self.ii, self.jj, self.ii_in, self.jj_in = z, z, z, z
        e = np.zeros((0, ht, wd, 2), np.float32)
        self.target, self.weight, self.target_in, self.weight_in = e, e, e, e
        self.prior_pose = None

    # -- the edge lists (add_factors :806-862 without the de-duplication / eviction, rm_factors :868-892) ---------------
    def add_edges(self, ii, jj):
        ii, jj = np.asarray(ii, np.int64), np.asarray(jj, np.int64)
        tgt = O.reproject(self.cam_T_world, self.disps, self.intr8, ii, jj)[0].astype(np.float32)
        self.ii, self.jj = np.concatenate([self.ii, ii]), np.concatenate([self.jj, jj])
        self.target = np.concatenate([self.target, tgt])
        self.weight = np.concatenate([self.weight, np.zeros_like(tgt)])

    def rm_edges(self, mask, store):
        mask = np.asarray(mask, bool)
        if store:
            self.ii_in, self.jj_in = np.concatenate([self.ii_in, self.ii[mask]]), np.concatenate([self.jj_in, self.jj[mask]])
            self.target_in = np.concatenate([self.target_in, self.target[mask]])
            self.weight_in = np.concatenate([self.weight_in, self.weight[mask]])
        self.ii, self.jj, self.target, self.weight = self.ii[~mask], self.jj[~mask], self.target[~mask], self.weight[~mask]

    # -- update() :370-470 -----------------------------------------------------------------------------------------------
    def update(self, op, itrs=2, compute_covariances=True):
        """op(coords1 [E,ht,wd,2], ii, jj) -> (delta [E,ht,wd,2], weight [E,ht,wd,2], damping [n_unique_ii,ht,wd])"""
        coords1 = O.reproject(self.cam_T_world, self.disps, self.intr8, self.ii, self.jj)[0].astype(np.float32)   # :378
        delta, weight, damping = op(coords1, self.ii, self.jj)
        kf0 = max(0, int(self.ii.min()))                                                     # :401
        self.target = (coords1 + delta.astype(np.float32)).astype(np.float32)              # :408
        self.weight = weight.astype(np.float32)
        self.damping[np.unique(self.ii)] = damping                                           # :411
        m = (self.ii_in >= kf0 - 3) & (self.jj_in >= kf0 - 3)                              # :420
        ii = np.concatenate([self.ii_in[m], self.ii])
        jj = np.concatenate([self.jj_in[m], self.jj])
        tgt = np.concatenate([self.target_in[m], self.target]).transpose(0, 3, 1, 2)       # :431
        wgt = np.concatenate([self.weight_in[m], self.weight]).transpose(0, 3, 1, 2)
        eta = (np.float32(0.2) * self.damping[np.unique(ii)] + np.float32(1e-7)).astype(np.float32)   # :428
        return self.ba(np.ascontiguousarray(tgt), np.ascontiguousarray(wgt), eta, ii, jj, kf0, itrs=itrs,
                       compute_covariances=compute_covariances)

    # -- ba() :1071-1232 -------------------------------------------------------------------------------------------------
    def ba(self, target, weight, eta, ii, jj, kf0, kf1=None, itrs=2, compute_covariances=True):
        if kf1 is None:
            kf1 = int(max(ii.max(), jj.max())) + 1                                           # :1078
        prior = self.prior_pose if (kf0 == 0 and self.prior_pose is not None) else None     # :1089-1095 (frame 0 in window)
        Hfull = E = Q = None
        for _ in range(itrs):
            H, v, Q, E, w, kx = O.reduced_camera_matrix(self.cam_T_world, self.disps, self.intr8, self.cam_T_body,
                                                        self.disps_sens, target, weight, eta, ii, jj, kf0, kf1)   # :1112
            assert kx.shape[0] == eta.shape[0], "damping rows (unique(ii), :428) != the kernel's K' rows: the reference would fail"
            delta, wTb, cTw, Hfull = O.ba_solve_retract(H, v, self.world_T_body, self.cam_T_body, kf0, kf1,
                                                        prior_pose=prior)                  # :1123-1158
            self.world_T_body[kf0:kf1] = wTb.astype(np.float32)
            self.cam_T_world[kf0:kf1] = cTw.astype(np.float32)
            self.disps = O.solve_depth(delta.astype(np.float32), self.disps, Q, E, w, ii, jj, kf0, kf1)   # :1161
            self.disps = np.maximum(self.disps, np.float32(0.001))                           # :1162
        if compute_covariances:
            sig, z, kx = O.ba_covariances(Hfull, E, Q, ii, jj, kf0, kf1, self.HW)            # :1164-1219
            self.world_T_body_cov[kf0:kf1] = sig.astype(np.float32)                          # :1222-1225
            z = z.reshape(-1, self.ht, self.wd).astype(np.float32)
            self.idepths_cov[kx] = z                                                         # :1227
            self.depths_cov[kx] = z / self.disps[kx] ** 4                                    # :1229
        return dict(kf0=kf0, kf1=kf1, M=int(ii.shape[0]))
