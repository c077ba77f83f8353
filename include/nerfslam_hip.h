/*
 * nerfslam_hip.h -- C ABI of libnerfslam_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the two hot paths of ToniRV/NeRF-SLAM:
 *   (1) the `droid_backends` operator table  (reference: src/droid.cpp:347-363)
 *   (2) the `pyngp` training-step surface    (reference: fusion/nerf_fusion.py:57-101,285-300)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - tensors are dense row-major ("contiguous", the only thing the reference checks:
 *     src/droid.cpp:129-130); shapes are given in the comments;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream). Nothing in this
 *     library synchronises the device or allocates memory: outputs and workspaces are
 *     caller-owned (the Python shim allocates them with torch);
 *   - return value: 0 on success, negative NS_E* code otherwise; ns_last_error() returns a
 *     thread-local message for the last failure;
 *   - int64 index tensors (`ii`, `jj`) keep the reference's dtype (torch.long).
 *
 * No torch / pybind types cross this boundary.
 */
#ifndef NERFSLAM_HIP_H
#define NERFSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS_OK 0
#define NS_EINVAL (-1)  /* bad argument (shape / dtype / null pointer)            */
#define NS_ELAUNCH (-2) /* hipGetLastError() after a launch reported a failure    */
#define NS_ENOSUP (-3)  /* configuration not supported by this build              */

#define NS_F16 1
#define NS_F32 2

#define NS_ACT_NONE 0
#define NS_ACT_RELU 1
#define NS_ACT_SIGMOID 2
#define NS_ACT_TANH 3

#define NS_CONV_FUSE_NONE 0
#define NS_CONV_FUSE_MUL_HI 1 /* out[co] *= e0[pixel][co - cout/2] for co >= cout/2       (r * h, gru.py:29-30)  */
#define NS_CONV_FUSE_GRU 2    /* out = e1 + e0 * (act(conv) - e1), e0 = z, e1 = h          (gru.py:31-33)        */

const char* ns_last_error(void);
int ns_version(void);          /* ABI version of this header: 1                          */
const char* ns_arch(void);     /* "gfx950"                                               */

/* ------------------------------------------------------------------------------------------
 * Correlation volumes
 * ---------------------------------------------------------------------------------------- */

/* corr_index_forward  (src/droid.cpp:280-288 -> src/correlation_kernels.cu:126-155, kernel :20-70)
 *   volume [B,h1,w1,h2,w2] dtype (NS_F16|NS_F32), coords [B,2,h1,w1] f32,
 *   corr   [B,2r+1,2r+1,h1,w1] dtype  (written completely; no zero-init needed).            */
int ns_corr_index_forward(const void* volume, const float* coords, void* corr, int dtype, int B, int h1,
                          int w1, int h2, int w2, int radius, void* stream);

/* corr_index_backward (src/droid.cpp:290-301 -> correlation_kernels.cu:157-185, kernel :73-124)
 *   volume_grad [B,h1,w1,h2,w2] f32 must be zero-filled by the caller; corr_grad [B,2r+1,2r+1,h1,w1]. */
int ns_corr_index_backward(const float* coords, const float* corr_grad, float* volume_grad, int B, int h1,
                           int w1, int h2, int w2, int radius, void* stream);

/* Fused CorrBlock.__call__ (networks/modules/corr.py:40-50): all `num_levels` (<=4) pyramid
 * levels in one launch, radius 3, f16.  pyr[l] is [E,h1,w1,h1>>l,w1>>l]; coords is the frontend's
 * native [E,h1,w1,2] layout when coords_interleaved=1, else [E,2,h1,w1]; it is divided by 2^l
 * inside the kernel (corr.py:47).  out [E, num_levels*49, h1, w1] f16 == torch.cat(out_pyramid,2). */
int ns_corr_lookup_pyramid(const void* const* pyr_host, int num_levels, const float* coords,
                           int coords_interleaved, void* out, int E, int h1, int w1, int tiled, void* stream);

/* Slot-addressed forms (this project's frontend, nerfslam/corr.py:CorrPool): the per-edge volumes live in a pool of
 * `capacity` slots and edge e uses volume index slot[e] (int32, device), so that edges enter / leave the factor graph
 * (visual_frontend.py:806-892: `torch.cat` / boolean-mask copies of ~61 MB per edge there) without moving a byte of
 * volume.  slot == NULL: identical to the plain entry points. */
int ns_corr_lookup_pyramid_slots(const void* const* pyr_host, int num_levels, const float* coords, int coords_interleaved,
                                 void* out, int E, int h1, int w1, int tiled, const int* slot, int capacity, void* stream);

/* Lookup fused with the update operator's correlation encoder (round 4): CorrBlock.__call__ (networks/modules/corr.py:40-50,
 * four levels, radius 3) followed by UpdateModule.corr_encoder[0:2] = Conv2d(196,128,1) + ReLU (networks/droid_net.py:83-87,
 * 133) in one launch; the [E,196,h1,w1] tensor is never written.  wfrag: the 196 x 128 weight as MFMA fragments, f16
 * [4 (32-channel tile)][13 (16-channel chunk of the 196 -> 208 inputs)][64 lanes][8]: element q of lane l of fragment (nt, c) =
 * W[32 nt + (l & 31)][16 c + 8 (l >> 5) + q], zero beyond input 195 (nerfslam/update_op.py packs it); bias [128] f32;
 * out [E,h1,w1,128] f16 channels-last = relu(W x + b).  Volumes, coords, slot, capacity as ns_corr_lookup_pyramid_slots. */
int ns_corr_lookup_encode_slots(const void* const* pyr_host, const float* coords, int coords_interleaved, const void* wfrag,
                                const float* bias, void* out, int E, int h1, int w1, int tiled, const int* slot, int capacity,
                                void* stream);
int ns_corr_volume_pyramid_slots(const void* fmap1, const void* fmap2, const int64_t* ii, const int64_t* jj,
                                 void* const* pyr_host, int num_levels, int E, int C, int ht, int wd, int tiled,
                                 const int* slot, void* stream);

/* CorrBlock.__init__ pyramid (corr.py:23-38): one 2x2 average-pool step over the last two dims,
 * f16 in/out, f32 accumulate, one rounding.  in [nslices,h,w] -> out [nslices,h/2,w/2].       */
int ns_corr_pool2x2(const void* in, void* out, long nslices, int h, int w, void* stream);

/* CorrBlock.corr + pyramid fused (corr.py:63-72 + :35-38), f16 MFMA, every output byte written once.
 *   fmap1 [n1,HW,C], fmap2 [n2,HW,C] f16 CHANNELS-LAST and already divided by 4 (corr.py:67-68);
 *   ii,jj [E] i64 frame ids into fmap1/fmap2 (both NULL: edge e uses row e of each);
 *   writes pyr_host[l] = [E,ht,wd,ht>>l,wd>>l] f16 for l < num_levels.  C must be 128.
 *   tiled = 1: levels 0 and 1 are written as 8x8-tiled slices, pyr_host[l] = [E, ht*wd, ceil(h_l/8), ceil(w_l/8), 8, 8]
 *   (one 128-byte line per tile; border tiles are padded, padding content unspecified) -- a private layout that
 *   ns_corr_lookup_pyramid reads with tiled = 1: the 8x8 tap window touches <= 4 lines instead of 8-9.          */
int ns_corr_volume_pyramid(const void* fmap1, const void* fmap2, const int64_t* ii, const int64_t* jj,
                           void* const* pyr_host, int num_levels, int E, int C, int ht, int wd, int tiled, void* stream);

/* altcorr_forward (src/droid.cpp:303-313 -> src/altcorr_kernel.cu:290-319, kernel :28-149)
 *   fmap1 [B,H1,W1,C] f32, fmap2 [B,H2,W2,C] f32 (channels-last), coords [B,N,H1,W1,2] f32,
 *   corr [B,N,(2r+1)^2,H1,W1] f32 (written completely).                                      */
int ns_altcorr_forward(const float* fmap1, const float* fmap2, const float* coords, float* corr, int B,
                       int H1, int W1, int H2, int W2, int C, int N, int radius, void* stream);

/* altcorr_backward (src/droid.cpp:315-327 -> altcorr_kernel.cu:150-355): fmap1_grad [B,H1,W1,C] (fully written) and
 * fmap2_grad [B,H2,W2,C] (ACCUMULATED into: zero it first, as the reference's torch::zeros does) for corr_grad
 * [B,N,49,H1,W1]; the reference's third output, coords_grad, is all zeros (:340, never written).  Radius 3 only
 * (NS_ENOSUP otherwise), C <= 512.  Dead in the reference's live path; float atomics on fmap2_grad like the reference. */
int ns_altcorr_backward(const float* fmap1, const float* fmap2, const float* coords, const float* corr_grad,
                        float* fmap1_grad, float* fmap2_grad, int B, int H1, int W1, int H2, int W2, int C, int N,
                        int radius, void* stream);

/* Fused AltCorrBlock.__call__ (networks/modules/corr.py:107-126): every pyramid level in one launch.
 *   fmaps_host[l] -> [nframes, H1>>l, W1>>l, C] f32 channels-last (features already divided by 4 and
 *   average-pooled as corr.py:98-105 does), ii,jj [E] i64 frame ids, coords [E,H1,W1,2] f32 (divided
 *   by 2^l in the kernel); out [E, num_levels*49, H1, W1] f32 == cat over levels of altcorr_forward. */
int ns_altcorr_pyramid(const float* const* fmaps_host, int num_levels, const int64_t* ii, const int64_t* jj,
                       const float* coords, float* out, int E, int H1, int W1, int C, void* stream);

/* The same for a HALF-precision pyramid -- what the reference's AltCorrBlock holds when it is given the frontend's half
 * features (corr.py:96-105 keeps `/ 4` and avg_pool2d in the input dtype; visual_frontend.py:209) -- on the matrix cores:
 * fmaps_host[l] -> [nframes, H1>>l, W1>>l, 128] f16 channels-last, 16-byte aligned.  Products of f16 values are exact in
 * f32: the result equals ns_altcorr_pyramid on the same (f16-representable) values up to summation order.  C = 128 only. */
int ns_altcorr_pyramid_f16(const void* const* fmaps_host, int num_levels, const int64_t* ii, const int64_t* jj,
                           const float* coords, float* out, int E, int H1, int W1, int C, void* stream);

/* On-the-fly correlation fused with the correlation encoder (round 4; config #5's counterpart of ns_corr_lookup_encode_slots):
 * AltCorrBlock.__call__ over four levels (networks/modules/corr.py:107-131) followed by Conv2d(196,128,1) + ReLU
 * (networks/droid_net.py:83-87,133) in one launch -- the 196 f32 planes per edge are never written.  fmaps: the half pyramid of
 * ns_altcorr_pyramid_f16 (four levels required); wfrag / bias as ns_corr_lookup_encode_slots; out [E,H1,W1,128] f16 channels-last.
 * The 196 correlation values are rounded to half before the convolution (what autocast hands it in the reference). */
int ns_altcorr_pyramid_encode_f16(const void* const* fmaps_host, const int64_t* ii, const int64_t* jj, const float* coords,
                                  const void* wfrag, const float* bias, void* out, int E, int H1, int W1, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Geometry
 * ---------------------------------------------------------------------------------------- */

/* motion features of the update operator (visual_frontend.py:379-386): out [E,4,ht,wd] =
 * clamp(cat(coords1 - pixel grid, target - coords1), -64, 64), channel-first, from interleaved [E,ht,wd,2] inputs. */
int ns_motion_features(const float* coords1, const float* target, float* out, int E, int ht, int wd, void* stream);

/* cvx_upsample (utils/flow_viz.py:166-183; visual_frontend.py:445-446): convex 8x upsampling of n maps [n,ht,wd] f32 with
 * the update operator's mask [n, 9*8*8, ht, wd] (f16 or f32 logits; plane index k*64 + sy*8 + sx, k = 3x3 neighbour in
 * row-major order): out [n, 8ht, 8wd] = sum_k softmax_k(mask)^pow * data(neighbour k); neighbours outside the image are
 * excluded from the softmax.                                                                                    */
int ns_cvx_upsample(const float* data, const void* mask, int mask_dtype, float* out, int n, int ht, int wd, float pow_,
                    void* stream);

/* The frontend's use of it (visual_frontend.py:445-446: inverse depths AND depth covariances of the updated keyframes kx,
 * same mask) as ONE launch on the keyframe buffers: data_a/data_b [buffer,ht,wd], out_a/out_b [buffer,8ht,8wd], kx [n] i64
 * (distinct), mask [n,576,ht,wd]; frame kx[f] of each output is overwritten.  data_b/out_b may both be NULL.  The mask
 * (98 % of the bytes) is read and soft-maxed once for both maps, and no gather / scatter copies are made.          */
int ns_cvx_upsample_keyframes(const float* data_a, const float* data_b, const int64_t* kx, const void* mask, int mask_dtype,
                              float* out_a, float* out_b, int n, int ht, int wd, float pow_, void* stream);
/* the same with a channels-last f16 mask [n,ht,wd,576] (what the update operator's 1x1 convolution writes) */
int ns_cvx_upsample_keyframes_nhwc(const float* data_a, const float* data_b, const int64_t* kx, const void* mask, float* out_a,
                                   float* out_b, int n, int ht, int wd, float pow_, void* stream);

/* frame_distance (src/droid.cpp:230-246 -> src/droid_kernels.cu:1572-1594, kernel :630-769)
 *   poses [n,7] (t,q xyzw), disps [n,ht,wd], intrinsics [4], ii,jj [num] i64 -> dist [num].   */
int ns_frame_distance(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                      const int64_t* jj, float* dist, int num, int ht, int wd, float beta, void* stream);

/* Reprojection of the frontend's update() (visual_frontend.py:909-918 -> networks/geom/projective_ops.py:98-145,
 * no Jacobians): coords [num,ht,wd,2] f32 (the layout ns_corr_lookup_pyramid reads), valid [num,ht,wd] or NULL. */
int ns_reproject(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
                 const int64_t* jj, float* coords, float* valid, int num, int ht, int wd, void* stream);

/* projmap (src/droid.cpp:249-264 -> droid_kernels.cu:1597-1622, kernel :539-628)
 *   coords [num,ht,wd,3] must be zero-filled (channel 2 is never written), valid [num,ht,wd,1]. */
int ns_projmap(const float* poses, const float* disps, const float* intrinsics, const int64_t* ii,
               const int64_t* jj, float* coords, float* valid, int num, int ht, int wd, void* stream);

/* iproj (src/droid.cpp:267-276 -> droid_kernels.cu:1652-1675, kernel :896-967)  points [nm,ht,wd,3]. */
int ns_iproj(const float* poses, const float* disps, const float* intrinsics, float* points, int nm, int ht,
             int wd, void* stream);

/* depth_filter (src/droid.cpp:330-344 -> droid_kernels.cu:1625-1649, kernel :773-892)
 *   counter [num,ht,wd] must be zero-filled.                                                 */
int ns_depth_filter(const float* poses, const float* disps, const float* intrinsics, const int64_t* inds,
                    const float* thresh, float* counter, int num, int nframes, int ht, int wd, void* stream);

/* solve_poses (src/droid.cpp:220-228 -> droid_kernels.cu:1837-1847, kernel :1015-1048):
 *   poses[k] <- Exp(dx[k-kf0]) * poses[k], dx = [tau, phi].                                   */
int ns_pose_retr(float* poses, const float* dx, int kf0, int kf1, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense bundle adjustment
 *
 * Graph plan.  The reference rebuilds, on the host and on every call, the expanded edge list
 * (self loops kf0..kf1-1 prepended), its unique source ids and the Schur pair list
 * (droid_kernels.cu:1702-1710, 1065-1103, 1359-1402).  Here the caller builds these small index
 * arrays once per graph change with ns_ba_plan_* (pure host code, no device work) and uploads
 * them; the kernels only read them.
 * ---------------------------------------------------------------------------------------- */

typedef struct ns_ba_plan {
  int M;        /* edges given to BA                                                          */
  int P;        /* kf1 - kf0 window poses                                                     */
  int K;        /* K' = #unique(cat(arange(kf0,kf1), ii))                                     */
  int kf0, kf1;
  int n_pairs;  /* Schur pairs (row n, row m, depth slot), n,m in window, same depth slot     */
  int n_rows;   /* P + M rows of E                                                            */
  int n_jobs;   /* Gram jobs of the Schur kernel (index part [10])                            */
  int max_src;  /* most edges leaving one source frame                                        */
} ns_ba_plan;

#define NS_BA_PLAN_PARTS 13   /* parts of the index block = entries of offsets_host           */
#define NS_GRAM_JOB_INTS 6    /* int32 per Gram job header                                    */
#define NS_GRAM_BLOCK 8       /* tiles of 16 values per block of a slot's Gram matrix         */

/* Sizes of the int32 index block the plan needs (in int32 elements).                          */
size_t ns_ba_plan_index_count(const int64_t* ii_host, const int64_t* jj_host, int M, int kf0, int kf1);

/* Fills `plan` and `index_host` (int32[ns_ba_plan_index_count]).  Layout of the index block
 * (all int32, offsets returned in `offsets_host[NS_BA_PLAN_PARTS]`):
 *   [0] kx[K]            sorted unique source-frame ids           (droid_kernels.cu:1706-1710)
 *   [1] kk[P+M]          depth slot of every E row
 *   [2] row_pose[P+M]    jj_expanded - kf0  (window pose of the row, may be <0 or >=P)
 *   [3] src_ptr[K+1]     CSR over edges grouped by depth slot      (accum_cuda :1065-1103)
 *   [4] src_edge[M]      edge ids in CSR order (stable: ascending edge id inside a slot)
 *   [5] pairs[3*n_pairs] (row n, row m, slot) in the reference's enumeration order (:1384-1399)
 *   [6] slot_rows_ptr[K+1], [7] slot_rows[...]  E rows (self loop + edges) per slot, any pose
 *   [8] win_rows_ptr[K+1],  [9] win_rows[...]   the rows of [7] whose pose lies in the window (what the Schur complement sums
 *                                                over, droid_kernels.cu:1375), same order
 *   [10] gram_jobs[NS_GRAM_JOB_INTS*n_jobs] (slot, first A tile, A tiles, first B tile, B tiles, 0): the slot's window rows x 6
 *        values, in tiles of 16, blocks of NS_GRAM_BLOCK tiles; first A tile == first B tile marks a diagonal block (upper triangle)
 *   [11] job_plane[256*n_jobs], [12] job_hrow[256*n_jobs]: per job the plane of E (row * 6 + component) and the row of H
 *        (6 * pose + component) of its 128 A and 128 B values, -1 for padding                                        */
int ns_ba_plan_build(const int64_t* ii_host, const int64_t* jj_host, int M, int kf0, int kf1,
                     ns_ba_plan* plan, int32_t* index_host, size_t* offsets_host);

/* reduced_camera_matrix (src/droid.cpp:167-196 -> droid_kernels.cu:1681-1768): K1 (:192-536),
 * accum (:971-991), EEt6x6 (:1118-1173), Ev6x1 (:1176-1210), SparseBlock (:1240-1316).
 *   poses [*,7], disps [*,ht,wd], intrinsics[4], extrinsics[7], disps_sens [*,ht,wd],
 *   targets, weights [M,2,ht,wd], eta [K,ht,wd], ii,jj [M] i64 (device), index = device copy of
 *   the plan's index block.
 * outputs: H [6P,6P] f32, v [6P] f32, Q [K,HW], E [P+M,6,HW], w [K,HW]  (all fully written).
 * workspace: ns_ba_workspace_bytes(plan, ht*wd) bytes, 256-byte aligned.  ws_zeroed = 0: contents
 * undefined (the call clears what it needs); 1: the caller promises the workspace was either
 * zero-filled or last used by a successful call of this function with the same plan (which leaves
 * its accumulators zeroed again) -- saves a memset node per linearisation.                       */
size_t ns_ba_workspace_bytes(const ns_ba_plan* plan, int HW);
int ns_reduced_camera_matrix(const float* poses, const float* disps, const float* intrinsics,
                             const float* extrinsics, const float* disps_sens, const float* targets,
                             const float* weights, const float* eta, const int64_t* ii, const int64_t* jj,
                             const ns_ba_plan* plan, const int32_t* index, const size_t* offsets_host, int ht,
                             int wd, float* H, float* v, float* Q, float* E, float* w, void* workspace,
                             int ws_zeroed, void* stream);

/* ns_reduced_camera_matrix `reps` times with HIP events between its five launches (edge table, fused lineariser, Gram kernel,
 * reduce / assembly, finalisation): us_out[5] = their mean durations in microseconds.  Synchronises the stream.  (bench.py's
 * BA roofline entries.)                                                                                              */
int ns_reduced_camera_matrix_timed(const float* poses, const float* disps, const float* intrinsics,
                                   const float* extrinsics, const float* disps_sens, const float* targets,
                                   const float* weights, const float* eta, const int64_t* ii, const int64_t* jj,
                                   const ns_ba_plan* plan, const int32_t* index, const size_t* offsets_host, int ht, int wd,
                                   float* H, float* v, float* Q, float* E, float* w, void* workspace, int ws_zeroed, void* stream,
                                   int reps, float* us_out);

/* solve_depth (src/droid.cpp:198-218 -> droid_kernels.cu:1772-1825): EvT6x1 (:1213-1238),
 * accum, dz = Q*(w - .), disp_retr (:1050-1063); disps updated in place, then optionally
 * clamped to >= clamp_min (visual_frontend.py:1162; pass a negative value to skip).           */
int ns_solve_depth(const float* dx, float* disps, const float* Q, const float* E, const float* w,
                   const ns_ba_plan* plan, const int32_t* index, const size_t* offsets_host, int ht, int wd,
                   float clamp_min, void* stream);

/* K1 alone -- projective_transform_kernel with the reference kernel's own per-edge outputs
 * (droid_kernels.cu:192-536): Hs [4,M,6,6], vs [2,M,6], Eiz/Ejz [M,6,HW], Cii/bz [M,HW].
 * etab_ws: scratch of M*80 floats.                                                            */
int ns_projective_transform(const float* targets, const float* weights, const float* poses, const float* disps,
                            const float* intrinsics, const float* extrinsics, const int64_t* ii,
                            const int64_t* jj, int M, int ht, int wd, float* Hs, float* vs, float* Eiz,
                            float* Ejz, float* Cii, float* bz, float* etab_ws, void* stream);

/* Device-resident replacement of the GTSAM round trip in ba() (visual_frontend.py:1123-1158),
 * one workgroup, f64 in LDS, 6P <= 192 (6P <= 108 with L^-1 / sigma_g); NS_ENOSUP above that: the host then
 * calls ns_ba_solve_large:
 *   (triu(H) mirrored [+ ep + lm*diag] [+ prior]) delta = v      (dense blocked Cholesky)
 *   mode 0:  world_T_body[kf0+i] <- world_T_body[kf0+i] * Exp(delta_i)   (delta = [omega, v]),
 *            cam_T_world[kf0+i]  <- cam_T_body * world_T_body[kf0+i]^-1
 *   mode 1:  solve only
 *   dx [P,6] f32 = delta.  prior_pose (7 floats, device) may be NULL; with a prior, 1/sigma^2 is
 *   added to the first pose block and -Log(prior^-1 * x0)/sigma^2 to its rhs.
 *   Optional outputs (NULL to skip): Hfull_out [6P,6P] f64 (the system actually solved),
 *   Linv_out [6P,6P] f32 (inverse Cholesky factor; needs Linv_ws [6P,6P] f64 scratch),
 *   sigma_g_out [P,6,6] f32 (diagonal blocks of the inverse, visual_frontend.py:1178-1189).
 *   info (int32, device): 0 ok, k>0 = pivot k not positive (dx is zero, nothing retracted).   */
int ns_ba_solve(const float* H, const float* v, float* world_T_body, float* cam_T_world,
                const float* cam_T_body, const float* prior_pose, float prior_sigma, float ep, float lm, int kf0,
                int kf1, int mode, float* dx, double* Hfull_out, float* Linv_out, double* Linv_ws,
                float* sigma_g_out, int32_t* info, void* stream);

/* ns_ba_solve for systems beyond one workgroup's LDS -- the global BA over the whole buffer (backend(),
 * visual_frontend.py:1255-1295: 6P = 1536 for 256 keyframes) and windows with covariances above 18 poses.  Same
 * arguments, semantics and outputs; blocked right-looking Cholesky in f64 through a caller-provided workspace of
 * ns_ba_solve_large_workspace_bytes(6P, want L^-1 / sigma_g) bytes (device memory); 6P <= ~19000.                  */
size_t ns_ba_solve_large_workspace_bytes(int n6, int want_inv);
int ns_ba_solve_large(const float* H, const float* v, float* world_T_body, float* cam_T_world,
                      const float* cam_T_body, const float* prior_pose, float prior_sigma, float ep, float lm, int kf0,
                      int kf1, int mode, float* dx, double* Hfull_out, float* Linv_out, float* sigma_g_out,
                      int32_t* info, void* workspace, size_t workspace_bytes, void* stream);

/* The retraction of ns_ba_solve mode 0 on its own (dx given).                                 */
int ns_ba_retract(const float* dx, float* world_T_body, float* cam_T_world, const float* cam_T_body, int kf0,
                  int kf1, void* stream);

/* Depth covariances of ba() (visual_frontend.py:1191-1219):
 *   z_cov[k,px] = Q + sum_j ( sum_{rows n of slot k, pose a in window} Q * E_n[:,px] . Linv[6a+:, j] )^2 */
int ns_ba_depth_cov(const float* Linv, const float* Q, const float* E, const ns_ba_plan* plan,
                    const int32_t* index, const size_t* offsets_host, int HW, float* z_cov, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mapping path: instant-ngp style NeRF training step.
 * Reference boundary: the `pyngp` calls of fusion/nerf_fusion.py:57-101, 285-303, 388-424 (the
 * arithmetic is an un-vendored fork of NVIDIA instant-ngp -> parity unpinned).
 * Hash grid: n_levels <= 16, 2 features/level, f16 table; positions in the unit cube.
 * ---------------------------------------------------------------------------------------- */

/* host-only: per-level scale / resolution / table offset (entries); offset_host has n_levels+1 slots */
int ns_ngp_grid_layout(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                       float* scale_host, int* res_host, uint32_t* offset_host);

/* multiresolution hash encoding: positions [N,3] f32, params f16 [entries*2] -> out f16,
 * [N, n_levels*2] (unit_major = 0) or [n_levels*2, N] (unit_major = 1: every store of a wave is one
 * contiguous 128-byte run, and the layout the MLP kernels consume)                                 */
int ns_ngp_encode_forward(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                          const float* positions, const void* params, void* out, int unit_major, long N, void* stream);

/* grad_params f32 [entries*2] += trilinear weights * dLdout f16 ([N, n_levels*2], or [n_levels*2, N] with
 * unit_major = 1).  No global atomics on table entries (csrc/ngp.hip: on this 8-XCD part they execute at the memory side):
 *   workspace == NULL   owner-computes kernel on every level: a workgroup owns a 16384-entry table slice in LDS and scans
 *                       all samples of the level for corners that fall into it;
 *   workspace != NULL   (workspace_bytes >= ns_ngp_encode_backward_workspace_bytes(..., max_samples >= N), zero-filled once
 *                       by the caller; packed fixed-point mode only; a SMALLER buffer is never written: the call then
 *                       takes the owner-computes kernels) the hashed levels are BINNED instead -- count / scatter /
 *                       accumulate passes over 64-bit (slice index, 2 x 25-bit Q(S) gradient) records, 8 B written + 8 B read
 *                       per (sample, corner); the dense coarse levels keep the owner-computes kernel.
 * All three paths (NS_ENC_BWD_ATOMIC=1 selects round 1's atomic scatter) produce the same integer sums bit for bit.
 * fixed_scale: 0 -> grad_params holds f32 pairs.  S > 0 -> PACKED fixed point: one 64-bit word per table
 * entry, word = round(g0*S) + (round(g1*S) << 32) (two signed Q(S) 32-bit fields): an order-independent
 * (bit-reproducible) sum; contributions saturate at |g| >= 2^24 / S in the binned path.  ns_ngp_adam
 * decodes the same format when given the same fixed_scale.                                             */
long ns_ngp_encode_backward_workspace_bytes(int n_levels, int n_features, int log2_hashmap, int base_res,
                                            float per_level_scale, long max_samples);
int ns_ngp_encode_backward(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                           const float* positions, const void* dLdout, int unit_major, float* grad_params,
                           float* workspace, size_t workspace_bytes, float fixed_scale, long N, void* stream);

/* density MLP 32->64->16 + colour MLP (16 + SH16)->64->64->16, f16 weights packed row-major
 * [W1 64x32 | W2 16x64 | W3 64x32 | W4 64x64 | W5 16x64]; featT [32,N] f16 UNIT-MAJOR (what
 * ns_ngp_encode_forward writes with unit_major = 1); out [N,4] f16 = (r,g,b raw, log-density).
 * Training: pass the four unit-major activation buffers (h1T [64,N], cinT [32,N], h3T, h4T [64,N]);
 * inference: all NULL.  N must be even.  (MFMA register chain, csrc/ngp_mlp.hip.)                 */
int ns_ngp_mlp_forward(const void* weights, const void* featT, const float* dirs, void* out, void* h1T, void* cinT,
                       void* h3T, void* h4T, long N, void* stream);

/* backward of the MLP pair: weights as in the forward; dLdout [N,4] f16; writes dLdfeatT [32,N] f16
 * (unit-major, read by ns_ngp_encode_backward with unit_major = 1) and ADDS the f32 weight gradients
 * (MFMA split-K GEMMs) into grad_weights[10240].  d5T [16,N], d4T/d3T/d1T [64,N], ddT [16,N] and
 * partial_ws [ksplit*10240] f32 are scratch.  N must be a multiple of 8.                          */
int ns_ngp_mlp_backward(const void* weights, const void* dLdout, const void* featT, const void* h1T,
                        const void* cinT, const void* h3T, const void* h4T, void* dLdfeatT, void* d5T, void* d4T,
                        void* d3T, void* ddT, void* d1T, float* partial_ws, int ksplit, float* grad_weights, long N,
                        void* stream);

/* Adam on f32 master parameters with an f16 working copy; zeroes `grad` behind itself. step >= 1.
 * fixed_scale > 0: `grad` is in the packed fixed-point format of ns_ngp_encode_backward.
 * Layout of the state (this entry point and ns_ngp_encode_backward_fused*): master, m1, m2 are either three dense f32 [n] arrays,
 * or -- told from the pointers, m1 == master + 2 floats and m2 == master + 4 floats -- the fields of ONE 32-byte record per table
 * entry (two parameters): record e at master + 8 e floats = [master.xy | m1.xy | m2.xy | 8 bytes unused]; n even.  A sparsely
 * touched entry then costs one 128-byte line instead of three; ns_ngp_adam leaves a record whose gradient is zero (and l2 = 0)
 * unread and its working copy as it is (every writer of the master writes the working copy too).     */
int ns_ngp_adam(float* master, void* half_params, float* grad, float* m1, float* m2, long n, int step, float lr,
                float beta1, float beta2, float eps, float l2, float grad_scale, float fixed_scale, void* stream);

/* training rays: ray r picks image / column / row = pcg(seed + 3r + {0,1,2}) mod {n_images, W, H}
 * (pcg = the 32-bit PCG output hash, csrc/ngp.hip:ns_pcg), direction = normalised c2w[:, :3] ((u + .5 - cx)/fx,
 * (v + .5 - cy)/fy, 1), origin = c2w[:, 3], t_range = slab test against [box_lo, box_hi]^3 clamped to `near`, and the
 * pixel's supervision (images [n,H,W,4] linear rgba, depths / depth_covs [n,H,W]; covariance floored at 1e-6). */
int ns_ngp_sample_rays(const float* images, const float* depths, const float* depth_covs, const float* c2w,
                       int n_images, int H, int W, float fx, float fy, float cx, float cy, float box_lo, float box_hi,
                       float near, unsigned seed, int R, float* rays_o, float* rays_d, float* t_range, float* gt_rgb,
                       float* gt_depth, float* gt_depth_cov, int* ray_img /* optional [R] */, void* stream);

/* camera-pose refinement (`optimize_extrinsics`, nerf_fusion.py:99,123; arithmetic [EXTERNAL]):
 * (1) dLdpos [N,3] = gradient of the loss w.r.t. the unit-cube sample positions through the hash encoding
 *     (dLdoutT [n_levels*2, N] f16 unit-major = what ns_ngp_mlp_backward wrote);
 * (2) per ray g_o = sum dL/dp, g_d = sum t dL/dp (dL/dp = dLdpos * pos_inv), added into cam_grad[ray_img][6] =
 *     (sum g_o, sum d x g_d): gradient w.r.t. the camera translation and a left rotation perturbation exp(w) R;
 * (3) Adam on the 6 dof of every image that received gradient, c2w [n,3,4] <- [exp(dw) R | t + dt]; clears cam_grad. */
int ns_ngp_encode_backward_input(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                 const float* positions, const void* params, const void* dLdoutT, float* dLdpos, long N,
                                 void* stream);
int ns_ngp_camera_gradient(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                           const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R, void* stream);
int ns_ngp_camera_step(float* c2w, float* cam_grad, float* m1, float* m2, int n_images, int step, float lr_pos,
                       float lr_rot, float beta1, float beta2, float eps, float grad_scale, void* stream);

/* occupancy-grid ray marching: bits = ncasc cascades of G^3 bits; rays_o/rays_d [R,3] (unit dirs),
 * t_range [R,2].  counter[3] (zeroed by the caller) receives (#samples requested by all rays,
 * #rays that received samples, end of the last reserved range = number of samples to process: a
 * ray whose range would cross max_samples is refused: ray_n = -1, it contributes neither samples
 * nor loss; the accepted ranges tile [0, counter[2]) without holes);
 * ray_start/ray_n [R]; pos/dirs [max_samples,3]; dt/tmid [max_samples].  Positions are written as
 * (p - pos_lo) * pos_inv (pass 0, 1 for scene coordinates; the trainer asks for unit-cube coordinates). */
int ns_ngp_march(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d,
                 const float* t_range, int R, float cone, float min_step, float max_step, float pos_lo,
                 float pos_inv, int max_per_ray, long max_samples, int* counter, int* ray_start, int* ray_n, float* pos, float* dirs, float* dt,
                 float* tmid, void* stream);

/* volume rendering; with dLdout != NULL also the loss (rgb L2 + depth_lambda * (d - gt)^2 / cov,
 * summed over rays into *loss) and its gradient w.r.t. net_out [S,4] (scaled by loss_scale / R). */
int ns_ngp_composite(const void* net_out, const float* dt, const float* tmid, const int* ray_start,
                     const int* ray_n, int R, const float* gt_rgb, const float* gt_depth, const float* gt_depth_cov,
                     float depth_lambda, float loss_scale, float* out_rgb, float* out_depth, float* loss,
                     void* dLdout, void* stream);

/* ------------------------------------------------------------------------------------------
 * Update operator of the tracker: channels-last f16 convolution on the MFMA units
 * ---------------------------------------------------------------------------------------- */

/* Replaces the nn.Conv2d (3x3 pad 1 / 1x1, stride 1) + bias + activation (+ the torch.cat in front of it) calls of
 * networks/droid_net.py:78-150 (UpdateModule, GraphAgg) and networks/modules/gru.py:5-34 (ConvGRU):
 *   out[n,y,x, out_offset + co] = act( bias[n*bias_nstride + co] + sum_{dy,dx,ci} w[co][ci][dy][dx] * in[n, y+dy, x+dx, ci] )
 * in  = the channel concatenation of nsrc (1..4) channels-last f16 tensors src_host[s] = [N,H,W,src_channels_host[s]]
 *       (device pointers in a HOST array; each channel count a multiple of 16), zero padding outside the image;
 *       src_strides_host[s] = pixel stride in elements when source s is a channel slice of a wider tensor (NULL: dense;
 *       stride a multiple of 8 and the slice's first element 16-byte aligned);
 * w   = weights packed by fragment: f16 [CI/16][ksize^2][COP/32][2 (h)][32 (i)][8 (e)] holding
 *       w[co = 32 ct + i][ci = 16 c + 8 h + e][tap], COP = ns_conv_packed_cout(cout), zero for co >= cout;
 * bias = f32 or NULL; bias_nstride = 0 for one bias vector, cout-or-more for a bias per image (the ConvGRU's global-context
 *       terms convz_glo(glo) etc. are exactly that);  act = NS_ACT_*;
 * out = channels-last f16 [N,H,W,out_stride], the result occupies channels [out_offset, out_offset + cout) (8-byte stores
 *       when stride and offset are multiples of 4 channels, scalar ones otherwise).                                 */
int ns_conv_packed_cout(int cout);
int ns_conv_nhwc_f16(const void* const* src_host, const int* src_channels_host, const int* src_strides_host, int nsrc,
                     int N, int H, int W,
                     const void* wpacked, int ksize, int cout, const float* bias, long bias_nstride, int act, void* out,
                     int out_stride, int out_offset, void* stream);

/* The same convolution with one elementwise step of the ConvGRU (networks/modules/gru.py:28-33) folded into its epilogue
 * (fuse = NS_CONV_FUSE_*): e0 / e1 are channels-last f16 tensors with pixel strides e0_stride / e1_stride (elements,
 * multiples of 4; a channel slice of a wider tensor is fine).  MUL_HI: the convz|convr launch writes [z | r * h] (e0 = h);
 * GRU: the convq launch writes the new hidden state (e0 = z, e1 = h).  Needs cout a multiple of the cout tile and a
 * 4-channel-aligned output slice.                                                                                  */
int ns_conv_nhwc_f16_fused(const void* const* src_host, const int* src_channels_host, const int* src_strides_host, int nsrc,
                           int N, int H, int W, const void* wpacked, int ksize, int cout, const float* bias, long bias_nstride,
                           int act, void* out, int out_stride, int out_offset, int fuse, const void* e0, int e0_stride,
                           const void* e1, int e1_stride, void* stream);

/* im2col of the flow encoder's first layer (networks/droid_net.py:96, Conv2d(4,128,7,padding=3) on the motion features):
 * flow [E,4,ht,wd] f32 -> out [E,ht,wd,208] f16 with out[..., (ci*7+ky)*7+kx] = flow[e,ci,y+ky-3,x+kx-3] (0 outside the
 * image; channels 196..207 zero), so that the layer is a 1x1 ns_conv_nhwc_f16 with weight.reshape(128,196).       */
int ns_flow_im2col(const float* flow, void* out, int E, int ht, int wd, void* stream);

/* layout glue of the update operator: the lookup's [E,C,HW] f16 planes (networks/modules/corr.py:52-57 output, C = 196) ->
 * channels-last [E,HW,CP] with channels C..CP-1 zero (CP = 208: 13 chunks of 16 for ns_conv_nhwc_f16), one pass.   */
int ns_planes_to_nhwc_f16(const void* src, void* dst, int E, int C, int CP, int HW, void* stream);

/* GraphAgg's scatter_mean over the source keyframe of every edge (networks/droid_net.py:64-70):
 * out[k,p,c] = mean_{m in [starts[k],starts[k+1])} src[members[m], p, c]; src [E,HW,src_stride] f16 (a channel slice is
 * fine), starts [K+1] / members [E] i32 on the device, out [K,HW,channels] f16, f32 accumulation.                  */
int ns_group_mean_nhwc_f16(const void* src, int src_stride, const int* starts, const int* members, void* out, int K, int HW,
                           int channels, void* stream);
/* ConvGRU global context (networks/modules/gru.py:25-33): glo[e,c] = mean over the HW pixels of wg[e,p,c] * net[e,p,c] (wg =
 * sigmoid(w(net)), both dense channels-last f16 [E,HW,128], f32 products and sums), then out[e,o] = bias[o] + sum_c glo[e,c]
 * W[c,o] (W f32 [128,nout] row-major: the three conv*_glo 1x1 convolutions side by side, nout = 384; bias may be NULL).
 * partial: f32 scratch [E, ns_gru_glo_parts(HW), 128]; out f32 [E,nout].  Two launches.                               */
int ns_gru_glo_parts(int HW);
int ns_gru_glo_bias(const void* wg, const void* net, const float* W, const float* bias, float* partial, float* out, int E, int HW,
                    int nout, void* stream);

/* ---- feature / context encoders on the MFMA convolution (networks/modules/extractor.py:118-198 `BasicEncoder`; host
 * side nerfslam/encoder_op.py).  Activations are dense channels-last f16 [N,H,W,C]; Ho = (H-1)/2+1, Wo = (W-1)/2+1.
 *
 * ns_enc_stem_im2col: image [N,3,H,W] (u8 if img_is_u8, else f32 holding 0..255) -> out [N,Ho,Wo,160] f16 patches of the
 *   7x7 / stride 2 / pad 3 stem over the NORMALISED image (x/255 - mean[c]) / std[c] (zero padded after normalisation):
 *   out[..., (ky*7+kx)*3 + c], channels 147..159 zero -> the stem is a 1x1 ns_conv_nhwc_f16 with
 *   weight.permute(0,2,3,1).reshape(32,147).  mean / std: 3 floats each in HOST memory.
 * ns_enc_im2col_3x3s2: x [N,H,W,C] -> out [N,Ho,Wo,9C], out[..., (ky*3+kx)*C + c] = x[n, 2oy+ky-1, 2ox+kx-1, c] (0 outside):
 *   a stride-2 3x3 convolution (extractor.py:16-17,158-159) becomes a 1x1 over 9C channels, and the block's stride-2 1x1
 *   shortcut (extractor.py:46-47) a 1x1 over the centre-tap slice [4C, 5C) of the same buffer.
 * ns_enc_in_stats: partial[n][p][0|1][c] = sum | sum of squares of x [N,HW,C] over part p of the pixels,
 *   p < ns_enc_in_parts(HW); C in {32, 64, 128}.  partial: f32 [N, parts, 2, C]; the last workgroup of an image to arrive
 *   leaves the totals in row 0.  ticket: N uint32 arrival counters owned by the CALLER, zero before the first launch (the
 *   kernel leaves them zero); launches that may be in flight at the same time must not share a row (may be NULL when
 *   ns_enc_in_parts(HW) == 1).
 * ns_enc_in_apply: out = relu(x' + relu(y')) with y' = (y - mean) / sqrt(var + eps) from ystats (nullptr: y' = y),
 *   x' likewise from xstats without the relu (x nullptr: out = relu(y')); biased variance, statistics combined in f64:
 *   InstanceNorm2d + relu, and the tail of a residual block (extractor.py:50-60), in one pass.                        */
int ns_enc_stem_im2col(const void* img, int img_is_u8, void* out, int N, int H, int W, const float* mean, const float* std,
                       void* stream);
int ns_enc_im2col_3x3s2(const void* x, void* out, int N, int H, int W, int C, void* stream);
int ns_enc_in_parts(int HW);
int ns_enc_in_stats(const void* x, float* partial, unsigned int* ticket, int N, int HW, int C, void* stream);
int ns_enc_in_apply(const void* y, const float* ystats, const void* x, const float* xstats, void* out, int N, int HW, int C,
                    float eps, void* stream);

/* ---- graph-captured training step (nerfslam/ngp.py): the `_ctl` forms read the per-step scalars from a device
 * control block instead of by-value arguments, so that a whole optimiser step is a fixed launch sequence:
 *   ctl[0] optimiser steps completed, ctl[1] rays of the current batch, ctl[2] ray-sampling seed, ctl[3] training views,
 *   ctl[4], ctl[5] float bits of Adam's bias corrections 1 - beta^(ctl[0] + 1) (maintained by ns_ngp_step_advance; the caller
 *   initialises them).
 * With ctl != NULL the by-value R / n_images are CAPACITIES (grid sizes), `step` and `seed` are ignored
 * (seed = ctl[2] * 0x9E3779B1 + (ctl[0] + step_offset) * 0x85EBCA77, bias corrections from ctl[0] + 1).  ctl == NULL: identical to the
 * plain entry points.  ns_ngp_step_advance ends a step: last[0..3] = march counters + ray count of this step,
 * ctl[1] = clamp(R * fill * max_samples / requested, min_rays, max_rays) rounded down to 128, ctl[0] += 1,
 * counter[0..2] = 0.  [no reference counterpart: the fork's trainer reads its counters back on the host] */
int ns_ngp_sample_rays_ctl(const float* images, const float* depths, const float* depth_covs, const float* c2w, int n_images,
                           int H, int W, float fx, float fy, float cx, float cy, float box_lo, float box_hi, float near,
                           unsigned seed, int R, float* rays_o, float* rays_d, float* t_range, float* gt_rgb, float* gt_depth,
                           float* gt_depth_cov, int* ray_img, const int* ctl, int step_offset, void* stream);
int ns_ngp_march_ctl(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d, const float* t_range,
                     int R, float cone, float min_step, float max_step, float pos_lo, float pos_inv, int max_per_ray,
                     long max_samples, int* counter, int* ray_start, int* ray_n, float* pos, float* dirs, float* dt, float* tmid,
                     const int* ctl, void* stream);
/* The same with the sample ranges handed out in WORKGROUP ORDER (16 rays per workgroup) instead of arrival order: the batch's
 * sample arrays -- and, at a full budget, the set of refused rays -- are then the same on every run, which makes the optimiser
 * step bit-reproducible (tests/test_ngp_gpu.py::test_training_is_bit_reproducible).  order_ws: 2 + ceil(R / 16) 64-bit words
 * (a workgroup's ray block is a ticket it draws on entry -- word 1 -- so the look-back never waits for a workgroup that has not
 * started), zero before the first launch (the kernel leaves them zero); one per set of sample arrays that may be marched concurrently.
 * NULL = ns_ngp_march_ctl.                                                                                                  */
int ns_ngp_march_ordered(const uint8_t* bits, int G, int ncasc, const float* rays_o, const float* rays_d, const float* t_range,
                         int R, float cone, float min_step, float max_step, float pos_lo, float pos_inv, int max_per_ray,
                         long max_samples, int* counter, int* ray_start, int* ray_n, float* pos, float* dirs, float* dt,
                         float* tmid, const int* ctl, unsigned long long* order_ws, void* stream);
int ns_ngp_composite_ctl(const void* net_out, const float* dt, const float* tmid, const int* ray_start, const int* ray_n, int R,
                         const float* gt_rgb, const float* gt_depth, const float* gt_depth_cov, float depth_lambda,
                         float loss_scale, float* out_rgb, float* out_depth, float* loss, void* dLdout, const int* ctl,
                         void* stream);
/* ns_ngp_composite_ctl with the loss PER RAY: ray_loss[R] is written (0 for rays beyond the batch's device-side count), `loss`
 * is not touched.  One atomicAdd per ray on the one address of `loss` serialises in the L2 (36 of the kernel's 46 us with ~4000
 * rays); the trainer sums the per-ray values when the loss is read (testbed.loss, nerf_fusion.py:312).                       */
int ns_ngp_composite_rays(const void* net_out, const float* dt, const float* tmid, const int* ray_start, const int* ray_n, int R,
                          const float* gt_rgb, const float* gt_depth, const float* gt_depth_cov, float depth_lambda,
                          float loss_scale, float* out_rgb, float* out_depth, float* loss, float* ray_loss, void* dLdout,
                          const int* ctl, void* stream);
int ns_ngp_camera_gradient_ctl(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                               const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R, const int* ctl,
                               void* stream);
/* two-stage camera gradient: per-ray 6-vectors into ray_scratch [R,6] f32, then one workgroup sums them per image in LDS and
 * adds the sums to cam_grad [n_images,6] (no global atomics; n_images <= 4096).  ray_scratch == NULL: the atomic form. */
int ns_ngp_camera_gradient_2stage(const float* dLdpos, const float* tmid, const float* rays_d, const int* ray_start,
                                  const int* ray_n, const int* ray_img, float pos_inv, float* cam_grad, int R, const int* ctl,
                                  float* ray_scratch, int n_images, void* stream);
int ns_ngp_camera_step_ctl(float* c2w, float* cam_grad, float* m1, float* m2, int n_images, int step, float lr_pos, float lr_rot,
                           float beta1, float beta2, float eps, float grad_scale, const int* ctl, void* stream);
int ns_ngp_adam_ctl(float* master, void* half_params, float* grad, float* m1, float* m2, long n, int step, float lr, float beta1,
                    float beta2, float eps, float l2, float grad_scale, float fixed_scale, const int* ctl, void* stream);
/* ns_ngp_adam_ctl with the layout of the optimiser state TOLD (round 6; ADVICE r05: not inferred from the pointers): record_floats
 * = 2 (three dense arrays), 8 (round 5's 32-byte record per table entry: [master.xy | m1.xy | m2.xy | unused], base 16-byte
 * aligned) or 6 (24-byte record, base 8-byte aligned); for 6 / 8, m1 == master + 2 and m2 == master + 4 floats is REQUIRED
 * (checked).  The entry points without `_rec` pass 0 = "tell from the pointers" (8 or 2), as rounds 2-5 did.            */
int ns_ngp_adam_rec_ctl(float* master, void* half_params, float* grad, float* m1, float* m2, int record_floats, long n, int step,
                        float lr, float beta1, float beta2, float eps, float l2, float grad_scale, float fixed_scale,
                        const int* ctl, void* stream);
int ns_ngp_step_advance(int* ctl, int* counter, int* last, float fill, long max_samples, int min_rays, int max_rays,
                        float beta1, float beta2, void* stream);
/* Double-buffered form of the above (round 3; nerfslam/ngp.py): opens the side branch of step k that samples and marches the
 * rays of step k + 1 into the OTHER set of {control block, counters, ray tables, sample arrays}: ctl_dst <- {step + 1, ray
 * count adapted from THIS step's sample count, seed, views, Adam bias corrections}, last <- this step's counters (lazy host
 * reads), counter_dst <- 0, *loss_dst <- 0 (the other set's loss accumulator).                                           */
int ns_ngp_step_prepare(const int* ctl_src, int* ctl_dst, const int* counter_src, int* counter_dst, int* last, float fill,
                        long max_samples, int min_rays, int max_rays, float beta1, float beta2, float* loss_dst /* cleared; may be
                        NULL */, void* stream);
/* the two halves of ns_ngp_step_advance: `_rays` (counters -> last, next ray count, counters cleared) may run as soon as the
 * backward pass is done, so that the next step's ns_ngp_sample_rays_ctl(step_offset = 1) + ns_ngp_march_ctl overlap this
 * step's optimiser pass; `_count` (ctl[0] += 1, Adam's bias corrections) closes the step after both have finished. */
int ns_ngp_step_rays(int* ctl, int* counter, int* last, float fill, long max_samples, int min_rays, int max_rays, void* stream);
int ns_ngp_step_count(int* ctl, float beta1, float beta2, void* stream);

/* `_n` forms of the per-sample NeRF kernels: N is the CAPACITY (grid size and row stride of the unit-major tensors), the
 * number of samples actually processed is read from device memory (*n_dev, e.g. the marcher's counter; rounded up to 8 by
 * the MLP kernels, whose tail slots must carry zero gradients).  n_dev == NULL: identical to the plain entry points.
 * They let the graph-captured training step skip the unused tail of its fixed sample budget. */
int ns_ngp_encode_forward_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                            const float* positions, const void* params, void* out, int unit_major, long N, const int* n_dev,
                            void* stream);
int ns_ngp_encode_backward_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                             const float* positions, const void* dLdout, int unit_major, float* grad_params, float* workspace,
                             size_t workspace_bytes, float fixed_scale, long N, const int* n_dev, void* stream);
int ns_ngp_encode_backward_input_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                   const float* positions, const void* params, const void* dLdoutT, float* dLdpos, long N,
                                   const int* n_dev, void* stream);
/* Pose refinement without a second gather of the table (round 3): the encode forward also writes, per level and feature, the
 * derivative of the feature with respect to the position divided by the level's scale -- jacT: unit-major [6 n_levels][N] f16,
 * row 6 l + 3 f + d -- from the corner values it has in registers; ns_ngp_encode_jacobian_dot_n then forms
 * dL/dpos[n] = sum_l scale_l sum_f dL/dfeature[2 l + f][n] J[6 l + 3 f + :][n]  (what ns_ngp_encode_backward_input_n computes
 * with 8 x n_levels gathers per sample; equal to it up to the f16 rounding of J).  jacT == NULL: ns_ngp_encode_forward_n.   */
int ns_ngp_encode_forward_j_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                              const float* positions, const void* params, void* out, int unit_major, void* jacT, long N,
                              const int* n_dev, void* stream);
int ns_ngp_encode_jacobian_dot_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                 const void* jacT, const void* dLdoutT, float* dLdpos, long N, const int* n_dev, void* stream);
/* Table gradient of the hash grid with the optimiser step fused into it (round 3; tiny-cuda-nn GridEncoding backward +
 * Adam with its skip-zero-gradient rule, [EXTERNAL] instant-ngp behind fusion/nerf_fusion.py:291-307):
 *   dLdoutT: unit-major [2 n_levels][N] f16 (what ns_ngp_mlp_dgrad_n writes); fixed_scale > 0 (packed Q fixed point);
 *   workspace: ns_ngp_encode_backward_fused_workspace_bytes(..., max_samples >= N) bytes of device memory, ZEROED ONCE by
 *     the caller and private to this entry point afterwards (it leaves its overflow counter cleared); the size is checked;
 *   master != NULL: Adam is applied to every touched table entry in the flush of the accumulation (master / m1 / m2 f32, dense
 *     arrays or interleaved 32-byte records as ns_ngp_adam describes -- the trainer keeps records;
 *     half_params the f16 working copy; bias corrections from `step`, or from the device control block `ctl` as
 *     ns_ngp_adam_ctl); grad_params is then not written and may be NULL;
 *   master == NULL: the packed sums are ADDED to grad_params (same bits as ns_ngp_encode_backward).
 *   parts: mask of 1 = scatter of the binned levels (records), 2 = their accumulation (+ Adam), 4 = accumulation of the
 *     owner-computes dense levels (partial planes), 8 = their reduction (+ Adam); 15 = the whole gradient.  1 before 2, 4
 *     before 8; the two halves touch disjoint table entries and disjoint parts of the workspace.  Round 4: by default EVERY
 *     level is binned (the scatter merges the runs of neighbouring lanes on the dense levels) and parts 4 / 8 launch
 *     nothing -- ns_ngp_encode_backward_fused_dense_levels() = 0; NS_ENC_DENSE_BINNED=0 | 1 restore round 3's two forms.
 * No count pass, no global atomics on table entries, nothing dropped: runs that outgrow their slot spill to a list.   */
size_t ns_ngp_encode_backward_fused_workspace_bytes(int n_levels, int n_features, int log2_hashmap, int base_res,
                                                    float per_level_scale, long max_samples);
/* number of (leading, dense) levels that parts 4 | 8 of ns_ngp_encode_backward_fused_n handle; 0: those parts are no-ops */
int ns_ngp_encode_backward_fused_dense_levels(int n_levels, int n_features, int log2_hashmap, int base_res,
                                              float per_level_scale);
int ns_ngp_encode_backward_fused_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                   const float* positions, const void* dLdoutT, float* grad_params, void* workspace,
                                   size_t workspace_bytes, float fixed_scale, long N, const int* n_dev, float* master,
                                   void* half_params, float* m1, float* m2, int step, float lr, float beta1, float beta2,
                                   float eps, float grad_scale, const int* ctl, int parts, void* stream);
/* the same with the record layout told (see ns_ngp_adam_rec_ctl): what the trainer calls */
int ns_ngp_encode_backward_fused_rec_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                       const float* positions, const void* dLdoutT, float* grad_params, void* workspace,
                                       size_t workspace_bytes, float fixed_scale, long N, const int* n_dev, float* master,
                                       void* half_params, float* m1, float* m2, int record_floats, int step, float lr, float beta1,
                                       float beta2, float eps, float grad_scale, const int* ctl, int parts, void* stream);

/* Replicated trainers (`--multi_gpu` with more than one mapper; reference boundary examples/slam_demo.py:63-77): what
 * the trainers exchange is the LIST of table entries a step touched, not the table.
 *   ns_ngp_encode_backward_fused_emit_n   parts 1 | 2 of ns_ngp_encode_backward_fused_n whose flush appends (entry, packed
 *       fixed-point sum) pairs -- 16 bytes each, at most one per entry -- to `list` and adds their number to *list_count (device
 *       int, zeroed by the caller); needs every level on the binned path (ns_ngp_encode_backward_fused_dense_levels() == 0).
 *   ns_ngp_sparse_table_update   `n_lists` lists of `stride` pairs each (list r holds counts[r] <= max_count valid pairs: the
 *       trainers' lists after the all-gather): 64-bit integer sums per entry in `acc` (int64 per table entry, zero on entry and
 *       zero again on return), then Adam ONCE per touched entry with the complete sum -- exactly ns_ngp_encode_backward_fused_n's
 *       fused update, from sums over all trainers.  Every trainer runs it on the same lists and obtains the same table bit for
 *       bit; no parameter travels back.                                                                              */
int ns_ngp_encode_backward_fused_emit_n(int n_levels, int n_features, int log2_hashmap, int base_res, float per_level_scale,
                                        const float* positions, const void* dLdoutT, void* workspace, size_t workspace_bytes,
                                        float fixed_scale, long N, const int* n_dev, void* list, int* list_count, int parts,
                                        void* stream);
int ns_ngp_sparse_table_update(const void* lists, const int* counts, int n_lists, long stride, long max_count, void* acc,
                               float* master, void* half_params, float* m1, float* m2, int step, float lr, float beta1,
                               float beta2, float eps, float grad_scale, float fixed_scale, const int* ctl, void* stream);
int ns_ngp_sparse_table_update_rec(const void* lists, const int* counts, int n_lists, long stride, long max_count, void* acc,
                                   float* master, void* half_params, float* m1, float* m2, int record_floats, int step, float lr,
                                   float beta1, float beta2, float eps, float grad_scale, float fixed_scale, const int* ctl,
                                   void* stream);
int ns_ngp_mlp_forward_n(const void* weights, const void* featT, const float* dirs, void* out, void* h1T, void* cinT, void* h3T,
                         void* h4T, long N, const int* n_dev, void* stream);
int ns_ngp_mlp_backward_n(const void* weights, const void* dLdout, const void* featT, const void* h1T, const void* cinT,
                          const void* h3T, const void* h4T, void* dLdfeatT, void* d5T, void* d4T, void* d3T, void* ddT, void* d1T,
                          float* partial_ws, int ksplit, float* grad_weights, long N, const int* n_dev, void* stream);
/* ReLU bit masks (round 3): the forward pass can also write one bit per hidden unit and sample -- relu_masks: 6 N uint32
 * ([3 layers h1, h3, h4][2 lane halves][N], the MFMA accumulator layout of the kernels) -- and ns_ngp_mlp_dgrad_m_n takes the
 * ReLU derivatives from them instead of re-reading the three 64-unit f16 activations (24 B instead of 384 B per sample, and no
 * dependent loads between the layers of the backward chain).  Same results bit for bit.                                  */
int ns_ngp_mlp_forward_m_n(const void* weights, const void* featT, const float* dirs, void* out, void* h1T, void* cinT, void* h3T,
                           void* h4T, void* relu_masks, long N, const int* n_dev, void* stream);
int ns_ngp_mlp_dgrad_m_n(const void* weights, const void* dLdout, const void* relu_masks, void* dLdfeatT, void* d5T, void* d4T,
                         void* d3T, void* ddT, void* d1T, long N, const int* n_dev, void* stream);
/* Fused backward pass of the two MLPs (round 3): the forward chain is recomputed from the features, the activation gradients
 * follow in registers, and every layer's weight gradient is contracted on chip (LDS transposes + MFMA) -- nothing but
 * dLdfeatT [32,N] f16 and the weight gradients leaves the kernel: 148 B of HBM traffic per sample instead of ~1.95 KB for
 * ns_ngp_mlp_forward_n (with activation buffers) + ns_ngp_mlp_dgrad_n + ns_ngp_mlp_wgrad_n.  Same dLdfeatT bit for bit; the
 * weight gradients differ by summation order.  grad_weights is ADDED to; partial_ws: wgs * 10240 floats; wgs workgroups
 * (512 fill the chip twice over).                                                                                      */
int ns_ngp_mlp_backward_fused_n(const void* weights, const void* featT, const float* dirs, const void* dLdout, void* dLdfeatT,
                                float* partial_ws, int wgs, float* grad_weights, long N, const int* n_dev, void* stream);
/* Weight gradients off the critical path (round 3, the trainer's default): ns_ngp_mlp_dgrad_m_n accepts NULL for its five
 * gradient buffers (it then writes dLdfeatT only: 96 B of traffic per sample) and ns_ngp_mlp_wgrad_recompute_n recomputes both
 * chains on chip from the features and the loss gradient and contracts the five weight gradients there -- a 46-KB-LDS kernel
 * that runs on a side stream NEXT TO the table gradient.  frags: ns_ngp_mlp_fragment_table_bytes() bytes, written by
 * ns_ngp_mlp_pack_fragments from the current f16 weights (once per optimiser step).                                        */
size_t ns_ngp_mlp_fragment_table_bytes(void);
/* forward pass / activation backward of the split form with the weights taken from the fragment table (no activation buffers,
 * bit masks only; dLdfeatT only): bit-identical to ns_ngp_mlp_forward_m_n / ns_ngp_mlp_dgrad_m_n, without the per-workgroup
 * element-wise gather of 24 / 20 weight fragments                                                                            */
int ns_ngp_mlp_forward_f_n(const void* frags, const void* featT, const float* dirs, void* out, void* relu_masks, long N,
                           const int* n_dev, void* stream);
int ns_ngp_mlp_dgrad_f_n(const void* frags, const void* dLdout, const void* relu_masks, void* dLdfeatT, long N, const int* n_dev,
                         void* stream);
int ns_ngp_mlp_pack_fragments(const void* weights, void* frags, void* stream);
int ns_ngp_mlp_wgrad_recompute_n(const void* frags, const void* featT, const float* dirs, const void* dLdout, float* partial_ws,
                                 int wgs, float* grad_weights, long N, const int* n_dev, void* stream);
/* The same in pieces (round 4; the optimiser of the MLP has no counterpart file in the reference: tiny-cuda-nn's Adam inside
 * the instant-ngp fork, [EXTERNAL]): ns_ngp_mlp_wgrad_partials_n is the weight-gradient launch alone -- it leaves
 * ns_ngp_mlp_wgrad_slabs(wgs, N) slabs of 10240 floats in partial_ws; ns_ngp_mlp_reduce adds them to grad_weights (replicated
 * trainers: the all-reduce follows); ns_ngp_mlp_step_fused is the MLP's whole optimiser step in ONE launch: slab reduce (same
 * order, same bits) + Adam (grad_weights is added in and cleared; bias corrections of `step`, or of ctl when given) + the f16
 * copy + both fragment tables updated in place (frags must have been packed once by ns_ngp_mlp_pack_fragments: the zero rows
 * of the tables are not rewritten).  Bit-identical to ns_ngp_mlp_reduce + ns_ngp_adam_ctl + ns_ngp_mlp_pack_fragments.          */
int ns_ngp_mlp_wgrad_slabs(int wgs, long N);
int ns_ngp_mlp_wgrad_partials_n(const void* frags, const void* featT, const float* dirs, const void* dLdout, float* partial_ws,
                                int wgs, long N, const int* n_dev, void* stream);
int ns_ngp_mlp_reduce(const float* partial_ws, int slabs, float* grad_weights, void* stream);
int ns_ngp_mlp_step_fused(const float* partial_ws, int slabs, float* grad_weights, float* master, void* half_params, float* m1,
                          float* m2, void* frags, int step, float lr, float beta1, float beta2, float eps, float l2,
                          float grad_scale, const int* ctl, void* stream);
/* the two halves of ns_ngp_mlp_backward_n: activation gradients (writes dLdfeatT and the d*T buffers), then the weight
 * gradients (reads them).  Separate entries so that the caller can put the second half on another stream, next to the
 * hash-grid backward that consumes dLdfeatT (nerfslam/ngp.py).                                                       */
int ns_ngp_mlp_dgrad_n(const void* weights, const void* dLdout, const void* h1T, const void* h3T, const void* h4T, void* dLdfeatT,
                       void* d5T, void* d4T, void* d3T, void* ddT, void* d1T, long N, const int* n_dev, void* stream);
int ns_ngp_mlp_wgrad_n(const void* featT, const void* h1T, const void* cinT, const void* h3T, const void* h4T, const void* d5T,
                       const void* d4T, const void* d3T, const void* ddT, const void* d1T, float* partial_ws, int ksplit,
                       float* grad_weights, long N, const int* n_dev, void* stream);

/* Occupancy-grid refresh (instant-ngp's update_density_grid, subset form; nerfslam/ngp.py:update_density_grid):
 *   ns_ngp_grid_cells   draws n cells uniformly over all cascades (PCG hash of seed + index; cell = (mip G + z) G^2 + y G + x)
 *                       and a jittered point in each, written in the unit cube of the render box [box_lo, box_hi]^3:
 *                       cells int32 [n], pos_unit f32 [n,3] -- the caller runs ns_ngp_encode_forward + ns_ngp_mlp_forward on them;
 *   ns_ngp_grid_update  net_out f16 [n,4] (log-density in column 3): grid *= decay; grid[cell] = max(grid[cell],
 *                       exp(log-density) * min_step); occupied = grid > min(mean(grid), max_threshold); bits[j] bit i = cell 8j+i.
 *                       partial_ws: 256 doubles.  Order-independent (integer atomicMax on the float bits, fixed-order mean).  */
int ns_ngp_grid_cells(int grid_size, int n_cascades, unsigned seed, int n, float box_lo, float box_hi, int* cells, float* pos_unit,
                      void* stream);
/* the same with the seed completed on the device from the step's control block (seed ^ pcg(steps completed + 1)): a refresh captured
 * into the optimiser step's HIP graph draws new cells on every replay                                                      */
int ns_ngp_grid_cells_ctl(int grid_size, int n_cascades, unsigned seed, int n, float box_lo, float box_hi, int* cells,
                          float* pos_unit, const int* ctl, void* stream);
int ns_ngp_grid_update(const void* net_out, const int* cells, int n, float min_step, float decay, float max_threshold,
                       float* density_grid, long n_cells_total, double* partial_ws, unsigned char* bits, void* stream);
/* The same refresh with the decay applied ONLY to the cells drawn in this update (grid[c] = max(decay grid[c], new maximum of c);
 * undrawn cells keep their value): the subset rule draws 4 % of the grid per update, decaying all of it lets unobserved dense
 * cells fade below the threshold between two draws.  tmp_grid: n_cells_total floats, zeroed once by the caller, left zeroed.   */
int ns_ngp_grid_update_sampled(const void* net_out, const int* cells, int n, float min_step, float decay, float max_threshold,
                               float* density_grid, float* tmp_grid, long n_cells_total, double* partial_ws, unsigned char* bits,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFSLAM_HIP_H */
