// ba_plan.hip -- host-side graph plan for the dense bundle adjustment (pure C++, no device work).
//
// The reference rebuilds these index structures on the host on EVERY call, with device<->host
// copies in the middle of the BA (src/droid_kernels.cu:1702-1710 `_unique`, :1065-1103 accum_cuda
// argsort + pointer loop, :1359-1402 schur_block pair enumeration).  Here they are built once per
// graph change from the host copy of (ii, jj) the frontend already owns, and uploaded once.
#include "common.h"

#include <algorithm>
#include <vector>

namespace {

struct PlanData {
  std::vector<int32_t> kx, kk, row_pose, src_ptr, src_edge, pairs, slot_rows_ptr, slot_rows, win_rows_ptr, win_rows, gram_jobs, job_plane, job_hrow;
  int K = 0;
};

int build(const int64_t* ii, const int64_t* jj, int M, int kf0, int kf1, PlanData& d) {
  const int P = kf1 - kf0;
  if (P < 0 || M < 0) return NS_EINVAL;
  const int NE = P + M;
  std::vector<int64_t> ii_exp(NE), jj_exp(NE);
  for (int t = 0; t < P; t++) ii_exp[t] = jj_exp[t] = kf0 + t;
  for (int e = 0; e < M; e++) {
    ii_exp[P + e] = ii[e];
    jj_exp[P + e] = jj[e];
  }
  // torch::_unique(ii_expanded, sorted=true, return_inverse=true)   (droid_kernels.cu:1706-1710)
  std::vector<int64_t> u(ii_exp);
  std::sort(u.begin(), u.end());
  u.erase(std::unique(u.begin(), u.end()), u.end());
  d.K = (int)u.size();
  d.kx.resize(d.K);
  for (int k = 0; k < d.K; k++) d.kx[k] = (int32_t)u[k];
  d.kk.resize(NE);
  d.row_pose.resize(NE);
  for (int n = 0; n < NE; n++) {
    d.kk[n] = (int32_t)(std::lower_bound(u.begin(), u.end(), ii_exp[n]) - u.begin());
    d.row_pose[n] = (int32_t)(jj_exp[n] - kf0);
  }
  // CSR of edges by depth slot (the job of accum_cuda, :1065-1103); ascending edge id inside a slot
  d.src_ptr.assign(d.K + 1, 0);
  for (int e = 0; e < M; e++) d.src_ptr[d.kk[P + e] + 1]++;
  for (int k = 0; k < d.K; k++) d.src_ptr[k + 1] += d.src_ptr[k];
  d.src_edge.resize(M);
  {
    std::vector<int32_t> fill(d.src_ptr.begin(), d.src_ptr.end() - 1);
    for (int e = 0; e < M; e++) d.src_edge[fill[d.kk[P + e]]++] = e;
  }
  // E rows per slot (self loops first, then edges in ascending id)
  d.slot_rows_ptr.assign(d.K + 1, 0);
  for (int n = 0; n < NE; n++) d.slot_rows_ptr[d.kk[n] + 1]++;
  for (int k = 0; k < d.K; k++) d.slot_rows_ptr[k + 1] += d.slot_rows_ptr[k];
  d.slot_rows.resize(NE);
  {
    std::vector<int32_t> fill(d.slot_rows_ptr.begin(), d.slot_rows_ptr.end() - 1);
    for (int n = 0; n < NE; n++) d.slot_rows[fill[d.kk[n]]++] = n;
  }
  // Schur pairs (:1368-1399): rows whose pose lies in the window, grouped by pose; a pair is two
  // rows of the same depth slot.  Only n <= m is emitted: the kernel mirrors the block (the
  // reference evaluates both orders separately in float, we keep H exactly symmetric).
  d.pairs.clear();
  for (int k = 0; k < d.K; k++) {
    const int b = d.slot_rows_ptr[k], e = d.slot_rows_ptr[k + 1];
    for (int a = b; a < e; a++) {
      const int n = d.slot_rows[a];
      if (d.row_pose[n] < 0 || d.row_pose[n] >= P) continue;
      for (int c = a; c < e; c++) {
        const int m = d.slot_rows[c];
        if (d.row_pose[m] < 0 || d.row_pose[m] >= P) continue;
        d.pairs.push_back(n);
        d.pairs.push_back(m);
        d.pairs.push_back(k);
      }
    }
  }
  // Schur complement as one Gram matrix per depth slot (ba_schur_gram_kernel): the slot's WINDOW rows (pose inside [0, P): the
  // only ones schur_block admits, :1375) in slot_rows order, six values each, cut into tiles of 16 values; blocks of
  // NS_GRAM_BLOCK tiles.  A job is (slot, first A tile, A tiles, first B tile, B tiles, 0): the diagonal job of a block computes the
  // upper triangle of its <= 8 x 8 tile pairs; a pair of blocks bi < bj is two jobs of <= 4 x 8 tile pairs (the accumulators
  // of a job have to fit one wave's registers).  A slot of <= 21 rows -- every slot of a tracking window -- is ONE job.
  d.win_rows_ptr.assign(d.K + 1, 0);
  d.win_rows.clear();
  d.gram_jobs.clear();
  for (int k = 0; k < d.K; k++) {
    for (int a = d.slot_rows_ptr[k]; a < d.slot_rows_ptr[k + 1]; a++) {
      const int n = d.slot_rows[a];
      if (d.row_pose[n] >= 0 && d.row_pose[n] < P) d.win_rows.push_back(n);
    }
    d.win_rows_ptr[k + 1] = (int32_t)d.win_rows.size();
    const int nr = d.win_rows_ptr[k + 1] - d.win_rows_ptr[k];
    const int nt = (6 * nr + 15) / 16;
    const int nblk = (nt + NS_GRAM_BLOCK - 1) / NS_GRAM_BLOCK;
    // per job: the plane of E (row * 6 + component) and the row of H (6 * pose + component) of each of its <= 128 A and <= 128 B
    // values, -1 for padding -- the kernels read these instead of walking win_rows_ptr -> win_rows -> row_pose (three dependent
    // round trips in front of a 3-us main loop at tracking sizes)
    auto emit = [&](int a0, int na, int b0, int nb) {
      const int32_t hdr[NS_GRAM_JOB_INTS] = {k, a0, na, b0, nb, 0};
      d.gram_jobs.insert(d.gram_jobs.end(), hdr, hdr + NS_GRAM_JOB_INTS);
      for (int side = 0; side < 2; side++) {
        const int t0 = side == 0 ? a0 : b0, ntl = side == 0 ? na : nb;
        for (int i = 0; i < 128; i++) {
          const int v = t0 * 16 + i;
          const bool live = i < ntl * 16 && v < 6 * nr;
          const int n = live ? d.win_rows[d.win_rows_ptr[k] + v / 6] : 0;
          d.job_plane.push_back(live ? n * 6 + v % 6 : -1);
          d.job_hrow.push_back(live ? 6 * d.row_pose[n] + v % 6 : -1);
        }
      }
    };
    for (int bi = 0; bi < nblk; bi++) {
      const int a0 = bi * NS_GRAM_BLOCK, na = std::min(NS_GRAM_BLOCK, nt - a0);
      emit(a0, na, a0, na);
      for (int bj = bi + 1; bj < nblk; bj++)
        for (int h = 0; h < na; h += NS_GRAM_BLOCK / 2)
          emit(a0 + h, std::min(NS_GRAM_BLOCK / 2, na - h), bj * NS_GRAM_BLOCK, std::min(NS_GRAM_BLOCK, nt - bj * NS_GRAM_BLOCK));
    }
  }
  return NS_OK;
}

size_t total_count(const PlanData& d) {
  return d.kx.size() + d.kk.size() + d.row_pose.size() + d.src_ptr.size() + d.src_edge.size() + d.pairs.size() +
         d.slot_rows_ptr.size() + d.slot_rows.size() + d.win_rows_ptr.size() + d.win_rows.size() + d.gram_jobs.size() +
         d.job_plane.size() + d.job_hrow.size();
}

}  // namespace

extern "C" size_t ns_ba_plan_index_count(const int64_t* ii_host, const int64_t* jj_host, int M, int kf0, int kf1) {
  PlanData d;
  if (build(ii_host, jj_host, M, kf0, kf1, d) != NS_OK) return 0;
  return total_count(d);
}

extern "C" int ns_ba_plan_build(const int64_t* ii_host, const int64_t* jj_host, int M, int kf0, int kf1,
                                ns_ba_plan* plan, int32_t* index_host, size_t* offsets_host) {
  NS_REQUIRE(plan && index_host && offsets_host, "ns_ba_plan_build: null pointer");
  NS_REQUIRE(M >= 0 && kf1 >= kf0, "ns_ba_plan_build: bad sizes M=%d kf0=%d kf1=%d", M, kf0, kf1);
  NS_REQUIRE(M == 0 || (ii_host && jj_host), "ns_ba_plan_build: null edge lists");
  PlanData d;
  int rc = build(ii_host, jj_host, M, kf0, kf1, d);
  if (rc != NS_OK) return rc;
  plan->M = M;
  plan->P = kf1 - kf0;
  plan->K = d.K;
  plan->kf0 = kf0;
  plan->kf1 = kf1;
  plan->n_pairs = (int)(d.pairs.size() / 3);
  plan->n_rows = plan->P + M;
  plan->n_jobs = (int)(d.gram_jobs.size() / NS_GRAM_JOB_INTS);
  plan->max_src = 0;
  for (int k = 0; k < d.K; k++) plan->max_src = std::max(plan->max_src, (int)(d.src_ptr[k + 1] - d.src_ptr[k]));
  const std::vector<int32_t>* parts[NS_BA_PLAN_PARTS] = {&d.kx,           &d.kk,        &d.row_pose,      &d.src_ptr,
                                                         &d.src_edge,     &d.pairs,     &d.slot_rows_ptr, &d.slot_rows,
                                                         &d.win_rows_ptr, &d.win_rows,  &d.gram_jobs,     &d.job_plane,
                                                         &d.job_hrow};
  size_t off = 0;
  for (int i = 0; i < NS_BA_PLAN_PARTS; i++) {
    offsets_host[i] = off;
    std::copy(parts[i]->begin(), parts[i]->end(), index_host + off);
    off += parts[i]->size();
  }
  return NS_OK;
}
