/* oracle.c -- CPU restatement of the NgsAMG multigrid apply path.  See oracle.h for scope, status
 * ("parity unpinned") and the reference lines each function follows.  TEST INFRASTRUCTURE ONLY. */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define MAXBS 6

#define ORC_MAX_OWNED 512
struct orc_handle {
  int n_levels;
  orc_level* lev;            /* copies of the descriptors (arrays are borrowed) */
  int cycle, clev_inv;
  double** x_level;          /* amg_matrix.cpp:19-26 */
  double** rhs_level;
  double** res_level;
  /* coarsest level: Cholesky factor of A_L restricted to free dofs */
  int64_t crs_n;             /* scalar size of the coarsest level */
  int64_t crs_nf;            /* number of free scalar dofs */
  int64_t* crs_idx;          /* free scalar dof indices */
  double* crs_L;             /* nf x nf lower factor */
  double* crs_tmp;
  double** x_old;            /* per level: snapshot of x for hybrid GS */
  void* owned[ORC_MAX_OWNED];   /* arrays allocated by orc_first_touch (everything else is borrowed) */
  int n_owned;
};

static struct orc_handle* g_cur = NULL;   /* set by the public entry points (the oracle is single-threaded at this level) */

static char g_err[512];
static int g_threads = 1;

const char* orc_last_error(void) { return g_err; }
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

static int fail(const char* m) { snprintf(g_err, sizeof(g_err), "%s", m); return 1; }

static inline int64_t vlen(const orc_level* L) { return L->A.n_rows * L->A.br; }

/* vector helpers of the cycles, threaded like the sparse kernels (the multi-threaded CPU baseline would otherwise
 * spend a fifth of its time in single-threaded memset / memcpy of 80 MB vectors) */
static void vzero(double* x, int64_t n) {
  if (g_threads <= 1) { memset(x, 0, sizeof(double) * n); return; }
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i < n; i++) x[i] = 0.0;
}
static void vcopy(double* dst, const double* src, int64_t n) {
  if (g_threads <= 1) { memcpy(dst, src, sizeof(double) * n); return; }
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i < n; i++) dst[i] = src[i];
}

/* ---------------------------------------------------------------- sparse kernels (NGSolve's part) */

/* y = A x (mode 0), y = b - A x (mode 1), y += s A x (mode 2)
 * = SparseMatrix<TM>::Mult / MultAdd as used by dof_map.cpp:651,708 and base_smoother.hpp:132-142 */
static void spmv(const orc_matrix* A, const double* x, double* y, int mode, double s, const double* b) {
  const int br = A->br, bc = A->bc;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i < A->n_rows; i++) {
    double acc[MAXBS] = {0, 0, 0, 0, 0, 0};
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      const double* a = A->val + k * br * bc;
      const double* xv = x + (int64_t)A->col[k] * bc;
      for (int r = 0; r < br; r++)
        for (int c = 0; c < bc; c++) acc[r] += a[r * bc + c] * xv[c];
    }
    for (int r = 0; r < br; r++) {
      if (mode == 0) y[i * br + r] = acc[r];
      else if (mode == 1) y[i * br + r] = b[i * br + r] - acc[r];
      else y[i * br + r] += s * acc[r];
    }
  }
}

/* ---------------------------------------------------------------- Jacobi (base_smoother.cpp:61-114) */

/* x += omega * Dinv * v */
static void diag_add(const orc_level* L, double* x, const double* v) {
  const int bs = L->A.br;
  const int64_t n = L->A.n_rows;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i < n; i++) {
    const double* d = L->dinv + i * bs * bs;
    for (int r = 0; r < bs; r++) {
      double t = 0;
      for (int c = 0; c < bs; c++) t += d[r * bs + c] * v[i * bs + c];
      x[i * bs + r] += L->omega * t;
    }
  }
}

static void jacobi_smooth(const orc_level* L, double* x, const double* b, double* res,
                          int res_updated, int update_res, int x_zero) {
  /* RichardsonSmoother::Smooth, base_smoother.cpp:61-74; SmoothBack is identical (:77-82) */
  if (!res_updated && x_zero) diag_add(L, x, b);
  else {
    if (!res_updated) spmv(&L->A, x, res, 1, 0.0, b);
    diag_add(L, x, res);
  }
  if (update_res) spmv(&L->A, x, res, 1, 0.0, b);
}

/* ---------------------------------------------------------------- Gauss-Seidel GSS3 (gssmoother.cpp) */

static inline int is_free(const orc_level* L, int64_t k) { return !L->free || L->free[k]; }

/* RHS form, gssmoother.cpp:209-212:  x_k += dinv_k (b_k - A_k: x)
 * xo != NULL: hybrid form -- entries whose column belongs to another block read the sweep-start values xo */
static inline void gs_row_rhs(const orc_level* L, int64_t k, double* x, const double* b, const double* xo) {
  const orc_matrix* A = &L->A;
  const int bs = A->br;
  double r[MAXBS] = {0, 0, 0, 0, 0, 0};
  for (int64_t p = A->rowptr[k]; p < A->rowptr[k + 1]; p++) {
    const double* a = A->val + p * bs * bs;
    const int64_t j = A->col[p];
    const double* xv = ((xo && L->gs_block[j] != L->gs_block[k]) ? xo : x) + j * bs;
    for (int i = 0; i < bs; i++)
      for (int j = 0; j < bs; j++) r[i] += a[i * bs + j] * xv[j];
  }
  double t[MAXBS];
  for (int i = 0; i < bs; i++) t[i] = b[k * bs + i] - r[i];
  const double* d = L->dinv + k * bs * bs;
  for (int i = 0; i < bs; i++) {
    double u = 0;
    for (int j = 0; j < bs; j++) u += d[i * bs + j] * t[j];
    x[k * bs + i] += u;
  }
}

/* RES form, gssmoother.cpp:274-278:  w = -dinv_k res_k ; res += A_k:^T w ; x_k -= w */
static inline void gs_row_res(const orc_level* L, int64_t k, double* x, double* res) {
  const orc_matrix* A = &L->A;
  const int bs = A->br;
  const double* d = L->dinv + k * bs * bs;
  double w[MAXBS];
  for (int i = 0; i < bs; i++) {
    double u = 0;
    for (int j = 0; j < bs; j++) u += d[i * bs + j] * res[k * bs + j];
    w[i] = -u;
  }
  for (int64_t p = A->rowptr[k]; p < A->rowptr[k + 1]; p++) {
    const double* a = A->val + p * bs * bs;
    double* rj = res + (int64_t)A->col[p] * bs;
    for (int j = 0; j < bs; j++) {
      double u = 0;
      for (int i = 0; i < bs; i++) u += a[i * bs + j] * w[i];   /* Trans(A_kj) * w */
      rj[j] += u;
    }
  }
  for (int i = 0; i < bs; i++) x[k * bs + i] -= w[i];
}

/* sweep over the free rows; natural order (reference) or the given visiting order (colour-major = what
 * the GPU multicolour kernel does); backwards = reversed */
static void gs_sweep(const orc_level* L, double* x, const double* b, double* res, int res_form, int backwards) {
  const int64_t n = L->A.n_rows;
  const double* xo = NULL;
  if (L->gs_block) {
    /* hybrid: freeze the off-block values; only the gather (RHS) form is restated, the reference's RES variant gives
     * the same x and the same final residual (hybrid_base_smoother.cpp:296-405) */
    double* snap = g_cur->x_old[(int)(L - g_cur->lev)];
    memcpy(snap, x, sizeof(double) * L->A.n_cols * L->A.bc);
    xo = snap;
    res_form = 0;
  }
  if (L->gs_order) {
    const int64_t m = L->gs_order_len;
    for (int64_t q = 0; q < m; q++) {
      int64_t k = L->gs_order[backwards ? m - 1 - q : q];
      if (!is_free(L, k)) continue;
      if (res_form) gs_row_res(L, k, x, res); else gs_row_rhs(L, k, x, b, xo);
    }
    return;
  }
  /* free range [first_free, next_free), gssmoother.cpp:111-139 -- equivalent to testing every row */
  if (!backwards) {
    for (int64_t k = 0; k < n; k++) if (is_free(L, k)) { if (res_form) gs_row_res(L, k, x, res); else gs_row_rhs(L, k, x, b, xo); }
  } else {
    for (int64_t k = n - 1; k >= 0; k--) if (is_free(L, k)) { if (res_form) gs_row_res(L, k, x, res); else gs_row_rhs(L, k, x, b, xo); }
  }
}

static void gs_smooth(const orc_level* L, int dir, double* x, const double* b, double* res,
                      int res_updated, int update_res, int x_zero) {
  /* GSS3::Smooth / SmoothBack, gssmoother.cpp:350-398 */
  if (L->gs_block) {
    gs_sweep(L, x, b, res, 0, dir);
    if (update_res) spmv(&L->A, x, res, 1, 0.0, b);
    return;
  }
  if (res_updated) {
    if (update_res) gs_sweep(L, x, b, res, 1, dir);
    else gs_sweep(L, x, b, res, 0, dir);
  } else {
    if (update_res) {
      /* CalcResiduum (base_smoother.hpp:132-142): res = b; if (!x_zero) res -= A x */
      if (x_zero) memcpy(res, b, sizeof(double) * vlen(L));
      else spmv(&L->A, x, res, 1, 0.0, b);
      gs_sweep(L, x, b, res, 1, dir);
    } else gs_sweep(L, x, b, res, 0, dir);
  }
}

/* ---------------------------------------------------------------- block Gauss-Seidel (block_gssmoother.cpp) */

#define ORC_MAXBLOCK 4096
/* one block, RHS form (:300-323): hr_j = b_j - A_j: x for the rows j of the block, hup = Dinv hr, x_B += hup */
/* xo != NULL: hybrid form (reference HybridBS, block_gssmoother.cpp:505-585 on top of hybrid_base_smoother.cpp:296-495) --
 * entries whose column belongs to another rank ("gs_block" = owner of every row) read the sweep-start values xo */
static void bgs_block_rhs(const orc_level* L, int32_t k, double* x, const double* b, double* hr, double* hup, const double* xo) {
  const orc_matrix* A = &L->A;
  const int bs = A->br;
  const int32_t p0 = L->block_ptr[k], m = L->block_ptr[k + 1] - p0;
  const int M = m * bs;
  for (int32_t q = 0; q < m; q++) {
    const int64_t j = L->block_rows[p0 + q];
    double r[MAXBS] = {0, 0, 0, 0, 0, 0};
    for (int64_t p = A->rowptr[j]; p < A->rowptr[j + 1]; p++) {
      const double* a = A->val + p * bs * bs;
      const int64_t cj = A->col[p];
      const double* xv = ((xo && L->gs_block[cj] != L->gs_block[j]) ? xo : x) + cj * bs;
      for (int i = 0; i < bs; i++)
        for (int c = 0; c < bs; c++) r[i] += a[i * bs + c] * xv[c];
    }
    for (int i = 0; i < bs; i++) hr[q * bs + i] = b[j * bs + i] - r[i];
  }
  const double* D = L->bdinv + L->bdinv_ptr[k];
  for (int i = 0; i < M; i++) { double u = 0; for (int c = 0; c < M; c++) u += D[(int64_t)c * M + i] * hr[c]; hup[i] = u; }
  for (int32_t q = 0; q < m; q++) { const int64_t j = L->block_rows[p0 + q]; for (int i = 0; i < bs; i++) x[j * bs + i] += hup[q * bs + i]; }
}

/* one block, RES form (:372-388): hr = res_B, hup = Dinv hr, res -= A_j:^T hup_j for the rows of the block, x_B += hup */
static void bgs_block_res(const orc_level* L, int32_t k, double* x, double* res, double* hr, double* hup) {
  const orc_matrix* A = &L->A;
  const int bs = A->br;
  const int32_t p0 = L->block_ptr[k], m = L->block_ptr[k + 1] - p0;
  const int M = m * bs;
  for (int32_t q = 0; q < m; q++) { const int64_t j = L->block_rows[p0 + q]; for (int i = 0; i < bs; i++) hr[q * bs + i] = res[j * bs + i]; }
  const double* D = L->bdinv + L->bdinv_ptr[k];
  for (int i = 0; i < M; i++) { double u = 0; for (int c = 0; c < M; c++) u += D[(int64_t)c * M + i] * hr[c]; hup[i] = u; }
  for (int32_t q = 0; q < m; q++) {
    const int64_t j = L->block_rows[p0 + q];
    const double* w = hup + q * bs;
    for (int64_t p = A->rowptr[j]; p < A->rowptr[j + 1]; p++) {
      const double* a = A->val + p * bs * bs;
      double* rc = res + (int64_t)A->col[p] * bs;
      for (int c = 0; c < bs; c++) { double u = 0; for (int i = 0; i < bs; i++) u += a[i * bs + c] * w[i]; rc[c] -= u; }   /* Trans(A_jc) * hup_j */
    }
    for (int i = 0; i < bs; i++) x[j * bs + i] += w[i];
  }
}

static int bgs_sweep(const orc_level* L, double* x, const double* b, double* res, int res_form, int backwards) {
  double hr[ORC_MAXBLOCK], hup[ORC_MAXBLOCK];
  const int32_t nb = L->n_blocks;
  const double* xo = NULL;
  if (L->gs_block) {           /* hybrid: freeze the off-rank values, gather form only (like gs_sweep) */
    double* snap = g_cur->x_old[(int)(L - g_cur->lev)];
    memcpy(snap, x, sizeof(double) * L->A.n_cols * L->A.bc);
    xo = snap;
    res_form = 0;
  }
  for (int32_t q = 0; q < nb; q++) {
    const int32_t pos = backwards ? nb - 1 - q : q;
    const int32_t k = L->block_order ? L->block_order[pos] : pos;
    if ((L->block_ptr[k + 1] - L->block_ptr[k]) * L->A.br > ORC_MAXBLOCK) return 1;
    if (res_form) bgs_block_res(L, k, x, res, hr, hup); else bgs_block_rhs(L, k, x, b, hr, hup, xo);
  }
  return 0;
}

static void bgs_smooth(const orc_level* L, int dir, double* x, const double* b, double* res,
                       int res_updated, int update_res, int x_zero) {
  /* BSmoother::Smooth / SmoothBack, block_gssmoother.cpp:434-498 */
  if (L->gs_block) {           /* hybrid block smoother: local sweep with frozen off-rank values, then the residual */
    bgs_sweep(L, x, b, res, 0, dir);
    if (update_res) spmv(&L->A, x, res, 1, 0.0, b);
    return;
  }
  if (update_res) {
    if (!res_updated) {          /* CalcResiduum(x, b, res, x_zero), base_smoother.hpp:132-142 */
      if (x_zero) memcpy(res, b, sizeof(double) * vlen(L));
      else spmv(&L->A, x, res, 1, 0.0, b);
    }
    bgs_sweep(L, x, b, res, 1, dir);
  } else bgs_sweep(L, x, b, res, 0, dir);      /* "if res_updated, just forget about the residual vector" */
}

/* ---------------------------------------------------------------- smoother dispatch + ProxySmoother */

static void base_smooth(const orc_level* L, int dir, double* x, const double* b, double* res,
                        int res_updated, int update_res, int x_zero) {
  if (L->sm_type == ORC_SM_JACOBI) jacobi_smooth(L, x, b, res, res_updated, update_res, x_zero);
  else if (L->sm_type == ORC_SM_BGS) bgs_smooth(L, dir, x, b, res, res_updated, update_res, x_zero);
  else gs_smooth(L, dir, x, b, res, res_updated, update_res, x_zero);
}

/* BaseSmoother::SmoothSymm, base_smoother.hpp:79-85 */
static void smooth_symm(const orc_level* L, double* x, const double* b, double* res, int ru, int ur, int xz) {
  base_smooth(L, 0, x, b, res, ru, ur, xz);
  base_smooth(L, 1, x, b, res, ur, ur, 0);
}

/* ProxySmoother (base_smoother.hpp:169-229), created iff sm_symm || sm_steps > 1 (amg_pc.cpp:1079-1082) */
static void level_smooth(const orc_level* L, int dir, double* x, const double* b, double* res,
                         int res_updated, int update_res, int x_zero) {
  const int k = L->sm_steps < 1 ? 1 : L->sm_steps;
  if (!L->sm_symm && k == 1) { base_smooth(L, dir, x, b, res, res_updated, update_res, x_zero); return; }
  if (L->sm_symm) {            /* SmoothSymmK for both directions */
    smooth_symm(L, x, b, res, res_updated, update_res, x_zero);
    for (int j = 1; j < k; j++) smooth_symm(L, x, b, res, update_res, update_res, 0);
  } else {                     /* SmoothK / SmoothBackK, base_smoother.hpp:87-103 */
    base_smooth(L, dir, x, b, res, res_updated, update_res, x_zero);
    for (int j = 1; j < k; j++) base_smooth(L, dir, x, b, res, update_res, update_res, 0);
  }
}

/* ---------------------------------------------------------------- transfers (dof_map.cpp:636-709) */

static void transfer_f2c(const orc_handle* h, int level, const double* xf, double* xc) {
  spmv(&h->lev[level].PT, xf, xc, 0, 0.0, NULL);
}
static void add_c2f(const orc_handle* h, int level, double fac, double* xf, const double* xc) {
  spmv(&h->lev[level].P, xc, xf, 2, fac, NULL);
}

/* ---------------------------------------------------------------- coarsest level */

static void coarse_solve(const orc_handle* h, const double* rhs, double* x) {
  const int64_t N = h->crs_n, nf = h->crs_nf;
  for (int64_t i = 0; i < N; i++) x[i] = 0.0;
  if (!h->clev_inv) return;    /* clev != inv: x_L = 0 (amg_matrix.cpp:242-246) */
  double* t = h->crs_tmp;
  const double* Lf = h->crs_L;
  for (int64_t i = 0; i < nf; i++) {
    double s = rhs[h->crs_idx[i]];
    for (int64_t k = 0; k < i; k++) s -= Lf[i * nf + k] * t[k];
    t[i] = s / Lf[i * nf + i];
  }
  for (int64_t i = nf - 1; i >= 0; i--) {
    double s = t[i];
    for (int64_t k = i + 1; k < nf; k++) s -= Lf[k * nf + i] * t[k];
    t[i] = s / Lf[i * nf + i];
  }
  for (int64_t i = 0; i < nf; i++) x[h->crs_idx[i]] = t[i];
}

/* ---------------------------------------------------------------- cycles (amg_matrix.cpp) */

static void smooth_v_from_level(orc_handle* h, int start, double* x, const double* b, double* res,
                                int res_updated, int update_res, int x_zero);

static void prep_level(orc_handle* h, int l) {
  /* x_l = 0; r_l = b_l   (amg_matrix.cpp:193-202) */
  const int64_t n = vlen(&h->lev[l]);
  vzero(h->x_level[l], n);
  vcopy(h->res_level[l], h->rhs_level[l], n);
}

static void cycle_v(orc_handle* h, double* x, const double* b) {
  const int L = h->n_levels;
  for (int l = 0; l + 1 < L; l++) {
    double* xl = l == 0 ? x : h->x_level[l];
    const double* bl = l == 0 ? b : h->rhs_level[l];
    double* rl = h->res_level[l];
    const int64_t n = vlen(&h->lev[l]);
    vzero(xl, n);
    vcopy(rl, bl, n);
    level_smooth(&h->lev[l], 0, xl, bl, rl, 1, 1, 1);
    transfer_f2c(h, l, rl, h->rhs_level[l + 1]);
  }
  if (L == 1) { coarse_solve(h, b, x); return; }
  coarse_solve(h, h->rhs_level[L - 1], h->x_level[L - 1]);
  for (int l = L - 2; l >= 0; l--) {
    double* xl = l == 0 ? x : h->x_level[l];
    const double* bl = l == 0 ? b : h->rhs_level[l];
    double* rl = h->res_level[l];
    add_c2f(h, l, 1.0, xl, h->x_level[l + 1]);
    level_smooth(&h->lev[l], 1, xl, bl, rl, 0, 0, 0);
  }
}

/* plain W-cycle: the reference's lambda (amg_matrix.cpp:45-104) additionally runs a V-like pass at
 * level 0 whose result it discards; the final result is the same. */
static void w_rec(orc_handle* h, int l, double* x0, const double* b0) {
  const int L = h->n_levels;
  if (l + 1 < L) {
    double* xl = l == 0 ? x0 : h->x_level[l];
    const double* bl = l == 0 ? b0 : h->rhs_level[l];
    double* rl = h->res_level[l];
    const int64_t n = vlen(&h->lev[l]);
    vzero(xl, n);
    vcopy(rl, bl, n);
    level_smooth(&h->lev[l], 0, xl, bl, rl, 1, 1, 1);
    transfer_f2c(h, l, rl, h->rhs_level[l + 1]);
    w_rec(h, l + 1, x0, b0);
    add_c2f(h, l, 1.0, xl, h->x_level[l + 1]);
    level_smooth(&h->lev[l], 1, xl, bl, rl, 0, 1, 0);
    level_smooth(&h->lev[l], 0, xl, bl, rl, 1, 1, 0);
    transfer_f2c(h, l, rl, h->rhs_level[l + 1]);
    w_rec(h, l + 1, x0, b0);
    add_c2f(h, l, 1.0, xl, h->x_level[l + 1]);
    level_smooth(&h->lev[l], 1, xl, bl, rl, 0, 0, 0);
  } else {
    if (L == 1) coarse_solve(h, b0, x0);
    else coarse_solve(h, h->rhs_level[L - 1], h->x_level[L - 1]);
  }
}

static void cycle_bs(orc_handle* h, double* x, const double* b) {
  const int L = h->n_levels;
  for (int l = 0; l + 1 < L; l++) {
    double* xl = l == 0 ? x : h->x_level[l];
    const double* bl = l == 0 ? b : h->rhs_level[l];
    double* rl = h->res_level[l];
    const int64_t n = vlen(&h->lev[l]);
    vzero(xl, n);
    vcopy(rl, bl, n);
    smooth_v_from_level(h, l, xl, bl, rl, 1, 1, 1);
    transfer_f2c(h, l, rl, h->rhs_level[l + 1]);
  }
  if (L == 1) { coarse_solve(h, b, x); return; }
  coarse_solve(h, h->rhs_level[L - 1], h->x_level[L - 1]);
  for (int l = L - 2; l >= 0; l--) {
    double* xl = l == 0 ? x : h->x_level[l];
    const double* bl = l == 0 ? b : h->rhs_level[l];
    double* rl = h->res_level[l];
    add_c2f(h, l, 1.0, xl, h->x_level[l + 1]);
    smooth_v_from_level(h, l, xl, bl, rl, 0, 0, 0);
  }
}

/* AMGMatrix::SmoothVFromLevel, amg_matrix.cpp:310-374 */
static void smooth_v_from_level(orc_handle* h, int start, double* x, const double* b, double* res,
                                int res_updated, int update_res, int x_zero) {
  const int L = h->n_levels;
  level_smooth(&h->lev[start], 0, x, b, res, res_updated, 1, x_zero);
  transfer_f2c(h, start, res, h->rhs_level[start + 1]);
  if (start + 2 < L)
    for (int l = start + 1; l + 1 < L; l++) {
      prep_level(h, l);
      level_smooth(&h->lev[l], 0, h->x_level[l], h->rhs_level[l], h->res_level[l], 1, 1, 1);
      transfer_f2c(h, l, h->res_level[l], h->rhs_level[l + 1]);
    }
  coarse_solve(h, h->rhs_level[L - 1], h->x_level[L - 1]);
  if (start + 2 < L)
    for (int l = L - 2; l > start; l--) {
      add_c2f(h, l, 1.0, h->x_level[l], h->x_level[l + 1]);
      level_smooth(&h->lev[l], 1, h->x_level[l], h->rhs_level[l], h->res_level[l], 0, 0, 0);
    }
  add_c2f(h, start, 1.0, x, h->x_level[start + 1]);
  level_smooth(&h->lev[start], 1, x, b, res, 0, update_res, 0);
}

static void do_cycle(orc_handle* h, double* x, const double* b) {
  /* AMGMatrix::Smooth dispatch, amg_matrix.hpp:37-43 */
  if (h->cycle == ORC_CYCLE_W) w_rec(h, 0, x, b);
  else if (h->cycle == ORC_CYCLE_BS) cycle_bs(h, x, b);
  else cycle_v(h, x, b);
}

/* ---------------------------------------------------------------- public API */

int orc_create(const orc_desc* d, orc_handle** out) {
  if (!d || !out || d->n_levels < 1) return fail("orc_create: bad descriptor");
  orc_handle* h = (orc_handle*)calloc(1, sizeof(orc_handle));
  h->n_levels = d->n_levels;
  h->cycle = d->cycle;
  h->clev_inv = d->clev_inv;
  h->lev = (orc_level*)malloc(sizeof(orc_level) * d->n_levels);
  memcpy(h->lev, d->levels, sizeof(orc_level) * d->n_levels);
  h->x_level = (double**)calloc(d->n_levels, sizeof(double*));
  h->rhs_level = (double**)calloc(d->n_levels, sizeof(double*));
  h->res_level = (double**)calloc(d->n_levels, sizeof(double*));
  h->x_old = (double**)calloc(d->n_levels, sizeof(double*));
  for (int l = 0; l < d->n_levels; l++) {
    const orc_level* L = &h->lev[l];
    /* n_cols > n_rows: trailing ghost columns of a rank-local matrix (stage-wise use only) */
    if (L->A.br != L->A.bc || L->A.br > MAXBS || L->A.n_rows > L->A.n_cols) { orc_destroy(h); return fail("orc_create: bad level matrix"); }
    if (l + 1 < d->n_levels) {
      const orc_level* Lc = &h->lev[l + 1];
      if (L->P.n_rows != L->A.n_rows || L->P.n_cols != Lc->A.n_rows || L->P.br != L->A.br || L->P.bc != Lc->A.br ||
          L->PT.n_rows != L->P.n_cols || L->PT.n_cols != L->P.n_rows || L->PT.br != L->P.bc || L->PT.bc != L->P.br)
        { orc_destroy(h); return fail("orc_create: P / PT shapes do not match the level matrices"); }
    }
    const int64_t n = L->A.n_cols * L->A.bc;
    h->x_old[l] = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    h->x_level[l] = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    h->rhs_level[l] = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    h->res_level[l] = (double*)calloc(n > 0 ? n : 1, sizeof(double));
  }
  /* coarsest-level factor: dense Cholesky of the free-free part (exact solve, amg_pc.cpp:904-922) */
  {
    const orc_level* L = &h->lev[h->n_levels - 1];
    const int bs = L->A.br;
    const int64_t N = vlen(L);
    h->crs_n = N;
    h->crs_idx = (int64_t*)malloc(sizeof(int64_t) * (N > 0 ? N : 1));
    int64_t* pos = (int64_t*)malloc(sizeof(int64_t) * (N > 0 ? N : 1));
    int64_t nf = 0;
    for (int64_t i = 0; i < L->A.n_rows; i++)
      for (int c = 0; c < bs; c++) {
        if (is_free(L, i)) { pos[i * bs + c] = nf; h->crs_idx[nf++] = i * bs + c; } else pos[i * bs + c] = -1;
      }
    h->crs_nf = nf;
    h->crs_tmp = (double*)calloc(nf > 0 ? nf : 1, sizeof(double));
    if (h->clev_inv) {
      if (nf > 8192) { free(pos); orc_destroy(h); return fail("orc_create: coarsest level too large for the dense solve"); }
      double* D = (double*)calloc((size_t)(nf > 0 ? nf * nf : 1), sizeof(double));
      for (int64_t i = 0; i < L->A.n_rows; i++)
        for (int64_t k = L->A.rowptr[i]; k < L->A.rowptr[i + 1]; k++) {
          const int64_t j = L->A.col[k];
          for (int r = 0; r < bs; r++)
            for (int c = 0; c < bs; c++) {
              const int64_t pr = pos[i * bs + r], pc = pos[j * bs + c];
              if (pr >= 0 && pc >= 0) D[pr * nf + pc] = L->A.val[(k * bs + r) * bs + c];
            }
        }
      for (int64_t j = 0; j < nf; j++) {
        double dd = D[j * nf + j];
        for (int64_t k = 0; k < j; k++) dd -= D[j * nf + k] * D[j * nf + k];
        if (!(dd > 0.0)) { free(D); free(pos); orc_destroy(h); return fail("orc_create: coarsest matrix is not SPD on its free dofs"); }
        dd = sqrt(dd);
        D[j * nf + j] = dd;
        for (int64_t i = j + 1; i < nf; i++) {
          double s = D[i * nf + j];
          for (int64_t k = 0; k < j; k++) s -= D[i * nf + k] * D[j * nf + k];
          D[i * nf + j] = s / dd;
        }
      }
      h->crs_L = D;
    }
    free(pos);
  }
  *out = h;
  return 0;
}

/* NUMA placement for the multi-threaded CPU baseline: private copies of the level data, every page first written by
 * the thread that will stream it (same static row schedule as spmv / diag_add).  The caller's arrays usually were
 * written by one thread, i.e. sit on one NUMA node. */
static void* ft_copy_rows(const void* src, const int64_t* rowptr, int64_t n_rows, size_t bytes_per_entry) {
  const int64_t nnz = rowptr ? rowptr[n_rows] : n_rows;
  char* dst = (char*)malloc((size_t)(nnz > 0 ? nnz : 1) * bytes_per_entry);
  if (!dst) return NULL;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i < n_rows; i++) {
    const int64_t a = rowptr ? rowptr[i] : i, b = rowptr ? rowptr[i + 1] : i + 1;
    memcpy(dst + (size_t)a * bytes_per_entry, (const char*)src + (size_t)a * bytes_per_entry, (size_t)(b - a) * bytes_per_entry);
  }
  return dst;
}

static int ft_matrix(orc_handle* h, orc_matrix* M) {
  if (!M->rowptr || M->n_rows == 0) return 0;
  int64_t* rp = (int64_t*)malloc(sizeof(int64_t) * (M->n_rows + 1));
  if (!rp) return 1;
#pragma omp parallel for schedule(static) num_threads(g_threads)
  for (int64_t i = 0; i <= M->n_rows; i++) rp[i] = M->rowptr[i];
  int32_t* col = (int32_t*)ft_copy_rows(M->col, M->rowptr, M->n_rows, sizeof(int32_t));
  double* val = (double*)ft_copy_rows(M->val, M->rowptr, M->n_rows, sizeof(double) * M->br * M->bc);
  if (!col || !val) { free(rp); free(col); free(val); return 1; }
  if (h->n_owned + 3 > ORC_MAX_OWNED) { free(rp); free(col); free(val); return 1; }
  h->owned[h->n_owned++] = rp; h->owned[h->n_owned++] = col; h->owned[h->n_owned++] = val;
  M->rowptr = rp; M->col = col; M->val = val;
  return 0;
}

int orc_first_touch(orc_handle* h) {
  if (!h) return fail("orc_first_touch: null handle");
  for (int l = 0; l < h->n_levels; l++) {
    orc_level* L = &h->lev[l];
    if (ft_matrix(h, &L->A) || ft_matrix(h, &L->P) || ft_matrix(h, &L->PT)) return fail("orc_first_touch: out of memory");
    if (L->dinv) {
      double* d = (double*)ft_copy_rows(L->dinv, NULL, L->A.n_rows, sizeof(double) * L->A.br * L->A.br);
      if (!d || h->n_owned + 1 > ORC_MAX_OWNED) { free(d); return fail("orc_first_touch: out of memory"); }
      h->owned[h->n_owned++] = d;
      L->dinv = d;
    }
  }
  return 0;
}

void orc_destroy(orc_handle* h) {
  if (!h) return;
  for (int i = 0; i < h->n_owned; i++) free(h->owned[i]);
  for (int l = 0; l < h->n_levels; l++) {
    if (h->x_level) free(h->x_level[l]);
    if (h->rhs_level) free(h->rhs_level[l]);
    if (h->res_level) free(h->res_level[l]);
    if (h->x_old) free(h->x_old[l]);
  }
  free(h->x_old);
  free(h->x_level); free(h->rhs_level); free(h->res_level);
  free(h->lev); free(h->crs_idx); free(h->crs_L); free(h->crs_tmp);
  free(h);
}

int orc_apply(orc_handle* h, const double* b, double* x) {
  if (!h) return fail("orc_apply: null handle");
  g_cur = h;
  do_cycle(h, x, b);
  return 0;
}

int orc_apply_add(orc_handle* h, double s, const double* b, double* x) {
  /* AMGMatrix::MultAdd, amg_matrix.cpp:385-389 */
  if (!h) return fail("orc_apply_add: null handle");
  g_cur = h;
  do_cycle(h, h->x_level[0], b);
  const int64_t n = vlen(&h->lev[0]);
  for (int64_t i = 0; i < n; i++) x[i] += s * h->x_level[0][i];
  return 0;
}

int orc_smooth_v_from_level(orc_handle* h, int level, double* x, const double* b, double* res,
                            int res_updated, int update_res, int x_zero) {
  if (!h || level < 0 || level + 1 >= h->n_levels) return fail("orc_smooth_v_from_level: bad level");
  g_cur = h;
  smooth_v_from_level(h, level, x, b, res, res_updated, update_res, x_zero);
  return 0;
}

int orc_smooth(orc_handle* h, int level, int dir, double* x, const double* b, double* res,
               int res_updated, int update_res, int x_zero) {
  if (!h || level < 0 || level >= h->n_levels) return fail("orc_smooth: bad level");
  g_cur = h;
  level_smooth(&h->lev[level], dir, x, b, res, res_updated, update_res, x_zero);
  return 0;
}

int orc_transfer_f2c(orc_handle* h, int level, const double* xf, double* xc) {
  if (!h || level < 0 || level + 1 >= h->n_levels) return fail("orc_transfer_f2c: bad level");
  transfer_f2c(h, level, xf, xc);
  return 0;
}

int orc_add_c2f(orc_handle* h, int level, double fac, double* xf, const double* xc) {
  if (!h || level < 0 || level + 1 >= h->n_levels) return fail("orc_add_c2f: bad level");
  add_c2f(h, level, fac, xf, xc);
  return 0;
}

int orc_coarse_solve(orc_handle* h, const double* rhs, double* x) {
  if (!h) return fail("orc_coarse_solve: null handle");
  coarse_solve(h, rhs, x);
  return 0;
}

int orc_matvec(orc_handle* h, int level, const double* x, double* y) {
  if (!h || level < 0 || level >= h->n_levels) return fail("orc_matvec: bad level");
  spmv(&h->lev[level].A, x, y, 0, 0.0, NULL);
  return 0;
}

static double dot(const double* a, const double* b, int64_t n) {
  double s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s) num_threads(g_threads)
  for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

int orc_pcg(orc_handle* h, const orc_matrix* Ain, const double* b, double* x, double tol, int maxit,
            double* errs, int* iters) {
  /* stand-in for ngsolve.krylovspace.CGSolver as the reference's tests drive it
   * (tests/h1/amg_utils.py:337-363): err = sqrt(|<C r, r>|), stop at err <= tol * err_0 */
  const orc_matrix* A = Ain ? Ain : (h ? &h->lev[0].A : NULL);
  if (!A) return fail("orc_pcg: no matrix");
  g_cur = h;
  const int64_t n = A->n_rows * A->br;
  double* d = (double*)malloc(sizeof(double) * n);
  double* w = (double*)malloc(sizeof(double) * n);
  double* s = (double*)malloc(sizeof(double) * n);
  spmv(A, x, d, 1, 0.0, b);
  if (h) do_cycle(h, w, d); else memcpy(w, d, sizeof(double) * n);
  memcpy(s, w, sizeof(double) * n);
  double wdn = dot(w, d, n);
  double err0 = sqrt(fabs(wdn));
  int it = 0;
  if (errs) errs[0] = err0;
  if (err0 > 0)
    for (it = 1; it <= maxit; it++) {
      spmv(A, s, w, 0, 0.0, NULL);
      const double wd = wdn;
      const double as_s = dot(s, w, n);
      const double alpha = wd / as_s;
      for (int64_t i = 0; i < n; i++) { x[i] += alpha * s[i]; d[i] -= alpha * w[i]; }
      if (h) do_cycle(h, w, d); else memcpy(w, d, sizeof(double) * n);
      wdn = dot(w, d, n);
      const double beta = wdn / wd;
      for (int64_t i = 0; i < n; i++) s[i] = beta * s[i] + w[i];
      const double err = sqrt(fabs(wdn));
      if (errs) errs[it] = err;
      if (err <= tol * err0) break;
    }
  if (it > maxit) it = maxit;
  if (iters) *iters = it;
  free(d); free(w); free(s);
  return 0;
}

int orc_gmres(orc_handle* h, const orc_matrix* Ain, const double* b, double* x, double tol, int maxit, int restart,
              double* errs, int* iters) {
  /* stand-in for ngsolve.krylovspace.GMRes as a caller of the preconditioner: restarted GMRES(m),
   * left-preconditioned (minimises |C (b - A x)| over the Krylov space of C A), MODIFIED Gram-Schmidt, Givens rotations;
   * err_k = |C r_k| (the recurrence value), stop at err_k <= tol * err_0.  Written independently of the device solver
   * (which orthogonalises by classical Gram-Schmidt with one re-orthogonalisation pass): equal histories up to rounding. */
  const orc_matrix* A = Ain ? Ain : (h ? &h->lev[0].A : NULL);
  if (!A || restart < 1) return fail("orc_gmres: bad arguments");
  g_cur = h;
  const int64_t n = A->n_rows * A->br;
  const int m = restart;
  double* V = (double*)malloc(sizeof(double) * n * (size_t)(m + 1));
  double* t = (double*)malloc(sizeof(double) * n);
  double* H = (double*)calloc((size_t)(m + 1) * m, sizeof(double));
  double* cs = (double*)calloc(m, sizeof(double));
  double* sn = (double*)calloc(m, sizeof(double));
  double* g = (double*)calloc(m + 1, sizeof(double));
  double* y = (double*)calloc(m, sizeof(double));
  int it = 0;
  double err0 = -1.0;
  while (it < maxit) {
    spmv(A, x, t, 1, 0.0, b);                                   /* t = b - A x */
    if (h) do_cycle(h, V, t); else memcpy(V, t, sizeof(double) * n);
    const double beta = sqrt(dot(V, V, n));
    if (err0 < 0.0) { err0 = beta; if (errs) errs[0] = err0; }
    if (beta == 0.0 || beta <= tol * err0) break;
    for (int64_t i = 0; i < n; i++) V[i] /= beta;
    for (int i = 0; i <= m; i++) g[i] = 0.0;
    g[0] = beta;
    int k = 0, done = 0;
    for (int j = 0; j < m && it < maxit; j++) {
      it++;
      double* vj = V + (size_t)j * n;
      double* w = V + (size_t)(j + 1) * n;
      spmv(A, vj, t, 0, 0.0, NULL);
      if (h) do_cycle(h, w, t); else memcpy(w, t, sizeof(double) * n);
      for (int i = 0; i <= j; i++) {                            /* modified Gram-Schmidt */
        const double* vi = V + (size_t)i * n;
        const double hij = dot(w, vi, n);
        H[(size_t)i * m + j] = hij;
        for (int64_t q = 0; q < n; q++) w[q] -= hij * vi[q];
      }
      const double hn = sqrt(dot(w, w, n));
      if (hn > 0.0) for (int64_t q = 0; q < n; q++) w[q] /= hn;
      double hj1 = hn;
      for (int i = 0; i < j; i++) {
        const double a = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = a;
      }
      const double den = hypot(H[(size_t)j * m + j], hj1);
      cs[j] = den > 0 ? H[(size_t)j * m + j] / den : 1.0;
      sn[j] = den > 0 ? hj1 / den : 0.0;
      H[(size_t)j * m + j] = den;
      if (j + 1 < m) H[(size_t)(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      k = j + 1;
      const double err = fabs(g[j + 1]);
      if (errs) errs[it] = err;
      if (err <= tol * err0 || hn == 0.0) { done = 1; break; }
    }
    for (int i = k - 1; i >= 0; i--) {
      double s = g[i];
      for (int q = i + 1; q < k; q++) s -= H[(size_t)i * m + q] * y[q];
      y[i] = H[(size_t)i * m + i] != 0.0 ? s / H[(size_t)i * m + i] : 0.0;
    }
    for (int i = 0; i < k; i++) { const double* vi = V + (size_t)i * n; for (int64_t q = 0; q < n; q++) x[q] += y[i] * vi[q]; }
    if (done) break;
  }
  if (iters) *iters = it;
  free(V); free(t); free(H); free(cs); free(sn); free(g); free(y);
  return 0;
}
