/* gss4.c -- CPU oracle for GSS4 (Gauss-Seidel on a subset of the rows with a compressed copy of the matrix).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference holds no vectors for this class either.
 *
 * Restated reference code: src/base/smoothers/gssmoother.cpp
 *   :456-507  GSS4::SetUp            xdofs, res_subset, compressed matrix cA (columns outside res_subset are dropped --
 *                                    res_subset holds every column of a subset row, so nothing is dropped in effect)
 *   :417-438, :511-527  diagonal     dinv_i = inverse (or pseudo-inverse) of repl_diag[xdofs[i]] or cA(i, i): the caller
 *                                    supplies the inverted blocks (numpy does the dense algebra in the tests)
 *   :443-453  iterate_rows           ascending, or descending when backwards
 *   :531-539  MultAdd                x(xdofs[i]) += s dinv_i b(xdofs[i])
 *   :543-561  SmoothRESInternal      w = -dinv_i res(xdofs[i]); res += cA_i:^T w; x(xdofs[i]) -= w
 *   :565-583  SmoothRHSInternal      r = b(xdofs[i]) - cA_i: x; x(xdofs[i]) += dinv_i r
 * Extension for the parity tests: an optional visiting order of the compressed rows (colour-major = the order of the GPU
 * kernels); NULL = the reference's order. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define G4_MAXBS 6

struct orc_gss4 {
  int bs;
  int64_t n, n_cols, m;
  int32_t* xdofs;        /* [m] */
  int64_t* rowptr;       /* [m+1] compressed matrix */
  int32_t* col;
  double* val;
  double* dinv;          /* [m*bs*bs] */
  int32_t* order;        /* [m] or NULL */
};

int orc_gss4_create(const orc_matrix* A, const uint8_t* subset, const double* dinv_full, orc_gss4** out) {
  if (!A || !out || !dinv_full || A->br != A->bc || A->br > G4_MAXBS) return 1;
  orc_gss4* g = (orc_gss4*)calloc(1, sizeof(orc_gss4));
  const int bs = A->br, bb = bs * bs;
  const int64_t n = A->n_rows;
  g->bs = bs; g->n = n; g->n_cols = A->n_cols;
  /* xdofs / res_subset (:458-476) */
  uint8_t* res_subset = (uint8_t*)calloc((size_t)A->n_cols, 1);
  int64_t cntx = 0;
  for (int64_t k = 0; k < n; k++)
    if (!subset || subset[k]) {
      cntx++;
      res_subset[k] = 1;
      for (int64_t p = A->rowptr[k]; p < A->rowptr[k + 1]; p++) res_subset[A->col[p]] = 1;
    }
  g->m = cntx;
  g->xdofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cntx ? cntx : 1));
  cntx = 0;
  for (int64_t k = 0; k < n; k++) if (!subset || subset[k]) g->xdofs[cntx++] = (int32_t)k;
  /* compress A (:477-497) */
  g->rowptr = (int64_t*)calloc((size_t)g->m + 1, sizeof(int64_t));
  for (int64_t i = 0; i < g->m; i++) {
    const int64_t k = g->xdofs[i];
    int64_t c = 0;
    for (int64_t p = A->rowptr[k]; p < A->rowptr[k + 1]; p++) if (res_subset[A->col[p]]) c++;
    g->rowptr[i + 1] = g->rowptr[i] + c;
  }
  const int64_t nnz = g->rowptr[g->m];
  g->col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
  g->val = (double*)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1) * bb);
  for (int64_t i = 0; i < g->m; i++) {
    const int64_t k = g->xdofs[i];
    int64_t c = g->rowptr[i];
    for (int64_t p = A->rowptr[k]; p < A->rowptr[k + 1]; p++)
      if (res_subset[A->col[p]]) {
        g->col[c] = A->col[p];
        memcpy(g->val + c * bb, A->val + p * bb, sizeof(double) * bb);
        c++;
      }
  }
  free(res_subset);
  g->dinv = (double*)malloc(sizeof(double) * (size_t)(g->m ? g->m : 1) * bb);
  for (int64_t i = 0; i < g->m; i++) memcpy(g->dinv + i * bb, dinv_full + (int64_t)g->xdofs[i] * bb, sizeof(double) * bb);
  *out = g;
  return 0;
}

void orc_gss4_destroy(orc_gss4* g) {
  if (!g) return;
  free(g->xdofs); free(g->rowptr); free(g->col); free(g->val); free(g->dinv); free(g->order); free(g);
}

int64_t orc_gss4_rows(const orc_gss4* g) { return g->m; }
int64_t orc_gss4_nnz(const orc_gss4* g) { return g->rowptr[g->m]; }

int orc_gss4_set_order(orc_gss4* g, const int32_t* order, int64_t len) {
  free(g->order); g->order = NULL;
  if (!order) return 0;
  if (len != g->m) return 1;
  g->order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(len ? len : 1));
  memcpy(g->order, order, sizeof(int32_t) * (size_t)len);
  return 0;
}

static inline int64_t visit(const orc_gss4* g, int64_t q, int backwards) {
  const int64_t p = backwards ? g->m - 1 - q : q;          /* iterate_rows (:443-453) */
  return g->order ? g->order[p] : p;
}

int orc_gss4_smooth(const orc_gss4* g, int backwards, double* x, const double* b) {
  const int bs = g->bs, bb = bs * bs;
  for (int64_t q = 0; q < g->m; q++) {
    const int64_t i = visit(g, q, backwards);
    const int64_t k = g->xdofs[i];
    double r[G4_MAXBS];
    for (int a = 0; a < bs; a++) r[a] = b[k * bs + a];
    for (int64_t p = g->rowptr[i]; p < g->rowptr[i + 1]; p++) {
      const double* v = g->val + p * bb;
      const double* xv = x + (int64_t)g->col[p] * bs;
      for (int a = 0; a < bs; a++) for (int c = 0; c < bs; c++) r[a] -= v[a * bs + c] * xv[c];
    }
    const double* d = g->dinv + i * bb;
    for (int a = 0; a < bs; a++) {
      double u = 0;
      for (int c = 0; c < bs; c++) u += d[a * bs + c] * r[c];
      x[k * bs + a] += u;
    }
  }
  return 0;
}

int orc_gss4_smooth_res(const orc_gss4* g, int backwards, double* x, double* res) {
  const int bs = g->bs, bb = bs * bs;
  for (int64_t q = 0; q < g->m; q++) {
    const int64_t i = visit(g, q, backwards);
    const int64_t k = g->xdofs[i];
    const double* d = g->dinv + i * bb;
    double w[G4_MAXBS];
    for (int a = 0; a < bs; a++) {
      double u = 0;
      for (int c = 0; c < bs; c++) u += d[a * bs + c] * res[k * bs + c];
      w[a] = -u;
    }
    for (int64_t p = g->rowptr[i]; p < g->rowptr[i + 1]; p++) {       /* AddRowTransToVector */
      const double* v = g->val + p * bb;
      double* rj = res + (int64_t)g->col[p] * bs;
      for (int c = 0; c < bs; c++) {
        double u = 0;
        for (int a = 0; a < bs; a++) u += v[a * bs + c] * w[a];
        rj[c] += u;
      }
    }
    for (int a = 0; a < bs; a++) x[k * bs + a] -= w[a];
  }
  return 0;
}

int orc_gss4_mult_add(const orc_gss4* g, double s, const double* b, double* x) {
  const int bs = g->bs, bb = bs * bs;
  for (int64_t i = 0; i < g->m; i++) {
    const int64_t k = g->xdofs[i];
    const double* d = g->dinv + i * bb;
    for (int a = 0; a < bs; a++) {
      double u = 0;
      for (int c = 0; c < bs; c++) u += d[a * bs + c] * b[k * bs + c];
      x[k * bs + a] += s * u;
    }
  }
  return 0;
}
