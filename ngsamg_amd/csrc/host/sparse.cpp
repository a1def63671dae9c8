// Sparse (block-)matrix kernels for the host setup: transpose, SpGEMM, Galerkin projection.
// Semantics follow the reference's setup helpers (not its code):
//   TransposeSPMImpl   src/base/linalg/utils_sparseMM.cpp:54-93   (explicit P^T, sorted columns)
//   MatMultABImpl      src/base/linalg/utils_sparseMM.cpp:107-238 (row-wise product, sorted columns)
//   RestrictMatrix     src/base/linalg/utils_sparseMM.hpp:93-109  ((P^T A) P)
// Implementation: Gustavson row products with one dense marker per OpenMP thread.
#include "bcsr.hpp"
#include <omp.h>
#include <algorithm>
#include <numeric>
#include <cstring>

namespace amgh {

BCSR transpose(const BCSR& A) {
  BCSR T;
  T.n_rows = A.n_cols; T.n_cols = A.n_rows; T.br = A.bc; T.bc = A.br;
  const int64_t nnz = A.nnz();
  T.rowptr.assign(T.n_rows + 1, 0);
  for (int64_t k = 0; k < nnz; k++) T.rowptr[A.col[k] + 1]++;
  for (int64_t i = 0; i < T.n_rows; i++) T.rowptr[i + 1] += T.rowptr[i];
  T.col.resize(nnz);
  T.val.resize(nnz * A.bsz());
  std::vector<int64_t> pos(T.rowptr.begin(), T.rowptr.end() - 1);
  const int br = A.br, bc = A.bc, bs = A.bsz();
  // rows visited in ascending order => columns of T ascend within each row
  for (int64_t i = 0; i < A.n_rows; i++)
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      int64_t p = pos[A.col[k]]++;
      T.col[p] = (int32_t)i;
      const double* src = &A.val[k * bs];
      double* dst = &T.val[p * bs];
      if (bs == 1) dst[0] = src[0];
      else for (int r = 0; r < br; r++) for (int c = 0; c < bc; c++) dst[c * br + r] = src[r * bc + c];
    }
  return T;
}

template <bool SCALAR>
static BCSR matmul_impl(const BCSR& A, const BCSR& B) {
  if (A.n_cols != B.n_rows || A.bc != B.br) throw Error("matmul: dimension mismatch");
  BCSR C;
  C.n_rows = A.n_rows; C.n_cols = B.n_cols; C.br = A.br; C.bc = B.bc;
  const int br = A.br, bk = A.bc, bc = B.bc, cbs = br * bc;
  const int64_t n = A.n_rows;
  const int nt = std::min(omp_get_max_threads(), 32);   // each thread owns dense markers of size n_cols: bound the memory
  std::vector<std::vector<int32_t>> tcol(nt);
  std::vector<std::vector<double>> tval(nt);
  std::vector<int64_t> rowlen(n, 0);
  std::vector<int64_t> tstart(nt + 1, 0);   // first row of each thread's contiguous chunk
  for (int t = 0; t <= nt; t++) tstart[t] = (n * t) / nt;
#pragma omp parallel num_threads(nt)
  {
    const int t = omp_get_thread_num();
    std::vector<int32_t> marker(B.n_cols, -1);   // marker[j] = slot of column j in the current row, tagged by row via `owner`
    std::vector<int64_t> owner(B.n_cols, -1);
    std::vector<int32_t> cols;
    std::vector<double> acc;
    std::vector<int32_t> order;
    auto& oc = tcol[t];
    auto& ov = tval[t];
    for (int64_t i = tstart[t]; i < tstart[t + 1]; i++) {
      cols.clear(); acc.clear();
      for (int64_t ka = A.rowptr[i]; ka < A.rowptr[i + 1]; ka++) {
        const int32_t k = A.col[ka];
        const double* a = &A.val[ka * br * bk];
        for (int64_t kb = B.rowptr[k]; kb < B.rowptr[k + 1]; kb++) {
          const int32_t j = B.col[kb];
          const double* b = &B.val[kb * bk * bc];
          int32_t slot;
          if (owner[j] != i) { owner[j] = i; slot = (int32_t)cols.size(); marker[j] = slot; cols.push_back(j); acc.resize(acc.size() + cbs, 0.0); }
          else slot = marker[j];
          double* c = &acc[(size_t)slot * cbs];
          if (SCALAR) c[0] += a[0] * b[0];
          else
            for (int r = 0; r < br; r++)
              for (int q = 0; q < bk; q++) {
                const double arq = a[r * bk + q];
                for (int s = 0; s < bc; s++) c[r * bc + s] += arq * b[q * bc + s];
              }
        }
      }
      const int len = (int)cols.size();
      order.resize(len);
      std::iota(order.begin(), order.end(), 0);
      std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cols[x] < cols[y]; });
      for (int q = 0; q < len; q++) {
        oc.push_back(cols[order[q]]);
        const double* c = &acc[(size_t)order[q] * cbs];
        ov.insert(ov.end(), c, c + cbs);
      }
      rowlen[i] = len;
    }
  }
  C.rowptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; i++) C.rowptr[i + 1] = C.rowptr[i] + rowlen[i];
  C.col.resize(C.rowptr[n]);
  C.val.resize(C.rowptr[n] * cbs);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
  for (int t = 0; t < nt; t++) {
    int64_t off = C.rowptr[tstart[t]];
    if (!tcol[t].empty()) {
      std::memcpy(&C.col[off], tcol[t].data(), tcol[t].size() * sizeof(int32_t));
      std::memcpy(&C.val[off * cbs], tval[t].data(), tval[t].size() * sizeof(double));
    }
  }
  return C;
}

BCSR matmul(const BCSR& A, const BCSR& B) {
  if (A.br == 1 && A.bc == 1 && B.bc == 1) return matmul_impl<true>(A, B);
  return matmul_impl<false>(A, B);
}

GalerkinHook& galerkin_hook() {
  static GalerkinHook h;
  return h;
}

namespace {
// layout of amgh_matrix (include/amgh.h)
struct MatView { int64_t n_rows, n_cols; int32_t br, bc; const int64_t* rowptr; const int32_t* col; const double* val; };
MatView view_of(const BCSR& M) { return MatView{M.n_rows, M.n_cols, M.br, M.bc, M.rowptr.data(), M.col.data(), M.val.data()}; }
}  // namespace

BCSR restrict_matrix(const BCSR& PT, const BCSR& A, const BCSR& P) {
  const GalerkinHook& hk = galerkin_hook();
  if (hk.run && hk.fetch && A.n_rows >= hk.min_rows) {
    const MatView vpt = view_of(PT), va = view_of(A), vp = view_of(P);
    void* res = nullptr;
    int64_t nr = 0, nnz = 0;
    const int rc = hk.run(&vpt, &va, &vp, &res, &nr, &nnz);
    if (rc == 0) {
      BCSR C;
      C.n_rows = PT.n_rows; C.n_cols = P.n_cols; C.br = PT.br; C.bc = P.bc;
      if (nr != C.n_rows) { hk.fetch(res, nullptr, nullptr, nullptr); throw Error("galerkin hook: result has the wrong number of rows"); }
      C.rowptr.resize(nr + 1);
      C.col.resize(nnz);
      C.val.resize(nnz * C.bsz());
      if (hk.fetch(res, C.rowptr.data(), C.col.data(), C.val.data()) != 0) throw Error("galerkin hook: fetch failed");
      if (C.rowptr.back() != nnz) throw Error("galerkin hook: inconsistent result");
      return C;
    }
    if (rc != 2) throw Error("galerkin hook failed");
  }
  BCSR PTA = matmul(PT, A);
  return matmul(PTA, P);
}

void spmv(const BCSR& A, const double* x, double* y) {
  const int br = A.br, bc = A.bc;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.n_rows; i++) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      const double* a = &A.val[k * br * bc];
      const double* xv = &x[(int64_t)A.col[k] * bc];
      for (int r = 0; r < br; r++) for (int c = 0; c < bc; c++) s[r] += a[r * bc + c] * xv[c];
    }
    for (int r = 0; r < br; r++) y[i * br + r] = s[r];
  }
}

}  // namespace amgh
