// Host-side block-CSR container and sparse kernels used by the AMG *setup* (cold path).
//
// Layout mirrors NGSolve's SparseMatrix<Mat<H,W>> as the reference uses it
// (reference SURVEY App. B; wire format src/base/distributed/mpiwrap_extension.hpp:133-156):
//   rowptr  : int64 [n_rows+1]          (reference: Array<size_t> firsti)
//   col     : int32 [nnz], ascending per row
//   val     : double [nnz * br * bc], one row-major br x bc block per stored entry
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <string>
#include <stdexcept>

namespace amgh {

struct BCSR {
  int64_t n_rows = 0, n_cols = 0;
  int br = 1, bc = 1;
  std::vector<int64_t> rowptr;
  std::vector<int32_t> col;
  std::vector<double> val;
  int64_t nnz() const { return rowptr.empty() ? 0 : rowptr.back(); }
  int bsz() const { return br * bc; }
};

// the caller's arrays as they are (C ABI entry points that only read the matrix: no copy of 10^8 entries)
struct CsrView {
  int64_t n_rows = 0, n_cols = 0;
  int br = 1, bc = 1;
  const int64_t* rowptr = nullptr;
  const int32_t* col = nullptr;
  const double* val = nullptr;
};

// C = A^T with sorted columns (reference TransposeSPMImpl, src/base/linalg/utils_sparseMM.cpp:54-93)
BCSR transpose(const BCSR& A);

// C = A * B with sorted columns (reference MatMultABImpl, utils_sparseMM.cpp:107-238)
BCSR matmul(const BCSR& A, const BCSR& B);

// Galerkin product in the reference's order (P^T A) P  (utils_sparseMM.hpp:93-109)
BCSR restrict_matrix(const BCSR& PT, const BCSR& A, const BCSR& P);
// accelerator hook of restrict_matrix (amgh.h: amgh_set_galerkin_hook); the matrix arguments are amgh_matrix views
struct GalerkinHook {
  int (*run)(const void* PT, const void* A, const void* P, void** result, int64_t* n_rows, int64_t* nnz) = nullptr;
  int (*fetch)(void* result, int64_t* rowptr, int32_t* col, double* val) = nullptr;
  int64_t min_rows = 0;
};
GalerkinHook& galerkin_hook();

// y = A x  (y overwritten); x, y are AoS block vectors
void spmv(const BCSR& A, const double* x, double* y);

// dense helpers (dense.cpp) ---------------------------------------------------------------
// in-place inverse of an n x n row-major matrix (Gauss-Jordan with partial pivoting); returns false if singular
bool dense_inverse(double* a, int n);
// symmetric eigen-decomposition (cyclic Jacobi); a is destroyed, evals[n], evecs row-major columns = vectors
void sym_eig(double* a, int n, double* evals, double* evecs);
// pseudo inverse with the reference's rule (utils_denseLA.hpp:1460-1570): try the direct inverse,
// else eigen-decomposition dropping eigenvalues <= max(rel_tol*mean, abs_tol)
void pseudo_inverse_try_normal(double* a, int n);
void pseudo_inverse_with_tol(double* a, int n);
// Cholesky-based SPD inverse; falls back to pseudo inverse if not SPD. returns true if Cholesky succeeded
bool spd_inverse(double* a, int n);

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

}  // namespace amgh
