// Small dense linear algebra for smoother diagonals and the coarsest-level inverse (setup, host).
// Rules follow the reference:
//   dinv_k = inv(A_kk), or CalcPseudoInverseTryNormal(A_kk) when `pinv`
//   (src/base/smoothers/gssmoother.cpp:143-170; src/base/utils/utils_denseLA.hpp:1460-1570;
//    tolerances RelZeroTol = 1e-12, AbsZeroTol = 1e-20, utils_denseLA.hpp:93-115).
#include "bcsr.hpp"
#include <cmath>
#include <algorithm>
#include <vector>

namespace amgh {

static constexpr double REL_ZERO_TOL = 1e-12;
static constexpr double ABS_ZERO_TOL = 1e-20;

// Gauss-Jordan with partial (row) pivoting. With check_tol the singularity test of the reference's
// TryDirectInverse is applied: pivot < max(AbsZeroTol * rest, RelZeroTol * max_diag) -> give up.
static bool gauss_jordan(double* a, int n, bool check_tol) {
  if (n == 0) return false;
  double eps = 0;
  for (int j = 0; j < n; j++) eps = std::max(eps, a[j * n + j]);
  eps *= REL_ZERO_TOL;
  std::vector<double> w(a, a + n * n), inv(n * n, 0.0);
  for (int j = 0; j < n; j++) inv[j * n + j] = 1.0;
  for (int j = 0; j < n; j++) {
    int r = j;
    double maxval = std::fabs(w[j * n + j]);
    for (int i = j + 1; i < n; i++)
      if (std::fabs(w[i * n + j]) > maxval) { maxval = std::fabs(w[i * n + j]); r = i; }
    double rest = 0;
    for (int i = j + 1; i < n; i++) rest += std::fabs(w[r * n + i]);
    if (check_tol) {
      if (maxval < std::max(ABS_ZERO_TOL * rest, eps)) return false;
    } else if (maxval == 0.0) return false;
    if (r != j)
      for (int k = 0; k < n; k++) { std::swap(w[j * n + k], w[r * n + k]); std::swap(inv[j * n + k], inv[r * n + k]); }
    double hr = 1.0 / w[j * n + j];
    for (int k = 0; k < n; k++) { w[j * n + k] *= hr; inv[j * n + k] *= hr; }
    for (int i = 0; i < n; i++) {
      if (i == j) continue;
      double f = w[i * n + j];
      if (f == 0.0) continue;
      for (int k = 0; k < n; k++) { w[i * n + k] -= f * w[j * n + k]; inv[i * n + k] -= f * inv[j * n + k]; }
    }
  }
  std::copy(inv.begin(), inv.end(), a);
  return true;
}

bool dense_inverse(double* a, int n) { return gauss_jordan(a, n, false); }

void sym_eig(double* a, int n, double* evals, double* evecs) {
  // cyclic Jacobi; evecs row-major with eigenvectors as COLUMNS
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) evecs[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, dg = 0;
    for (int i = 0; i < n; i++) { dg += a[i * n + i] * a[i * n + i]; for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j]; }
    if (off <= 1e-32 * (dg + off) || off == 0.0) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = a[p * n + q];
        if (apq == 0.0) continue;
        double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = evecs[k * n + p], vkq = evecs[k * n + q];
          evecs[k * n + p] = c * vkp - s * vkq; evecs[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) evals[i] = a[i * n + i];
}

static void pseudo_inverse_eig(double* a, int n) {
  std::vector<double> w(a, a + n * n), ev(n), V(n * n);
  // symmetrise (the blocks are symmetric up to rounding)
  for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) { double m = 0.5 * (w[i * n + j] + w[j * n + i]); w[i * n + j] = w[j * n + i] = m; }
  sym_eig(w.data(), n, ev.data(), V.data());
  double tol = 0;
  for (int i = 0; i < n; i++) tol += ev[i];
  tol = std::max(REL_ZERO_TOL * tol / n, ABS_ZERO_TOL);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
    double s = 0;
    for (int k = 0; k < n; k++) if (ev[k] > tol) s += V[i * n + k] * V[j * n + k] / ev[k];
    a[i * n + j] = s;
  }
}

// CalcPseudoInverseWithTol (utils_denseLA.hpp:1474-1519): always the eigenvalue form, no attempt at a direct inverse
void pseudo_inverse_with_tol(double* a, int n) {
  if (n == 1) { a[0] = (std::fabs(a[0]) > ABS_ZERO_TOL) ? 1.0 / a[0] : 0.0; return; }
  pseudo_inverse_eig(a, n);
}

void pseudo_inverse_try_normal(double* a, int n) {
  if (n == 1) { a[0] = (std::fabs(a[0]) > ABS_ZERO_TOL) ? 1.0 / a[0] : 0.0; return; }
  std::vector<double> keep(a, a + n * n);
  if (gauss_jordan(a, n, true)) return;
  std::copy(keep.begin(), keep.end(), a);
  pseudo_inverse_eig(a, n);
}

bool spd_inverse(double* a, int n) {
  // Cholesky A = L L^T, then A^-1 = L^-T L^-1
  std::vector<double> L(a, a + n * n);
  bool ok = true;
  for (int j = 0; j < n && ok; j++) {
    double d = L[j * n + j];
    for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) { ok = false; break; }
    d = std::sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = L[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  if (!ok) { pseudo_inverse_eig(a, n); return false; }
  // invert L (lower) in place into Li
  std::vector<double> Li(n * n, 0.0);
  for (int j = 0; j < n; j++) {
    Li[j * n + j] = 1.0 / L[j * n + j];
    for (int i = j + 1; i < n; i++) {
      double s = 0;
      for (int k = j; k < i; k++) s -= L[i * n + k] * Li[k * n + j];
      Li[i * n + j] = s / L[i * n + i];
    }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = i; k < n; k++) s += Li[k * n + i] * Li[k * n + j];
      a[i * n + j] = a[j * n + i] = s;
    }
  return true;
}

}  // namespace amgh
