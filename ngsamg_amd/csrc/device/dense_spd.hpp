// Dense inverse of a (large) coarsest-level matrix ON THE DEVICE (included by amgx.hip).
//
// Reference: BaseAMGPC::CoarseLevelInv (src/base/precond/amg_pc.cpp:843-928) always inverts the coarsest matrix on its
// free dofs (sparse Cholesky / master inverse) and AMGMatrix::SmoothV applies it (amg_matrix.cpp:228-233).  The host setup
// of this build hands over a dense inverse for up to 4096 unknowns; beyond that (coarsening that stalls, max_levels
// reached, 6 dofs per vertex) the inverse is formed here, where it is cheap: a blocked Gauss-Jordan sweep without
// pivoting (the matrix is SPD on its free dofs, every Schur complement stays SPD) in 64 x 64 tiles whose trailing update
//      T(i,j) -= T(i,k) T(k,j)          for all tiles i != k, j != k          (2 n^3 flops in total)
// is a genuine contraction and runs on the matrix cores (v_mfma_f64_16x16x4_f64, 64 x 64 x 64 per workgroup from LDS) --
// the one place on this path where north_star's "MFMA where it is a real contraction" applies.  One application is then
// the same dense GEMV as for small coarse levels (n^2 * 8 bytes streamed: 0.5 GB / ~90 us at n = 8192).
//   step k:  P = T(k,k)^-1 (one workgroup, LDS);  T(k,j) <- P T(k,j);  trailing update;  T(i,k) <- -T(i,k) P;  T(k,k) <- P
#pragma once

namespace amgx {

constexpr int GJ_T = 64;                 // tile edge
typedef double gj_double4 __attribute__((ext_vector_type(4)));

// dense image of the block-CSR matrix restricted to free block rows; identity on non-free and padding dofs
__global__ __launch_bounds__(BLOCK) void gj_zero_kernel(int64_t len, double* __restrict__ D) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t < len) D[t] = 0.0;
}
__global__ __launch_bounds__(BLOCK) void gj_scatter_kernel(int64_t n_rows, int bs, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const double* __restrict__ val, const uint8_t* __restrict__ free_rows, int64_t ld,
                                                           double* __restrict__ D) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rows) return;
  if (free_rows && !free_rows[i]) return;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int64_t j = col[k];
    if (j >= n_rows || (free_rows && !free_rows[j])) continue;
    for (int r = 0; r < bs; ++r)
      for (int c = 0; c < bs; ++c) D[(i * bs + r) * ld + j * bs + c] = val[((int64_t)k * bs + r) * bs + c];
  }
}
// diagonal entries of the rows that take no part (non-free dofs, padding): value v (1 before the inversion, 0 after it)
__global__ __launch_bounds__(BLOCK) void gj_fix_diag_kernel(int64_t n_pad, int64_t n, int bs, const uint8_t* __restrict__ free_rows, int64_t ld,
                                                            double v, double* __restrict__ D) {
  const int64_t d = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (d >= n_pad) return;
  const bool out = d >= n || (free_rows && !free_rows[d / bs]);
  if (out) D[d * ld + d] = v;
}

// P = T(k,k)^-1 by unpivoted Gauss-Jordan in LDS; status[0] = min over the steps of pivot / largest diagonal entry seen
// (<= 0 or tiny: the matrix is not positive definite on its free dofs)
__global__ __launch_bounds__(256) void gj_pivot_kernel(int k, int64_t ld, const double* __restrict__ D, double* __restrict__ P, double* __restrict__ status) {
  __shared__ double a[GJ_T][GJ_T + 1];
  __shared__ double colp[GJ_T], rowp[GJ_T];
  __shared__ double dmax_s, minratio_s;
  const double* T = D + ((int64_t)k * GJ_T) * ld + (int64_t)k * GJ_T;
  for (int e = threadIdx.x; e < GJ_T * GJ_T; e += 256) a[e / GJ_T][e % GJ_T] = T[(int64_t)(e / GJ_T) * ld + (e % GJ_T)];
  if (threadIdx.x == 0) { dmax_s = 0.0; minratio_s = 1e300; }
  __syncthreads();
  if (threadIdx.x == 0) { double m = 0.0; for (int i = 0; i < GJ_T; ++i) m = fmax(m, fabs(a[i][i])); dmax_s = m; }
  __syncthreads();
  for (int p = 0; p < GJ_T; ++p) {
    // row p and column p of the current state are staged first, so that the in-place update below reads nothing another
    // thread writes in the same step
    if (threadIdx.x < GJ_T) { colp[threadIdx.x] = a[threadIdx.x][p]; rowp[threadIdx.x] = a[p][threadIdx.x]; }
    __syncthreads();
    const double piv = rowp[p];
    if (threadIdx.x == 0) minratio_s = fmin(minratio_s, dmax_s > 0.0 ? piv / dmax_s : -1.0);
    const double ip = piv != 0.0 ? 1.0 / piv : 0.0;
    // Gauss-Jordan step: a'[p][p] = 1/piv, a'[p][j] = a[p][j]/piv, a'[i][p] = -a[i][p]/piv, a'[i][j] = a[i][j] - a[i][p] a[p][j]/piv
    for (int e = threadIdx.x; e < GJ_T * GJ_T; e += 256) {
      const int i = e / GJ_T, j = e % GJ_T;
      double v;
      if (i == p) v = (j == p) ? ip : rowp[j] * ip;
      else if (j == p) v = -colp[i] * ip;
      else v = a[i][j] - colp[i] * (rowp[j] * ip);
      a[i][j] = v;
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < GJ_T * GJ_T; e += 256) P[e] = a[e / GJ_T][e % GJ_T];
  if (threadIdx.x == 0) status[0] = fmin(status[0], minratio_s);
}

// C(64 x 64) = alpha * A(64 x 64) B(64 x 64) + beta * C0 from LDS tiles, on the matrix cores.
// wave w owns rows [16 w, 16 w + 16); operand layout of v_mfma_f64_16x16x4_f64 (tools/mfma_lab.hip): A[m = lane & 15][k = lane >> 4],
// B[k = lane >> 4][n = lane & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15]
__device__ __forceinline__ void gj_tile_mma(const double (*As)[GJ_T + 1], const double (*Bs)[GJ_T + 1], gj_double4 acc[4]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = gj_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k4 = 0; k4 < GJ_T; k4 += 4) {
    const double av = As[16 * w + m][k4 + kq];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[k4 + kq][16 * ct + m], acc[ct], 0, 0, 0);
  }
}
__device__ __forceinline__ void gj_load_tile(double (*S)[GJ_T + 1], const double* __restrict__ T, int64_t ld) {
  for (int e = threadIdx.x; e < GJ_T * GJ_T; e += 256) S[e / GJ_T][e % GJ_T] = T[(int64_t)(e / GJ_T) * ld + (e % GJ_T)];
}
// mode 0: row panel    T(k,j) <- P T(k,j)               (grid: tiles j, j != k skipped inside)
// mode 1: column panel T(i,k) <- -T(i,k) P              (grid: tiles i)
// mode 2: trailing     T(i,j) <- T(i,j) - T(i,k) T(k,j) (grid: nt x nt)
__global__ __launch_bounds__(256) void gj_tile_kernel(int mode, int k, int nt, int64_t ld, double* __restrict__ D, const double* __restrict__ P) {
  __shared__ double As[GJ_T][GJ_T + 1];
  __shared__ double Bs[GJ_T][GJ_T + 1];
  int ti, tj;
  if (mode == 0) { ti = k; tj = blockIdx.x; if (tj == k) return; }
  else if (mode == 1) { ti = blockIdx.x; tj = k; if (ti == k) return; }
  else { ti = blockIdx.y; tj = blockIdx.x; if (ti == k || tj == k) return; }
  double* C = D + ((int64_t)ti * GJ_T) * ld + (int64_t)tj * GJ_T;
  if (mode == 0) { gj_load_tile(As, P, GJ_T); gj_load_tile(Bs, C, ld); }
  else if (mode == 1) { gj_load_tile(As, C, ld); gj_load_tile(Bs, P, GJ_T); }
  else { gj_load_tile(As, D + ((int64_t)ti * GJ_T) * ld + (int64_t)k * GJ_T, ld); gj_load_tile(Bs, D + ((int64_t)k * GJ_T) * ld + (int64_t)tj * GJ_T, ld); }
  __syncthreads();
  gj_double4 acc[4];
  gj_tile_mma(As, Bs, acc);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * w + (lane >> 4) + 4 * r, colq = 16 * ct + (lane & 15);
      double* c = C + (int64_t)row * ld + colq;
      if (mode == 0) *c = acc[ct][r];
      else if (mode == 1) *c = -acc[ct][r];
      else *c -= acc[ct][r];
    }
}
__global__ __launch_bounds__(256) void gj_store_pivot_kernel(int k, int64_t ld, double* __restrict__ D, const double* __restrict__ P) {
  double* T = D + ((int64_t)k * GJ_T) * ld + (int64_t)k * GJ_T;
  for (int e = threadIdx.x; e < GJ_T * GJ_T; e += 256) T[(int64_t)(e / GJ_T) * ld + (e % GJ_T)] = P[e];
}

// In-place inverse of the SPD matrix D (n_pad x n_pad, n_pad a multiple of 64, row-major with stride ld) on `stream`.
// Returns the smallest pivot ratio met (<= 1e-14: not positive definite enough to trust the unpivoted sweep).
static double dense_spd_inverse(double* D, int64_t n_pad, int64_t ld, hipStream_t stream) {
  if (n_pad % GJ_T) throw Err("dense_spd_inverse: size must be a multiple of 64");
  const int nt = (int)(n_pad / GJ_T);
  DevBuf<double> P, status;
  P.alloc(GJ_T * GJ_T);
  status.alloc(1);
  const double big = 1e300;
  HIPCHK(hipMemcpyAsync(status.p, &big, sizeof(double), hipMemcpyHostToDevice, stream));
  for (int k = 0; k < nt; ++k) {
    hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, stream, k, ld, D, P.p, status.p);
    if (nt > 1) {
      hipLaunchKernelGGL(gj_tile_kernel, dim3(nt), dim3(256), 0, stream, 0, k, nt, ld, D, P.p);
      hipLaunchKernelGGL(gj_tile_kernel, dim3(nt, nt), dim3(256), 0, stream, 2, k, nt, ld, D, P.p);
      hipLaunchKernelGGL(gj_tile_kernel, dim3(nt), dim3(256), 0, stream, 1, k, nt, ld, D, P.p);
    }
    hipLaunchKernelGGL(gj_store_pivot_kernel, dim3(1), dim3(256), 0, stream, k, ld, D, P.p);
    if ((k & 31) == 31) HIPCHK(hipStreamSynchronize(stream));
  }
  HIPCHK(hipGetLastError());
  double st = 0.0;
  HIPCHK(hipMemcpyAsync(&st, status.p, sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  return st;
}

}  // namespace amgx
