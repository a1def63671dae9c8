// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the AMG apply path.
//
// Every kernel here is HBM-bound (arithmetic intensity <= 0.25 flop/B): the design rules are
// coalesced 16-B/lane streaming of the matrix, gathers of x served by the XCD-local L2, and enough
// independent loads in flight per wave.  No MFMA: SpMV with a single right-hand side is not a contraction.
//
// Matrix formats on the device (built once at amgx_create, see amgx.hip):
//   SELL-64-pair ("sliced ELL"): slices of 64 consecutive rows = one wavefront; inside a slice the entries
//       are stored column-major in PAIRS, so lane r reads a double2 (values j, j+1 of its row) and an int2
//       (their columns): 16 B + 8 B per lane per step, perfectly coalesced (1 KiB + 512 B per wave
//       instruction).  An odd trailing column is stored as singles.  Used for matrices with near-uniform
//       row length (FEM level matrices, prolongations).
//   CSR-vector: G = 2..64 lanes cooperate on one row, wave-level shuffle reduction.  Used for irregular /
//       long rows (P^T, coarse level matrices) and for all block (3x3, 6x6, 3x6, 6x3) matrices.
//   colour-major SELL copy of A for multicolour Gauss-Seidel (slices never cross a colour).
//
// The blockIdx -> row-block mapping is XCD-aware: hardware deals workgroups round-robin over the 8 XCDs
// (MI355X_MICROARCH.md "Workgroup dispatch"), so logical block = f(blockIdx) is chosen such that each XCD
// walks one contiguous eighth of the rows and the x-gathers of neighbouring rows hit that XCD's own L2.
// This is a speed-only assumption; any placement gives the same result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amgx {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

enum Epilogue : int {
  EP_MULT = 0,   // y = A x
  EP_RES = 1,    // y = b - A x
  EP_AXPY = 2,   // y = yin + s * A x        (yin may alias y)
  EP_JAC = 3     // y = yin + omega * dinv * (b - A yin)   with x == yin gathered, y != yin
};

struct EpArgs {
  const double* b;
  const double* yin;
  const double* dinv;
  double s;      // AXPY factor or Jacobi omega
};

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ---------------------------------------------------------------------------------------------------
// scalar epilogue
template <int EP>
__device__ __forceinline__ void store_scalar(int64_t row, double acc, double* y, const EpArgs& ep) {
  if (EP == EP_MULT) y[row] = acc;
  else if (EP == EP_RES) y[row] = ep.b[row] - acc;
  else if (EP == EP_AXPY) y[row] = ep.yin[row] + ep.s * acc;
  else y[row] = ep.yin[row] + ep.s * (ep.dinv[row] * (ep.b[row] - acc));
}

// ---------------------------------------------------------------------------------------------------
// SELL-64-pair, scalar, one thread per row, one wave per slice
template <int EP>
__global__ __launch_bounds__(BLOCK) void sell_spmv_kernel(int64_t n_rows, int n_slices,
                                                          const int64_t* __restrict__ slice_ptr,
                                                          const int32_t* __restrict__ cols,
                                                          const double* __restrict__ vals,
                                                          const double* __restrict__ x, double* y, EpArgs ep) {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = lb * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)((slice_ptr[s + 1] - base) >> 6);
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
  const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base);
  const int np = w >> 1;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
  for (int p = 0; p < np; ++p) {
    const double2 v = v2[p * WAVE + lane];
    const int2 c = c2[p * WAVE + lane];
    acc0 += v.x * x[c.x];
    acc1 += v.y * x[c.y];
  }
  if (w & 1) {
    const int64_t o = base + (int64_t)(w - 1) * WAVE + lane;
    acc0 += vals[o] * x[cols[o]];
  }
  const int64_t row = (int64_t)s * WAVE + lane;
  if (row < n_rows) store_scalar<EP>(row, acc0 + acc1, y, ep);
}

// ---------------------------------------------------------------------------------------------------
// CSR-vector, scalar: G lanes per row
template <int G, int EP>
__global__ __launch_bounds__(BLOCK) void csrvec_spmv_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ cols,
                                                            const double* __restrict__ vals,
                                                            const double* __restrict__ x, double* y, EpArgs ep) {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t t = (int64_t)lb * BLOCK + threadIdx.x;
  const int64_t row = t / G;
  const int sub = (int)(t % G);
  double acc = 0.0;
  if (row < n_rows) {
    const int e = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < e; k += G) acc += vals[k] * x[cols[k]];
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
  if (row < n_rows && sub == 0) store_scalar<EP>(row, acc, y, ep);
}

// ---------------------------------------------------------------------------------------------------
// CSR-vector, block BR x BC: G lanes per block row, each lane owns whole blocks
template <int BR, int BC, int G, int EP>
__global__ __launch_bounds__(BLOCK) void bcsrvec_spmv_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ cols,
                                                             const double* __restrict__ vals,
                                                             const double* __restrict__ x, double* y, EpArgs ep) {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t t = (int64_t)lb * BLOCK + threadIdx.x;
  const int64_t row = t / G;
  const int sub = (int)(t % G);
  double acc[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) acc[r] = 0.0;
  if (row < n_rows) {
    const int e = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < e; k += G) {
      const double* __restrict__ a = vals + (int64_t)k * (BR * BC);
      const double* __restrict__ xv = x + (int64_t)cols[k] * BC;
      double xr[BC];
#pragma unroll
      for (int c = 0; c < BC; ++c) xr[c] = xv[c];
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int c = 0; c < BC; ++c) acc[r] += a[r * BC + c] * xr[c];
    }
  }
#pragma unroll
  for (int r = 0; r < BR; ++r)
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, G);
  if (row < n_rows && sub == 0) {
    double* yo = y + row * BR;
    if (EP == EP_MULT) {
#pragma unroll
      for (int r = 0; r < BR; ++r) yo[r] = acc[r];
    } else if (EP == EP_RES) {
#pragma unroll
      for (int r = 0; r < BR; ++r) yo[r] = ep.b[row * BR + r] - acc[r];
    } else if (EP == EP_AXPY) {
#pragma unroll
      for (int r = 0; r < BR; ++r) yo[r] = ep.yin[row * BR + r] + ep.s * acc[r];
    } else {
      // Jacobi: only square blocks reach this branch (BR == BC)
      double tt[BR];
#pragma unroll
      for (int r = 0; r < BR; ++r) tt[r] = ep.b[row * BR + r] - acc[r];
      const double* __restrict__ d = ep.dinv + row * (BR * BR);
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        double u = 0.0;
#pragma unroll
        for (int c = 0; c < BR; ++c) u += d[r * BR + c] * tt[c];
        yo[r] = ep.yin[row * BR + r] + ep.s * u;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// multicolour Gauss-Seidel, scalar: one colour per launch, colour-major SELL copy of A.
//   x_k += dinv_k * (b_k - A_k: x)        (RHS form, reference gssmoother.cpp:209-212)
// Rows of one colour have no mutual couplings, so the in-place update is race-free.
__global__ __launch_bounds__(BLOCK) void gs_color_kernel(int slice_begin, int slice_end,
                                                         const int64_t* __restrict__ slice_ptr,
                                                         const int32_t* __restrict__ cols,
                                                         const double* __restrict__ vals,
                                                         const int32_t* __restrict__ rowid,
                                                         const double* __restrict__ dinv,
                                                         const double* __restrict__ b, double* x) {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = slice_begin + lb * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (s >= slice_end) return;
  const int row = rowid[(int64_t)s * WAVE + lane];
  if (row < 0) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)((slice_ptr[s + 1] - base) >> 6);
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
  const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base);
  const int np = w >> 1;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
  for (int p = 0; p < np; ++p) {
    const double2 v = v2[p * WAVE + lane];
    const int2 c = c2[p * WAVE + lane];
    acc0 += v.x * x[c.x];
    acc1 += v.y * x[c.y];
  }
  if (w & 1) {
    const int64_t o = base + (int64_t)(w - 1) * WAVE + lane;
    acc0 += vals[o] * x[cols[o]];
  }
  x[row] += dinv[row] * (b[row] - (acc0 + acc1));
}

// multicolour Gauss-Seidel, block BS x BS: CSR rows through a colour-major row list, G lanes per row
template <int BS, int G>
__global__ __launch_bounds__(BLOCK) void bgs_color_kernel(int list_begin, int list_end,
                                                          const int32_t* __restrict__ rowlist,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ cols,
                                                          const double* __restrict__ vals,
                                                          const double* __restrict__ dinv,
                                                          const double* __restrict__ b, double* x) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t q = list_begin + t / G;
  const int sub = (int)(t % G);
  const bool active = q < list_end;
  const int row = active ? rowlist[q] : 0;
  double acc[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) acc[r] = 0.0;
  if (active) {
    const int e = rowptr[row + 1];
    for (int k = rowptr[row] + sub; k < e; k += G) {
      const double* __restrict__ a = vals + (int64_t)k * (BS * BS);
      const double* xv = x + (int64_t)cols[k] * BS;
      double xr[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) xr[c] = xv[c];
#pragma unroll
      for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) acc[r] += a[r * BS + c] * xr[c];
    }
  }
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, G);
  if (active && sub == 0) {
    double tt[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) tt[r] = b[(int64_t)row * BS + r] - acc[r];
    const double* __restrict__ d = dinv + (int64_t)row * (BS * BS);
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      double u = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) u += d[r * BS + c] * tt[c];
      x[(int64_t)row * BS + r] += u;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// x = (ADD ? x : 0) + omega * Dinv * v     (DiagonalMatrix<TM> apply, base_smoother.cpp:61-74)
template <int BS, bool ADD>
__global__ __launch_bounds__(BLOCK) void diag_apply_kernel(int64_t n, const double* __restrict__ dinv,
                                                           const double* __restrict__ v, double* x, double omega) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  if (BS == 1) {
    const double u = omega * (dinv[i] * v[i]);
    x[i] = ADD ? x[i] + u : u;
  } else {
    double vv[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) vv[c] = v[i * BS + c];
    const double* __restrict__ d = dinv + i * (BS * BS);
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      double u = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) u += d[r * BS + c] * vv[c];
      u *= omega;
      x[i * BS + r] = ADD ? x[i * BS + r] + u : u;
    }
  }
}

// y += s * x
__global__ __launch_bounds__(BLOCK) void axpy_kernel(int64_t n, double s, const double* __restrict__ x, double* y) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) y[i] += s * x[i];
}

// deterministic pseudo-random fill in [-1, 1) (measurement hook: kernels are timed on non-trivial data)
__global__ __launch_bounds__(BLOCK) void fill_kernel(int64_t n, uint64_t seed, double* v) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  v[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// dense y = M x, one wave per row (coarsest-level inverse, n <= a few hundred)
__global__ __launch_bounds__(BLOCK) void dense_gemv_kernel(int n, const double* __restrict__ M,
                                                           const double* __restrict__ x, double* y) {
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & (WAVE - 1);
  if (row >= n) return;
  double acc = 0.0;
  for (int c = lane; c < n; c += WAVE) acc += M[(int64_t)row * n + c] * x[c];
#pragma unroll
  for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
  if (lane == 0) y[row] = acc;
}

}  // namespace amgx
