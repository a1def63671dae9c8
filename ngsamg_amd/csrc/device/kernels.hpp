// Hand-written HIP kernels (gfx950 / CDNA4, wave64) for the AMG apply path.
//
// Every kernel here is HBM-bound in the roofline sense (arithmetic intensity <= 0.25 flop/B).  Design
// rules, in the order they turned out to matter on MI355X: coalesced 16-B/lane streaming of the matrix;
// few, SMALL batches of loads in flight per wave with the next batch requested before the current one is consumed
// (rows are short: a wave lives for only a few memory round trips, and bursts of gathers stall the CU's address path);
// own-row epilogue operands requested ahead of the row product; gathers of x served by L2 / Infinity Cache.
// No MFMA: SpMV with a single right-hand side is not a contraction.  Measured (tools/mfma_lab.hip, profiles/r02/mfma_ab.txt):
// the 6x6 block-row product through v_mfma_f64_16x16x4_f64 (two block rows per instruction, 2 of 16 result columns used)
// is 1.67x SLOWER than the VALU form at the cfg-5 shape and bit-identical.
//
// Matrix formats on the device (built once at amgx_create, see amgx.hip):
//   SELL-64-pair ("sliced ELL"), G lanes per row: slices of 64/G consecutive rows = one wavefront; inside a slice the
//       entries are stored column-major in PAIRS, so a lane reads a double2 (two values of its row) and their packed
//       16-bit (or 32-bit) columns per step: perfectly coalesced 1 KiB + 256 B per wave instruction.  An odd trailing
//       column is stored as singles.  Variants: diagonal-first rows, omega*Dinv in the diagonal slot of the
//       pre-smoothing image, length-sorted 512-row windows for ragged rows (sell_win_spmv_kernel).
//   BSELL: the same idea for square bs x bs block matrices, one lane per scalar row of a block row.
//   CSR-vector / block CSR row-per-lane: irregular or long rows (P^T of block levels, rectangular transfer blocks).
//   colour-major SELL copy of A for multicolour Gauss-Seidel (slices never cross a colour); aggregate blocks with dense
//       inverses for block Gauss-Seidel (bgs_block_kernel).
//
// blockIdx -> row-block mapping: measured on MI355X (profiles/r01/spmv_lab_round*.log), an XCD-aware remap
// (each XCD walking one contiguous eighth of the rows) is 5-6 % SLOWER than the natural round-robin order
// for these streaming kernels: the x-gathers are served by the Infinity Cache either way, and eight distant
// HBM streams are worse than one front advancing through the matrix.  The remap helper is kept (it is a
// speed-only choice, any placement gives the same result) but the kernels use the natural order.
// The matrix streams (values, column indices) are read exactly once per launch and are loaded
// non-temporally so that they do not evict the vectors from L2 / Infinity Cache (+6...13 % measured).
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
  EP_JAC = 3,    // y = yin + omega * dinv * (b - A yin)   with x == yin gathered, y != yin
  EP_PRE = 4,    // Jacobi pre-smoothing from x = 0 in one pass over the column-scaled image A' = A * omega*Dinv:
                 //   y = b - A' b  (= b - A x with x = omega*Dinv*b),  y2 = omega * dinv * b  (= x)
  EP_CRES = 5    // y = c .* v - A x  with c = ep.dinv, v = ep.b (own-row entries): residual right after a block-hybrid
                 //   Gauss-Seidel sweep from zero, A = the part of the matrix the sweep did not use (see gsb_sweep_kernel)
};

struct EpArgs {
  const double* b;
  const double* yin;
  const double* dinv;
  double s;      // AXPY factor or Jacobi omega
  double* y2;    // second output of EP_PRE; block EP_JAC: optional output of the residual b - A x (nullptr: none)
  int nt;        // bit 0: stream the epilogue's own-row operands / results non-temporally (read / written once per cycle)
                 // bit 1 (EP_PRE): second output is z = x + omega*Dinv*r, the pre-smoothed iterate smoothed once more
                 //        without a coarse correction (cycle with the post-smoothing folded into the prolongation)
                 // bit 2: request the own-row operands BEFORE the row product (their latency then overlaps the matrix
                 //        stream instead of extending every wave's life by one dependent memory round trip)
};
constexpr int EPF_NT = 1, EPF_FOLD = 2, EPF_HOIST = 4;

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

template <class T>
__device__ __forceinline__ T ld_nt(const T* p) { return __builtin_nontemporal_load(p); }

// ---------------------------------------------------------------------------------------------------
// scalar epilogue
// have_xd: xd is the value of the gathered vector at the own row (EP_JAC: yin[row], EP_PRE: b[row]) and need not be loaded
template <int EP>
__device__ __forceinline__ void store_scalar(int64_t row, double acc, double* y, const EpArgs& ep, bool have_xd = false, double xd = 0.0) {
  if (EP == EP_MULT) y[row] = acc;
  else if (EP == EP_RES) y[row] = ep.b[row] - acc;
  else if (EP == EP_AXPY) y[row] = ep.yin[row] + ep.s * acc;
  else if (EP == EP_CRES) y[row] = ep.dinv[row] * ep.b[row] - acc;
  else if (EP == EP_JAC) {
    if (ep.nt & EPF_NT) __builtin_nontemporal_store((have_xd ? xd : ep.yin[row]) + ep.s * (ld_nt(ep.dinv + row) * (ld_nt(ep.b + row) - acc)), y + row);
    else y[row] = (have_xd ? xd : ep.yin[row]) + ep.s * (ep.dinv[row] * (ep.b[row] - acc));
  } else {
    const double bi = have_xd ? xd : ep.b[row];
    const double r = bi - acc;
    y[row] = r;
    const double di = (ep.nt & EPF_NT) ? ld_nt(ep.dinv + row) : ep.dinv[row];
    double xi = ep.s * (di * bi);
    if (ep.nt & EPF_FOLD) xi += ep.s * (di * r);
    if (ep.nt & EPF_NT) __builtin_nontemporal_store(xi, ep.y2 + row);
    else ep.y2[row] = xi;
  }
}

// own-row operands of an epilogue, loaded ahead of the row product (EPF_HOIST)
struct EpOps { double b, d, yin; };
template <int EP>
__device__ __forceinline__ EpOps ep_operands(int64_t row, const EpArgs& ep, bool have_xd, bool skip_d = false) {
  EpOps o{0.0, 0.0, 0.0};
  const bool nt = ep.nt & EPF_NT;
  if (EP == EP_RES) o.b = ep.b[row];
  else if (EP == EP_CRES) { o.b = ep.b[row]; o.d = ep.dinv[row]; }
  else if (EP == EP_AXPY) o.yin = ep.yin[row];
  else if (EP == EP_JAC) {
    o.d = nt ? ld_nt(ep.dinv + row) : ep.dinv[row];
    o.b = nt ? ld_nt(ep.b + row) : ep.b[row];
    if (!have_xd) o.yin = ep.yin[row];
  } else if (EP == EP_PRE) {
    if (!skip_d) o.d = nt ? ld_nt(ep.dinv + row) : ep.dinv[row];
    if (!have_xd) o.b = ep.b[row];
  }
  return o;
}
template <int EP>
__device__ __forceinline__ void store_scalar_ops(int64_t row, double acc, double* y, const EpArgs& ep, const EpOps& o, bool have_xd, double xd) {
  const bool nt = ep.nt & EPF_NT;
  if (EP == EP_MULT) y[row] = acc;
  else if (EP == EP_RES) y[row] = o.b - acc;
  else if (EP == EP_CRES) y[row] = o.d * o.b - acc;
  else if (EP == EP_AXPY) y[row] = o.yin + ep.s * acc;
  else if (EP == EP_JAC) {
    const double v = (have_xd ? xd : o.yin) + ep.s * (o.d * (o.b - acc));
    if (nt) __builtin_nontemporal_store(v, y + row); else y[row] = v;
  } else {
    const double bi = have_xd ? xd : o.b;
    const double r = bi - acc;
    y[row] = r;
    double xi = ep.s * (o.d * bi);
    if (ep.nt & EPF_FOLD) xi += ep.s * (o.d * r);
    if (nt) __builtin_nontemporal_store(xi, ep.y2 + row); else ep.y2[row] = xi;
  }
}

// ---------------------------------------------------------------------------------------------------
// SELL-64-pair view.  Column indices come in two encodings, chosen per slice at build time:
//   32-bit:  col32[o]                                         (bit 0 of slice_ptr[s] clear)
//   16-bit:  col = (rowrel ? row : 0) + cbase[column] + col16[o]   (bit 0 set)
// where `column` = (base >> 6) + j numbers the 64-entry columns of all slices.  On FEM matrices the offset
// col - row is (nearly) the same for all 64 rows of a slice column, so the 16-bit form almost always applies
// and the index stream shrinks from 4 to ~2 bytes per entry (-13 % kernel time measured).
struct SellMat {
  const int64_t* slice_ptr;   // [n_slices+1] element offset (multiple of 64) | encoding flag in bit 0
  const int32_t* col32;
  const uint16_t* col16;
  const int32_t* cbase;
  const double* val;
  int rowrel;
  int diag_first;             // G == 1 only: entry 0 of every row is its diagonal, so the gathered x[row] comes for free
  int wdiag;                  // pre-smoothing image A' only (implies diag_first): the diagonal slot holds omega*Dinv_i instead of
                              //   A'_ii (which is omega for a free row, 0 otherwise), so the epilogue needs no dinv stream
  int xcd;                    // 1: workgroup b of the launch works on unit xcd_remap(b): every XCD (= every L2) walks ONE contiguous
                              //   eighth of the rows.  For long-row levels whose gathered vector does not fit an L2 next to the rows in flight
                              //   (the 1.24 M x 52 level of cfg 2: 10 MB of x, 40 % of the rows in flight at once); level 0 keeps 0 (xcd_remap)
};
__device__ __forceinline__ int sell_unit(const SellMat& M) { return M.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; }

// SELL row product, software-pipelined in batches of K pair-steps: the matrix loads (values + packed indices) of batch
// i+1 are requested BEFORE the gathers of batch i are consumed, so a wave always has a matrix batch in flight while it
// waits for gathered x values.  Batch size: measured (profiles/r01/unroll_ab.txt) -- the rows of this path are short
// (7 pair-steps for A at cfg 2, 1..8 for Q) and small batches win: K = 2, 3 beat K = 4 by 5 % and K = 8 by 7 %; a
// `#pragma unroll 4` loop was worst (3 of 7 steps in a one-at-a-time remainder loop).
#ifndef SELL_BATCH
#define SELL_BATCH 2
#endif
#ifndef SELL_PIPELINE
#define SELL_PIPELINE 1
#endif

template <int K>
struct SellRegs {
  double v0[K], v1[K];
  uint32_t ca[K], cb2[K];    // C16: ca = packed 16-bit pair; 32-bit: ca, cb2 = the two columns
};

template <int K, bool C16>
__device__ __forceinline__ void sell_load(SellRegs<K>& R, const double* __restrict__ vb, const void* __restrict__ cpv, int p, int lane) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int q = p + k;
    R.v0[k] = ld_nt(vb + (q * WAVE + lane) * 2);
    R.v1[k] = ld_nt(vb + (q * WAVE + lane) * 2 + 1);
    if (C16) R.ca[k] = ld_nt(static_cast<const uint32_t*>(cpv) + q * WAVE + lane);
    else {
      const int32_t* __restrict__ cp = static_cast<const int32_t*>(cpv);
      R.ca[k] = (uint32_t)ld_nt(cp + (q * WAVE + lane) * 2);
      R.cb2[k] = (uint32_t)ld_nt(cp + (q * WAVE + lane) * 2 + 1);
    }
  }
}

template <int K, bool C16>
__device__ __forceinline__ void sell_consume(const SellRegs<K>& R, const int32_t* __restrict__ cb, int r0, int p,
                                             const double* __restrict__ x, double& acc0, double& acc1, double* xd) {
  double x0[K], x1[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int q = p + k;
    const int c0 = C16 ? r0 + cb[2 * q] + (int)(R.ca[k] & 0xffffu) : (int)R.ca[k];
    const int c1 = C16 ? r0 + cb[2 * q + 1] + (int)(R.ca[k] >> 16) : (int)R.cb2[k];
    x0[k] = x[c0];
    x1[k] = x[c1];
  }
  if (p == 0) { xd[0] = x0[0]; xd[1] = R.v0[0]; }
#pragma unroll
  for (int k = 0; k < K; ++k) { acc0 += R.v0[k] * x0[k]; acc1 += R.v1[k] * x1[k]; }
}

// remainder batch of REM < K steps; `last` (may be null) is a full batch whose gathers are consumed after the
// remainder's matrix loads went out
template <int K, int REM, bool C16>
__device__ __forceinline__ void sell_tail(int rem, const SellRegs<K>* last, int p_last, const double* __restrict__ vb,
                                          const void* __restrict__ cpv, const int32_t* __restrict__ cb, int r0, int p, int lane,
                                          const double* __restrict__ x, double& acc0, double& acc1, double* xd) {
  if (rem == REM) {
    SellRegs<REM> R;
    sell_load<REM, C16>(R, vb, cpv, p, lane);
    if (last) sell_consume<K, C16>(*last, cb, r0, p_last, x, acc0, acc1, xd);
    sell_consume<REM, C16>(R, cb, r0, p, x, acc0, acc1, xd);
  } else if constexpr (REM > 1) sell_tail<K, REM - 1, C16>(rem, last, p_last, vb, cpv, cb, r0, p, lane, x, acc0, acc1, xd);
}

template <bool C16, int K = SELL_BATCH>
__device__ __forceinline__ void sell_pairs(int np, const double* __restrict__ vb, const void* __restrict__ cpv, const int32_t* __restrict__ cb,
                                           int r0, int lane, const double* __restrict__ x, double& acc0, double& acc1, double* xd) {
  const int nfull = np / K, rem = np - nfull * K;
#if SELL_PIPELINE
  if (nfull > 0) {
    SellRegs<K> A;
    sell_load<K, C16>(A, vb, cpv, 0, lane);
    for (int b = 1; b < nfull; ++b) {
      SellRegs<K> B;
      sell_load<K, C16>(B, vb, cpv, b * K, lane);
      sell_consume<K, C16>(A, cb, r0, (b - 1) * K, x, acc0, acc1, xd);
      A = B;
    }
    if (K > 1 && rem) sell_tail<K, (K > 1 ? K - 1 : 1), C16>(rem, &A, (nfull - 1) * K, vb, cpv, cb, r0, nfull * K, lane, x, acc0, acc1, xd);
    else sell_consume<K, C16>(A, cb, r0, (nfull - 1) * K, x, acc0, acc1, xd);
  } else if (K > 1 && rem) sell_tail<K, (K > 1 ? K - 1 : 1), C16>(rem, nullptr, 0, vb, cpv, cb, r0, 0, lane, x, acc0, acc1, xd);
#else
  for (int b = 0; b < nfull; ++b) {
    SellRegs<K> A;
    sell_load<K, C16>(A, vb, cpv, b * K, lane);
    sell_consume<K, C16>(A, cb, r0, b * K, x, acc0, acc1, xd);
  }
  if (K > 1 && rem) sell_tail<K, (K > 1 ? K - 1 : 1), C16>(rem, nullptr, 0, vb, cpv, cb, r0, nfull * K, lane, x, acc0, acc1, xd);
#endif
}

// dot product of SELL row (slice s, lane) with x; row = global row id of this lane (for row-relative columns)
// xd (optional, 2 doubles): receives the x value gathered for entry 0 of this lane's row and the matrix value of that
// entry (meaningful when M.diag_first)
// (xd is always a real local of the caller: a conditionally-null pointer kept the pair in scratch memory)
// K = pair-steps per batch of the software pipeline (see SELL_BATCH)
// (sp0, sp1 = slice_ptr[s], slice_ptr[s + 1]: callers with a long prologue load them first, see sell_pre_restrict_kernel)
// x32 (optional): the vector slices in the 32-bit encoding gather from (local-window images: 16-bit slices index the LDS window,
// 32-bit slices -- chunks whose window would not fit -- carry global columns)
template <int K = SELL_BATCH>
__device__ __forceinline__ double sell_row_dot_sp(const SellMat& M, int64_t sp0, int64_t sp1, int lane, int row, const double* x, double* xd,
                                                  const double* x32 = nullptr) {
  const int64_t base = sp0 & ~(int64_t)63;
  const int w = (int)(((sp1 & ~(int64_t)63) - base) >> 6);
  const int np = w >> 1;
  const double* __restrict__ vb = M.val + base;
  double acc0 = 0.0, acc1 = 0.0;
  const bool c16 = sp0 & 1;
  const int32_t* __restrict__ cb = M.cbase + (base >> 6);
  const int r0 = (c16 && M.rowrel) ? row : 0;
  // odd trailing column: its matrix loads go out first, its gather comes last
  double vs = 0.0;
  int cs = 0;
  if (w & 1) {
    const int64_t o = (int64_t)(w - 1) * WAVE + lane;
    vs = ld_nt(vb + o);
    cs = c16 ? (int)ld_nt(M.col16 + base + o) : ld_nt(M.col32 + base + o);
  }
  const double* __restrict__ xw32 = x32 ? x32 : x;
  if (c16) sell_pairs<true, K>(np, vb, M.col16 + base, cb, r0, lane, x, acc0, acc1, xd);
  else sell_pairs<false, K>(np, vb, M.col32 + base, nullptr, 0, lane, xw32, acc0, acc1, xd);
  if (w & 1) {
    const double x0 = c16 ? x[r0 + cb[w - 1] + cs] : xw32[cs];
    if (np == 0) { xd[0] = x0; xd[1] = vs; }
    acc0 += vs * x0;
  }
  return acc0 + acc1;
}
template <int K = SELL_BATCH>
__device__ __forceinline__ double sell_row_dot(const SellMat& M, int s, int lane, int row, const double* x, double* xd) {
  return sell_row_dot_sp<K>(M, M.slice_ptr[s], M.slice_ptr[s + 1], lane, row, x, xd);
}
template <int K = SELL_BATCH>
__device__ __forceinline__ double sell_row_dot(const SellMat& M, int s, int lane, int row, const double* x) {
  double xd[2];
  return sell_row_dot<K>(M, s, lane, row, x, xd);
}

// SELL-64-pair, scalar, G lanes per row (G = 1: one thread per row), one wave per slice of 64/G rows
template <int G, int EP>
__global__ __launch_bounds__(BLOCK) void sell_spmv_kernel(int64_t n_rows, int slice0, int n_slices, SellMat M,
                                                          const double* __restrict__ x, double* y, EpArgs ep) {
  const int lane = threadIdx.x & (WAVE - 1);
  // the slice index is wave-uniform: tell the compiler, so slice pointers and column bases use scalar loads
  // (slice0 .. n_slices: the launch covers a range of slices -- interior / boundary rows of a rank-partitioned level)
  const int s = __builtin_amdgcn_readfirstlane(slice0 + sell_unit(M) * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= n_slices) return;
  const int row = s * (WAVE / G) + lane / G;
  double xd[2] = {0.0, 0.0};
  const bool use_xd = G == 1 && (EP == EP_JAC || EP == EP_PRE) && M.diag_first;
  const bool wdiag = EP == EP_PRE && use_xd && M.wdiag;
  const bool writer = (lane % G) == 0 && row < n_rows;
  const bool hoist = (ep.nt & EPF_HOIST) && EP != EP_MULT;
  EpOps ops{0.0, 0.0, 0.0};
  if (hoist && writer) ops = ep_operands<EP>(row, ep, use_xd, wdiag);
  // levels with several lanes per row are small (latency rather than bandwidth), but deeper batches do not pay there
  // either: same-box A/B of the whole cycle, batch 2 / 3 / 4 / 6 on those levels: 0.692 / 0.693-0.716 / 0.687-0.692 / 0.736-0.761 ms
#ifndef SELL_BATCH_MULTI
#define SELL_BATCH_MULTI SELL_BATCH
#endif
  double acc = sell_row_dot<(G > 1 ? SELL_BATCH_MULTI : SELL_BATCH)>(M, s, lane, row, x, xd);
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
  if (writer) {
    if (!hoist) ops = ep_operands<EP>(row, ep, use_xd, wdiag);
    if (wdiag) {
      // diagonal slot = wd = omega*Dinv_i: take it out of the product and put the true A'_ii b_i (omega*b_i or 0) back
      const double wd = xd[1], bi = xd[0];
      acc = acc - wd * bi + (wd != 0.0 ? ep.s * bi : 0.0);
      ops.d = wd / ep.s;                       // store_scalar_ops forms ep.s * (d * .)
    }
    store_scalar_ops<EP>(row, acc, y, ep, ops, use_xd, xd[0]);
  }
}

// Windowed SELL, one thread per row: a workgroup owns WB consecutive rows, stored in order of decreasing length
// (host: upload_matrix), so that the 64 rows of a slice have (nearly) the same length.  The row sums go through LDS
// back to natural order, so the epilogue's own-row reads and the store stay coalesced.
#ifndef SELL_WIN_SIZE
#define SELL_WIN_SIZE 512      // A/B (profiles/r01/unroll_ab.txt): 256 is 2 % slower, 1024 ties
#endif
constexpr int SELL_WIN = SELL_WIN_SIZE;
template <int WB, int EP>
__global__ __launch_bounds__(WB) void sell_win_spmv_kernel(int64_t n_rows, int win0, SellMat M, const uint16_t* __restrict__ rowloc,
                                                           const double* __restrict__ x, double* y, EpArgs ep) {
  __shared__ double buf[WB];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wb = win0 + sell_unit(M);          // window index (win0: first window of the launch, see sell_spmv_kernel)
  const int s = __builtin_amdgcn_readfirstlane(wb * (WB / WAVE) + (threadIdx.x >> 6));
  const int64_t slot = (int64_t)s * WAVE + lane;
  const int64_t row = (int64_t)wb * WB + threadIdx.x;
  const bool hoist = (ep.nt & EPF_HOIST) && EP != EP_MULT;
  EpOps ops{0.0, 0.0, 0.0};
  if (hoist && row < n_rows) ops = ep_operands<EP>(row, ep, false);
  if (slot < n_rows) buf[rowloc[slot]] = sell_row_dot(M, s, lane, 0, x);
  __syncthreads();
  if (row < n_rows) {
    if (!hoist) ops = ep_operands<EP>(row, ep, false);
    store_scalar_ops<EP>(row, buf[threadIdx.x], y, ep, ops, false, 0.0);
  }
}

// ---------------------------------------------------------------------------------------------------
// CSR-vector, scalar: G lanes per row
template <int G, int EP>
__global__ __launch_bounds__(BLOCK) void csrvec_spmv_kernel(int64_t row0, int64_t n_rows, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ cols,
                                                            const double* __restrict__ vals,
                                                            const double* __restrict__ x, double* y, EpArgs ep) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t row = row0 + t / G;            // rows [row0, n_rows)
  const int sub = (int)(t % G);
  double acc = 0.0;
  if (row < n_rows) {
    const int e = rowptr[row + 1];
    // plain (cached) loads on purpose: a 128-B line of a row is touched in two consecutive steps of the lane
    // group, so the non-temporal policy of the SELL kernels doubles the HBM traffic here (measured: 2x slower)
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
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
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
    } else if (EP == EP_JAC) {
      // Jacobi: only square blocks reach this branch (BR == BC)
      double tt[BR];
#pragma unroll
      for (int r = 0; r < BR; ++r) tt[r] = ep.b[row * BR + r] - acc[r];
      if (ep.y2) {                       // block EP_JAC with a second output: the residual b - A x itself
#pragma unroll
        for (int r = 0; r < BR; ++r) ep.y2[row * BR + r] = tt[r];
      }
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
// BSELL: sliced ELL for square BS x BS block matrices, ONE LANE PER SCALAR ROW.  A slice holds RB = 64/BS block rows
// (63 / 60 / 64 active lanes for BS = 3 / 6 / 2); per block step k the values are stored structure-of-arrays,
//   [k][column pair cp][lane 0..63][2]   (+ [k][last column][lane] when BS is odd),
// so every wave instruction streams one aligned 1 KiB (or 512 B) chunk, exactly like the scalar SELL kernel; the block
// column index is stored once per (k, block row).  Used when the slice padding stays small (FEM fine levels); irregular
// coarse levels keep the CSR row-per-lane kernels.
struct BSellMat {
  const int64_t* slice_ptr;   // [n_slices+1] cumulative block steps
  const int32_t* col;         // [steps * RB]
  const double* val;          // [steps * BS * 64]
  int xmode;                  // row product: 0 = plain loop, BS 8-byte gathers per lane, 1 = 16-byte gathers, 2 = one load + lane exchange,
                              // 3 / 4 / 5 = column indices one batch of 2 / 4 / 3 steps ahead (bsell_row_dot_ahead)
};

#ifndef BSELL_UNROLL
#define BSELL_UNROLL 2
#endif
// one block step of a BSELL row product: acc += (row r of the BS x BS block at vk) . x_block.
// V2 (even block sizes, x 16-byte aligned: c * BS * 8 is then a multiple of 16): the x block is read with 16-byte loads.  Counters at
// cfg 3 (profiles/r04/pmc_cfg3_gs_sq.csv): the 6x6 kernels spend 46 ... 78 % of their wave cycles stalled on instruction ISSUE -- per
// block step six 8-byte gathers (ten distinct lines each: one per block row of the wave) next to three streaming loads.  Halving the
// gather instructions (V2) or replacing them by one load + a lane exchange (bsell_block_step_shfl) did NOT pay (BSellMat::xmode 1 / 2,
// measured neutral / 7-13 % slower, see amgx.hip bsell_xmode): both stay opt-in.
template <int BS, bool V2>
__device__ __forceinline__ void bsell_block_step(const double* __restrict__ vk, const double* __restrict__ xv, int lane, double& acc) {
  if (V2 && (BS % 2) == 0) {
    const double2* __restrict__ x2 = reinterpret_cast<const double2*>(xv);
    double2 xx[BS / 2];
#pragma unroll
    for (int cp = 0; cp < BS / 2; ++cp) xx[cp] = x2[cp];
#pragma unroll
    for (int cp = 0; cp < BS / 2; ++cp) {
      const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
      acc += v0 * xx[cp].x + v1 * xx[cp].y;
    }
  } else {
#pragma unroll
    for (int cp = 0; cp < BS / 2; ++cp) {
      const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
      acc += v0 * xv[2 * cp] + v1 * xv[2 * cp + 1];
    }
    if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xv[BS - 1];
  }
}
// the same step with ONE 8-byte gather per lane: lane r of a block row reads x[c * BS + r] and the BS lanes of the block row exchange
// their values through the cross-lane network (base = first lane of the block row).  Same products in the same order.
template <int BS>
__device__ __forceinline__ void bsell_block_step_shfl(const double* __restrict__ vk, const double* __restrict__ xv, int lane, int r, int base,
                                                      double& acc) {
  const double mine = xv[r];
  double xb[BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) xb[c] = __shfl(mine, base + c, WAVE);
#pragma unroll
  for (int cp = 0; cp < BS / 2; ++cp) {
    const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
    acc += v0 * xb[2 * cp] + v1 * xb[2 * cp + 1];
  }
  if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xb[BS - 1];
}

// Row product of one BSELL slice with the column indices one batch AHEAD: the plain loop asks for a step's column index, waits,
// asks for the gathered block and waits again (ISA of the round-4 kernel: s_waitcnt on the index loads before the gathers, vmcnt(0)
// at the end of every iteration) -- two dependent memory round trips per iteration and nothing in flight across iterations.  Here
// the indices of batch n + 1 are requested before the values and gathers of batch n, so an iteration waits for ONE round trip.
// Same products in the same order.
template <int BS, bool V2, int U>
__device__ __forceinline__ double bsell_row_dot_ahead(const double* __restrict__ vb, const int32_t* __restrict__ cb, int w, int rbl, int lane,
                                                      const double* x) {
  constexpr int RB = WAVE / BS;
  double acc = 0.0;
  int cn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) cn[u] = u < w ? cb[u * RB + rbl] : 0;
  int k = 0;
  for (; k + U <= w; k += U) {
    int cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cc[u] = cn[u];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int kk = k + U + u; cn[u] = kk < w ? cb[kk * RB + rbl] : 0; }
#pragma unroll
    for (int u = 0; u < U; ++u) bsell_block_step<BS, V2>(vb + (int64_t)(k + u) * (BS * WAVE), x + (int64_t)cc[u] * BS, lane, acc);
  }
#pragma unroll
  for (int u = 0; u < U - 1; ++u)
    if (k + u < w) bsell_block_step<BS, V2>(vb + (int64_t)(k + u) * (BS * WAVE), x + (int64_t)cn[u] * BS, lane, acc);
  return acc;
}

template <int BS, int EP>
__global__ __launch_bounds__(BLOCK) void bsell_spmv_kernel(int64_t n_rows, int slice0, int n_slices, BSellMat M,
                                                           const double* __restrict__ x, double* y, EpArgs ep) {
  constexpr int RB = WAVE / BS;
  const int lane = threadIdx.x & (WAVE - 1);
  // (slice0 .. n_slices: interior / boundary block rows of a rank-partitioned level, see sell_spmv_kernel)
  const int s = __builtin_amdgcn_readfirstlane(slice0 + blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= n_slices) return;
  const int rbl = lane / BS < RB ? lane / BS : RB - 1;      // idle lanes (lane >= RB*BS) shadow the last block row
  const int r = lane % BS;
  const int64_t brow = (int64_t)s * RB + rbl;
  const bool active = lane < RB * BS && brow < n_rows;
  const int64_t k0 = M.slice_ptr[s];
  const int w = (int)(M.slice_ptr[s + 1] - k0);
  const double* __restrict__ vb = M.val + k0 * (BS * WAVE);
  const int32_t* __restrict__ cb = M.col + k0 * RB;
  const int64_t i = brow * BS + r;
  // own-row epilogue operands requested ahead of the matrix stream (EPF_HOIST, see sell_spmv_kernel)
  const bool hoist = (ep.nt & EPF_HOIST) && active && EP != EP_MULT;
  double ob = 0.0, oy = 0.0, od[BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) od[c] = 0.0;
  if (hoist) {
    if (EP == EP_RES || EP == EP_JAC) ob = ep.b[i];
    if (EP == EP_AXPY || EP == EP_JAC) oy = ep.yin[i];
    if (EP == EP_JAC) {
#pragma unroll
      for (int c = 0; c < BS; ++c) od[c] = ep.dinv[brow * (BS * BS) + r * BS + c];
    }
  }
  double acc = 0.0;
  const bool x16 = M.xmode == 1 && (BS % 2) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;      // (wave-uniform)
  if (M.xmode >= 3) {
    if (M.xmode == 3) acc = bsell_row_dot_ahead<BS, false, 2>(vb, cb, w, rbl, lane, x);
    else if (M.xmode == 4) acc = bsell_row_dot_ahead<BS, false, 4>(vb, cb, w, rbl, lane, x);
    else acc = bsell_row_dot_ahead<BS, false, 3>(vb, cb, w, rbl, lane, x);
  } else if (M.xmode == 2) {
    const int xbase = lane - r;
    for (int k = 0; k < w; ++k) {
      const int c = cb[k * RB + rbl];
      bsell_block_step_shfl<BS>(vb + (int64_t)k * (BS * WAVE), x + (int64_t)c * BS, lane, r, xbase, acc);
    }
  } else if (x16) {
#pragma unroll BSELL_UNROLL
    for (int k = 0; k < w; ++k) {
      const int c = cb[k * RB + rbl];
      bsell_block_step<BS, true>(vb + (int64_t)k * (BS * WAVE), x + (int64_t)c * BS, lane, acc);
    }
  } else {
#pragma unroll BSELL_UNROLL
    for (int k = 0; k < w; ++k) {
      const int c = cb[k * RB + rbl];        // cached load: the RB*4-byte column chunks of consecutive steps share cache lines
      bsell_block_step<BS, false>(vb + (int64_t)k * (BS * WAVE), x + (int64_t)c * BS, lane, acc);
    }
  }
  double out = 0.0;
  if (EP == EP_JAC) {
    const double t = active ? (hoist ? ob : ep.b[i]) - acc : 0.0;
    if (active && ep.y2) ep.y2[i] = t;
    const int base = lane - r;
    double u = 0.0;
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      const double tc = __shfl(t, base + c, WAVE);
      if (active) u += (hoist ? od[c] : ep.dinv[brow * (BS * BS) + r * BS + c]) * tc;
    }
    if (active) out = (hoist ? oy : ep.yin[i]) + ep.s * u;
  } else if (active) {
    if (EP == EP_MULT) out = acc;
    else if (EP == EP_RES) out = (hoist ? ob : ep.b[i]) - acc;
    else out = (hoist ? oy : ep.yin[i]) + ep.s * acc;
  }
  if (active) y[i] = out;
}

// scalar row r of a block row in CSR storage (BR x BC blocks) times x, blocks k0, k0 + W, ... < e.  The small levels these
// kernels serve are pure latency: a rolled loop waits for the column index and then for the gathered x of every block (two
// dependent round trips per step, 12 steps for a 50-block row at W = 4 = the 15 us such a launch took); here the indices,
// the matrix rows and the gathers of UN blocks are requested together.
// UN: same-box A/B at cfg 5 / cfg 3 (6x6 blocks): the Gauss-Seidel colour kernels are fastest with 2 blocks in flight
// (143 vs 136-140 applications/s with 1 or 4), the transfer / residual kernels with 4 (235 vs 226 with 1 or 2).
template <int BR, int BC, int W, int UN>
__device__ __forceinline__ double bcsr_row_dot(int k0, int e, const int32_t* __restrict__ cols, const double* __restrict__ vals, int r,
                                               const double* x) {
  double acc = 0.0;
  int k = k0;
  for (; k + (UN - 1) * W < e; k += UN * W) {
    int cc[UN];
    double av[UN][BC], xv[UN][BC];
#pragma unroll
    for (int u = 0; u < UN; ++u) cc[u] = cols[k + u * W];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double* __restrict__ a = vals + (int64_t)(k + u * W) * (BR * BC) + r * BC;
#pragma unroll
      for (int c = 0; c < BC; ++c) av[u][c] = a[c];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double* xp = x + (int64_t)cc[u] * BC;
#pragma unroll
      for (int c = 0; c < BC; ++c) xv[u][c] = xp[c];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int c = 0; c < BC; ++c) acc += av[u][c] * xv[u][c];
  }
  for (; k < e; k += W) {
    const double* __restrict__ a = vals + (int64_t)k * (BR * BC) + r * BC;
    const double* xp = x + (int64_t)cols[k] * BC;
#pragma unroll
    for (int c = 0; c < BC; ++c) acc += a[c] * xp[c];
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------
// Block CSR, ROW-PER-LANE inside the block (BR x BC blocks, BR >= 2): lane (g, r) of a block row's lane group owns
// scalar row r and the blocks k = g, g+W, ...; it reads its BC contiguous values of each block, so the BR lanes of a
// group read one contiguous block and a wave streams 64/(BR*W) block rows with full cache-line use.  (The generic
// bcsrvec kernel lets ONE lane read a whole block with BR*BC scalar loads at a block-sized lane stride: 2.7 TB/s on 6x6.)
template <int BR, int BC, int W, int EP>
__global__ __launch_bounds__(BLOCK) void bcsr_rowlane_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ cols,
                                                             const double* __restrict__ vals,
                                                             const double* __restrict__ x, double* y, EpArgs ep) {
  constexpr int LPR = BR * W;                // lanes per block row
  constexpr int RPW = WAVE / LPR;            // block rows per wave
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int rloc = lane / LPR;
  const int g = (lane % LPR) / BR;
  const int r = lane % BR;
  const int64_t row = wave * RPW + rloc;
  const bool active = rloc < RPW && row < n_rows;
  double acc = 0.0;
  if (active) acc = bcsr_row_dot<BR, BC, W, 4>(rowptr[row] + g, rowptr[row + 1], cols, vals, r, x);
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) acc += __shfl_down(acc, o * BR, WAVE);
  // lanes with g == 0 now hold (A x)_r of their block row
  const int64_t i = row * BR + r;
  double out = 0.0;
  if (EP == EP_JAC) {           // square blocks only
    const double t = (active && g == 0) ? ep.b[i] - acc : 0.0;
    if (active && g == 0 && ep.y2) ep.y2[i] = t;
    const int base = lane - r;
    double u = 0.0;
#pragma unroll
    for (int c = 0; c < BR; ++c) {
      const double tc = __shfl(t, base + c, WAVE);
      if (active && g == 0) u += ep.dinv[row * (BR * BR) + r * BR + c] * tc;
    }
    if (active && g == 0) out = ep.yin[i] + ep.s * u;
  } else if (active && g == 0) {
    if (EP == EP_MULT) out = acc;
    else if (EP == EP_RES) out = ep.b[i] - acc;
    else out = ep.yin[i] + ep.s * acc;
  }
  if (active && g == 0) y[i] = out;
}

// ---------------------------------------------------------------------------------------------------
// multicolour Gauss-Seidel, scalar: one colour per launch, colour-major SELL copy of A.
//   x_k += dinv_k * (b_k - A_k: x)        (RHS form, reference gssmoother.cpp:209-212)
// Rows of one colour have no mutual couplings, so the in-place update is race-free.
template <int G>
__global__ __launch_bounds__(BLOCK) void gs_color_kernel(int slice_begin, int slice_end, SellMat M,
                                                         const int32_t* __restrict__ rowid,
                                                         const double* __restrict__ dinv,
                                                         const double* __restrict__ b, double* x) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = __builtin_amdgcn_readfirstlane(slice_begin + blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= slice_end) return;
  // G lanes per row (SELL-G slices of 64/G rows): short dependent chains on the coarse levels, where a colour has
  // only a few hundred rows of ~50 entries and one thread per row would be pure latency
  const int row = rowid[(int64_t)s * (WAVE / G) + lane / G];
  // own-row operands requested before the row product (one dependent round trip less per wave, see EPF_HOIST)
  const bool writer = row >= 0 && (lane % G) == 0;
  double dv = 0.0, bv = 0.0, xv = 0.0;
  if (writer) { dv = dinv[row]; bv = b[row]; xv = x[row]; }
  double acc = row >= 0 ? sell_row_dot(M, s, lane, row, x) : 0.0;
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
  if (writer) x[row] = xv + dv * (bv - acc);
}

// ---------------------------------------------------------------------------------------------------
// Block-hybrid Gauss-Seidel: ONE launch per sweep.  A workgroup owns B = TH / G consecutive rows (G lanes per row) and
// sweeps them like one rank of the reference's hybrid smoother (HybridGSSmoother, gssmoother.cpp:709-861, with
// workgroups in the role of the ranks): Gauss-Seidel inside the block in colour order, couplings that leave the block use
// the values from the START of the sweep (xin); the diagonal is the l1-modified one where the off-block weight is large
// (hybrid_smoother_utils.hpp:111-142, applied on the host: amgh_hybrid_dinv).
//   * the block's own x lives in LDS (natural order: coalesced load and store); every lane loads its share of its row
//     (<= 2*GSB_WP + 1 entries, block-local SELL-G slices whose slots are sorted by colour) into REGISTERS up front, so the
//     matrix streams from HBM exactly like in the SpMV kernels while nothing depends on the sweep yet;
//   * off-block entries are gathered from xin (L2 / Infinity Cache) and summed immediately; in-block entries keep
//     (value, LDS slot); the colour phases then touch LDS only: acc = sum v * xs[slot], x_k += dinv_k (b_k - acc),
//     one workgroup barrier per colour;
//   * out of place (xin != xout) unless FROM_ZERO (x = 0 everywhere: no global gathers at all).
// Replaces 8...30 dependent colour launches per sweep whose colour-major slices touched ~8x the cache lines per gather.
#ifndef GSB_MINW
#define GSB_MINW 1
#endif
constexpr int GSB_WP = 8;                    // pair-steps per lane held in registers (<= 16 entries + one odd trailing)
struct GsbArgs {
  const int32_t* rowid;                      // [n_blocks * B] slot -> row, -1 = padding
  const uint8_t* slotcolor;                  // [n_blocks * B] colour of the slot's row, 255 = not swept (non-free / padding)
  const double* dinv;
  const double* b;
  int n_colors;
  int backward;
  // local-window form (LW): per block the sorted list of the distinct OFF-block columns its rows touch; slices in the 16-bit encoding
  // then carry codes -- code < B: in-block row, code >= B: entry code - B of the list -- and the sweep-start values of the list are
  // staged in LDS; slices in the 32-bit encoding (blocks whose list would not fit) carry global columns as in the plain form
  const int32_t* lw_cptr;
  const int32_t* lw_ccol;
};
constexpr int GSB_LW_CAP = 3072;             // off-block columns per block the LDS stage holds (24 KB)

// WP = pair-steps a lane holds (2 * WP + 1 entries): GSB_WP in general; the sweep from zero reads the short `lowin` rows and
// is instantiated with WP = 2 where they fit (fewer registers: more resident workgroups to hide each other's colour phases)
template <int TH, int G, bool FROM_ZERO, int WP = GSB_WP, bool LW = false>
// (second launch-bounds argument = waves per SIMD the register budget must allow: the mid-width general sweep asks for two 1024-lane
//  or three 512-lane workgroups per CU)
__global__ __launch_bounds__(TH, (TH == 256 ? GSB_MINW : (WP <= 5 && !FROM_ZERO ? (TH == 1024 ? 8 : 6) : 1))) void gsb_sweep_kernel(int64_t n_rows, int block0, SellMat M, GsbArgs a,
                                                        const double* __restrict__ xin, double* xout) {
  constexpr int B = TH / G;                  // rows per block
  constexpr int RPS = WAVE / G;              // rows per slice
  __shared__ double xs[B], bs[B], ds[B];     // x of the block; b and the (modified) inverse diagonal in natural row order
  __shared__ double xw[LW ? GSB_LW_CAP : 1]; // local-window form: sweep-start values of the block's off-block columns
  const int blk = block0 + blockIdx.x;
  const int64_t r0 = (int64_t)blk * B;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int s = __builtin_amdgcn_readfirstlane(blk * (TH / WAVE) + (tid >> 6));
  double xwv[LW ? (GSB_LW_CAP + TH - 1) / TH : 1];
  int wk0 = 0, wk1 = 0;
  if (LW) {
    wk0 = a.lw_cptr[blk]; wk1 = a.lw_cptr[blk + 1];
#pragma unroll
    for (int q = 0; q < (GSB_LW_CAP + TH - 1) / TH; ++q) {
      const int k = wk0 + tid + q * TH;
      xwv[q] = k < wk1 ? xin[a.lw_ccol[k]] : 0.0;
    }
  }
  // ---- phase 1: every load that depends on nothing goes out before anything is consumed (a wait inside this phase
  // would serialise ~30 round trips per wave: measured 456 us -> see profiles/r02/gs_block_ab.txt)
  double x_own = 0.0, b_nat = 0.0, d_nat = 0.0;
  if (tid < B && r0 + tid < n_rows) {        // natural order: coalesced, and independent of the slot -> row map
    if (!FROM_ZERO) x_own = xin[r0 + tid];
    b_nat = a.b[r0 + tid];
    d_nat = a.dinv[r0 + tid];
  }
  const int slot = s * RPS + lane / G;
  const int row = a.rowid[slot];
  const int col_raw = (int)a.slotcolor[slot];
  const int64_t sp0 = M.slice_ptr[s];
  const int64_t base = sp0 & ~(int64_t)63;
  const int w = (int)(((M.slice_ptr[s + 1] & ~(int64_t)63) - base) >> 6);
  const int np = w >> 1;
  const bool c16 = sp0 & 1;
  const double* __restrict__ vb = M.val + base;
  const int32_t* __restrict__ cb = M.cbase + (base >> 6);
  double v[2 * WP + 1];
  int cl[2 * WP + 1];
  uint32_t ra[WP], rb[WP];           // raw index words: 16-bit form = one packed pair in ra; 32-bit form = ra, rb
#pragma unroll
  for (int p = 0; p < WP; ++p) {
    v[2 * p] = 0.0; v[2 * p + 1] = 0.0; ra[p] = 0; rb[p] = 0;
    if (p < np) {                                            // wave-uniform
      v[2 * p] = ld_nt(vb + (p * WAVE + lane) * 2);
      v[2 * p + 1] = ld_nt(vb + (p * WAVE + lane) * 2 + 1);
      if (c16) ra[p] = ld_nt(reinterpret_cast<const uint32_t*>(M.col16 + base) + p * WAVE + lane);
      else {
        ra[p] = (uint32_t)ld_nt(M.col32 + base + (p * WAVE + lane) * 2);
        rb[p] = (uint32_t)ld_nt(M.col32 + base + (p * WAVE + lane) * 2 + 1);
      }
    }
  }
  int cbv[2 * WP + 1];                   // column bases of the slice: one group of scalar loads (the array has slack at its end)
#pragma unroll
  for (int j = 0; j < 2 * WP + 1; ++j) cbv[j] = 0;
  if (c16) {
#pragma unroll
    for (int j = 0; j < 2 * WP + 1; ++j) cbv[j] = cb[j];
  }
  uint32_t rt = 0;
  v[2 * WP] = 0.0;
  if (w & 1) {
    const int64_t o = (int64_t)(w - 1) * WAVE + lane;
    v[2 * WP] = ld_nt(vb + o);
    rt = c16 ? (uint32_t)ld_nt(M.col16 + base + o) : (uint32_t)ld_nt(M.col32 + base + o);
  }
  // ---- phase 2: the column decode (needs the index words)
  const int mycol = row >= 0 ? col_raw : 255;
  const bool writer = mycol != 255 && (lane % G) == 0;
#pragma unroll
  for (int p = 0; p < WP; ++p) {
    cl[2 * p] = (int)r0; cl[2 * p + 1] = (int)r0;
    if (p < np) {
      if (c16) { cl[2 * p] = cbv[2 * p] + (int)(ra[p] & 0xffffu); cl[2 * p + 1] = cbv[2 * p + 1] + (int)(ra[p] >> 16); }
      else { cl[2 * p] = (int)ra[p]; cl[2 * p + 1] = (int)rb[p]; }
    }
  }
  // (the odd trailing column of a slice of width w <= 2*WP + 1 is column w - 1: a wave-uniform pick from the group)
  int cbt = 0;
#pragma unroll
  for (int j = 0; j < 2 * WP + 1; j += 2) cbt = (w - 1 == j) ? cbv[j] : cbt;
  cl[2 * WP] = (w & 1) ? (c16 ? cbt + (int)rt : (int)rt) : (int)r0;
  // ---- phase 3: off-block part (frozen values): all gathers requested, then summed; in-block part keeps (value, LDS slot)
  double xg[2 * WP + 1];
  if (LW) {
    // the window goes to LDS first (its loads were requested at the very top); EVERY thread of the block stages its share, whatever
    // the encoding of its own slice (an empty slice is never flagged 16-bit)
#pragma unroll
    for (int q = 0; q < (GSB_LW_CAP + TH - 1) / TH; ++q) {
      const int k = tid + q * TH;
      if (wk0 + k < wk1) xw[k] = xwv[q];
    }
    __syncthreads();
  }
  if (LW && c16) {
    // local-window slice: codes
#pragma unroll
    for (int j = 0; j < 2 * WP + 1; ++j) {
      // (slots beyond the slice's width carry the r0 dummy of the decode above with value 0: in-block slot 0)
      const int code = (j < 2 * np || (j == 2 * WP && (w & 1))) ? cl[j] : 0;
      const bool inb = code < B;
      xg[j] = inb ? 0.0 : xw[code - B];
      cl[j] = inb ? code : -1;
    }
  } else {
#pragma unroll
  for (int j = 0; j < 2 * WP + 1; ++j) {
    const int loc = cl[j] - (int)r0;
    const bool inb = loc >= 0 && loc < B && cl[j] < n_rows;       // (ghost columns of a rank-partitioned level are never in-block)
    xg[j] = 0.0;
    if (!FROM_ZERO) { if (!inb) xg[j] = xin[cl[j]]; }
    cl[j] = inb ? loc : -1;
  }
  }
  double acc_off = 0.0;
#pragma unroll
  for (int j = 0; j < 2 * WP + 1; ++j) {
    const bool inb = cl[j] >= 0;
    if (!FROM_ZERO) acc_off += inb ? 0.0 : v[j] * xg[j];
    if (!inb) { v[j] = 0.0; cl[j] = 0; }
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc_off += __shfl_xor(acc_off, o, G);
  if (tid < B) { xs[tid] = x_own; bs[tid] = b_nat; ds[tid] = d_nat; }
  const int own = row >= 0 ? row - (int)r0 : 0;
  __syncthreads();                                           // xs, bs, ds are loaded
  const double dv = writer ? ds[own] : 0.0, bv = writer ? bs[own] : 0.0;
  for (int q = 0; q < a.n_colors; ++q) {
    const int c = a.backward ? a.n_colors - 1 - q : q;
    if (__any(mycol == c)) {                                 // slices hold one or two colours (slots are colour-sorted)
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < 2 * WP + 1; ++j) acc += v[j] * xs[cl[j]];
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
      if (writer && mycol == c) xs[own] = xs[own] + dv * (bv - acc_off - acc);
    }
    __syncthreads();
  }
  if (tid < B && r0 + tid < n_rows) xout[r0 + tid] = xs[tid];
}

// r = -(U x) on the colour-major rows: the residual right after a forward sweep from x = 0, where
// (b - L x - D x)_k = 0 holds for every swept row (see Handle::pre_smooth), so only the entries coupling to HIGHER
// colours are needed.  One launch over all colours (no ordering required).
template <int G>
__global__ __launch_bounds__(BLOCK) void gs_upper_residual_kernel(int n_slices, SellMat M, const int32_t* __restrict__ rowid,
                                                                  const double* __restrict__ x, double* __restrict__ r) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= n_slices) return;
  const int row = rowid[(int64_t)s * (WAVE / G) + lane / G];
  double acc = row >= 0 ? sell_row_dot(M, s, lane, row, x) : 0.0;
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
  if (row >= 0 && (lane % G) == 0) r[row] = -acc;
}

// multicolour Gauss-Seidel, block BS x BS: CSR rows through a colour-major row list, row-per-lane inside the block
// (lane (g, r) owns scalar row r and the blocks k = g, g+W, ... of its block row, like bcsr_rowlane_kernel)
template <int BS, int W>
__global__ __launch_bounds__(BLOCK) void bgs_color_kernel(int list_begin, int list_end,
                                                          const int32_t* __restrict__ rowlist,
                                                          const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ cols,
                                                          const double* __restrict__ vals,
                                                          const double* __restrict__ dinv,
                                                          const double* __restrict__ b, double* x) {
  constexpr int LPR = BS * W;
  constexpr int RPW = WAVE / LPR;
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int rloc = lane / LPR;
  const int g = (lane % LPR) / BS;
  const int r = lane % BS;
  const int64_t q = list_begin + wave * RPW + rloc;
  const bool active = rloc < RPW && q < list_end;
  const int row = active ? rowlist[q] : 0;
  double acc = 0.0;
  if (active) acc = bcsr_row_dot<BS, BS, W, (BS >= 6 ? 2 : 4)>(rowptr[row] + g, rowptr[row + 1], cols, vals, r, x);
#pragma unroll
  for (int o = W >> 1; o > 0; o >>= 1) acc += __shfl_down(acc, o * BS, WAVE);
  const int64_t i = (int64_t)row * BS + r;
  const double t = (active && g == 0) ? b[i] - acc : 0.0;
  const int base = lane - r;
  double u = 0.0;
#pragma unroll
  for (int c = 0; c < BS; ++c) {
    const double tc = __shfl(t, base + c, WAVE);
    if (active && g == 0) u += dinv[(int64_t)row * (BS * BS) + r * BS + c] * tc;
  }
  if (active && g == 0) x[i] += u;
}

// multicolour Gauss-Seidel, block BS x BS, on a COLOUR-MAJOR BSELL copy of A (slices never cross a colour): the matrix
// streams in aligned 1 KiB chunks like bsell_spmv_kernel (the CSR row-list kernel above reads 8-byte elements at block
// strides: 2-3 TB/s).  x_B += Dinv_B (b_B - A_B: x), the block solve goes through wave shuffles.  rowid[slot] = block row
// (or -1: padding).  In place: rows of one colour are not coupled, a block row is read and written by one lane group.
template <int BS>
__global__ __launch_bounds__(BLOCK) void bgs_bsell_color_kernel(int slice_begin, int slice_end, BSellMat M,
                                                                const int32_t* __restrict__ rowid, const double* __restrict__ dinv,
                                                                const double* __restrict__ b, double* x) {
  constexpr int RB = WAVE / BS;
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = __builtin_amdgcn_readfirstlane(slice_begin + blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= slice_end) return;
  const int rbl = lane / BS < RB ? lane / BS : RB - 1;
  const int r = lane % BS;
  const int brow = rowid[(int64_t)s * RB + rbl];
  const bool active = lane < RB * BS && brow >= 0;
  const int64_t k0 = M.slice_ptr[s];
  const int w = (int)(M.slice_ptr[s + 1] - k0);
  const double* __restrict__ vb = M.val + k0 * (BS * WAVE);
  const int32_t* __restrict__ cb = M.col + k0 * RB;
  const int64_t i = (int64_t)(brow >= 0 ? brow : 0) * BS + r;
  double bv = 0.0, od[BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) od[c] = 0.0;
  if (active) {
    bv = b[i];
#pragma unroll
    for (int c = 0; c < BS; ++c) od[c] = dinv[(int64_t)brow * (BS * BS) + r * BS + c];
  }
  double acc = 0.0;
#pragma unroll 2
  for (int k = 0; k < w; ++k) {
    const int c = cb[k * RB + rbl];
    const double* xv = x + (int64_t)c * BS;
    const double* __restrict__ vk = vb + (int64_t)k * (BS * WAVE);
#pragma unroll
    for (int cp = 0; cp < BS / 2; ++cp) {
      const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
      acc += v0 * xv[2 * cp] + v1 * xv[2 * cp + 1];
    }
    if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xv[BS - 1];
  }
  const double t = active ? bv - acc : 0.0;
  const int base = lane - r;
  double u = 0.0;
#pragma unroll
  for (int c = 0; c < BS; ++c) u += od[c] * __shfl(t, base + c, WAVE);
  if (active) x[i] += u;
}

// r_B = -(U x)_B on the colour-major block rows after a forward sweep from x = 0 (see gs_upper_residual_kernel); M holds
// the couplings to higher colours only, one launch over all colours
template <int BS>
__global__ __launch_bounds__(BLOCK) void bgs_bsell_upper_residual_kernel(int n_slices, BSellMat M, const int32_t* __restrict__ rowid,
                                                                         const double* __restrict__ x, double* __restrict__ r) {
  constexpr int RB = WAVE / BS;
  const int lane = threadIdx.x & (WAVE - 1);
  const int s = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (s >= n_slices) return;
  const int rbl = lane / BS < RB ? lane / BS : RB - 1;
  const int rr = lane % BS;
  const int brow = rowid[(int64_t)s * RB + rbl];
  const bool active = lane < RB * BS && brow >= 0;
  const int64_t k0 = M.slice_ptr[s];
  const int w = (int)(M.slice_ptr[s + 1] - k0);
  const double* __restrict__ vb = M.val + k0 * (BS * WAVE);
  const int32_t* __restrict__ cb = M.col + k0 * RB;
  double acc = 0.0;
#pragma unroll 2
  for (int k = 0; k < w; ++k) {
    const int c = cb[k * RB + rbl];
    const double* xv = x + (int64_t)c * BS;
    const double* __restrict__ vk = vb + (int64_t)k * (BS * WAVE);
#pragma unroll
    for (int cp = 0; cp < BS / 2; ++cp) {
      const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
      acc += v0 * xv[2 * cp] + v1 * xv[2 * cp + 1];
    }
    if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xv[BS - 1];
  }
  if (active) r[(int64_t)brow * BS + rr] = -acc;
}

// ---------------------------------------------------------------------------------------------------
// Block-hybrid Gauss-Seidel for SQUARE-BLOCK levels (BS = 2, 3, 6): ONE launch per sweep instead of one per colour.
// The reference's hybrid smoother with workgroups in the role of the ranks (HybridGSSmoother<Mat<BS,BS>>,
// gssmoother.cpp:709-861, 891-896), like gsb_sweep_kernel for scalar levels: a workgroup owns BB consecutive block rows;
//   phase 0: b' = b - (A_off + A_oth) x_old   -- everything that multiplies sweep-start values, with all waves busy:
//            A_off = the couplings that leave the block (frozen) + the diagonal block, streamed as BSELL slices exactly like
//            bsell_spmv_kernel; A_oth = the in-block couplings to the colours this sweep reaches LATER (higher colours in a
//            forward sweep, lower ones in a backward sweep), all their slices at once, x from LDS; b' stays in LDS
//   colour phases: x_B += Dinv_B (b'_B - A_in,B: x) on the in-block couplings to the colours already swept (BSELL slices
//            sorted by colour inside the block, LOCAL block columns), x in LDS, one workgroup barrier per colour -- these run
//            one after the other inside a workgroup and carry ~9 % of A for line blocks; Dinv = inverse of the l1-modified
//            block diagonal (amgh_hybrid_dinv_block, hybrid_smoother_utils.hpp:86-141)
// A = A_off + A_lowin + A_upin (three images, every entry once): forward sweep IN = lowin, OTH = upin; backward the reverse.
// Every entry of A is read once per sweep.  Out of place (other workgroups read the old values), except from zero where
// nothing outside the block is read.
// blk_ptr / blk_rows: the block rows of every sweep block (ascending inside a block).  Runs of consecutive rows, or compact
// blocks grown over the matrix graph (amgh_compact_blocks): the local index of a row = its position in its block's list.
// MODE 0: general sweep (xin -> xout).  MODE 1: from x = 0, nothing outside the block is read (phase 0 skipped).
// MODE 2 (block-COLOURED sweeps from zero, see below): own rows start from 0, phase 0 streams OFF (= the couplings to the sweep
//         blocks of LOWER block colours, already swept in this pass) with the values in xin; the in-block `other` image multiplies zeros.
// Block-coloured form (blk_list != null, DevBGSB::bc): the sweep blocks carry a colouring of the BLOCK graph and a sweep is one
// launch per block colour over blk_list[block0 ...): blocks of one colour are not coupled, so the launch works IN PLACE
// (xin == xout) and a block reads the new values of every block swept before it -- Gauss-Seidel between the blocks as well as
// inside them, i.e. exact Gauss-Seidel in the order (block colour, block, in-block colour), instead of the hybrid form's frozen
// couplings.  PCG at 30^3 nodes: 15 iterations (sequential order 16, hybrid line blocks 19).
template <int BS, int MODE>
__global__ __launch_bounds__(BLOCK) void bgsb_sweep_kernel(int BB, int block0, const int32_t* __restrict__ blk_list,
                                                           const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ blk_rows,
                                                           BSellMat OFF, const int32_t* __restrict__ off_ptr,
                                                           BSellMat IN, BSellMat OTH, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_row,
                                                           int n_colors, int dir, const double* __restrict__ dinv, const double* __restrict__ b,
                                                           const double* xin, double* xout) {
  constexpr bool FROM_ZERO = MODE != 0;
  extern __shared__ double bgsb_sh[];
  constexpr int RB = WAVE / BS;
  double* xs = bgsb_sh;
  double* bsh = bgsb_sh + (size_t)BB * BS;
  int* rowsh = reinterpret_cast<int*>(bgsb_sh + (size_t)2 * BB * BS);      // global block row of every local row
  const int blk = blk_list ? blk_list[block0 + blockIdx.x] : block0 + blockIdx.x;
  const int p0 = blk_ptr[blk];
  const int nb = blk_ptr[blk + 1] - p0;
  for (int e = threadIdx.x; e < nb; e += BLOCK) rowsh[e] = blk_rows[p0 + e];
  __syncthreads();
  for (int e = threadIdx.x; e < nb * BS; e += BLOCK) {
    const int64_t g = (int64_t)rowsh[e / BS] * BS + e % BS;
    xs[e] = FROM_ZERO ? 0.0 : xin[g];
    bsh[e] = b[g];
  }
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
  const int rbl = lane / BS < RB ? lane / BS : RB - 1;
  const int r = lane % BS;
  const bool lane_on = lane < RB * BS;
  if (MODE != 1) {
    const int s0 = off_ptr[blk], s1 = off_ptr[blk + 1];
    for (int s = s0 + wave; s < s1; s += WAVES_PER_BLOCK) {
      const int lrow = (s - s0) * RB + rbl;
      const bool active = lane_on && lrow < nb;
      const int64_t k0 = OFF.slice_ptr[s];
      const int w = (int)(OFF.slice_ptr[s + 1] - k0);
      const double* __restrict__ vb = OFF.val + k0 * (BS * WAVE);
      const int32_t* __restrict__ cb = OFF.col + k0 * RB;
      double acc = 0.0;
      if (OFF.xmode >= 3) {
        if (OFF.xmode == 3) acc = bsell_row_dot_ahead<BS, false, 2>(vb, cb, w, rbl, lane, xin);
        else if (OFF.xmode == 4) acc = bsell_row_dot_ahead<BS, false, 4>(vb, cb, w, rbl, lane, xin);
        else acc = bsell_row_dot_ahead<BS, false, 3>(vb, cb, w, rbl, lane, xin);
      } else if (OFF.xmode == 2) {
        for (int k = 0; k < w; ++k) {
          const int c = cb[k * RB + rbl];
          bsell_block_step_shfl<BS>(vb + (int64_t)k * (BS * WAVE), xin + (int64_t)c * BS, lane, r, lane - r, acc);
        }
      } else if (OFF.xmode == 1 && (BS % 2) == 0 && (reinterpret_cast<uintptr_t>(xin) & 15) == 0) {
#pragma unroll 2
        for (int k = 0; k < w; ++k) {
          const int c = cb[k * RB + rbl];
          bsell_block_step<BS, true>(vb + (int64_t)k * (BS * WAVE), xin + (int64_t)c * BS, lane, acc);
        }
      } else {
#pragma unroll 2
      for (int k = 0; k < w; ++k) {
        const int c = cb[k * RB + rbl];
        bsell_block_step<BS, false>(vb + (int64_t)k * (BS * WAVE), xin + (int64_t)c * BS, lane, acc);
      }
      }
      if (active) bsh[lrow * BS + r] -= acc;
    }
    __syncthreads();
    // the other in-block image: all colours at once, sweep-start values from LDS (a row sits in one slice of it)
    const int t0 = in_ptr[blk * n_colors], t1 = MODE == 0 ? in_ptr[(blk + 1) * n_colors] : t0;
    for (int s = t0 + wave; s < t1; s += WAVES_PER_BLOCK) {
      const int lrow = in_row[(int64_t)s * RB + rbl];
      const bool active = lane_on && lrow >= 0;
      const int64_t k0 = OTH.slice_ptr[s];
      const int w = (int)(OTH.slice_ptr[s + 1] - k0);
      const double* __restrict__ vb = OTH.val + k0 * (BS * WAVE);
      const int32_t* __restrict__ cb = OTH.col + k0 * RB;
      double acc = 0.0;
#pragma unroll 2
      for (int k = 0; k < w; ++k) {
        const int cl = cb[k * RB + rbl];
        const double* xv = xs + cl * BS;
        const double* __restrict__ vk = vb + (int64_t)k * (BS * WAVE);
#pragma unroll
        for (int cp = 0; cp < BS / 2; ++cp) {
          const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
          acc += v0 * xv[2 * cp] + v1 * xv[2 * cp + 1];
        }
        if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xv[BS - 1];
      }
      if (active) bsh[lrow * BS + r] -= acc;
    }
    __syncthreads();
  }
  for (int q = 0; q < n_colors; ++q) {
    const int c = dir ? n_colors - 1 - q : q;
    const int s0 = in_ptr[blk * n_colors + c], s1 = in_ptr[blk * n_colors + c + 1];
    for (int s = s0 + wave; s < s1; s += WAVES_PER_BLOCK) {
      const int lrow = in_row[(int64_t)s * RB + rbl];
      const bool active = lane_on && lrow >= 0;
      const int64_t k0 = IN.slice_ptr[s];
      const int w = (int)(IN.slice_ptr[s + 1] - k0);
      const double* __restrict__ vb = IN.val + k0 * (BS * WAVE);
      const int32_t* __restrict__ cb = IN.col + k0 * RB;
      double od[BS];
#pragma unroll
      for (int cc = 0; cc < BS; ++cc) od[cc] = active ? dinv[(int64_t)rowsh[lrow] * (BS * BS) + r * BS + cc] : 0.0;
      double acc = 0.0;
#pragma unroll 2
      for (int k = 0; k < w; ++k) {
        const int cl = cb[k * RB + rbl];
        const double* xv = xs + cl * BS;
        const double* __restrict__ vk = vb + (int64_t)k * (BS * WAVE);
#pragma unroll
        for (int cp = 0; cp < BS / 2; ++cp) {
          const double v0 = ld_nt(vk + cp * (2 * WAVE) + lane * 2), v1 = ld_nt(vk + cp * (2 * WAVE) + lane * 2 + 1);
          acc += v0 * xv[2 * cp] + v1 * xv[2 * cp + 1];
        }
        if (BS & 1) acc += ld_nt(vk + (BS / 2) * (2 * WAVE) + lane) * xv[BS - 1];
      }
      const double t = active ? bsh[lrow * BS + r] - acc : 0.0;
      const int base = lane - r;
      double u = 0.0;
#pragma unroll
      for (int cc = 0; cc < BS; ++cc) u += od[cc] * __shfl(t, base + cc, WAVE);
      if (active) xs[lrow * BS + r] += u;
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < nb * BS; e += BLOCK) xout[(int64_t)rowsh[e / BS] * BS + e % BS] = xs[e];
}

// ---------------------------------------------------------------------------------------------------
// Column-blocked restriction  b_c = P^T r  for large scalar levels (reference ProlMap::TransferF2C,
// dof_map.cpp:636-654).  The gather form over P^T touches ~40 different cache lines of r per coarse row and is
// TA/L2-bound (2.9 TB/s); here a workgroup owns a chunk of RESTRICT_CHUNK consecutive FINE rows, stages that
// piece of r in LDS with coalesced loads and reduces the chunk-local transpose of P from LDS:
//   part[slot] = sum_e w[e] * r_lds[fi[e]]       (slot = (chunk, coarse column) pair, entries 10 B each)
// A second tiny kernel adds the few partial sums of every coarse row in a fixed order (deterministic, no atomics).
constexpr int RESTRICT_CHUNK = 1024;          // fine rows per workgroup
constexpr int RESTRICT_MAX_ENTRIES = 4096;    // entries of P per chunk (P has <= sp_max_per_row entries per row)

// phase A: every thread streams entries (coalesced, independent loads) and leaves the products in LDS;
// phase B: one thread per (chunk, coarse column) slot adds its contiguous segment of products.
__global__ __launch_bounds__(BLOCK) void restrict_chunk_kernel(int64_t n_fine, const int32_t* __restrict__ chunk_slot,
                                                               const int32_t* __restrict__ slot_ptr,
                                                               const double* __restrict__ w, const uint16_t* __restrict__ fi,
                                                               const double* __restrict__ r, double* __restrict__ part,
                                                               const int32_t* __restrict__ dest) {
  __shared__ double rl[RESTRICT_CHUNK];
  __shared__ double pr[RESTRICT_MAX_ENTRIES];
  const int c = blockIdx.x;
  const int64_t row0 = (int64_t)c * RESTRICT_CHUNK;
  const int nrow = (int)((n_fine - row0) < RESTRICT_CHUNK ? (n_fine - row0) : RESTRICT_CHUNK);
  const int s0 = chunk_slot[c], s1 = chunk_slot[c + 1];
  const int e0 = slot_ptr[s0], e1 = slot_ptr[s1];
  for (int i = threadIdx.x; i < nrow; i += BLOCK) rl[i] = r[row0 + i];
  __syncthreads();
  for (int e = e0 + threadIdx.x; e < e1; e += BLOCK) pr[e - e0] = ld_nt(w + e) * rl[ld_nt(fi + e)];
  __syncthreads();
  for (int slot = s0 + threadIdx.x; slot < s1; slot += BLOCK) {
    const int a = slot_ptr[slot] - e0, b = slot_ptr[slot + 1] - e0;
    double acc = 0.0;
    for (int k = a; k < b; ++k) acc += pr[k];
    part[dest ? dest[slot] : slot] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused Jacobi pre-smoothing + restriction for big scalar levels (the level-0 hot spot):
//   x = omega*Dinv*b,  r = b - A'b  (EP_PRE, one pass over the column-scaled image),  part = chunk-local P^T r
// One 1024-thread workgroup owns 16 consecutive SELL slices = FUSED_CHUNK rows; r never goes to HBM: it is left in
// LDS, multiplied with the chunk-local transpose of P (entries: fp64 weight + 16-bit local row) and reduced per
// (chunk, coarse column) slot; restrict_sum_kernel then adds the ~10 partial sums of every coarse row in a fixed
// order.  Replaces: 80 MB write of r + the P^T gather kernel (134 us at cfg 2, TA/L2-bound).
// FB = workgroup size = rows per chunk (1024 or 512); 4 entries of P per row at most
// MODE 0: Jacobi pre-smoothing as described above.  MODE 1: residual after a block-hybrid Gauss-Seidel sweep from zero,
// r = c .* x - A_rest x (EP_CRES; b = the swept x, dinv = c, nothing written to x), with the same chunk-local restriction.
// EPT: entries of P per thread the chunk may hold (4: prolongations with <= 3 entries per row; 6: the up to 5 of the reference's
// "classic" rows; chosen per level by build_restrict from the fullest chunk)
// G = lanes per row (SELL-G image, long rows of the coarser levels): the chunk then holds FUSED_BLOCK / G rows, the G partial
// row sums are combined by a wave shuffle and the first lane of every group runs the epilogue
template <int FUSED_BLOCK, int MODE = 0, int EPT = 4, int G = 1>
__global__ __launch_bounds__(FUSED_BLOCK) void sell_pre_restrict_kernel(int64_t n_rows, int chunk0, int n_slices, SellMat M,
                                                                        const double* __restrict__ b, const double* __restrict__ dinv,
                                                                        double omega, int nt, double* __restrict__ x, double* r_out,
                                                                        const int32_t* __restrict__ chunk_slot,
                                                                        const int32_t* __restrict__ slot_ptr,
                                                                        const double* __restrict__ w, const uint16_t* __restrict__ fi,
                                                                        double* __restrict__ part,
                                                                        const int32_t* __restrict__ dest,
                                                                        const int32_t* __restrict__ slice_list = nullptr) {
  constexpr int FUSED_MAX_ENTRIES = EPT * FUSED_BLOCK;
  constexpr int RPC = FUSED_BLOCK / G;         // rows per chunk
  __shared__ double rl[RPC];
  __shared__ double pr[FUSED_MAX_ENTRIES];
  const int lane = threadIdx.x & (WAVE - 1);
  const int c = chunk0 + sell_unit(M);         // chunk0: first chunk of the launch (interior / boundary chunks of a rank-partitioned level)
  // compact chunks (G == 1, cluster_slices): the chunk's slices come from a list (-1: none) instead of being consecutive
  const int sq = c * (FUSED_BLOCK / WAVE) + (threadIdx.x >> 6);
  const int s_raw = slice_list ? slice_list[sq] : sq;
  const int s = __builtin_amdgcn_readfirstlane(s_raw < 0 ? n_slices : s_raw);
  const int row = s * (WAVE / G) + lane / G;
  const int lrow = (threadIdx.x >> 6) * (WAVE / G) + lane / G;     // row inside the chunk
  const bool writer = (lane % G) == 0;
  // the chunk-local restriction data of this thread is requested FIRST, so that it arrives while the row product
  // streams A'; the epilogue after the barriers then touches LDS only
  // (nothing in this prologue is CONSUMED before the row product: a subtraction on a freshly loaded slot pointer here made
  // every wave wait for all of its restriction data -- two full memory round trips -- before it requested a byte of A')
  const bool has_slice = s < n_slices;
  const int64_t sp0 = has_slice ? M.slice_ptr[s] : 0, sp1 = has_slice ? M.slice_ptr[s + 1] : 0;
  const int s0 = chunk_slot[c], s1 = chunk_slot[c + 1];
  const int e0 = slot_ptr[s0], e1 = slot_ptr[s1];
  double wq[FUSED_MAX_ENTRIES / FUSED_BLOCK];
  int fq[FUSED_MAX_ENTRIES / FUSED_BLOCK];
#pragma unroll
  for (int q = 0; q < FUSED_MAX_ENTRIES / FUSED_BLOCK; ++q) {
    const int e = e0 + threadIdx.x + q * FUSED_BLOCK;
    wq[q] = e < e1 ? ld_nt(w + e) : 0.0;
    fq[q] = e < e1 ? (int)ld_nt(fi + e) : 0;
  }
  const int myslot = s0 + threadIdx.x;
  // dest (optional, AMGX_RSUM_SORT=1): the partials of one coarse row stored next to each other, so that
  // restrict_sum_kernel streams them -- scattered stores here instead of scattered loads there; measured slower overall
  int mydest = myslot, pa_raw = 0, pb_raw = 0;
  if (myslot < s1) {
    if (dest) mydest = dest[myslot];
    pa_raw = slot_ptr[myslot];
    pb_raw = slot_ptr[myslot + 1];
  }
  double r = 0.0;
  if (MODE == 1) {
    if (s < n_slices) {
      double ci = 0.0, xi = 0.0;
      if (writer && row < n_rows) { ci = dinv[row]; xi = b[row]; }
      double xdd[2];
      double acc = sell_row_dot_sp(M, sp0, sp1, lane, row, b, xdd);
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
      if (writer && row < n_rows) { r = ci * xi - acc; if (r_out) r_out[row] = r; }
    }
  } else if (s < n_slices) {
    double bi = 0.0, di = 0.0;
    double xd[2] = {0.0, 0.0};
    const bool wdiag = G == 1 && M.wdiag && M.diag_first;
    const bool hoist = G > 1 || (nt & EPF_HOIST);
    if (!wdiag && hoist && writer && row < n_rows) { bi = b[row]; di = (nt & EPF_NT) ? ld_nt(dinv + row) : dinv[row]; }
    double acc = sell_row_dot_sp(M, sp0, sp1, lane, row, b, xd);
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
    if (writer && row < n_rows) {
      if (wdiag) {
        // diagonal slot = omega*Dinv_i (no dinv stream, b_i from the gather): see sell_spmv_kernel
        bi = xd[0];
        const double wd = xd[1];
        acc = acc - wd * bi + (wd != 0.0 ? omega * bi : 0.0);
        di = wd / omega;
      } else if (!hoist) { bi = b[row]; di = (nt & EPF_NT) ? ld_nt(dinv + row) : dinv[row]; }
      r = bi - acc;
      double xi = omega * (di * bi);
      if (nt & EPF_FOLD) xi += omega * (di * r);
      if (nt & EPF_NT) __builtin_nontemporal_store(xi, x + row);
      else x[row] = xi;
      if (r_out) r_out[row] = r;
    }
  }
  if (writer) rl[lrow] = r;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < FUSED_MAX_ENTRIES / FUSED_BLOCK; ++q) {
    const int e = threadIdx.x + q * FUSED_BLOCK;
    if (e0 + e < e1) pr[e] = wq[q] * rl[fq[q]];
  }
  __syncthreads();
  if (myslot < s1) {
    const int pa = pa_raw - e0, pb = pb_raw - e0;
    double acc = 0.0;
    for (int k = pa; k < pb; ++k) acc += pr[k];
    part[mydest] = acc;
  }
  for (int slot = myslot + FUSED_BLOCK; slot < s1; slot += FUSED_BLOCK) {     // chunks with more than 1024 slots
    const int a = slot_ptr[slot] - e0, bnd = slot_ptr[slot + 1] - e0;
    double acc = 0.0;
    for (int k = a; k < bnd; ++k) acc += pr[k];
    part[dest ? dest[slot] : slot] = acc;
  }
}

// The residual after a block-hybrid Gauss-Seidel sweep from zero, r = c .* x - A_rest x, fused with the chunk-local
// restriction like sell_pre_restrict_kernel<.., 1>, for the WINDOWED SELL form (rows of a 512-row window stored by
// decreasing length): A_rest has ragged rows (each row lost its in-block lower-colour couplings), plain slices pad ~27 %.
template <int WB, int EPT = 4>
__global__ __launch_bounds__(WB) void sell_win_cres_restrict_kernel(int64_t n_rows, SellMat M, const uint16_t* __restrict__ rowloc,
                                                                    const double* __restrict__ x, const double* __restrict__ cvec,
                                                                    const int32_t* __restrict__ chunk_slot, const int32_t* __restrict__ slot_ptr,
                                                                    const double* __restrict__ w, const uint16_t* __restrict__ fi,
                                                                    double* __restrict__ part, const int32_t* __restrict__ dest) {
  constexpr int MAXE = EPT * WB;
  __shared__ double buf[WB];
  __shared__ double pr[MAXE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int c = sell_unit(M);
  const int s = __builtin_amdgcn_readfirstlane(c * (WB / WAVE) + (threadIdx.x >> 6));
  const int64_t slot = (int64_t)s * WAVE + lane;
  const int64_t row = (int64_t)c * WB + threadIdx.x;
  const int s0 = chunk_slot[c], s1 = chunk_slot[c + 1];
  const int e0 = slot_ptr[s0], e1 = slot_ptr[s1];
  double wq[MAXE / WB];
  int fq[MAXE / WB];
#pragma unroll
  for (int q = 0; q < MAXE / WB; ++q) {
    const int e = e0 + threadIdx.x + q * WB;
    wq[q] = e < e1 ? ld_nt(w + e) : 0.0;
    fq[q] = e < e1 ? (int)ld_nt(fi + e) : 0;
  }
  const int myslot = s0 + threadIdx.x;
  int mydest = myslot, pa_raw = 0, pb_raw = 0;               // (consumed after the row product: see sell_pre_restrict_kernel)
  if (myslot < s1) {
    if (dest) mydest = dest[myslot];
    pa_raw = slot_ptr[myslot];
    pb_raw = slot_ptr[myslot + 1];
  }
  double ci = 0.0, xi = 0.0;
  if (row < n_rows) { ci = cvec[row]; xi = x[row]; }
  if (slot < n_rows) buf[rowloc[slot]] = sell_row_dot(M, s, lane, 0, x);
  __syncthreads();
  const double r = row < n_rows ? ci * xi - buf[threadIdx.x] : 0.0;
  buf[threadIdx.x] = r;                      // (same thread, same entry: the residuals replace the row sums in place)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXE / WB; ++q) {
    const int e = threadIdx.x + q * WB;
    if (e0 + e < e1) pr[e] = wq[q] * buf[fq[q]];
  }
  __syncthreads();
  if (myslot < s1) {
    const int pa = pa_raw - e0, pb = pb_raw - e0;
    double acc = 0.0;
    for (int k = pa; k < pb; ++k) acc += pr[k];
    part[mydest] = acc;
  }
  for (int sl = myslot + WB; sl < s1; sl += WB) {
    const int a = slot_ptr[sl] - e0, bnd = slot_ptr[sl + 1] - e0;
    double acc = 0.0;
    for (int k = a; k < bnd; ++k) acc += pr[k];
    part[dest ? dest[sl] : sl] = acc;
  }
}

// Jacobi pre-smoothing from zero + residual + chunk-local restriction (sell_pre_restrict_kernel, MODE 0) for the WINDOWED SELL form
// of A': the coarser levels of a reference-shaped hierarchy have ragged rows (34 ... 63 entries at the 1.24 M-row level of cfg 2),
// plain 64-row slices pad 20 %, length-sorted 512-row windows 3.6 %.  No diagonal-first trick here (the rows of a window are stored
// by decreasing length): b and dinv of the own row are read in natural order, which on these levels is 1 % of the matrix stream.
template <int WB, int EPT = 4>
__global__ __launch_bounds__(WB) void sell_win_pre_restrict_kernel(int64_t n_rows, int win0, SellMat M, const uint16_t* __restrict__ rowloc,
                                                                   const double* __restrict__ b, const double* __restrict__ dinv, double omega, int nt,
                                                                   double* __restrict__ x,
                                                                   const int32_t* __restrict__ chunk_slot, const int32_t* __restrict__ slot_ptr,
                                                                   const double* __restrict__ w, const uint16_t* __restrict__ fi,
                                                                   double* __restrict__ part, const int32_t* __restrict__ dest) {
  constexpr int MAXE = EPT * WB;
  __shared__ double buf[WB];
  __shared__ double pr[MAXE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int c = win0 + sell_unit(M);
  const int s = __builtin_amdgcn_readfirstlane(c * (WB / WAVE) + (threadIdx.x >> 6));
  const int64_t slot = (int64_t)s * WAVE + lane;
  const int64_t row = (int64_t)c * WB + threadIdx.x;
  const int s0 = chunk_slot[c], s1 = chunk_slot[c + 1];
  const int e0 = slot_ptr[s0], e1 = slot_ptr[s1];
  double wq[MAXE / WB];
  int fq[MAXE / WB];
#pragma unroll
  for (int q = 0; q < MAXE / WB; ++q) {
    const int e = e0 + threadIdx.x + q * WB;
    wq[q] = e < e1 ? ld_nt(w + e) : 0.0;
    fq[q] = e < e1 ? (int)ld_nt(fi + e) : 0;
  }
  const int myslot = s0 + threadIdx.x;
  int mydest = myslot, pa_raw = 0, pb_raw = 0;               // (consumed after the row product: see sell_pre_restrict_kernel)
  if (myslot < s1) {
    if (dest) mydest = dest[myslot];
    pa_raw = slot_ptr[myslot];
    pb_raw = slot_ptr[myslot + 1];
  }
  double bi = 0.0, di = 0.0;
  if (row < n_rows) { bi = b[row]; di = (nt & EPF_NT) ? ld_nt(dinv + row) : dinv[row]; }
  if (slot < n_rows) buf[rowloc[slot]] = sell_row_dot(M, s, lane, 0, b);
  __syncthreads();
  double r = 0.0;
  if (row < n_rows) {
    r = bi - buf[threadIdx.x];
    double xi = omega * (di * bi);
    if (nt & EPF_FOLD) xi += omega * (di * r);
    if (nt & EPF_NT) __builtin_nontemporal_store(xi, x + row);
    else x[row] = xi;
  }
  buf[threadIdx.x] = r;                      // (same thread, same entry: the residuals replace the row sums in place)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXE / WB; ++q) {
    const int e = threadIdx.x + q * WB;
    if (e0 + e < e1) pr[e] = wq[q] * buf[fq[q]];
  }
  __syncthreads();
  if (myslot < s1) {
    const int pa = pa_raw - e0, pb = pb_raw - e0;
    double acc = 0.0;
    for (int k = pa; k < pb; ++k) acc += pr[k];
    part[mydest] = acc;
  }
  for (int sl = myslot + WB; sl < s1; sl += WB) {
    const int a = slot_ptr[sl] - e0, bnd = slot_ptr[sl + 1] - e0;
    double acc = 0.0;
    for (int k = a; k < bnd; ++k) acc += pr[k];
    part[dest ? dest[sl] : sl] = acc;
  }
}

// Fused Jacobi pre-smoothing + residual + chunk-local restriction (sell_pre_restrict_kernel, MODE 0) for the LONG-ROW levels of a
// reference-shaped hierarchy, with the gathered vector staged in LDS ("local window" image).  On those levels (cfg 2, level 1:
// 1.24 M rows x 52 entries) 30 % of the plain kernel's time are scattered gathers: a wave step reads entry k of 64 different rows,
// up to 64 cache lines, and every gathered double drags a line from L2 to the CU (tools/gather_probe.py: 186 us with the real
// columns, 130 us with perfectly coalesced ones).  Here a 512-lane workgroup owns LW_ROWS = 256 consecutive rows (two lanes per
// row); the image stores, per chunk, the sorted list of the DISTINCT columns its rows touch (~3 900 of 13 300 entries) and 16-bit
// indices into that list.  The workgroup loads b at those columns once -- sorted, hence in runs, a quarter of the gathers -- into
// LDS and the row products gather from LDS (ds_read_b64) instead of L2.  Everything else as in sell_pre_restrict_kernel.
// Counters (profiles/r04/pmc_l1_*.csv): the plain kernel spends 40 % of its wave cycles stalled on instruction ISSUE (SQ_WAIT_INST_ANY;
// level 0: 24 %) -- the address path takes one cycle per distinct line of a gather -- while its fabric traffic equals the algorithmic
// bytes (no over-fetch) and the vector L1 hits 84 %.  Measured at cfg 2: level-1 down kernel 209 -> 157 us.
constexpr int LW_CAP = 4608;                 // distinct columns per chunk the LDS window holds (36 KB); chunks beyond it: 32-bit global columns
// G = lanes per row (2: 256-row chunks; 4: 128-row chunks for levels with ~80+ entries per row, whose 256-row chunks would not fit).
// MODE 0: Jacobi pre-smoothing from zero as described above (b = right-hand side, dinv, x receives omega*Dinv*b [+ fold]).
// MODE 1: residual after a block-hybrid Gauss-Seidel sweep from zero, r = c .* x - A_rest x (b = the swept x, dinv = c; nothing is
//         written to x) -- sell_win_cres_restrict_kernel's job on a local-window image of A_rest.
template <int EPT = 4, int G = 2, int MODE = 0>
__global__ __launch_bounds__(512, (EPT <= 2 ? 6 : 4)) void sell_lw_pre_restrict_kernel(int64_t n_rows, int chunk0, int n_slices, SellMat M,
                                                                   const int32_t* __restrict__ lw_cptr, const int32_t* __restrict__ lw_ccol,
                                                                   const double* __restrict__ b, const double* __restrict__ dinv,
                                                                   double omega, int nt, double* __restrict__ x,
                                                                   const int32_t* __restrict__ chunk_slot, const int32_t* __restrict__ slot_ptr,
                                                                   const double* __restrict__ w, const uint16_t* __restrict__ fi,
                                                                   double* __restrict__ part, const int32_t* __restrict__ dest) {
  constexpr int FB = 512;
  constexpr int ROWS = FB / G;
  constexpr int MAXE = EPT * FB;
  __shared__ double xw[LW_CAP];
  __shared__ double rl[ROWS];
  __shared__ double pr[MAXE];
  const int lane = threadIdx.x & (WAVE - 1);
  const int c = chunk0 + sell_unit(M);
  const int s = __builtin_amdgcn_readfirstlane(c * (FB / WAVE) + (threadIdx.x >> 6));
  const int row = s * (WAVE / G) + lane / G;
  const int lrow = (threadIdx.x >> 6) * (WAVE / G) + lane / G;
  const bool writer = (lane % G) == 0;
  // stage the window: xw[k] = b[ccol[k]]  (all loads of the prologue are requested before anything is consumed)
  const int k0 = lw_cptr[c], k1 = lw_cptr[c + 1];
  double xv[LW_CAP / FB];
#pragma unroll
  for (int q = 0; q < LW_CAP / FB; ++q) {
    const int k = k0 + threadIdx.x + q * FB;
    xv[q] = k < k1 ? b[lw_ccol[k]] : 0.0;
  }
  const bool has_slice = s < n_slices;
  const int64_t sp0 = has_slice ? M.slice_ptr[s] : 0, sp1 = has_slice ? M.slice_ptr[s + 1] : 0;
  const int s0 = chunk_slot[c], s1 = chunk_slot[c + 1];
  const int e0 = slot_ptr[s0], e1 = slot_ptr[s1];
  double wq[EPT];
  int fq[EPT];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = e0 + threadIdx.x + q * FB;
    wq[q] = e < e1 ? ld_nt(w + e) : 0.0;
    fq[q] = e < e1 ? (int)ld_nt(fi + e) : 0;
  }
  const int myslot = s0 + threadIdx.x;
  int mydest = myslot, pa_raw = 0, pb_raw = 0;
  if (myslot < s1) {
    if (dest) mydest = dest[myslot];
    pa_raw = slot_ptr[myslot];
    pb_raw = slot_ptr[myslot + 1];
  }
  double bi = 0.0, di = 0.0;
  if (writer && has_slice && row < n_rows) { bi = b[row]; di = (MODE == 0 && (nt & EPF_NT)) ? ld_nt(dinv + row) : dinv[row]; }
#pragma unroll
  for (int q = 0; q < LW_CAP / FB; ++q) {
    const int k = threadIdx.x + q * FB;
    if (k0 + k < k1) xw[k] = xv[q];
  }
  __syncthreads();
  double r = 0.0;
  if (has_slice) {
    double xd[2] = {0.0, 0.0};
    double acc = sell_row_dot_sp(M, sp0, sp1, lane, 0, xw, xd, b);       // 16-bit slices: indices into the window; 32-bit: global columns
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
    if (writer && row < n_rows) {
      if (MODE == 1) r = di * bi - acc;
      else {
        r = bi - acc;
        double xi = omega * (di * bi);
        if (nt & EPF_FOLD) xi += omega * (di * r);
        if (nt & EPF_NT) __builtin_nontemporal_store(xi, x + row);
        else x[row] = xi;
      }
    }
  }
  if (writer) rl[lrow] = r;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int e = threadIdx.x + q * FB;
    if (e0 + e < e1) pr[e] = wq[q] * rl[fq[q]];
  }
  __syncthreads();
  if (myslot < s1) {
    const int pa = pa_raw - e0, pb = pb_raw - e0;
    double acc = 0.0;
    for (int k = pa; k < pb; ++k) acc += pr[k];
    part[mydest] = acc;
  }
  for (int slot = myslot + FB; slot < s1; slot += FB) {
    const int a = slot_ptr[slot] - e0, bnd = slot_ptr[slot + 1] - e0;
    double acc = 0.0;
    for (int k = a; k < bnd; ++k) acc += pr[k];
    part[dest ? dest[slot] : slot] = acc;
  }
}

// x = z + Q x_c (windowed SELL, EP_AXPY and friends) with the gathered COARSE vector staged in LDS: the local-window form of
// sell_win_spmv_kernel.  Counters of the plain kernel on Q (profiles/r04/pmc_*): 40 % (level 0) / 58 % (level 1) of the wave cycles
// stalled on instruction issue -- 14 / 30 gathers per row whose 64 lanes hit 8 ... 30 different lines, one address-path cycle each.
// The 512 fine rows of a window interpolate from only ~400 coarse columns: their values are loaded once (sorted list, ~3 KB) and
// the row products gather from LDS.  Windows whose list would not fit keep 32-bit global columns.
constexpr int QW_CAP = 2048;                 // distinct coarse columns per window the LDS stage holds (16 KB)
template <int WB, int EP>
__global__ __launch_bounds__(WB) void sell_lw_win_spmv_kernel(int64_t n_rows, int win0, SellMat M, const uint16_t* __restrict__ rowloc,
                                                              const int32_t* __restrict__ lw_cptr, const int32_t* __restrict__ lw_ccol,
                                                              const double* __restrict__ x, double* y, EpArgs ep) {
  __shared__ double buf[WB];
  __shared__ double xw[QW_CAP];
  const int lane = threadIdx.x & (WAVE - 1);
  const int wb = win0 + sell_unit(M);
  const int s = __builtin_amdgcn_readfirstlane(wb * (WB / WAVE) + (threadIdx.x >> 6));
  const int64_t slot = (int64_t)s * WAVE + lane;
  const int64_t row = (int64_t)wb * WB + threadIdx.x;
  const int k0 = lw_cptr[wb], k1 = lw_cptr[wb + 1];
  double xv[QW_CAP / WB];
#pragma unroll
  for (int q = 0; q < QW_CAP / WB; ++q) {
    const int k = k0 + threadIdx.x + q * WB;
    xv[q] = k < k1 ? x[lw_ccol[k]] : 0.0;
  }
  const bool hoist = (ep.nt & EPF_HOIST) && EP != EP_MULT;
  EpOps ops{0.0, 0.0, 0.0};
  if (hoist && row < n_rows) ops = ep_operands<EP>(row, ep, false);
#pragma unroll
  for (int q = 0; q < QW_CAP / WB; ++q) {
    const int k = threadIdx.x + q * WB;
    if (k0 + k < k1) xw[k] = xv[q];
  }
  __syncthreads();
  if (slot < n_rows) {
    double xd[2];
    buf[rowloc[slot]] = sell_row_dot_sp(M, M.slice_ptr[s], M.slice_ptr[s + 1], lane, 0, xw, xd, x);
  }
  __syncthreads();
  if (row < n_rows) {
    if (!hoist) ops = ep_operands<EP>(row, ep, false);
    store_scalar_ops<EP>(row, buf[threadIdx.x], y, ep, ops, false, 0.0);
  }
}

// adds the partial sums of every coarse row: RSUM_G lanes per row (the partials of a row sit in different chunks,
// i.e. in unrelated cache lines: lanes in parallel instead of one thread walking them)
constexpr int RSUM_G = 8;
// RSUM_R consecutive coarse rows per lane group: the kernel is pure latency (counters: 86 % of the wave cycles parked on memory, three
// dependent round trips per wave -- row pointers, slot indices, partial sums -- and 4 TB/s of fabric traffic), so every lane carries the
// loads of RSUM_R rows through each round trip instead of one
constexpr int RSUM_R = 4;
__global__ __launch_bounds__(BLOCK) void restrict_sum_kernel(int64_t n_coarse, const int32_t* __restrict__ optr,
                                                             const int32_t* __restrict__ oidx,
                                                             const double* __restrict__ part, double* __restrict__ bc) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t J0 = (t / RSUM_G) * RSUM_R;
  const int sub = (int)(t % RSUM_G);
  int kb[RSUM_R], ke[RSUM_R];
#pragma unroll
  for (int r = 0; r < RSUM_R; ++r) {
    const int64_t J = J0 + r;
    kb[r] = J < n_coarse ? optr[J] + sub : 0;
    ke[r] = J < n_coarse ? optr[J + 1] : 0;
  }
  double acc[RSUM_R];
  if (oidx) {
    // a row has ~10 partials over RSUM_G lanes: the first two per lane are requested together (index, then value), so
    // the common case is two dependent round trips instead of four
    int i0[RSUM_R], i1[RSUM_R];
#pragma unroll
    for (int r = 0; r < RSUM_R; ++r) {
      i0[r] = kb[r] < ke[r] ? oidx[kb[r]] : -1;
      i1[r] = kb[r] + RSUM_G < ke[r] ? oidx[kb[r] + RSUM_G] : -1;
    }
#pragma unroll
    for (int r = 0; r < RSUM_R; ++r) {
      const double p0 = i0[r] >= 0 ? part[i0[r]] : 0.0, p1 = i1[r] >= 0 ? part[i1[r]] : 0.0;
      acc[r] = p0 + p1;
    }
#pragma unroll
    for (int r = 0; r < RSUM_R; ++r)
      for (int k = kb[r] + 2 * RSUM_G; k < ke[r]; k += RSUM_G) acc[r] += part[oidx[k]];
  } else {
#pragma unroll
    for (int r = 0; r < RSUM_R; ++r) { acc[r] = 0.0; for (int k = kb[r]; k < ke[r]; k += RSUM_G) acc[r] += part[k]; }      // partials stored row by row (dest)
  }
  // fixed combination order: deterministic
#pragma unroll
  for (int r = 0; r < RSUM_R; ++r) {
#pragma unroll
    for (int o = RSUM_G >> 1; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, RSUM_G);
    if (J0 + r < n_coarse && sub == 0) bc[J0 + r] = acc[r];
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

// v = 0 / dst = src as kernels of our own (two entries per lane; in a captured cycle they are plain kernel nodes, where the
// runtime's memset / memcpy nodes are blit launches with their own tail paths)
__global__ __launch_bounds__(BLOCK) void vec_zero_kernel(int64_t n, double* __restrict__ v) {
  const int64_t i = 2 * ((int64_t)blockIdx.x * BLOCK + threadIdx.x);
  if (i + 1 < n) {
    if ((reinterpret_cast<uintptr_t>(v) & 15) == 0) *reinterpret_cast<double2*>(v + i) = make_double2(0.0, 0.0);
    else { v[i] = 0.0; v[i + 1] = 0.0; }
  } else if (i < n) v[i] = 0.0;
}
__global__ __launch_bounds__(BLOCK) void vec_copy_kernel(int64_t n, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t i = 2 * ((int64_t)blockIdx.x * BLOCK + threadIdx.x);
  if (i + 1 < n) {
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0)
      *reinterpret_cast<double2*>(dst + i) = *reinterpret_cast<const double2*>(src + i);
    else { const double a = src[i], b = src[i + 1]; dst[i] = a; dst[i + 1] = b; }
  } else if (i < n) dst[i] = src[i];
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

// ---------------------------------------------------------------------------------------------------
// Coarse tail of the Jacobi V-cycle in ONE workgroup: the levels with <= TAIL_MAX_ROWS rows are launch-latency bound
// (~4.7 us per dependent graph node, 11+ nodes), so their whole down-sweep, the dense coarse solve and the up-sweep run
// as a list of row-parallel operations separated by workgroup barriers.  Matrices are plain CSR here.
constexpr int TAIL_BLOCK = 1024;
constexpr int TAIL_G = 8;
constexpr int TAIL_MAX_ROWS = 256;     // Jacobi: one workgroup is latency-bound beyond this (1261-row level: 4x slower than separate kernels)
constexpr int TAIL_MAX_ROWS_GS = 256;  // Gauss-Seidel: measured no gain for larger levels either (54 colour phases in one workgroup ~ 118 launches)
enum TailType : int { T_SPMV = 0, T_DENSE = 1, T_GS = 2, T_ZERO = 3 };

struct TailOp {
  int type;          // TailType
  int ep;            // Epilogue for T_SPMV
  int n;             // rows
  const int32_t* rowptr;
  const int32_t* col;
  const double* val; // CSR values, or the dense n x n matrix
  const double* x;
  double* y;
  EpArgs args;
  // T_GS: multicolour sweep  y[row] += dinv[row] * (b[row] - A[row,:] y)  over the colour-major row list
  const int32_t* rowlist;
  const int32_t* cptr;   // [n_colors+1] ranges of rowlist
  int n_colors;
  int backward;
  int lds_ok;            // T_GS: n <= TAIL_BLOCK / TAIL_G rows of <= TAIL_G * TAIL_GS_K entries: x in LDS, rows in registers
  const int32_t* rowcolor;   // T_GS with lds_ok: colour of every row (-1: not swept)
};
constexpr int TAIL_GS_K = 8;

// own-row operands requested before the row product, consumed after it (one dependent round trip less per operation)
__device__ __forceinline__ EpOps tail_operands(int ep, int64_t row, const EpArgs& a) {
  switch (ep) {
    case EP_MULT: return EpOps{0.0, 0.0, 0.0};
    case EP_RES: return ep_operands<EP_RES>(row, a, false);
    case EP_AXPY: return ep_operands<EP_AXPY>(row, a, false);
    case EP_JAC: return ep_operands<EP_JAC>(row, a, false);
    case EP_CRES: return ep_operands<EP_CRES>(row, a, false);
    default: return ep_operands<EP_PRE>(row, a, false);
  }
}
__device__ __forceinline__ void tail_store(int ep, int64_t row, double acc, double* y, const EpArgs& a, const EpOps& o) {
  switch (ep) {
    case EP_MULT: store_scalar_ops<EP_MULT>(row, acc, y, a, o, false, 0.0); break;
    case EP_RES: store_scalar_ops<EP_RES>(row, acc, y, a, o, false, 0.0); break;
    case EP_AXPY: store_scalar_ops<EP_AXPY>(row, acc, y, a, o, false, 0.0); break;
    case EP_JAC: store_scalar_ops<EP_JAC>(row, acc, y, a, o, false, 0.0); break;
    case EP_CRES: store_scalar_ops<EP_CRES>(row, acc, y, a, o, false, 0.0); break;
    default: store_scalar_ops<EP_PRE>(row, acc, y, a, o, false, 0.0); break;
  }
}

// The kernel is one chain of dependent steps on a single CU, so every memory round trip counts: the operation list is
// copied to LDS once, an operation requests its row entries (<= TAIL_SP_K per lane in registers) and own-row operands
// together, then all gathers, and only then starts to add.
constexpr int TAIL_MAX_OPS = 48;
constexpr int TAIL_SP_K = 8;
__global__ __launch_bounds__(TAIL_BLOCK) void tail_kernel(int n_ops, const TailOp* __restrict__ ops) {
  __shared__ double xs[TAIL_BLOCK / TAIL_G];
  __shared__ TailOp lops[TAIL_MAX_OPS];
  const int tid = threadIdx.x;
  {
    constexpr int WORDS = (int)(sizeof(TailOp) / sizeof(int));
    const int total = (n_ops < TAIL_MAX_OPS ? n_ops : TAIL_MAX_OPS) * WORDS;
    const int* src = reinterpret_cast<const int*>(ops);
    int* dst = reinterpret_cast<int*>(lops);
    for (int k = tid; k < total; k += TAIL_BLOCK) dst[k] = src[k];
    __syncthreads();
  }
  for (int i = 0; i < n_ops; ++i) {
    const TailOp op = i < TAIL_MAX_OPS ? lops[i] : ops[i];
    if (op.type == T_ZERO) {
      for (int k = tid; k < op.n; k += TAIL_BLOCK) op.y[k] = 0.0;
    } else if (op.type == T_GS && op.lds_ok) {
      // small level: every lane group owns ONE row for the whole sweep, its entries sit in registers and x in LDS, so a
      // colour phase is LDS reads + one barrier instead of a chain of dependent global loads (2 us per colour)
      const int row = tid / TAIL_G, sub = tid % TAIL_G;
      const bool act = row < op.n;
      const int mycol = act ? op.rowcolor[row] : -1;
      double v[TAIL_GS_K];
      int cl[TAIL_GS_K];
      const int e0 = act ? op.rowptr[row] : 0, e1 = act ? op.rowptr[row + 1] : 0;
#pragma unroll
      for (int k = 0; k < TAIL_GS_K; ++k) {
        const int e = e0 + sub + k * TAIL_G;
        v[k] = e < e1 ? op.val[e] : 0.0;
        cl[k] = e < e1 ? op.col[e] : 0;
      }
      double dv = 0.0, bv = 0.0;
      if (act && sub == 0) { dv = op.args.dinv[row]; bv = op.args.b[row]; xs[row] = op.y[row]; }
      __syncthreads();
      for (int q = 0; q < op.n_colors; ++q) {
        const int c = op.backward ? op.n_colors - 1 - q : q;
        if (mycol == c) {                                   // uniform inside a lane group: the shuffles are safe
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < TAIL_GS_K; ++k) acc += v[k] * xs[cl[k]];
#pragma unroll
          for (int o = TAIL_G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, TAIL_G);
          if (sub == 0) xs[row] += dv * (bv - acc);
        }
        __syncthreads();
      }
      if (act && sub == 0) op.y[row] = xs[row];
    } else if (op.type == T_GS) {
      const int sub = tid % TAIL_G;
      for (int q = 0; q < op.n_colors; ++q) {
        const int c = op.backward ? op.n_colors - 1 - q : q;
        const int r1 = op.cptr[c + 1];
        for (int p = op.cptr[c] + tid / TAIL_G; p < r1; p += TAIL_BLOCK / TAIL_G) {
          const int row = op.rowlist[p];
          double acc = 0.0;
          const int e = op.rowptr[row + 1];
          for (int k = op.rowptr[row] + sub; k < e; k += TAIL_G) acc += op.val[k] * op.y[op.col[k]];
#pragma unroll
          for (int o = TAIL_G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, TAIL_G);
          if (sub == 0) op.y[row] += op.args.dinv[row] * (op.args.b[row] - acc);
        }
        __syncthreads();   // the next colour reads what this one wrote
      }
    } else if (op.type == T_DENSE) {
      const int lane = tid & (WAVE - 1);
      for (int row = tid >> 6; row < op.n; row += TAIL_BLOCK / WAVE) {
        double acc = 0.0;
        for (int c = lane; c < op.n; c += WAVE) acc += op.val[(int64_t)row * op.n + c] * op.x[c];
#pragma unroll
        for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
        if (lane == 0) op.y[row] = acc;
      }
    } else {
      const int sub = tid % TAIL_G;
      // all lanes of a group run the same trip count (row is group-uniform), so the shuffles are safe
      for (int row = tid / TAIL_G; row < op.n; row += TAIL_BLOCK / TAIL_G) {
        const int e0 = op.rowptr[row], e = op.rowptr[row + 1];
        EpOps eo{0.0, 0.0, 0.0};
        if (sub == 0) eo = tail_operands(op.ep, row, op.args);
        double v[TAIL_SP_K], xv[TAIL_SP_K];
        int cl[TAIL_SP_K];
#pragma unroll
        for (int k = 0; k < TAIL_SP_K; ++k) {
          const int q = e0 + sub + k * TAIL_G;
          v[k] = q < e ? op.val[q] : 0.0;
          cl[k] = q < e ? op.col[q] : -1;
        }
#pragma unroll
        for (int k = 0; k < TAIL_SP_K; ++k) xv[k] = cl[k] >= 0 ? op.x[cl[k]] : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < TAIL_SP_K; ++k) acc += v[k] * xv[k];
        for (int q = e0 + sub + TAIL_SP_K * TAIL_G; q < e; q += TAIL_G) acc += op.val[q] * op.x[op.col[q]];    // rows beyond 64 entries
#pragma unroll
        for (int o = TAIL_G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, TAIL_G);
        if (sub == 0) tail_store(op.ep, row, acc, op.y, op.args, eo);
      }
    }
    __syncthreads();     // workgroup-scope release/acquire: the next operation reads what this one wrote
  }
}

// ---------------------------------------------------------------------------------------------------
// Block Gauss-Seidel (reference BSmoother::Smooth_impl, block_gssmoother.cpp:287-328): one workgroup per block,
// one launch per colour of the block graph.
//   hr_j = b_j - A_j: x   for the rows j of the block        (G lanes per scalar row, reduced by shuffles)
//   hu   = Dinv_B hr      (dense M x M, column-major: lane i reads column entries D[j*M + i] -> coalesced)
//   x_B += hu
// hr is complete (barrier) before any x_B is written, exactly like the reference's two loops.
constexpr int BGS_MAX_M = 1024;      // scalar dofs per block (host side rejects bigger blocks)
// Both phases are latency-bound (a block has only ~50-120 scalar dofs): phase 2 is therefore split over all TH threads,
// S = TH / M slices of the j-range per output row, partial sums combined through LDS (a single thread per row walking
// all M columns cost ~30 dependent round trips per block).
// G = lanes per scalar row in phase 1; the host picks (TH, G) so that M * G <= TH where possible (one pass): the coarse
// levels have few, big blocks with long rows (6x6: M ~ 120, 45-60 blocks per row) and are pure latency otherwise.
template <int BS, int TH, int G>
__global__ __launch_bounds__(TH) void bgs_block_kernel(int list_begin, const int32_t* __restrict__ blocklist,
                                                       const int32_t* __restrict__ block_ptr, const int32_t* __restrict__ block_rows,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                       const double* __restrict__ vals, const int64_t* __restrict__ dinv_ptr,
                                                       const double* __restrict__ dinv, const double* __restrict__ b, double* x) {
  __shared__ double hr[BGS_MAX_M];
  __shared__ double part[TH];
  constexpr int DMAX = 16;                                  // entries of the inverse a thread keeps in registers
  const int k = blocklist[list_begin + blockIdx.x];
  const int p0 = block_ptr[k];
  const int M = (block_ptr[k + 1] - p0) * BS;
  const double* __restrict__ D = dinv + dinv_ptr[k];
  // A block is a chain of dependent round trips (block table -> rows -> row pointers -> entries -> gathers -> inverse ->
  // x); everything that does not depend on the row products is requested up front: this thread's slice of the inverse
  // and the old x value of its output row.
  const bool fits = M <= TH;
  const int S = fits ? TH / M : 1;                          // slices of the column range of the inverse
  const int di = fits ? threadIdx.x % M : 0, dsl = fits ? threadIdx.x / M : S;
  const bool dreg_ok = fits && (M + S - 1) / S <= DMAX;
  double dreg[DMAX];
  double xold = 0.0;
  int64_t xrow = 0;
  if (dreg_ok) {
    if (dsl < S) {
#pragma unroll
      for (int q = 0; q < DMAX; ++q) { const int jj = dsl + q * S; dreg[q] = jj < M ? D[(int64_t)jj * M + di] : 0.0; }
    }
    if (threadIdx.x < M) {
      xrow = (int64_t)block_rows[p0 + threadIdx.x / BS] * BS + threadIdx.x % BS;
      xold = x[xrow];                                       // x_B is not written before the barrier below
    }
  }
  const int sub = threadIdx.x % G;
  for (int t0 = 0; t0 < M; t0 += TH / G) {             // trip count is workgroup-uniform: shuffles are safe
    const int t = t0 + threadIdx.x / G;
    double acc = 0.0, bv = 0.0;
    if (t < M) {
      const int64_t row = block_rows[p0 + t / BS];
      const int rr = t % BS;
      if (sub == 0) bv = b[row * BS + rr];
      const int e = rowptr[row + 1];
#pragma unroll 4
      for (int p = rowptr[row] + sub; p < e; p += G) {
        const double* __restrict__ a = vals + ((int64_t)p * BS + rr) * BS;
        const double* xv = x + (int64_t)cols[p] * BS;
#pragma unroll
        for (int c = 0; c < BS; ++c) acc += a[c] * xv[c];
      }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
    if (t < M && sub == 0) hr[t] = bv - acc;
  }
  __syncthreads();
  if (fits) {
    double u = 0.0;
    if (dsl < S) {
      if (dreg_ok) {
#pragma unroll
        for (int q = 0; q < DMAX; ++q) { const int jj = dsl + q * S; if (jj < M) u += dreg[q] * hr[jj]; }
      } else {
#pragma unroll 8
        for (int jj = dsl; jj < M; jj += S) u += D[(int64_t)jj * M + di] * hr[jj];
      }
    }
    part[threadIdx.x] = u;
    __syncthreads();
    if (threadIdx.x < M) {
      double tot = 0.0;
      for (int q = 0; q < S; ++q) tot += part[q * M + threadIdx.x];      // fixed order: deterministic
      if (dreg_ok) x[xrow] = xold + tot;
      else {
        const int64_t row = block_rows[p0 + threadIdx.x / BS];
        x[row * BS + threadIdx.x % BS] += tot;              // x_B is only read in the first phase, which is complete
      }
    }
  } else {
    for (int i = threadIdx.x; i < M; i += TH) {
      double u = 0.0;
#pragma unroll 8
      for (int jj = 0; jj < M; ++jj) u += D[(int64_t)jj * M + i] * hr[jj];
      const int64_t row = block_rows[p0 + i / BS];
      x[row * BS + i % BS] += u;
    }
  }
}

// dst[i] = src[perm[i]] (gather) / dst[perm[i]] = src[i] (scatter) on block vectors with bs entries per block row:
// translation between the caller's numbering and the colour-major numbering of Gauss-Seidel levels (amgx.hip, LevelPerm)
__global__ __launch_bounds__(BLOCK) void perm_gather_kernel(int64_t len, int bs, const int32_t* __restrict__ perm,
                                                            const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= len) return;
  const int64_t i = t / bs;
  const int c = (int)(t - i * bs);
  dst[t] = src[(int64_t)perm[i] * bs + c];
}
__global__ __launch_bounds__(BLOCK) void perm_scatter_kernel(int64_t len, int bs, const int32_t* __restrict__ perm,
                                                             const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= len) return;
  const int64_t i = t / bs;
  const int c = (int)(t - i * bs);
  dst[(int64_t)perm[i] * bs + c] = src[t];
}

// ---------------------------------------------------------------------------------------------------
// Halo exchange buffers (reference DCCMap, src/base/linalg/dcc_map.cpp:249-302).  Block vectors are AoS (entry =
// bs*dof + comp, dcc_map.cpp:237-243); `idx` lists block rows, one thread moves one scalar entry, so consecutive
// threads write (pack) / read (unpack) consecutive buffer entries: the buffer side is always coalesced.
//   halo_pack_kernel      : buf[t] = vec[idx[t / bs] * bs + t % bs]           BufferM (:280-288): owner entries -> send buffer
//   halo_unpack_add_kernel: vec[idx[t / bs] * bs + t % bs] += buf[t]          ApplyM  (:266-274): vec += received
//   ghost segments are contiguous per peer in this layout, so BufferG (:252-260, copy ghost entries and ZERO them) is a
//   plain device copy + halo_zero_kernel, and ApplyG (:294-302, vec = received) is the receive itself (no kernel).
// An index list never repeats a row inside one call (a peer's list is a set; lists of different peers are unpacked by
// separate launches), so the += is race-free.
__global__ __launch_bounds__(BLOCK) void halo_pack_kernel(int64_t len, int bs, const int32_t* __restrict__ idx,
                                                          const double* __restrict__ vec, double* __restrict__ buf) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= len) return;
  const int64_t i = t / bs;
  buf[t] = vec[(int64_t)idx[i] * bs + (t - i * bs)];
}
__global__ __launch_bounds__(BLOCK) void halo_unpack_add_kernel(int64_t len, int bs, const int32_t* __restrict__ idx,
                                                                const double* __restrict__ buf, double* __restrict__ vec) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t >= len) return;
  const int64_t i = t / bs;
  vec[(int64_t)idx[i] * bs + (t - i * bs)] += buf[t];
}
__global__ __launch_bounds__(BLOCK) void halo_zero_kernel(int64_t len, double* __restrict__ v) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t < len) v[t] = 0.0;
}
// 64-bit index gather (all-gather compaction, level-k [owned | ghost] pick from the replicated tail solution)
__global__ __launch_bounds__(BLOCK) void index_gather_kernel(int64_t len, const int64_t* __restrict__ idx,
                                                             const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (t < len) dst[t] = src[idx[t]];
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

// ---------------------------------------------------------------------------------------------------
// Rigid-body transfer blocks.  The prolongation blocks of the elasticity (and vector-H1) hierarchies are not general
// matrices: P_ik = w_ik Q(t_ik) with the rigid-body transformation Q(t) = [I, -skew(t); 0, I] of the offset t between the
// fine vertex and the coarse vertex (reference src/elasticity/elasticity_energy.hpp:447-490, the aux-smoothed
// prolongation multiplies it with a SCALAR weight, vertex_factory_impl.hpp:1968-2020).  Streaming the 6 x 6 block costs
// 292 bytes per entry, its generator (column, w, t) 36: the transfers of cfg 5 shrink from 1.9 GB to 0.24 GB per pass and
// the block is rebuilt in registers (u_f = w (u_c + o_c x t), o_f = w o_c).  amgx_create detects the structure block by
// block and keeps the general block-CSR form for anything else.
//   DIM 3: coarse block size 6, fine 3 (displacements only) or 6;  DIM 2: coarse 3, fine 2 or 3 (w I for vector H1 = t = 0);
//   DIM 0: w I with equal block sizes
struct RbMat {
  const int32_t* ptr;         // [n_rows + 1]
  const int32_t* col;         // [nnz]
  const double* w;            // [nnz]
  const double* t;            // [DIM][nnz] (structure of arrays)
  int64_t nnz;
};
template <int BF, int BC, int DIM>
__device__ __forceinline__ void rb_forward(const double* __restrict__ xc, double w, double t0, double t1, double t2, double* acc) {
  // acc[0..BF) += w Q(t) xc
  if (DIM == 3) {
    const double o0 = xc[3], o1 = xc[4], o2 = xc[5];
    acc[0] += w * (xc[0] + o1 * t2 - o2 * t1);
    acc[1] += w * (xc[1] + o2 * t0 - o0 * t2);
    acc[2] += w * (xc[2] + o0 * t1 - o1 * t0);
    if (BF == 6) { acc[3] += w * o0; acc[4] += w * o1; acc[5] += w * o2; }
  } else if (DIM == 2) {
    const double o = xc[2];
    acc[0] += w * (xc[0] - t1 * o);
    acc[1] += w * (xc[1] + t0 * o);
    if (BF == 3) acc[2] += w * o;
  } else {
#pragma unroll
    for (int r = 0; r < BF; ++r) acc[r] += w * xc[r];
  }
}
template <int BF, int BC, int DIM>
__device__ __forceinline__ void rb_transposed(const double* __restrict__ rf, double w, double t0, double t1, double t2, double* acc) {
  // acc[0..BC) += w Q(t)^T rf
  if (DIM == 3) {
    const double f0 = rf[0], f1 = rf[1], f2 = rf[2];
    acc[0] += w * f0; acc[1] += w * f1; acc[2] += w * f2;
    double m0 = t1 * f2 - t2 * f1, m1 = t2 * f0 - t0 * f2, m2 = t0 * f1 - t1 * f0;
    if (BF == 6) { m0 += rf[3]; m1 += rf[4]; m2 += rf[5]; }
    acc[3] += w * m0; acc[4] += w * m1; acc[5] += w * m2;
  } else if (DIM == 2) {
    acc[0] += w * rf[0]; acc[1] += w * rf[1];
    double m = -t1 * rf[0] + t0 * rf[1];
    if (BF == 3) m += rf[2];
    acc[2] += w * m;
  } else {
#pragma unroll
    for (int r = 0; r < BC; ++r) acc[r] += w * rf[r];
  }
}
// y = (yin) + s * P x_c : one thread per fine block row (rows have ~3 entries); EP_MULT or EP_AXPY
template <int BF, int BC, int DIM, int EP>
__global__ __launch_bounds__(BLOCK) void rb_prolong_kernel(int64_t n_rows, RbMat M, const double* __restrict__ x, double* y, EpArgs ep) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n_rows) return;
  double acc[BF];
#pragma unroll
  for (int r = 0; r < BF; ++r) acc[r] = 0.0;
  const int e = M.ptr[i + 1];
  for (int k = M.ptr[i]; k < e; ++k) {
    const double* __restrict__ xc = x + (int64_t)M.col[k] * BC;
    const double w = M.w[k];
    const double t0 = DIM >= 2 ? M.t[k] : 0.0, t1 = DIM >= 2 ? M.t[M.nnz + k] : 0.0, t2 = DIM == 3 ? M.t[2 * M.nnz + k] : 0.0;
    rb_forward<BF, BC, DIM>(xc, w, t0, t1, t2, acc);
  }
#pragma unroll
  for (int r = 0; r < BF; ++r) y[i * BF + r] = EP == EP_AXPY ? ep.yin[i * BF + r] + ep.s * acc[r] : acc[r];
}
// y = P^T r : G lanes per coarse block row (rows have 30 ... 60 entries), entries (fine row, w, t)
template <int BF, int BC, int DIM, int G, int EP>
__global__ __launch_bounds__(BLOCK) void rb_restrict_kernel(int64_t n_rows, RbMat M, const double* __restrict__ x, double* y, EpArgs ep) {
  const int64_t tg = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t J = tg / G;
  const int sub = (int)(tg % G);
  double acc[BC];
#pragma unroll
  for (int r = 0; r < BC; ++r) acc[r] = 0.0;
  if (J < n_rows) {
    const int e = M.ptr[J + 1];
    for (int k = M.ptr[J] + sub; k < e; k += G) {
      const double* __restrict__ rf = x + (int64_t)M.col[k] * BF;
      const double w = M.w[k];
      const double t0 = DIM >= 2 ? M.t[k] : 0.0, t1 = DIM >= 2 ? M.t[M.nnz + k] : 0.0, t2 = DIM == 3 ? M.t[2 * M.nnz + k] : 0.0;
      rb_transposed<BF, BC, DIM>(rf, w, t0, t1, t2, acc);
    }
  }
#pragma unroll
  for (int r = 0; r < BC; ++r)
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, G);
  if (J < n_rows && sub == 0) {
#pragma unroll
    for (int r = 0; r < BC; ++r) y[J * BC + r] = EP == EP_AXPY ? ep.yin[J * BC + r] + ep.s * acc[r] : acc[r];
  }
}

// ---------------------------------------------------------------------------------------------------
// Collapsed coarse levels.  The V-cycle restricted to the levels >= l_c is a fixed LINEAR operator x_lc = B b_lc
// (smoothers, transfers and the coarse inverse of amg_matrix.cpp:183-302 are all linear).  Where those levels are
// launch-latency bound (a few hundred ... thousand unknowns, 3 .. 60 dependent launches of ~5 us) the operator is
// formed ONCE at amgx_create by running the device's own sub-cycle on the unit vectors, and one application becomes
// ONE dense GEMV that streams n^2 * 8 bytes (n = 1261 at cfg 2: 12.7 MB, ~6 us) -- 288 GB of HBM buy latency.
//   dense_unit_kernel      : v = e_j
//   dense_transpose_kernel : B[i][j] = Bt[j][i]   (Bt row j = B e_j as the sub-cycle delivers it)
//   dense_op_gemv_kernel   : y = B x, one wave per row, 16-byte loads, rows padded to an even length `ld` (x needs 16-byte alignment)
__global__ __launch_bounds__(BLOCK) void dense_unit_kernel(int64_t n, int64_t j, double* __restrict__ v) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) v[i] = i == j ? 1.0 : 0.0;
}
__global__ __launch_bounds__(BLOCK) void dense_transpose_kernel(int n, int ld, const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ double t[16][17];
  const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if (by + ty < n && bx + tx < n) t[ty][tx] = src[(int64_t)(by + ty) * ld + bx + tx];
  __syncthreads();
  if (bx + ty < n && by + tx < n) dst[(int64_t)(bx + ty) * ld + by + tx] = t[tx][ty];
}
__global__ __launch_bounds__(BLOCK) void dense_op_gemv_kernel(int n, int ld, const double* __restrict__ M,
                                                              const double* __restrict__ x, double* __restrict__ y) {
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & (WAVE - 1);
  if (row >= n) return;
  const double2* __restrict__ m2 = reinterpret_cast<const double2*>(M + (int64_t)row * ld);
  const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x);
  const int np = n >> 1;                        // full pairs per row; an odd last entry is added by lane 0 (x has exactly n entries)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int c = lane;
  // (a wave walks its row in dependent round trips: eight 16-byte loads of M in flight per lane halve their number -- n = 2 824 at cfg 2:
  //  3 instead of 6)
  for (; c + 7 * WAVE < np; c += 8 * WAVE) {
    double2 m[8], v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = m2[c + q * WAVE];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = x2[c + q * WAVE];
    a0 += (m[0].x * v[0].x + m[0].y * v[0].y) + (m[4].x * v[4].x + m[4].y * v[4].y);
    a1 += (m[1].x * v[1].x + m[1].y * v[1].y) + (m[5].x * v[5].x + m[5].y * v[5].y);
    a2 += (m[2].x * v[2].x + m[2].y * v[2].y) + (m[6].x * v[6].x + m[6].y * v[6].y);
    a3 += (m[3].x * v[3].x + m[3].y * v[3].y) + (m[7].x * v[7].x + m[7].y * v[7].y);
  }
  for (; c + 3 * WAVE < np; c += 4 * WAVE) {    // four independent 16-byte loads in flight per lane
    const double2 m0 = m2[c], m1 = m2[c + WAVE], mm2 = m2[c + 2 * WAVE], m3 = m2[c + 3 * WAVE];
    const double2 v0 = x2[c], v1 = x2[c + WAVE], v2 = x2[c + 2 * WAVE], v3 = x2[c + 3 * WAVE];
    a0 += m0.x * v0.x + m0.y * v0.y;
    a1 += m1.x * v1.x + m1.y * v1.y;
    a2 += mm2.x * v2.x + mm2.y * v2.y;
    a3 += m3.x * v3.x + m3.y * v3.y;
  }
  for (; c < np; c += WAVE) { const double2 m0 = m2[c], v0 = x2[c]; a0 += m0.x * v0.x + m0.y * v0.y; }
  if ((n & 1) && lane == 0) a1 += M[(int64_t)row * ld + (n - 1)] * x[n - 1];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
  if (lane == 0) y[row] = acc;
}

}  // namespace amgx
