// Device-side builders of the big scalar level images (included by amgx.hip after the host builders).
//
// amgx_create used to form every image on the host: at cfg 2 (148 M entries) the SELL images of A and A', the product
// Q = (I - omega Dinv A) P (reference: the sparse products of utils_sparseMM.cpp:107-238 on the setup side) and its windowed
// image were 2.6 s of host loops behind 4 GB of pageable host-to-device copies.  Here the CSR arrays of A and P go to the
// device once and kernels write the images where they are used:
//     db_width / db_scan / db_fill     CSR -> SELL-64-pair image (one thread per row; natural order or a row list)
//     db_window_sort                   row order of the windowed form (rows of a 512-row window by decreasing length, stable)
//     db_fold<false|true>              row-wise sparse product Q = P - omega Dinv (A P): count, then fill (columns ascending)
// Every kernel restates its host builder (build_sell, upload_matrix's window sort, fold_prolongation) decision by decision and
// in the same floating-point order without contraction, so the images are bit-identical: AMGX_VERIFY_IMAGES=1 builds both and
// compares every array; AMGX_HOST_IMAGES=1 keeps the host builders.
#pragma once

namespace amgx {

struct DevCsrSrc {                      // a scalar CSR matrix on the device, 64-bit row pointers as in amgx_matrix
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  DevBuf<int64_t> rowptr;
  DevBuf<int32_t> col;
  DevBuf<double> val;
  void upload(const amgx_matrix& A) {
    n_rows = A.n_rows; n_cols = A.n_cols; nnz = A.rowptr[A.n_rows];
    rowptr.upload(A.rowptr, (size_t)A.n_rows + 1);
    col.upload(A.col, (size_t)std::max<int64_t>(1, nnz));
    val.upload(A.val, (size_t)std::max<int64_t>(1, nnz));
  }
};

constexpr int DB_BLOCK = 256;                    // 4 slices (waves) per workgroup
constexpr int DB_FOLD_CAP = 64;                  // distinct coarse columns of one row of Q the product kernel holds (else: host builder)

// width of every slice of R = 64 / G list entries (G lanes per row): sp[s + 1] = 64 * w, w = steps of the longest row
// (build_sell: G == 1: its length; G > 1: 2 * ceil(ceil(len / 2) / G)), sp[0] = 0
__global__ __launch_bounds__(DB_BLOCK) void db_width_kernel(int64_t m, const int32_t* __restrict__ rows, const int64_t* __restrict__ rowptr,
                                                            int G, int64_t ns, int64_t* __restrict__ sp) {
  const int64_t s = (int64_t)blockIdx.x * (DB_BLOCK / WAVE) + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (s >= ns) return;
  const int64_t q = s * (WAVE / G) + l / G;
  int len = 0;
  if (q < m) {
    const int64_t r = rows ? (int64_t)rows[q] : q;
    if (r >= 0) len = (int)(rowptr[r + 1] - rowptr[r]);
  }
  for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
  const int w = G == 1 ? len : 2 * (((len + 1) / 2 + G - 1) / G);
  if (l == 0) { sp[s + 1] = (int64_t)w * WAVE; if (s == 0) sp[0] = 0; }
}

// in-place inclusive sum of v[1..n] (v[0] stays): one workgroup, a contiguous piece per thread
__global__ __launch_bounds__(1024) void db_scan_kernel(int64_t n, int64_t* __restrict__ v) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t a = 1 + (int64_t)t * chunk, b = min(n + 1, a + chunk);
  int64_t s = 0;
  for (int64_t i = a; i < b; ++i) s += v[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int64_t x = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  int64_t run = t ? part[t - 1] : 0;
  for (int64_t i = a; i < b; ++i) { run += v[i]; v[i] = run; }
}

struct DbFill {
  int64_t m, n_cols, ns;
  const int32_t* rows;
  int G, rowrel, diag_first;
  const int64_t* rowptr; const int32_t* col; const double* val;
  const double* colscale; double omega;        // image of A' = A * diag(omega * colscale): value = val * (omega * colscale[col])
  const double* wdiag; int64_t wdiag_rows;      // diagonal-first image whose diagonal slot carries omega * wdiag[row] (patch_sell_diag)
  const int64_t* sp;
  int32_t* col32; uint16_t* col16; int32_t* cbase; double* out;
  uint8_t* comp; unsigned long long* counters;  // [0] slices in the 16-bit form, [1] bytes one product streams
  const int32_t* vcol;                          // (local-window images: col = window-local index) the global column colscale is read at
  const uint8_t* no16;                          // per row: slices holding such a row keep the 32-bit encoding (build_sell's no16)
};

// one wave per slice, G lanes per list entry: build_sell's fill pass
__global__ __launch_bounds__(DB_BLOCK) void db_fill_kernel(DbFill a) {
  const int64_t s = (int64_t)blockIdx.x * (DB_BLOCK / WAVE) + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (s >= a.ns) return;
  const int64_t base = a.sp[s];
  const int w = (int)((a.sp[s + 1] - base) / WAVE);
  const int wp = w & ~1;
  const int G = a.G;
  const int64_t q = s * (WAVE / G) + l / G;
  const int64_t r = q < a.m ? (a.rows ? (int64_t)a.rows[q] : q) : -1;
  const int64_t rb = r >= 0 ? a.rowptr[r] : 0;
  const int len = r >= 0 ? (int)(a.rowptr[r + 1] - rb) : 0;
  int dp = 0;
  if (a.diag_first && r >= 0)
    for (int k = 0; k < len; ++k) if (a.col[rb + k] == r) { dp = k; break; }
  const int64_t rk = a.rows ? r : q;
  const int64_t rr = a.rowrel ? rk : 0;
  const int32_t padcol = (r >= 0 && len) ? a.col[rb] : 0;
  bool comp = !(a.no16 && r >= 0 && a.no16[r]);
  for (int j = 0; j < w; ++j) {
    const int e = G == 1 ? j : 2 * ((j >> 1) * G + (l % G)) + (j & 1);      // (lane, column) -> entry of the lane's row
    const bool has = e < len;
    int64_t c = 0;
    double v = 0.0;
    if (has) {
      const int se = !a.diag_first ? e : (e == 0 ? dp : (e <= dp ? e - 1 : e));
      const int64_t ks = rb + se;
      c = a.col[ks];
      v = a.val[ks];
      if (a.colscale) v = v * (a.omega * a.colscale[a.vcol ? (int64_t)a.vcol[ks] : c]);
      if (a.wdiag && e == 0 && r < a.wdiag_rows) v = a.omega * a.wdiag[r];
    }
    long long mn = has ? (long long)(c - (a.rowrel ? r : 0)) : LLONG_MAX;
    for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
    const int64_t cb = mn == LLONG_MAX ? 0 : (int64_t)mn;
    if (cb < INT32_MIN / 2 || cb > INT32_MAX / 2) comp = false;
    if (l == 0) a.cbase[base / WAVE + j] = (int32_t)cb;
    const int64_t o = j < wp ? base + (int64_t)(j >> 1) * (2 * WAVE) + l * 2 + (j & 1) : base + (int64_t)(w - 1) * WAVE + l;
    if (has) {
      a.col32[o] = (int32_t)c;
      a.out[o] = v;
      const int64_t d = c - rr - cb;
      if (d < 0 || d > 65535) comp = false;
      a.col16[o] = (uint16_t)d;
    } else {
      a.col32[o] = padcol;
      a.out[o] = 0.0;
      if (a.rows && r < 0) { a.col16[o] = 0; continue; }          // lane never executes
      int64_t d = (int64_t)padcol - rr - cb;
      if (d < 0 || d > 65535) {
        d = max((int64_t)0, -(rr + cb));
        if (d > 65535 || rr + cb + d >= a.n_cols) comp = false;
      }
      a.col16[o] = (uint16_t)d;
    }
  }
  const bool all = __all(comp ? 1 : 0) != 0;
  if (l == 0) {
    const bool flag = all && w > 0;
    a.comp[s] = flag ? 1 : 0;
    atomicAdd(a.counters + 0, flag ? 1ull : 0ull);
    atomicAdd(a.counters + 1, flag ? (unsigned long long)((int64_t)w * WAVE * 10 + 4 * w) : (unsigned long long)((int64_t)w * WAVE * 12));
  }
}

// encoding flag of slice s into bit 0 of its offset (after every slice has read the plain offsets)
__global__ __launch_bounds__(BLOCK) void db_flag_kernel(int64_t ns, const uint8_t* __restrict__ comp, int64_t* __restrict__ sp) {
  const int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (s < ns && comp[s]) sp[s] |= 1;
}

// flags[0]: a row without a stored diagonal; flags[1]: a row whose dinv is not the plain inverse of its diagonal
__global__ __launch_bounds__(BLOCK) void db_diag_check_kernel(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const double* __restrict__ val, const double* __restrict__ dinv, int* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  bool found = false;
  double aii = 0.0;
  for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) if (col[k] == i) { aii = val[k]; found = true; break; }
  if (!found) flags[0] = 1;
  if (dinv && dinv[i] != 0.0 && !(fabs(dinv[i] * aii - 1.0) < 1e-13)) flags[1] = 1;
}

// rows of every window of WIN consecutive rows in order of decreasing length, equal lengths in row order (std::stable_sort)
template <int WIN>
__global__ __launch_bounds__(WIN) void db_window_sort_kernel(int64_t n_rows, const int64_t* __restrict__ rowptr, int32_t* __restrict__ rows,
                                                             uint16_t* __restrict__ rowloc) {
  __shared__ int lens[WIN];
  const int64_t w0 = (int64_t)blockIdx.x * WIN;
  const int t = threadIdx.x;
  const int cnt = (int)min((int64_t)WIN, n_rows - w0);
  const int len = t < cnt ? (int)(rowptr[w0 + t + 1] - rowptr[w0 + t]) : -1;
  lens[t] = len;
  __syncthreads();
  if (t >= cnt) return;
  int rank = 0;
  for (int u = 0; u < cnt; ++u) { const int lu = lens[u]; rank += (lu > len) || (lu == len && u < t); }
  rows[w0 + rank] = (int32_t)(w0 + t);
  rowloc[w0 + rank] = (uint16_t)t;
}

struct DbFold {
  int64_t n, p_rows;
  const int64_t* arp; const int32_t* acol; const double* aval;
  const int64_t* prp; const int32_t* pcol; const double* pval;
  const double* dinv; double omega;
};

// fold_prolongation, one thread per row.  FILL = false: qrp[i + 1] = distinct columns of row i.  FILL = true (qrp scanned):
// columns ascending, value = P_ic - omega * (Dinv_i * sum_j A_ij P_jc), the sum in the order of the entries of A_i and P_j.
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void db_fold_kernel(DbFold a, int64_t* __restrict__ qrp, int32_t* __restrict__ qcol, double* __restrict__ qval,
                                                        int* __restrict__ overflow) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  int32_t cols[DB_FOLD_CAP];
  double own[FILL ? DB_FOLD_CAP : 1], acc[FILL ? DB_FOLD_CAP : 1];
  int nc = 0;
  bool over = false;
  auto slot = [&](int32_t c) -> int {
    for (int q = 0; q < nc; ++q) if (cols[q] == c) return q;
    if (nc == DB_FOLD_CAP) { over = true; return -1; }
    cols[nc] = c;
    if (FILL) { own[nc] = 0.0; acc[nc] = 0.0; }
    return nc++;
  };
  for (int64_t k = a.prp[i]; k < a.prp[i + 1]; ++k) {
    const int sl = slot(a.pcol[k]);
    if (FILL && sl >= 0) own[sl] += a.pval[k];
  }
  const double d = a.dinv[i];
  if (d != 0.0)
    for (int64_t k = a.arp[i]; k < a.arp[i + 1]; ++k) {
      const int64_t j = a.acol[k];
      if (j >= a.p_rows) continue;
      const double av = a.aval[k];
      for (int64_t q = a.prp[j]; q < a.prp[j + 1]; ++q) {
        const int sl = slot(a.pcol[q]);
        if (FILL && sl >= 0) acc[sl] += av * a.pval[q];
      }
    }
  if (over) *overflow = 1;
  if (!FILL) { qrp[i + 1] = nc; if (i == 0) qrp[0] = 0; return; }
  if (over) return;
  int64_t o = qrp[i];
  for (int p = 0; p < nc; ++p) {                 // ascending columns: the smallest of what is left
    int best = -1;
    for (int q = 0; q < nc; ++q) if (cols[q] != INT32_MAX && (best < 0 || cols[q] < cols[best])) best = q;
    double u = 0.0;
    u += d * acc[best];
    qcol[o] = cols[best];
    qval[o] = own[best] - a.omega * u;
    cols[best] = INT32_MAX;
    ++o;
  }
}

// Split of A for the block-hybrid Gauss-Seidel sweep from zero (build_gsb): part 0 = in-block couplings to lower colours,
// part 1 = everything else but the diagonal; couplings to non-free columns are dropped, non-free rows are empty.
struct DbSplit {
  int64_t n; int B;
  const int64_t* rowptr; const int32_t* col; const double* val;
  const int32_t* color; const double* dinv;
};
__device__ __forceinline__ int db_split_part(const DbSplit& a, int64_t i, int ci, int64_t j) {
  if (j == i) return -1;
  const int cj = j < a.n ? a.color[j] : 0;            // ghost columns count as live
  if (cj < 0) return -1;
  const int64_t b0 = (i / a.B) * a.B, b1 = b0 + a.B;
  return (j >= b0 && j < b1 && j < a.n && cj < ci) ? 0 : 1;
}
__global__ __launch_bounds__(BLOCK) void db_split_count_kernel(DbSplit a, int64_t* __restrict__ rp0, int64_t* __restrict__ rp1) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  int64_t c0 = 0, c1 = 0;
  const int ci = a.color[i];
  if (ci >= 0)
    for (int64_t k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
      const int pt = db_split_part(a, i, ci, a.col[k]);
      if (pt == 0) ++c0; else if (pt == 1) ++c1;
    }
  rp0[i + 1] = c0; rp1[i + 1] = c1;
  if (i == 0) { rp0[0] = 0; rp1[0] = 0; }
}
// cv[i] = 1 / dinv_i - a_ii on swept rows (0 elsewhere); *bad = 1: a swept row without a diagonal inverse
__global__ __launch_bounds__(BLOCK) void db_split_fill_kernel(DbSplit a, const int64_t* __restrict__ rp0, const int64_t* __restrict__ rp1,
                                                              int32_t* __restrict__ c0, double* __restrict__ v0, int32_t* __restrict__ c1,
                                                              double* __restrict__ v1, double* __restrict__ cv, int* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= a.n) return;
  const int ci = a.color[i];
  double cvi = 0.0;
  if (ci >= 0) {
    int64_t o0 = rp0[i], o1 = rp1[i];
    double aii = 0.0;
    for (int64_t k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
      const int64_t j = a.col[k];
      if (j == i) { aii = a.val[k]; continue; }
      const int pt = db_split_part(a, i, ci, j);
      if (pt == 0) { c0[o0] = (int32_t)j; v0[o0] = a.val[k]; ++o0; }
      else if (pt == 1) { c1[o1] = (int32_t)j; v1[o1] = a.val[k]; ++o1; }
    }
    if (a.dinv[i] != 0.0) cvi = 1.0 / a.dinv[i] - aii; else *bad = 1;
  }
  cv[i] = cvi;
}

// ---------------------------------------------------------------------------------------------------
// BSELL images of square-block matrices (build_bsell, build_bsell_sel) from the block-CSR arrays on the device.
// sel: which entries of a block row an image keeps, the column it stores for them and a factor on the values --
//   BB_ALL   every entry, global block column                                  (the level matrix A; padding column = the row itself)
//   BB_OFF   everything but the in-block couplings to another colour            (block-hybrid Gauss-Seidel: streamed with sweep-start values)
//   BB_IN    in-block couplings to LOWER colours, block-local column           (colour phases of a forward sweep)
//   BB_UPIN  in-block couplings to HIGHER colours, block-local column
//   BB_REST  everything but BB_IN, negated, diagonal blocks scaled by fac - 1  (r = rest x after the sweep from zero)
//   block-coloured form (bcolor = block colour of every row's sweep block):
//   BB_OFFLO  the couplings to swept rows of OTHER blocks with a LOWER block colour (no diagonal blocks, no couplings to rows that are
//             never swept: in a sweep from zero both multiply zeros -- or stale values of blocks the sweep has not reached yet)
//   BB_RESTBC the couplings to swept rows of blocks with a HIGHER block colour + BB_UPIN, negated, global columns
//             (r = rest x after that sweep)
enum : int { BB_ALL = 0, BB_OFF = 1, BB_IN = 2, BB_UPIN = 3, BB_REST = 4, BB_OFFLO = 5, BB_RESTBC = 6 };
struct DbBsell {
  int64_t m, n, ns;                     // list entries (padded to whole slices), block rows of the matrix, slices
  int bs, sel;
  const int32_t* rows;                  // block row of every list entry (-1: padding slot); null: natural order
  const int64_t* rowptr; const int32_t* col; const double* val;
  const int32_t* blk_of; const int32_t* lpos; const int32_t* color; const double* fac; const int32_t* bcolor;
  int64_t* sp;                          // [ns + 1] cumulative block steps
  int32_t* ocol; double* oval;
};
__device__ __forceinline__ bool db_bsell_keep(const DbBsell& a, int64_t i, int64_t j) {
  if (a.sel == BB_ALL) return true;
  const bool same = j < a.n && j != i && a.blk_of[i] == a.blk_of[j] && a.color[i] >= 0 && a.color[j] >= 0;
  const bool lower = same && a.color[j] < a.color[i], upper = same && a.color[j] > a.color[i];
  const bool other = (a.sel == BB_OFFLO || a.sel == BB_RESTBC) && j < a.n && j != i && a.blk_of[i] != a.blk_of[j] && a.color[i] >= 0 && a.color[j] >= 0;
  const bool high = other && a.bcolor[j] > a.bcolor[i], low = other && a.bcolor[j] < a.bcolor[i];
  switch (a.sel) {
    case BB_OFF: return !lower && !upper;
    case BB_IN: return lower;
    case BB_UPIN: return upper;
    case BB_OFFLO: return low;
    case BB_RESTBC: return upper || high;
    default: return !lower;
  }
}
// steps of every slice: the longest kept row among its RB = 64 / bs list entries (lane = list entry)
__global__ __launch_bounds__(DB_BLOCK) void db_bsell_width_kernel(DbBsell a) {
  const int64_t s = (int64_t)blockIdx.x * (DB_BLOCK / WAVE) + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (s >= a.ns) return;
  const int RB = WAVE / a.bs;
  int w = 0;
  const int64_t q = s * RB + l;
  if (l < RB && q < a.m) {
    const int64_t i = a.rows ? (int64_t)a.rows[q] : (q < a.n ? q : -1);
    if (i >= 0)
      for (int64_t k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) if (db_bsell_keep(a, i, a.col[k])) ++w;
  }
  for (int o = 32; o > 0; o >>= 1) w = max(w, __shfl_xor(w, o));
  if (l == 0) { a.sp[s + 1] = w; if (s == 0) a.sp[0] = 0; }
}
// one wave per slice, lane = (list entry rb, scalar row rr of its block row); ocol / oval hold the padding already
__global__ __launch_bounds__(DB_BLOCK) void db_bsell_fill_kernel(DbBsell a) {
  const int64_t s = (int64_t)blockIdx.x * (DB_BLOCK / WAVE) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= a.ns) return;
  const int bs = a.bs, RB = WAVE / bs;
  const int rb = lane / bs, rr = lane % bs;
  if (rb >= RB) return;
  const int64_t q = s * RB + rb;
  const int64_t i = q < a.m ? (a.rows ? (int64_t)a.rows[q] : (q < a.n ? q : -1)) : -1;
  const int64_t k0 = a.sp[s];
  const int w = (int)(a.sp[s + 1] - k0);
  if (a.sel == BB_ALL && rr == 0) {       // build_bsell pads with a valid block column: the row itself (0 for padding slots)
    const int32_t pc = (int32_t)min(max(i, (int64_t)0), a.n - 1);
    int len = 0;
    if (i >= 0) len = (int)(a.rowptr[i + 1] - a.rowptr[i]);
    for (int k = len; k < w; ++k) a.ocol[(k0 + k) * RB + rb] = pc;
  }
  if (i < 0) return;
  int64_t kk = k0;
  for (int64_t k = a.rowptr[i]; k < a.rowptr[i + 1]; ++k) {
    const int32_t j = a.col[k];
    if (!db_bsell_keep(a, i, j)) continue;
    if (rr == 0) a.ocol[kk * RB + rb] = (a.sel == BB_IN || a.sel == BB_UPIN) ? a.lpos[j] : j;
    const double sc = a.sel == BB_REST ? ((i == j && a.color[i] >= 0) ? a.fac[i] - 1.0 : -1.0) : (a.sel == BB_RESTBC ? -1.0 : 1.0);
    const double* __restrict__ blk = a.val + k * (bs * bs) + rr * bs;
    double* __restrict__ vk = a.oval + kk * (bs * WAVE);
    for (int c = 0; c < bs; ++c) {
      if ((bs & 1) && c == bs - 1) vk[(bs / 2) * (2 * WAVE) + lane] = sc * blk[c];
      else vk[(c / 2) * (2 * WAVE) + lane * 2 + (c & 1)] = sc * blk[c];
    }
    ++kk;
  }
}
__global__ __launch_bounds__(BLOCK) void db_fill_i32_kernel(int64_t n, int32_t v, int32_t* __restrict__ p) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------
// host side

static bool dev_images_wanted(const amgx_matrix& A) {
  if (std::getenv("AMGX_HOST_IMAGES")) return false;
  if (A.br != 1 || A.bc != 1 || A.n_rows <= 0) return false;
  const int64_t nnz = A.rowptr[A.n_rows];
  if (nnz <= 0 || nnz >= (int64_t)2147483647) return false;
  int64_t min_rows = 65536;
  if (const char* e = std::getenv("AMGX_DEV_IMAGES_MIN_ROWS")) min_rows = std::atoll(e);
  if (A.n_rows < min_rows) return false;
  // upload_matrix's choice of lanes per row: the device builder is the one-thread-per-row form
  const double avg = (double)nnz / (double)A.n_rows;
  int G = 1;
  while (G < 16 && A.n_rows * G < ((int64_t)1 << 20) && avg > 3.0 * G) G <<= 1;
  G = std::max(G, sell_long_row_lanes(avg));
  if (const char* e = std::getenv("AMGX_SELL_MAX_LANES")) G = std::max(1, std::min(G, std::atoi(e)));
  return G == 1;
}

struct DbDiagInfo { bool all_diag = false, plain = false; };

static DbDiagInfo dev_diag_check(const DevCsrSrc& A, const double* d_dinv) {
  DevBuf<int> fl;
  fl.alloc(2);
  HIPCHK(hipMemset(fl.p, 0, 2 * sizeof(int)));
  hipLaunchKernelGGL(db_diag_check_kernel, dim3((unsigned)((A.n_rows + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, A.n_rows, A.rowptr.p, A.col.p, A.val.p, d_dinv, fl.p);
  HIPCHK(hipGetLastError());
  int h[2];
  HIPCHK(hipMemcpy(h, fl.p, sizeof(h), hipMemcpyDeviceToHost));
  return DbDiagInfo{h[0] == 0, h[1] == 0};
}

// slice offsets of the list `rows` (or the natural order) into sp [ns + 1]; returns the stored entries
static int64_t dev_slice_offsets(const DevCsrSrc& A, const int32_t* d_rows, int64_t m, DevBuf<int64_t>& sp, int G = 1) {
  const int R = WAVE / G;
  const int64_t ns = (m + R - 1) / R;
  sp.alloc((size_t)ns + 1);
  hipLaunchKernelGGL(db_width_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(DB_BLOCK), 0, 0, m, d_rows, A.rowptr.p, G, ns, sp.p);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, 0, ns, sp.p);
  HIPCHK(hipGetLastError());
  int64_t stored = 0;
  HIPCHK(hipMemcpy(&stored, sp.p + ns, sizeof(int64_t), hipMemcpyDeviceToHost));
  return stored;
}

// build_sell + upload_sell for a matrix on the device: the image of the list `d_rows` [m] (null: natural order) with G lanes per
// row; sp / stored = its slice offsets (dev_slice_offsets), consumed.  stream_bytes as HostSell::stream_bytes.
static void dev_build_sell(const DevCsrSrc& A, const int32_t* d_rows, int64_t m, int G, bool rowrel, bool diag_first, const double* d_colscale,
                           double omega, const double* d_wdiag, DevBuf<int64_t>& sp, int64_t stored, DevMatrix::Sell& S, int64_t* stream_bytes,
                           const int32_t* d_col_stored = nullptr, const uint8_t* d_no16 = nullptr) {
  const int R = WAVE / G;
  const int64_t ns = (m + R - 1) / R;
  S.col32.alloc((size_t)std::max<int64_t>(1, stored));
  S.col16.alloc((size_t)std::max<int64_t>(1, stored));
  S.val.alloc((size_t)std::max<int64_t>(1, stored));
  S.cbase.alloc((size_t)(stored / WAVE + 32));       // (+ 32: slack behind the last slice, see upload_sell)
  HIPCHK(hipMemset(S.cbase.p, 0, (size_t)(stored / WAVE + 32) * sizeof(int32_t)));
  DevBuf<uint8_t> comp;
  comp.alloc((size_t)std::max<int64_t>(1, ns));
  DevBuf<unsigned long long> counters;
  counters.alloc(2);
  HIPCHK(hipMemset(counters.p, 0, 2 * sizeof(unsigned long long)));
  // (d_col_stored: the image stores these columns -- window-local indices -- while A.col stays the column of the value scaling)
  DbFill f{m, A.n_cols, ns, d_rows, G, rowrel ? 1 : 0, diag_first ? 1 : 0, A.rowptr.p, d_col_stored ? d_col_stored : A.col.p, A.val.p,
           d_colscale, omega, d_wdiag, A.n_rows, sp.p, S.col32.p, S.col16.p, S.cbase.p, S.val.p, comp.p, counters.p,
           d_col_stored ? A.col.p : nullptr, d_no16};
  if (ns > 0) {
    hipLaunchKernelGGL(db_fill_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(DB_BLOCK), 0, 0, f);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(db_flag_kernel, dim3((unsigned)((ns + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, ns, comp.p, sp.p);
    HIPCHK(hipGetLastError());
  }
  unsigned long long cnt[2];
  HIPCHK(hipMemcpy(cnt, counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
  const int64_t n_comp = (int64_t)cnt[0];
  if (n_comp == ns) S.col32.release();               // only read by 32-bit slices
  if (n_comp == 0) { S.col16.release(); S.cbase.release(); }
  S.slice_ptr = std::move(sp);
  S.rowrel = rowrel ? 1 : 0;
  S.diag_first = diag_first ? 1 : 0;
  S.wdiag = 0;
  S.win = 0;
  if (stream_bytes) *stream_bytes = 8 * (ns + 1) + (int64_t)cnt[1];
}

// upload_matrix for a scalar matrix already on the device (same decisions, same image).  Returns false, D untouched, where the
// host builder would not produce a one-thread-per-row SELL image (the caller then runs the host builder).
//   d_colscale / omega: image of A * diag(omega * colscale);  d_wdiag: the diagonal slot carries omega * wdiag[row] (needs info.plain)
static bool dev_upload_matrix(const DevCsrSrc& A, DevMatrix& D, bool rowrel_ok, double max_pad, int win, const DbDiagInfo* info,
                              const double* d_colscale = nullptr, double omega = 0.0, const double* d_wdiag = nullptr) {
  const int64_t m = A.n_rows;
  if (m <= 0 || A.nnz <= 0) return false;
  DevBuf<int64_t> sp;
  int64_t stored = dev_slice_offsets(A, nullptr, m, sp);
  if (!((double)stored <= max_pad * (double)A.nnz)) return false;                       // host: sellG = 0 -> CSR kernels
  const bool windowed = win > 0 && (double)stored > 1.10 * (double)A.nnz && !std::getenv("AMGX_NO_SELL_WINDOW");
  DevBuf<int32_t> rows;
  DevBuf<uint16_t> rowloc;
  if (windowed) {
    if (win != SELL_WIN) throw Err("windowed SELL: unexpected window size");
    rows.alloc((size_t)m); rowloc.alloc((size_t)m);
    hipLaunchKernelGGL((db_window_sort_kernel<SELL_WIN>), dim3((unsigned)((m + SELL_WIN - 1) / SELL_WIN)), dim3(SELL_WIN), 0, 0, m, A.rowptr.p, rows.p, rowloc.p);
    HIPCHK(hipGetLastError());
    stored = dev_slice_offsets(A, rows.p, m, sp);
  }
  const int64_t ns = (m + WAVE - 1) / WAVE;
  const bool rowrel = !windowed && rowrel_ok && A.n_cols >= A.n_rows;
  const bool diag_first = !windowed && rowrel_ok && A.n_rows <= A.n_cols && !std::getenv("AMGX_NO_DIAG_FIRST") && info && info->all_diag;
  const bool wdiag = d_wdiag && diag_first;
  DevMatrix::Sell& S = D.sell;
  int64_t bytes = 0;
  dev_build_sell(A, windowed ? rows.p : nullptr, m, 1, rowrel, diag_first, d_colscale, omega, wdiag ? d_wdiag : nullptr, sp, stored, S, &bytes);
  S.wdiag = wdiag ? 1 : 0;
  D.n_rows = A.n_rows; D.n_cols = A.n_cols; D.br = 1; D.bc = 1; D.nnz = A.nnz;
  D.fmt = FMT_SELL;
  D.lanes = 1;
  D.n_slices = (int)ns;
  D.stored = stored;
  D.stream_bytes = bytes;
  if (windowed) {
    D.stream_bytes += 2 * A.n_rows;
    S.win = win;
    S.rowloc = std::move(rowloc);
  }
  return true;
}

// Q = (I - omega Dinv A) P on the device (fold_prolongation); false = a row with more than DB_FOLD_CAP columns or too many entries
static bool dev_fold_prolongation(const DevCsrSrc& A, const DevCsrSrc& P, const double* d_dinv, double omega, DevCsrSrc& Q) {
  const int64_t n = A.n_rows;
  DbFold a{n, P.n_rows, A.rowptr.p, A.col.p, A.val.p, P.rowptr.p, P.col.p, P.val.p, d_dinv, omega};
  DevBuf<int> over;
  over.alloc(1);
  HIPCHK(hipMemset(over.p, 0, sizeof(int)));
  Q.rowptr.alloc((size_t)n + 1);
  const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL((db_fold_kernel<false>), dim3(grid), dim3(BLOCK), 0, 0, a, Q.rowptr.p, (int32_t*)nullptr, (double*)nullptr, over.p);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, 0, n, Q.rowptr.p);
  HIPCHK(hipGetLastError());
  int ho = 0;
  int64_t nnz = 0;
  HIPCHK(hipMemcpy(&ho, over.p, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&nnz, Q.rowptr.p + n, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (ho || nnz >= (int64_t)2147483647) return false;
  Q.n_rows = n; Q.n_cols = P.n_cols; Q.nnz = nnz;
  Q.col.alloc((size_t)std::max<int64_t>(1, nnz));
  Q.val.alloc((size_t)std::max<int64_t>(1, nnz));
  hipLaunchKernelGGL((db_fold_kernel<true>), dim3(grid), dim3(BLOCK), 0, 0, a, Q.rowptr.p, Q.col.p, Q.val.p, over.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return true;
}

// ---- local-window images (build_sell_lw, build_sell_lw_windowed) on the device --------------------------------------------
// Per unit of U consecutive rows (a chunk of 512 / G rows, or a window of SELL_WIN rows): the sorted list of its distinct columns
// and, per entry, the index of its column in that list.  The host sorts the unit's columns; here a workgroup marks them in a
// bitmap over [min column, max column] in LDS and ranks every column by the set bits below it -- the same list, the same
// indices.  A unit with more than `cap` distinct columns keeps its global columns and flags its rows (no16), as on the host; a
// unit whose column RANGE does not fit the bitmap makes the builder decline the image (the host builder takes over).
constexpr int DB_LW_BITS = 1 << 19;
constexpr int DB_LW_WORDS = DB_LW_BITS / 32;
struct DbLw {
  int64_t n; int U, cap;
  const int64_t* rowptr; const int32_t* col;
  int32_t* lcol; uint8_t* no16; int32_t* cnt; int32_t* lists;      // cnt [units + 1], lists [units * cap]
  unsigned long long* counters;                                     // [0] units beyond cap, [1] units the bitmap cannot hold
};
__global__ __launch_bounds__(512) void db_lw_list_kernel(DbLw a) {
  extern __shared__ unsigned char db_lw_sh[];               // bitmap 64 KB | prefix counts 32 KB | reduction scratch 4 KB
  uint32_t* bits = reinterpret_cast<uint32_t*>(db_lw_sh);
  uint16_t* pre = reinterpret_cast<uint16_t*>(db_lw_sh + (size_t)DB_LW_WORDS * 4);
  int* red = reinterpret_cast<int*>(db_lw_sh + (size_t)DB_LW_WORDS * 6);
  const int t = threadIdx.x;
  const int64_t u = blockIdx.x;
  const int64_t r0 = u * a.U, r1 = min(a.n, r0 + (int64_t)a.U);
  const int64_t e0 = a.rowptr[r0], e1 = a.rowptr[r1];
  if (t == 0) a.cnt[u + 1] = 0;
  if (e1 == e0) return;
  int mn = INT32_MAX, mx = -1;
  for (int64_t k = e0 + t; k < e1; k += 512) { const int c = a.col[k]; mn = min(mn, c); mx = max(mx, c); }
  red[t] = mn; red[512 + t] = mx;
  __syncthreads();
  for (int o = 256; o > 0; o >>= 1) {
    if (t < o) { red[t] = min(red[t], red[t + o]); red[512 + t] = max(red[512 + t], red[512 + t + o]); }
    __syncthreads();
  }
  const int cmin = red[0], cmax = red[512];
  __syncthreads();
  const int64_t range = (int64_t)cmax - cmin + 1;
  if (range > DB_LW_BITS) { if (t == 0) atomicAdd(a.counters + 1, 1ull); return; }
  const int nw = (int)((range + 31) >> 5);
  for (int w = t; w < nw; w += 512) bits[w] = 0u;
  __syncthreads();
  for (int64_t k = e0 + t; k < e1; k += 512) { const int d = a.col[k] - cmin; atomicOr(&bits[d >> 5], 1u << (d & 31)); }
  __syncthreads();
  const int per = (nw + 511) / 512;
  const int w0 = min(nw, t * per), w1 = min(nw, w0 + per);
  int sum = 0;
  for (int w = w0; w < w1; ++w) sum += __popc(bits[w]);
  red[t] = sum;
  __syncthreads();
  for (int o = 1; o < 512; o <<= 1) {                      // inclusive scan
    const int x = t >= o ? red[t - o] : 0;
    __syncthreads();
    red[t] += x;
    __syncthreads();
  }
  const int total = red[511];
  const int off = red[t] - sum;
  if (total > a.cap) {
    for (int64_t r = r0 + t; r < r1; r += 512) a.no16[r] = 1;
    for (int64_t k = e0 + t; k < e1; k += 512) a.lcol[k] = a.col[k];
    if (t == 0) atomicAdd(a.counters + 0, 1ull);
    return;
  }
  int run = off;
  for (int w = w0; w < w1; ++w) { pre[w] = (uint16_t)run; run += __popc(bits[w]); }
  __syncthreads();
  for (int64_t k = e0 + t; k < e1; k += 512) {
    const int d = a.col[k] - cmin, w = d >> 5;
    a.lcol[k] = (int32_t)pre[w] + __popc(bits[w] & ((1u << (d & 31)) - 1u));
  }
  int32_t* __restrict__ out = a.lists + u * (int64_t)a.cap;
  for (int w = w0; w < w1; ++w) {
    uint32_t b = bits[w];
    int idx = pre[w];
    while (b) { const int bit = __ffs((int)b) - 1; out[idx++] = cmin + w * 32 + bit; b &= b - 1u; }
  }
  if (t == 0) a.cnt[u + 1] = total;
}
// in-place inclusive sum of v[1..n] (v[0] stays), 32-bit (db_scan_kernel's shape)
__global__ __launch_bounds__(1024) void db_scan32_kernel(int64_t n, int32_t* __restrict__ v) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t a = 1 + (int64_t)t * chunk, b = min(n + 1, a + chunk);
  int64_t s = 0;
  for (int64_t i = a; i < b; ++i) s += v[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int64_t x = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += x;
    __syncthreads();
  }
  int64_t run = t ? part[t - 1] : 0;
  for (int64_t i = a; i < b; ++i) { run += v[i]; v[i] = (int32_t)run; }
}
__global__ __launch_bounds__(256) void db_lw_compact_kernel(int cap, const int32_t* __restrict__ cptr, const int32_t* __restrict__ lists,
                                                            int32_t* __restrict__ ccol) {
  const int64_t u = blockIdx.x;
  const int c0 = cptr[u], c1 = cptr[u + 1];
  for (int k = threadIdx.x; k < c1 - c0; k += 256) ccol[c0 + k] = lists[u * (int64_t)cap + k];
}

// build_sell_lw (windowed = false: natural row order, G lanes per row, units of 512 / G rows, values scaled by omega * colscale[column])
// and build_sell_lw_windowed (windowed = true: G = 1, rows of every SELL_WIN window by decreasing length) for a CSR matrix on
// the device.  false: nothing built (too many units beyond `cap`, or a unit the bitmap cannot hold) -- the host builder decides.
static bool dev_build_lw(const DevCsrSrc& A, bool windowed, int G, int64_t cap, bool test_cap, const double* d_colscale, double omega, DevMatrix& D,
                         DevBuf<int32_t>& d_cptr, DevBuf<int32_t>& d_ccol) {
  const int64_t n = A.n_rows, nnz = A.nnz;
  if (n <= 0 || nnz <= 0 || cap > 65535) return false;
  const int U = windowed ? SELL_WIN : 512 / G;
  const int64_t nu = (n + U - 1) / U;
  if (nu > 0x7fffffffLL || nu * cap > ((int64_t)1 << 33)) return false;
  DevBuf<int32_t> lcol, lists, cnt;
  DevBuf<uint8_t> no16;
  DevBuf<unsigned long long> counters;
  lcol.alloc((size_t)nnz); lists.alloc((size_t)(nu * cap)); cnt.alloc((size_t)nu + 1); no16.alloc((size_t)n); counters.alloc(2);
  HIPCHK(hipMemset(no16.p, 0, (size_t)n));
  HIPCHK(hipMemset(cnt.p, 0, (size_t)(nu + 1) * sizeof(int32_t)));
  HIPCHK(hipMemset(counters.p, 0, 2 * sizeof(unsigned long long)));
  DbLw a{n, U, (int)cap, A.rowptr.p, A.col.p, lcol.p, no16.p, cnt.p, lists.p, counters.p};
  constexpr size_t lw_lds = (size_t)DB_LW_WORDS * 6 + 1024 * sizeof(int);
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&db_lw_list_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw_lds));
  hipLaunchKernelGGL(db_lw_list_kernel, dim3((unsigned)nu), dim3(512), lw_lds, 0, a);
  HIPCHK(hipGetLastError());
  unsigned long long hc[2];
  HIPCHK(hipMemcpy(hc, counters.p, sizeof(hc), hipMemcpyDeviceToHost));
  if (hc[1]) return false;
  if ((int64_t)hc[0] * 20 > nu && !test_cap) return false;           // (more than 5 % of the units without a window)
  hipLaunchKernelGGL(db_scan32_kernel, dim3(1), dim3(1024), 0, 0, nu, cnt.p);
  HIPCHK(hipGetLastError());
  int32_t total = 0;
  HIPCHK(hipMemcpy(&total, cnt.p + nu, sizeof(int32_t), hipMemcpyDeviceToHost));
  DevBuf<int32_t> ccol;
  ccol.alloc((size_t)std::max<int32_t>(1, total));
  if (total == 0) HIPCHK(hipMemset(ccol.p, 0, sizeof(int32_t)));
  hipLaunchKernelGGL(db_lw_compact_kernel, dim3((unsigned)nu), dim3(256), 0, 0, (int)cap, cnt.p, lists.p, ccol.p);
  HIPCHK(hipGetLastError());
  lists.release();
  DevBuf<int32_t> rows;
  DevBuf<uint16_t> rowloc;
  if (windowed) {
    rows.alloc((size_t)n); rowloc.alloc((size_t)n);
    hipLaunchKernelGGL((db_window_sort_kernel<SELL_WIN>), dim3((unsigned)nu), dim3(SELL_WIN), 0, 0, n, A.rowptr.p, rows.p, rowloc.p);
    HIPCHK(hipGetLastError());
  }
  DevBuf<int64_t> sp;
  const int64_t stored = dev_slice_offsets(A, windowed ? rows.p : nullptr, n, sp, G);
  const int64_t ns = (n + WAVE / G - 1) / (WAVE / G);
  int64_t bytes = 0;
  dev_build_sell(A, windowed ? rows.p : nullptr, n, G, false, false, d_colscale, omega, nullptr, sp, stored, D.sell, &bytes, lcol.p, no16.p);
  D.n_rows = n; D.n_cols = A.n_cols; D.br = D.bc = 1; D.nnz = nnz;
  D.fmt = FMT_SELL; D.lanes = G;
  D.n_slices = (int)ns;
  D.stored = stored;
  D.stream_bytes = bytes + 4 * (int64_t)std::max<int32_t>(1, total) + 4 * (nu + 1) + (windowed ? 2 * n : 0);
  if (windowed) { D.sell.win = SELL_WIN; D.sell.rowloc = std::move(rowloc); }
  d_cptr = std::move(cnt);
  d_ccol = std::move(ccol);
  HIPCHK(hipDeviceSynchronize());
  return true;
}

template <class T>
static std::vector<T> db_download(const DevBuf<T>& b, size_t count) {
  std::vector<T> h(count);
  if (count) HIPCHK(hipMemcpy(h.data(), b.p, count * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}

// ---- block matrices ---------------------------------------------------------------------------------
struct DevBcsrSrc {                      // a square-block CSR matrix on the device
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int bs = 1;
  DevBuf<int64_t> rowptr;
  DevBuf<int32_t> col;
  DevBuf<double> val;
  void upload(const amgx_matrix& A) {
    n_rows = A.n_rows; n_cols = A.n_cols; bs = A.br; nnz = A.rowptr[A.n_rows];
    rowptr.upload(A.rowptr, (size_t)A.n_rows + 1);
    col.upload(A.col, (size_t)std::max<int64_t>(1, nnz));
    val.upload(A.val, (size_t)std::max<int64_t>(1, nnz) * bs * bs);
  }
};
struct DbBgsbMaps { const int32_t* blk_of = nullptr; const int32_t* lpos = nullptr; const int32_t* color = nullptr; const double* fac = nullptr; const int32_t* bcolor = nullptr; };

static bool dev_bsell_wanted(const amgx_matrix& A) {
  if (std::getenv("AMGX_HOST_IMAGES") || std::getenv("AMGX_NO_BSELL")) return false;
  if (A.br != A.bc || (A.br != 2 && A.br != 3 && A.br != 6) || A.n_rows <= 0) return false;
  const int64_t nnz = A.rowptr[A.n_rows];
  if (nnz <= 0 || nnz >= (int64_t)2147483647) return false;
  int64_t min_rows = 65536;
  if (const char* e = std::getenv("AMGX_DEV_IMAGES_MIN_ROWS")) min_rows = std::atoll(e);
  return A.n_rows * A.br >= min_rows;
}

// build_bsell / build_bsell_sel on the device.  d_rows [m]: list of block rows, m a multiple of 64 / bs (null: natural order, all
// rows).  max_pad > 0: decline (false, D untouched) if the stored block steps exceed max_pad * entries (build_bsell's contract).
static bool dev_build_bsell(const DevBcsrSrc& A, const int32_t* d_rows, int64_t m, int sel, const DbBgsbMaps& mp, int32_t padcol, double max_pad, DevMatrix& D) {
  const int bs = A.bs, RB = WAVE / bs;
  if (!d_rows) m = A.n_rows;
  else if (m % RB) throw Err("dev_build_bsell: the row list must be padded to whole slices");
  const int64_t ns = (m + RB - 1) / RB;
  DevBuf<int64_t> sp;
  sp.alloc((size_t)ns + 1);
  DbBsell a{m, A.n_rows, ns, bs, sel, d_rows, A.rowptr.p, A.col.p, A.val.p, mp.blk_of, mp.lpos, mp.color, mp.fac, mp.bcolor, sp.p, nullptr, nullptr};
  if (ns == 0) HIPCHK(hipMemset(sp.p, 0, sizeof(int64_t)));
  else {
    hipLaunchKernelGGL(db_bsell_width_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(DB_BLOCK), 0, 0, a);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, 0, ns, sp.p);
    HIPCHK(hipGetLastError());
  }
  int64_t steps = 0;
  HIPCHK(hipMemcpy(&steps, sp.p + ns, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (max_pad > 0.0 && (A.nnz == 0 || (double)steps * RB > max_pad * (double)A.nnz)) return false;
  DevBuf<int32_t> col;
  DevBuf<double> val;
  const size_t ncol = (size_t)std::max<int64_t>(1, steps * RB), nval = (size_t)std::max<int64_t>(1, steps * bs * WAVE);
  col.alloc(ncol); val.alloc(nval);
  hipLaunchKernelGGL(db_fill_i32_kernel, dim3((unsigned)((ncol + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, 0, (int64_t)ncol, padcol, col.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemset(val.p, 0, nval * sizeof(double)));
  if (ns > 0) {
    a.ocol = col.p; a.oval = val.p;
    hipLaunchKernelGGL(db_bsell_fill_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(DB_BLOCK), 0, 0, a);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipDeviceSynchronize());
  D.fmt = FMT_BSELL; D.br = D.bc = bs;
  D.n_rows = A.n_rows; D.n_cols = A.n_cols;
  D.n_slices = (int)ns;
  D.stored = steps * RB;
  D.stream_bytes = steps * ((int64_t)bs * WAVE * 8 + RB * 4) + 8 * (ns + 1);
  D.bsell.slice_ptr = std::move(sp); D.bsell.col = std::move(col); D.bsell.val = std::move(val);
  return true;
}

static void verify_same_bsell(const DevMatrix& a, const DevMatrix& b, const char* what) {
  auto fail = [&](const char* part) { throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": device-built and host-built BSELL images differ in " + part); };
  if (a.fmt != b.fmt || a.n_rows != b.n_rows || a.n_cols != b.n_cols || a.br != b.br || a.n_slices != b.n_slices || a.stored != b.stored ||
      a.stream_bytes != b.stream_bytes) fail("the descriptor");
  if (a.fmt != FMT_BSELL) return;
  const size_t ns = (size_t)a.n_slices;
  const auto sx = db_download(a.bsell.slice_ptr, ns + 1), sy = db_download(b.bsell.slice_ptr, ns + 1);
  if (sx != sy) fail("slice_ptr");
  const size_t steps = (size_t)sx[ns], RB = (size_t)(WAVE / a.br);
  if (db_download(a.bsell.col, steps * RB) != db_download(b.bsell.col, steps * RB)) fail("col");
  const auto vx = db_download(a.bsell.val, steps * a.br * WAVE), vy = db_download(b.bsell.val, steps * a.br * WAVE);
  if (std::memcmp(vx.data(), vy.data(), vx.size() * sizeof(double)) != 0) fail("val");
}

// AMGX_VERIFY_IMAGES: every array of two SELL images bit by bit (the 16-bit columns only where a slice uses them)
static void verify_same_sell(const DevMatrix::Sell& x, const DevMatrix::Sell& y, size_t ns, int64_t n_rows, const char* what) {
  auto fail = [&](const char* part) { throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": device-built and host-built images differ in " + part); };
  if (x.rowrel != y.rowrel || x.diag_first != y.diag_first || x.wdiag != y.wdiag || x.win != y.win) fail("the flags");
  if (x.slice_ptr.n != y.slice_ptr.n || x.slice_ptr.n != ns + 1) fail("the number of slices");
  const auto spx = db_download(x.slice_ptr, ns + 1), spy = db_download(y.slice_ptr, ns + 1);
  if (spx != spy) fail("slice_ptr");
  const size_t st = (size_t)(spx[ns] & ~(int64_t)63);
  const auto vx = db_download(x.val, st), vy = db_download(y.val, st);
  if (std::memcmp(vx.data(), vy.data(), st * sizeof(double)) != 0) fail("val");
  if ((x.col32.p == nullptr) != (y.col32.p == nullptr) || (x.col16.p == nullptr) != (y.col16.p == nullptr)) fail("the column encodings present");
  if (x.col32.p) {
    const auto cx = db_download(x.col32, st), cy = db_download(y.col32, st);
    if (cx != cy) fail("col32");
  }
  if (x.col16.p) {
    const auto cx = db_download(x.col16, st), cy = db_download(y.col16, st);
    const auto bx = db_download(x.cbase, st / WAVE), by = db_download(y.cbase, st / WAVE);
    if (bx != by) fail("cbase");
    for (size_t s = 0; s < ns; ++s) {
      if (!(spx[s] & 1)) continue;
      const size_t o0 = (size_t)(spx[s] & ~(int64_t)63), o1 = (size_t)(spx[s + 1] & ~(int64_t)63);
      if (std::memcmp(cx.data() + o0, cy.data() + o0, (o1 - o0) * sizeof(uint16_t)) != 0) fail("col16");
    }
  }
  if (x.win) {
    const auto rx = db_download(x.rowloc, (size_t)n_rows), ry = db_download(y.rowloc, (size_t)n_rows);
    if (rx != ry) fail("rowloc");
  }
}
static void verify_same_image(const DevMatrix& a, const DevMatrix& b, const char* what) {
  if (a.fmt != b.fmt || a.n_rows != b.n_rows || a.n_cols != b.n_cols || a.nnz != b.nnz || a.lanes != b.lanes || a.n_slices != b.n_slices ||
      a.stored != b.stored || a.stream_bytes != b.stream_bytes)
    throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": device-built and host-built images differ in the descriptor");
  if (a.fmt == FMT_SELL) { verify_same_sell(a.sell, b.sell, (size_t)a.n_slices, a.n_rows, what); return; }
  // (both builders fell back to the CSR kernels)
  if (a.rowptr.n != b.rowptr.n || a.col.n != b.col.n || a.val.n != b.val.n || db_download(a.rowptr, a.rowptr.n) != db_download(b.rowptr, b.rowptr.n) ||
      db_download(a.col, a.col.n) != db_download(b.col, b.col.n))
    throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": CSR images differ in the pattern");
  const auto vx = db_download(a.val, a.val.n), vy = db_download(b.val, b.val.n);
  if (std::memcmp(vx.data(), vy.data(), vx.size() * sizeof(double)) != 0) throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": CSR images differ in the values");
}

// local-window images: the SELL image, the unit offsets and the column lists
static void verify_same_lw(const DevMatrix& a, const DevBuf<int32_t>& ap, const DevBuf<int32_t>& ac, const DevMatrix& b, const DevBuf<int32_t>& bp,
                           const DevBuf<int32_t>& bc, const char* what) {
  verify_same_image(a, b, what);
  if (a.sell.win != b.sell.win) throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": window sizes differ");
  if (a.sell.win && db_download(a.sell.rowloc, (size_t)a.n_rows) != db_download(b.sell.rowloc, (size_t)b.n_rows))
    throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": row orders of the windows differ");
  if (ap.n != bp.n || db_download(ap, ap.n) != db_download(bp, bp.n)) throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": unit offsets differ");
  const auto pa = db_download(ap, ap.n);
  const size_t used = pa.empty() ? 0 : (size_t)pa.back();
  if (ac.n < used || bc.n < used) throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": column lists are too short");
  if (db_download(ac, used) != db_download(bc, used)) throw Err(std::string("AMGX_VERIFY_IMAGES: ") + what + ": column lists differ");
}

}  // namespace amgx
