// Sparse matrix products of the SETUP on the device: C = A B for scalar CSR matrices and the Galerkin product
// A_c = (P^T A) P built from two of them, the intermediate staying in HBM.
// Semantics follow the reference's setup helpers (not its code):
//   MatMultABImpl   src/base/linalg/utils_sparseMM.cpp:107-238  (row-wise product, sorted columns)
//   RestrictMatrix  src/base/linalg/utils_sparseMM.hpp:93-109   ((P^T A) P)
// and reproduce the host library's product (csrc/host/sparse.cpp, matmul_impl) BIT FOR BIT: entry (i, j) is
//   c = 0;  for k ascending over the columns of row i of A with b_kj stored:  c = fma(a_ik, b_kj, c)
// -- the order and the fused multiply-add of the host loop (g++ contracts c += a * b to vfmadd there).
//
// Mapping: a GROUP of G lanes owns one row of C.  The entries of row i of A are taken one after the other (that is the
// summation order); the G lanes take the entries of B's row k side by side -- their columns are distinct, so every lane works
// on its own slot of the row's hash table (open addressing, LDS: keys + accumulators) and no two lanes ever add into one slot
// inside a step; between steps the LDS operations of the wave (G <= 64) or a workgroup barrier (G = 256) keep the order.
// Two passes over the same code (count, then fill at the scanned offsets) instead of a scratch copy of all products; the fill
// pass ranks the distinct columns of the row (ascending output, as TransposeSPMImpl / MatMultABImpl keep their rows) and writes
// the row in place.  Rows are binned by the number of products sum_k |B_k| (an upper bound of the distinct columns):
//   <= 256 -> 16 lanes, 256 slots (16 rows per workgroup);  <= 2048 -> one wave, 2048 slots;  <= 8192 -> one workgroup;
// a longer row makes the call report "not supported" and the caller keeps its own product.
//
// BLOCK matrices (br x bk blocks times bk x bc blocks, elasticity: 6x3 . 3x3, 6x3 . 3x6, 6x6 . 6x6, ...): the count pass is the
// scalar one (it only reads the pattern); the fill pass (spg_block_row_kernel) gives a row one WAVE, bins the rows by their
// DISTINCT columns (known from the count pass: the accumulators are br * bc doubles per slot, so the table must follow the row's
// real size, not the bound) and spreads the (entry of B's row, element of the result block) pairs over the lanes:
//   c[r][s] = fma(a[r][q], b[q][s], c[r][s])  for q = 0 .. bk-1 in order, blocks k ascending  -- the host loop's order.
#pragma once
#include <cstdint>
#include <type_traits>

namespace spg {

constexpr int SPG_BLOCK = 256;

struct SpgArgs {
  int64_t n;                       // rows of A (= rows of C)
  const int64_t* arp; const int32_t* acol; const double* aval;
  const int64_t* brp; const int32_t* bcol; const double* bval;
  const int64_t* bound;            // [n + 1]: bound[i + 1] = products of row i
  int64_t lo, hi;                  // rows with lo < bound <= hi belong to this launch
  const int32_t* rows; int64_t n_list;      // row list of the launch (null: every row, filtered by lo / hi)
  int64_t* crp;                    // count pass: crp[i + 1] = distinct columns; fill pass: scanned offsets
  int32_t* ccol; double* cval;
};

// bound[i + 1] = sum over the entries k of row i of A of |row k of B|
__global__ __launch_bounds__(SPG_BLOCK) void spg_bound_kernel(int64_t n, const int64_t* __restrict__ arp, const int32_t* __restrict__ acol,
                                                              const int64_t* __restrict__ brp, int64_t* __restrict__ bound) {
  const int64_t i = (int64_t)blockIdx.x * SPG_BLOCK + threadIdx.x;
  if (i >= n) return;
  int64_t s = 0;
  for (int64_t k = arp[i]; k < arp[i + 1]; ++k) { const int64_t r = acol[k]; s += brp[r + 1] - brp[r]; }
  bound[i + 1] = s;
  if (i == 0) bound[0] = 0;
}

// rows with lo < bound <= hi, in any order (the output position of a row does not depend on it)
__global__ __launch_bounds__(SPG_BLOCK) void spg_list_kernel(int64_t n, const int64_t* __restrict__ bound, int64_t lo, int64_t hi,
                                                             int32_t* __restrict__ rows, unsigned long long* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * SPG_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int64_t b = bound[i + 1];
  if (b > lo && b <= hi) rows[atomicAdd(count, 1ull)] = (int32_t)i;
}

template <int G>
__device__ __forceinline__ void spg_group_sync() {
  if (G > 64) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // one wave: LDS operations complete in program order
}

__device__ __forceinline__ unsigned spg_hash(int32_t j) { return (unsigned)j * 2654435761u; }

template <int G, int CAP, bool FILL>
__global__ __launch_bounds__(SPG_BLOCK) void spg_row_kernel(SpgArgs a) {
  constexpr int GROUPS = SPG_BLOCK / G;
  extern __shared__ unsigned char spg_sh[];
  // per group: keys [CAP] | count, pad | accumulators [CAP] | slot list [CAP] (fill pass)
  constexpr size_t PER = (size_t)CAP * 4 + 16 + (FILL ? (size_t)CAP * 8 + (size_t)CAP * 4 : 0);
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  unsigned char* base = spg_sh + (size_t)g * PER;
  int32_t* keys = reinterpret_cast<int32_t*>(base);
  int* cnt = reinterpret_cast<int*>(base + (size_t)CAP * 4);
  double* acc = reinterpret_cast<double*>(base + (size_t)CAP * 4 + 16);
  int32_t* list = reinterpret_cast<int32_t*>(base + (size_t)CAP * 4 + 16 + (size_t)CAP * 8);
  const int64_t q = (int64_t)blockIdx.x * GROUPS + g;
  // (G = 256: one group per workgroup, so the early exits below are workgroup-uniform and the barriers stay legal)
  const int64_t total = a.rows ? a.n_list : a.n;
  if (q >= total) return;
  const int64_t i = a.rows ? a.rows[q] : q;
  if (!a.rows) { const int64_t b = a.bound[i + 1]; if (!(b > a.lo && b <= a.hi)) return; }
  for (int s = gl; s < CAP; s += G) keys[s] = -1;
  if (gl == 0) *cnt = 0;
  spg_group_sync<G>();
  for (int64_t ka = a.arp[i]; ka < a.arp[i + 1]; ++ka) {
    const int64_t k = a.acol[ka];
    const double av = FILL ? a.aval[ka] : 0.0;
    const int64_t e = a.brp[k + 1];
    for (int64_t kb = a.brp[k] + gl; kb < e; kb += G) {
      const int32_t j = a.bcol[kb];
      unsigned h = spg_hash(j) & (CAP - 1);
      while (true) {
        const int32_t prev = atomicCAS(&keys[h], -1, j);
        if (prev == -1) {                       // new column of the row
          atomicAdd(cnt, 1);
          if (FILL) acc[h] = 0.0;
          break;
        }
        if (prev == j) break;
        h = (h + 1) & (CAP - 1);
      }
      if (FILL) acc[h] = fma(av, a.bval[kb], acc[h]);
    }
    spg_group_sync<G>();
  }
  const int nc = *cnt;
  if (!FILL) {
    if (gl == 0) a.crp[i + 1] = nc;
    return;
  }
  // slots in use -> list (any order), then every entry finds its rank among the row's columns
  spg_group_sync<G>();
  if (gl == 0) *cnt = 0;
  spg_group_sync<G>();
  for (int s = gl; s < CAP; s += G) if (keys[s] != -1) list[atomicAdd(cnt, 1)] = s;
  spg_group_sync<G>();
  const int64_t o = a.crp[i];
  for (int e = gl; e < nc; e += G) {
    const int s = list[e];
    const int32_t key = keys[s];
    int rank = 0;
    for (int u = 0; u < nc; ++u) rank += keys[list[u]] < key;
    a.ccol[o + rank] = key;
    a.cval[o + rank] = acc[s];
  }
}

__host__ __device__ inline size_t spg_block_lds(int cap, int cbs) { return (size_t)cap * 12 + 16 + (size_t)cap * cbs * 8; }

struct SpgBlockArgs {
  SpgArgs a;
  int br, bk, bc;                  // A blocks br x bk, B blocks bk x bc (row-major)
  int cap;                         // slots of a row's table (power of two)
  const int64_t* ncnt;             // [n + 1]: ncnt[i + 1] = distinct columns of row i (count pass)
};

// fill pass for block matrices: one wave per row of the list, `waves` rows per workgroup.
// LDS per wave: keys [cap] | count, pad | slot list [cap] | ranks [cap] | accumulators [cap * br * bc]
__global__ __launch_bounds__(SPG_BLOCK) void spg_block_row_kernel(SpgBlockArgs g) {
  extern __shared__ unsigned char spg_sh[];
  const SpgArgs& a = g.a;
  const int cbs = g.br * g.bc, abs_ = g.br * g.bk, bbs = g.bk * g.bc;
  const int cap = g.cap, mask = cap - 1;
  const size_t PER = spg_block_lds(cap, cbs);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
  unsigned char* base = spg_sh + (size_t)wv * PER;
  int32_t* keys = reinterpret_cast<int32_t*>(base);
  int* cnt = reinterpret_cast<int*>(base + (size_t)cap * 4);
  int32_t* list = reinterpret_cast<int32_t*>(base + (size_t)cap * 4 + 16);
  int32_t* rnk = list + cap;
  double* acc = reinterpret_cast<double*>(base + (size_t)cap * 12 + 16);
  const int64_t q = (int64_t)blockIdx.x * waves + wv;
  if (q >= a.n_list) return;
  const int64_t i = a.rows[q];
  for (int s = lane; s < cap; s += 64) keys[s] = -1;
  for (int s = lane; s < cap * cbs; s += 64) acc[s] = 0.0;
  if (lane == 0) *cnt = 0;
  spg_group_sync<64>();
  for (int64_t ka = a.arp[i]; ka < a.arp[i + 1]; ++ka) {
    const int64_t k = a.acol[ka];
    const double* __restrict__ ab = a.aval + ka * abs_;
    const int64_t b0 = a.brp[k];
    const int tasks = (int)(a.brp[k + 1] - b0) * cbs;
    for (int t = lane; t < tasks; t += 64) {
      const int64_t kb = b0 + t / cbs;
      const int e = t % cbs, r = e / g.bc, sc = e % g.bc;
      const int32_t j = a.bcol[kb];
      unsigned h = spg_hash(j) & mask;
      while (true) {
        const int32_t prev = atomicCAS(&keys[h], -1, j);
        if (prev == -1 || prev == j) break;
        h = (h + 1) & mask;
      }
      const double* __restrict__ bb = a.bval + kb * bbs;
      double c = acc[h * cbs + e];
      for (int qq = 0; qq < g.bk; ++qq) c = fma(ab[r * g.bk + qq], bb[qq * g.bc + sc], c);
      acc[h * cbs + e] = c;
    }
    spg_group_sync<64>();
  }
  for (int s = lane; s < cap; s += 64) if (keys[s] != -1) list[atomicAdd(cnt, 1)] = s;
  spg_group_sync<64>();
  const int nc = *cnt;
  const int64_t o = a.crp[i];
  for (int en = lane; en < nc; en += 64) {
    const int32_t key = keys[list[en]];
    int rank = 0;
    for (int u = 0; u < nc; ++u) rank += keys[list[u]] < key;
    rnk[en] = rank;
    a.ccol[o + rank] = key;
  }
  spg_group_sync<64>();
  for (int t = lane; t < nc * cbs; t += 64) {
    const int en = t / cbs, e = t % cbs;
    a.cval[(o + rnk[en]) * cbs + e] = acc[list[en] * cbs + e];
  }
}

// fill pass for block rows whose accumulators do not fit LDS: one wave per row, the table (keys, rank of every slot) in LDS, the
// accumulators ARE the row's place in the result (zeroed first; read-modify-write through L2, a device-scope fence between the
// steps orders the update of one block element by different lanes in consecutive steps).
// LDS: keys [cap] | count, pad | slot list [cap] | rank of the slot [cap]
__global__ __launch_bounds__(64) void spg_block_row_gmem_kernel(SpgBlockArgs g) {
  extern __shared__ unsigned char spg_sh[];
  const SpgArgs& a = g.a;
  const int cbs = g.br * g.bc, abs_ = g.br * g.bk, bbs = g.bk * g.bc;
  const int cap = g.cap, mask = cap - 1;
  const int lane = threadIdx.x;
  int32_t* keys = reinterpret_cast<int32_t*>(spg_sh);
  int* cnt = reinterpret_cast<int*>(spg_sh + (size_t)cap * 4);
  int32_t* list = reinterpret_cast<int32_t*>(spg_sh + (size_t)cap * 4 + 16);
  int32_t* rnk = list + cap;
  const int64_t q = blockIdx.x;
  if (q >= a.n_list) return;
  const int64_t i = a.rows[q];
  for (int s = lane; s < cap; s += 64) keys[s] = -1;
  if (lane == 0) *cnt = 0;
  spg_group_sync<64>();
  for (int64_t ka = a.arp[i]; ka < a.arp[i + 1]; ++ka) {          // the row's pattern
    const int64_t k = a.acol[ka];
    const int64_t e1 = a.brp[k + 1];
    for (int64_t kb = a.brp[k] + lane; kb < e1; kb += 64) {
      const int32_t j = a.bcol[kb];
      unsigned h = spg_hash(j) & mask;
      while (true) {
        const int32_t prev = atomicCAS(&keys[h], -1, j);
        if (prev == -1 || prev == j) break;
        h = (h + 1) & mask;
      }
    }
  }
  spg_group_sync<64>();
  for (int s = lane; s < cap; s += 64) if (keys[s] != -1) list[atomicAdd(cnt, 1)] = s;
  spg_group_sync<64>();
  const int nc = *cnt;
  const int64_t o = a.crp[i];
  for (int en = lane; en < nc; en += 64) {
    const int s = list[en];
    const int32_t key = keys[s];
    int rank = 0;
    for (int u = 0; u < nc; ++u) rank += keys[list[u]] < key;
    rnk[s] = rank;
    a.ccol[o + rank] = key;
  }
  double* __restrict__ crow = a.cval + o * cbs;
  for (int t = lane; t < nc * cbs; t += 64) crow[t] = 0.0;
  spg_group_sync<64>();
  __threadfence();
  for (int64_t ka = a.arp[i]; ka < a.arp[i + 1]; ++ka) {
    const int64_t k = a.acol[ka];
    const double* __restrict__ ab = a.aval + ka * abs_;
    const int64_t b0 = a.brp[k];
    const int tasks = (int)(a.brp[k + 1] - b0) * cbs;
    for (int t = lane; t < tasks; t += 64) {
      const int64_t kb = b0 + t / cbs;
      const int e = t % cbs, r = e / g.bc, sc = e % g.bc;
      const int32_t j = a.bcol[kb];
      unsigned h = spg_hash(j) & mask;
      while (keys[h] != j) h = (h + 1) & mask;
      const double* __restrict__ bb = a.bval + kb * bbs;
      double* cp = crow + (int64_t)rnk[h] * cbs + e;
      double c = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int qq = 0; qq < g.bk; ++qq) c = fma(ab[r * g.bk + qq], bb[qq * g.bc + sc], c);
      __hip_atomic_store(cp, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
  }
}

}  // namespace spg

namespace amgx {

template <int G, int CAP, bool FILL>
static void spg_launch(spg::SpgArgs a, int64_t n_list) {
  if (n_list == 0) return;
  constexpr int GROUPS = spg::SPG_BLOCK / G;
  constexpr size_t PER = (size_t)CAP * 4 + 16 + (FILL ? (size_t)CAP * 8 + (size_t)CAP * 4 : 0);
  constexpr size_t lds = PER * GROUPS;
  // (per launch: the attribute belongs to the current device's copy of the kernel)
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spg::spg_row_kernel<G, CAP, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t grid = (n_list + GROUPS - 1) / GROUPS;
  if (grid > 0x7fffffffLL) throw Err("spgemm: too many rows for one launch");
  a.n_list = n_list;
  hipLaunchKernelGGL((spg::spg_row_kernel<G, CAP, FILL>), dim3((unsigned)grid), dim3(spg::SPG_BLOCK), lds, 0, a);
  HIPCHK(hipGetLastError());
}

// a (block-)CSR matrix on the device, 64-bit row pointers as in amgx_matrix
struct SpCsr {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int br = 1, bc = 1;
  DevBuf<int64_t> rowptr;
  DevBuf<int32_t> col;
  DevBuf<double> val;
  void upload(const amgx_matrix& A) {
    n_rows = A.n_rows; n_cols = A.n_cols; br = A.br; bc = A.bc; nnz = A.rowptr[A.n_rows];
    rowptr.upload(A.rowptr, (size_t)A.n_rows + 1);
    col.upload(A.col, (size_t)std::max<int64_t>(1, nnz));
    val.upload(A.val, (size_t)std::max<int64_t>(1, nnz) * br * bc);
  }
};

// the largest table (power of two) one wave can hold next to its accumulators in 96 KB of LDS
static int spg_block_cap(int cbs) {
  int cap = 8192;
  while (cap > 64 && spg::spg_block_lds(cap, cbs) > 96 * 1024) cap >>= 1;
  return cap;
}

// C = A B on the device ((block-)CSR, columns ascending per row), bit-identical to the host library's matmul.
// false: a row with more than 8192 products, or an index range the 32-bit arrays cannot hold -- the caller keeps its own product.
static bool dev_spgemm(const SpCsr& A, const SpCsr& B, SpCsr& C) {
  if (A.bc != B.br) throw Err("spgemm: block sizes do not match");
  const int64_t n = A.n_rows;
  const int cbs = A.br * B.bc;
  const bool scalar = A.br == 1 && A.bc == 1 && B.bc == 1;
  C.n_rows = n; C.n_cols = B.n_cols; C.nnz = 0; C.br = A.br; C.bc = B.bc;
  C.rowptr.alloc((size_t)n + 1);
  HIPCHK(hipMemset(C.rowptr.p, 0, (size_t)(n + 1) * sizeof(int64_t)));
  if (n == 0) { C.col.alloc(1); C.val.alloc(1); return true; }
  if (n > 0x7fffffffLL) return false;
  const unsigned grid = (unsigned)((n + spg::SPG_BLOCK - 1) / spg::SPG_BLOCK);
  DevBuf<int64_t> bound;
  bound.alloc((size_t)n + 1);
  hipLaunchKernelGGL(spg::spg_bound_kernel, dim3(grid), dim3(spg::SPG_BLOCK), 0, 0, n, A.rowptr.p, A.col.p, B.rowptr.p, bound.p);
  HIPCHK(hipGetLastError());
  // row lists of the three size classes (+ the rows no class holds)
  const int64_t lim[5] = {0, 256, 2048, 8192, INT64_MAX};
  DevBuf<int32_t> rows[4];
  DevBuf<unsigned long long> cnt;
  cnt.alloc(4);
  HIPCHK(hipMemset(cnt.p, 0, 4 * sizeof(unsigned long long)));
  for (int t = 1; t < 4; ++t) {            // (class 0 runs over all rows and filters by its bound: no list)
    rows[t].alloc((size_t)n);
    hipLaunchKernelGGL(spg::spg_list_kernel, dim3(grid), dim3(spg::SPG_BLOCK), 0, 0, n, bound.p, lim[t], lim[t + 1], rows[t].p, cnt.p + t);
    HIPCHK(hipGetLastError());
  }
  unsigned long long hc[4];
  HIPCHK(hipMemcpy(hc, cnt.p, sizeof(hc), hipMemcpyDeviceToHost));
  if (hc[3]) return false;
  spg::SpgArgs a{n, A.rowptr.p, A.col.p, A.val.p, B.rowptr.p, B.col.p, B.val.p, bound.p, 0, 0, nullptr, 0, C.rowptr.p, nullptr, nullptr};
  auto pass = [&](auto fill) {
    constexpr bool FILL = decltype(fill)::value;
    spg::SpgArgs s = a;
    s.lo = lim[0]; s.hi = lim[1]; s.rows = nullptr;
    spg_launch<16, 256, FILL>(s, n);
    s.rows = rows[1].p;
    spg_launch<64, 2048, FILL>(s, (int64_t)hc[1]);
    s.rows = rows[2].p;
    spg_launch<256, 8192, FILL>(s, (int64_t)hc[2]);
  };
  pass(std::false_type{});
  DevBuf<int64_t> ncnt;
  if (!scalar) {                            // distinct columns per row, kept beside the scanned offsets
    ncnt.alloc((size_t)n + 1);
    HIPCHK(hipMemcpy(ncnt.p, C.rowptr.p, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice));
  }
  hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, 0, n, C.rowptr.p);
  HIPCHK(hipGetLastError());
  int64_t nnz = 0;
  HIPCHK(hipMemcpy(&nnz, C.rowptr.p + n, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (nnz >= (int64_t)2147483647) return false;
  C.nnz = nnz;
  C.col.alloc((size_t)std::max<int64_t>(1, nnz));
  C.val.alloc((size_t)std::max<int64_t>(1, nnz) * cbs);
  a.ccol = C.col.p; a.cval = C.val.p;
  if (scalar) {
    pass(std::true_type{});
  } else {
    // block fill: rows by distinct columns -- <= 48: four rows (waves) per workgroup with 64-slot tables; then one row per
    // workgroup with 128 slots, 512 slots, the largest table one wave holds (filled to 3/4 at most); more: not for this kernel
    const int big = spg_block_cap(cbs);
    struct Tier { int64_t hi; int cap, waves; };
    std::vector<Tier> tiers{{48, 64, 4}};
    for (int cap : {128, 512}) if (cap < big) tiers.push_back({(int64_t)cap * 3 / 4, cap, 1});
    tiers.push_back({(int64_t)big * 3 / 4, big, 1});
    tiers.push_back({8192 * 3 / 4, 8192, 0});          // (waves = 0: accumulators in the result itself, spg_block_row_gmem_kernel)
    tiers.push_back({INT64_MAX, 0, 0});
    int64_t lo = 0;
    for (const Tier& tr : tiers) {
      HIPCHK(hipMemset(cnt.p, 0, sizeof(unsigned long long)));
      hipLaunchKernelGGL(spg::spg_list_kernel, dim3(grid), dim3(spg::SPG_BLOCK), 0, 0, n, ncnt.p, lo, tr.hi, rows[1].p, cnt.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpy(hc, cnt.p, sizeof(unsigned long long), hipMemcpyDeviceToHost));
      lo = tr.hi;
      if (!hc[0]) continue;
      if (!tr.cap) return false;
      spg::SpgBlockArgs g{a, A.br, A.bc, B.bc, tr.cap, ncnt.p};
      g.a.rows = rows[1].p;
      g.a.n_list = (int64_t)hc[0];
      if (g.a.n_list > 0x7fffffffLL) throw Err("spgemm: too many rows for one launch");
      if (tr.waves == 0) {
        const size_t lds = (size_t)tr.cap * 12 + 16;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spg::spg_block_row_gmem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(spg::spg_block_row_gmem_kernel, dim3((unsigned)g.a.n_list), dim3(64), lds, 0, g);
        HIPCHK(hipGetLastError());
        continue;
      }
      const size_t lds = spg::spg_block_lds(tr.cap, cbs) * tr.waves;
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spg::spg_block_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int64_t wg = (g.a.n_list + tr.waves - 1) / tr.waves;
      hipLaunchKernelGGL(spg::spg_block_row_kernel, dim3((unsigned)wg), dim3(64 * tr.waves), lds, 0, g);
      HIPCHK(hipGetLastError());
    }
  }
  HIPCHK(hipDeviceSynchronize());
  return true;
}

}  // namespace amgx
