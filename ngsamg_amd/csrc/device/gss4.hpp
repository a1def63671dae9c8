// gss4.hpp -- Gauss-Seidel on a SUBSET of the rows of a (block-)sparse matrix, on a compressed device copy.
// (included at the end of amgx.hip)
//
// Reference: GSS4<TM> (src/base/smoothers/gssmoother.hpp:99-143, gssmoother.cpp:407-583): "Compresses rows/cols from orig.
// sparse mat that it needs.  Meant to be used only for small subsets of rows."  The hybrid smoother uses it for its EX
// stage (rows shared with other ranks, gssmoother.cpp:697-698, 721-782).
//   SetUp (:456-507)          xdofs = rows of the subset, cA = their rows of A            -> gid[], cA (CSR over compressed rows)
//   CalcDiags / ctor (:417-438, :511-527)  dinv_i = inverse of the (replacement) diagonal  -> the caller passes the inverses
//   SmoothRHSInternal (:565-583)   x_k += dinv_k (b_k - cA_k: x)
//   SmoothRESInternal (:543-561)   w = -dinv_k res_k; res += cA_k:^T w; x_k -= w
//   MultAdd (:531-539)             x_k += s dinv_k b_k
// The reference visits the rows one after the other; here rows of one colour (no mutual coupling) go in parallel, colours
// in ascending (Smooth) or descending (SmoothBack) order -- the order of the multicolour smoother of the levels (amgx.h,
// AMGX_SM_GS).  Only the compressed data lives on the device: memory and work are O(rows of the subset).
//
// RES form without scatter: a sweep visits every row once, so the reference's recurrence
//     dx_k = dinv_k (res_k^old - sum_{k' visited before k} (A_k'k)^T dx_k'),   x_k += dx_k,   res -= sum_k (A_k:)^T dx_k
// is one Gauss-Seidel sweep FROM ZERO on S = (cA^T restricted to subset x subset) with right-hand side res^old, followed by
// one product with T = cA^T (rows = every row the subset couples to).  Both are gathers over CSR copies built at set-up:
// deterministic, no atomics, and literal also where A is not symmetric.
#pragma once

namespace amgx {

// rows q0..q1 of a row list over a CSR matrix with BS x BS blocks; W lane groups per row (lane (g, r) owns scalar row r
// and the blocks k = g, g + W, ... of its block row, like bgs_color_kernel)
//   MODE 0: xo[oi] += dinv_ci (bo[bi] - M_ci: xg)       one Gauss-Seidel colour
//   MODE 1: xo[oi] -= M_ci: xg                          residual update
//   MODE 2: xo[oi] += s * dinv_ci bo[bi]                MultAdd (no matrix)
//   MODE 3: xo[oi] += bo[bi]                            x += dx (no matrix)
// ci = list ? list[q] : q;  bi = bmap ? bmap[ci] : ci;  oi = omap ? omap[ci] : ci   (block indices)
template <int BS, int W, int MODE>
__global__ __launch_bounds__(BLOCK) void gss4_rows_kernel(int q0, int q1, const int32_t* __restrict__ list,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                          const double* __restrict__ vals, const double* __restrict__ dinv,
                                                          const double* xg, const double* bo, const int32_t* __restrict__ bmap,
                                                          double* xo, const int32_t* __restrict__ omap, double s) {
  constexpr int LPR = BS * W;
  constexpr int RPW = WAVE / LPR;
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int rloc = lane / LPR;
  const int g = (lane % LPR) / BS;
  const int r = lane % BS;
  const int64_t q = q0 + wave * RPW + rloc;
  const bool active = rloc < RPW && q < q1;
  const int ci = active ? (list ? list[q] : (int)q) : 0;
  double acc = 0.0;
  if (MODE <= 1) {
    if (active) acc = bcsr_row_dot<BS, BS, W, (BS >= 6 ? 2 : 4)>(rowptr[ci] + g, rowptr[ci + 1], cols, vals, r, xg);
#pragma unroll
    for (int o = W >> 1; o > 0; o >>= 1) acc += __shfl_down(acc, o * BS, WAVE);
  }
  const bool writer = active && g == 0;
  const int64_t bi = (int64_t)((bmap && active) ? bmap[ci] : ci) * BS + r;
  const int64_t oi = (int64_t)((omap && active) ? omap[ci] : ci) * BS + r;
  if (MODE == 1) { if (writer) xo[oi] -= acc; return; }
  if (MODE == 3) { if (writer) xo[oi] += bo[bi]; return; }
  const double t = writer ? (MODE == 0 ? bo[bi] - acc : s * bo[bi]) : 0.0;
  const int base = lane - r;
  double u = 0.0;
#pragma unroll
  for (int c = 0; c < BS; ++c) {
    const double tc = __shfl(t, base + c, WAVE);
    if (writer) u += dinv[(int64_t)ci * (BS * BS) + r * BS + c] * tc;
  }
  if (writer) xo[oi] += u;
}

struct Gss4Csr {
  int64_t n_rows = 0, nnz = 0;
  int W = 1;
  DevBuf<int32_t> rowptr, col;
  DevBuf<double> val;
  void upload(const std::vector<int32_t>& rp, const std::vector<int32_t>& cc, const std::vector<double>& vv, int bs) {
    n_rows = (int64_t)rp.size() - 1;
    nnz = (int64_t)cc.size();
    rowptr.upload(rp); col.upload(cc); val.upload(vv);
    const double avg = n_rows ? (double)nnz / (double)n_rows : 0.0;
    W = 1;
    while (W < 8 && bs * W * 2 <= WAVE && avg > 3.0 * W) W <<= 1;
  }
};

struct Gss4 {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int bs = 1;
  int64_t n = 0, n_cols = 0;            // block rows / columns of A
  int64_t m = 0, mt = 0;                // rows of the subset (xdofs); rows the subset couples to (rows of T)
  int n_colors = 0;
  std::vector<int> color_ptr;           // [n_colors+1] ranges of `list`
  DevBuf<int32_t> gid, list, tgid;      // xdofs; colour-major list of compressed rows; global ids of the rows of T
  Gss4Csr cA, S, T;                     // cA: columns global;  S, T: columns compressed (transposed blocks)
  DevBuf<double> dinv, dxc;             // [m*bs*bs] inverse diagonal blocks, [m*bs] dx of the RES form
  DevBuf<double> stage[2];

  static int grid_for_rows(int64_t rows, int bs, int W) {
    const int rpw = WAVE / (bs * W);
    const int64_t waves = (rows + rpw - 1) / rpw;
    return (int)std::max<int64_t>(1, (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  }

  template <int BS, int MODE>
  void launch_w(int W, int q0, int q1, const int32_t* lst, const Gss4Csr* M, const double* xg, const double* bo, const int32_t* bmap,
                double* xo, const int32_t* omap, double s) {
    if (q1 <= q0) return;
    const int32_t* rp = M ? M->rowptr.p : nullptr;
    const int32_t* cc = M ? M->col.p : nullptr;
    const double* vv = M ? M->val.p : nullptr;
    const int grid = grid_for_rows(q1 - q0, BS, W);
#define GSS4_LAUNCH(WW) hipLaunchKernelGGL((gss4_rows_kernel<BS, WW, MODE>), dim3(grid), dim3(BLOCK), 0, stream, q0, q1, lst, rp, cc, vv, \
                                          (const double*)dinv.p, xg, bo, bmap, xo, omap, s)
    switch (W) {
      case 1: GSS4_LAUNCH(1); break;
      case 2: GSS4_LAUNCH(2); break;
      case 4: GSS4_LAUNCH(4); break;
      default:
        if constexpr (BS * 8 <= WAVE) { GSS4_LAUNCH(8); } else { GSS4_LAUNCH(4); }
        break;
    }
#undef GSS4_LAUNCH
    HIPCHK(hipGetLastError());
  }
  template <int MODE>
  void launch(int q0, int q1, const int32_t* lst, const Gss4Csr* M, const double* xg, const double* bo, const int32_t* bmap, double* xo,
              const int32_t* omap, double s = 0.0) {
    const int W = M ? M->W : 1;
    switch (bs) {
      case 1: launch_w<1, MODE>(W, q0, q1, lst, M, xg, bo, bmap, xo, omap, s); break;
      case 2: launch_w<2, MODE>(W, q0, q1, lst, M, xg, bo, bmap, xo, omap, s); break;
      case 3: launch_w<3, MODE>(W, q0, q1, lst, M, xg, bo, bmap, xo, omap, s); break;
      case 6: launch_w<6, MODE>(std::min(W, 4), q0, q1, lst, M, xg, bo, bmap, xo, omap, s); break;
      default: throw Err("GSS4: block size must be 1, 2, 3 or 6");
    }
  }

  // Smooth (backwards = 0) / SmoothBack: x_k += dinv_k (b_k - cA_k: x), colour by colour
  void smooth_rhs(int backwards, double* x, const double* b) {
    for (int q = 0; q < n_colors; ++q) {
      const int c = backwards ? n_colors - 1 - q : q;
      launch<0>(color_ptr[c], color_ptr[c + 1], list.p, &cA, x, b, gid.p, x, gid.p);
    }
  }
  // SmoothRES / SmoothBackRES: see the header comment
  void smooth_res(int backwards, double* x, double* res) {
    if (m == 0) return;
    HIPCHK(hipMemsetAsync(dxc.p, 0, (size_t)m * bs * sizeof(double), stream));
    for (int q = 0; q < n_colors; ++q) {
      const int c = backwards ? n_colors - 1 - q : q;
      launch<0>(color_ptr[c], color_ptr[c + 1], list.p, &S, dxc.p, res, gid.p, dxc.p, nullptr);
    }
    launch<3>(0, (int)m, nullptr, nullptr, nullptr, dxc.p, nullptr, x, gid.p);
    launch<1>(0, (int)mt, nullptr, &T, dxc.p, nullptr, nullptr, res, tgid.p);
  }
  void mult_add(double s, const double* b, double* x) { launch<2>(0, (int)m, nullptr, nullptr, nullptr, b, gid.p, x, gid.p, s); }
};

static Gss4* gss4_create(const amgx_gss4_desc* d) {
  if (!d) throw Err("amgx_gss4_create: null descriptor");
  const amgx_matrix& A = d->A;
  if (!A.rowptr || (A.rowptr[A.n_rows] && (!A.col || !A.val))) throw Err("amgx_gss4_create: matrix arrays missing");
  if (A.br != A.bc || !(A.br == 1 || A.br == 2 || A.br == 3 || A.br == 6)) throw Err("amgx_gss4_create: square blocks of size 1, 2, 3 or 6 expected");
  if (A.n_rows > A.n_cols) throw Err("amgx_gss4_create: more rows than columns");
  if (!d->dinv) throw Err("amgx_gss4_create: dinv missing");
  if (!d->color || d->n_colors < 0) throw Err("amgx_gss4_create: colouring missing");
  const int bs = A.br, bb = bs * bs;
  const int64_t n = A.n_rows;
  std::unique_ptr<Gss4> g(new Gss4());
  g->device = d->device;
  HIPCHK(hipSetDevice(d->device));
  g->bs = bs; g->n = n; g->n_cols = A.n_cols; g->n_colors = d->n_colors;
  // ---- xdofs and the compressed numbering (SetUp, gssmoother.cpp:456-507)
  std::vector<int32_t> xdofs, cidx(A.n_cols, -1);
  for (int64_t k = 0; k < n; ++k) {
    const bool in = !d->subset || d->subset[k];
    if (in != (d->color[k] >= 0)) throw Err("amgx_gss4_create: exactly the rows of the subset carry a colour");
    if (d->color[k] >= d->n_colors) throw Err("amgx_gss4_create: colour index out of range");
    if (in) { cidx[k] = (int32_t)xdofs.size(); xdofs.push_back((int32_t)k); }
  }
  const int64_t m = (int64_t)xdofs.size();
  g->m = m;
  // ---- cA (columns global), validation of the colouring, and the transposed copies S (subset x subset) and T
  std::vector<int32_t> rp(m + 1, 0), cc;
  std::vector<double> vv;
  std::vector<int64_t> tcount(A.n_cols, 0);
  for (int64_t i = 0; i < m; ++i) {
    const int64_t k = xdofs[i];
    for (int64_t p = A.rowptr[k]; p < A.rowptr[k + 1]; ++p) {
      const int64_t j = A.col[p];
      if (j < 0 || j >= A.n_cols) throw Err("amgx_gss4_create: column index out of range");
      if (j != k && j < n && d->color[j] >= 0 && d->color[j] == d->color[k]) throw Err("amgx_gss4_create: invalid colouring: two coupled rows of the subset share a colour");
      cc.push_back((int32_t)j);
      vv.insert(vv.end(), A.val + p * bb, A.val + (p + 1) * bb);
      tcount[j]++;
    }
    rp[i + 1] = (int32_t)cc.size();
  }
  g->cA.upload(rp, cc, vv, bs);
  std::vector<int32_t> tgid, tpos(A.n_cols, -1);
  for (int64_t j = 0; j < A.n_cols; ++j) if (tcount[j]) { tpos[j] = (int32_t)tgid.size(); tgid.push_back((int32_t)j); }
  const int64_t mt = (int64_t)tgid.size();
  g->mt = mt;
  std::vector<int32_t> trp(mt + 1, 0), srp(m + 1, 0);
  for (int64_t t = 0; t < mt; ++t) trp[t + 1] = trp[t] + (int32_t)tcount[tgid[t]];
  for (int64_t i = 0; i < m; ++i) srp[i + 1] = srp[i] + (int32_t)tcount[xdofs[i]];
  std::vector<int32_t> tcc(trp[mt]), scc(srp[m]);
  std::vector<double> tvv((size_t)trp[mt] * bb), svv((size_t)srp[m] * bb);
  std::vector<int32_t> tfill(trp.begin(), trp.end() - 1), sfill(srp.begin(), srp.end() - 1);
  for (int64_t i = 0; i < m; ++i)               // ascending i: the rows of the transposed copies come out sorted by column
    for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
      const int64_t j = cc[p];
      const double* a = vv.data() + (size_t)p * bb;
      auto put = [&](std::vector<int32_t>& oc, std::vector<double>& ov, int32_t at) {
        oc[at] = (int32_t)i;
        for (int r = 0; r < bs; ++r) for (int c = 0; c < bs; ++c) ov[(size_t)at * bb + r * bs + c] = a[c * bs + r];     // Trans(A_kj)
      };
      put(tcc, tvv, tfill[tpos[j]]++);
      if (cidx[j] >= 0) put(scc, svv, sfill[cidx[j]]++);
    }
  g->T.upload(trp, tcc, tvv, bs);
  g->S.upload(srp, scc, svv, bs);
  g->tgid.upload(tgid);
  g->gid.upload(xdofs);
  // ---- colour-major list of compressed rows
  g->color_ptr.assign(d->n_colors + 1, 0);
  for (int64_t i = 0; i < m; ++i) g->color_ptr[d->color[xdofs[i]] + 1]++;
  for (int c = 0; c < d->n_colors; ++c) g->color_ptr[c + 1] += g->color_ptr[c];
  std::vector<int32_t> lst(m);
  {
    std::vector<int> pos(g->color_ptr.begin(), g->color_ptr.end() - 1);
    for (int64_t i = 0; i < m; ++i) lst[pos[d->color[xdofs[i]]]++] = (int32_t)i;
  }
  g->list.upload(lst);
  std::vector<double> dv((size_t)m * bb);
  for (int64_t i = 0; i < m; ++i) std::copy(d->dinv + (size_t)xdofs[i] * bb, d->dinv + (size_t)(xdofs[i] + 1) * bb, dv.begin() + (size_t)i * bb);
  g->dinv.upload(dv);
  g->dxc.alloc((size_t)std::max<int64_t>(1, m * bs));
  return g.release();
}

}  // namespace amgx

struct amgx_gss4_t { amgx::Gss4* g; };

namespace {
template <class F>
int gss4_guard(amgx_gss4 gg, F&& f) {
  try {
    if (!gg || !gg->g) throw amgx::Err("null handle");
    HIPCHK(hipSetDevice(gg->g->device));
    f(*gg->g);
    return 0;
  } catch (const std::exception& e) {
    if (gg && gg->g) gg->g->err = e.what(); else g_create_err = e.what();
    return 1;
  }
}
// host vectors are staged whole (the calls touch rows scattered over the vector)
struct Gss4Staged {
  amgx::Gss4& g;
  bool host;
  Gss4Staged(amgx::Gss4& gg, int flags) : g(gg), host(!(flags & AMGX_DEVICE_PTR)) {}
  double* in(int slot, const double* p, int64_t len) {
    if (!host) return const_cast<double*>(p);
    if ((int64_t)g.stage[slot].n < len) g.stage[slot].alloc(len);
    HIPCHK(hipMemcpyAsync(g.stage[slot].p, p, len * sizeof(double), hipMemcpyHostToDevice, g.stream));
    return g.stage[slot].p;
  }
  void out(int slot, double* p, int64_t len) {
    if (host) HIPCHK(hipMemcpyAsync(p, g.stage[slot].p, len * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  }
  void finish() { if (host) HIPCHK(hipStreamSynchronize(g.stream)); }
};
}  // namespace

extern "C" {

int amgx_gss4_create(const amgx_gss4_desc* desc, amgx_gss4* out) {
  try {
    if (!out) throw amgx::Err("amgx_gss4_create: null output");
    amgx::Gss4* g = amgx::gss4_create(desc);
    *out = new amgx_gss4_t{g};
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_gss4_destroy(amgx_gss4 g) {
  if (!g) return 0;
  if (g->g) { (void)hipSetDevice(g->g->device); (void)hipDeviceSynchronize(); delete g->g; }
  delete g;
  return 0;
}

const char* amgx_gss4_last_error(amgx_gss4 g) { return (g && g->g) ? g->g->err.c_str() : g_create_err.c_str(); }

int amgx_gss4_set_stream(amgx_gss4 gg, void* s) {
  return gss4_guard(gg, [&](amgx::Gss4& g) {
    hipStream_t ns = (hipStream_t)s;
    if (ns != g.stream) { HIPCHK(hipStreamSynchronize(g.stream)); g.stream = ns; }
  });
}

int amgx_gss4_synchronize(amgx_gss4 gg) { return gss4_guard(gg, [&](amgx::Gss4& g) { HIPCHK(hipStreamSynchronize(g.stream)); }); }

int amgx_gss4_info(amgx_gss4 gg, int64_t* n_rows, int64_t* n_touched, int64_t* nnz) {
  return gss4_guard(gg, [&](amgx::Gss4& g) {
    if (n_rows) *n_rows = g.m;
    if (n_touched) *n_touched = g.mt;
    if (nnz) *nnz = g.cA.nnz;
  });
}

int amgx_gss4_smooth(amgx_gss4 gg, int dir, double* x, const double* b, int flags) {
  return gss4_guard(gg, [&](amgx::Gss4& g) {
    if (!x || !b || x == b) throw amgx::Err("amgx_gss4_smooth: bad vectors");
    Gss4Staged st(g, flags);
    double* dx = st.in(0, x, g.n_cols * g.bs);
    const double* db = st.in(1, b, g.n * g.bs);
    g.smooth_rhs(dir != 0, dx, db);
    st.out(0, x, g.n_cols * g.bs);
    st.finish();
  });
}

int amgx_gss4_smooth_res(amgx_gss4 gg, int dir, double* x, double* res, int flags) {
  return gss4_guard(gg, [&](amgx::Gss4& g) {
    if (!x || !res || x == res) throw amgx::Err("amgx_gss4_smooth_res: bad vectors");
    Gss4Staged st(g, flags);
    double* dx = st.in(0, x, g.n * g.bs);
    double* dr = st.in(1, res, g.n_cols * g.bs);
    g.smooth_res(dir != 0, dx, dr);
    st.out(0, x, g.n * g.bs);
    st.out(1, res, g.n_cols * g.bs);
    st.finish();
  });
}

int amgx_gss4_mult_add(amgx_gss4 gg, double s, const double* b, double* x, int flags) {
  return gss4_guard(gg, [&](amgx::Gss4& g) {
    if (!x || !b || x == b) throw amgx::Err("amgx_gss4_mult_add: bad vectors");
    Gss4Staged st(g, flags);
    double* dx = st.in(0, x, g.n * g.bs);
    const double* db = st.in(1, b, g.n * g.bs);
    g.mult_add(s, db, dx);
    st.out(0, x, g.n * g.bs);
    st.finish();
  });
}

}  // extern "C"
