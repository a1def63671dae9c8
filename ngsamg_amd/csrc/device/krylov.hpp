// Device-resident Krylov solvers around the preconditioner: preconditioned CG and restarted GMRES with
// hand-written BLAS-1 kernels -- the callers of the hot path on the reference side are NGSolve's CGSolver / GMRes
// (reference tests/h1/amg_utils.py:346, ngsolve.krylovspace); with the vectors resident in HBM one iteration is the
// preconditioner application + one SpMV + a few fused vector passes, and the host only reads one scalar per iteration.
//
// Reductions are deterministic: a fixed grid of workgroups writes partial sums, one workgroup adds them in a fixed order.
// The scalars of the recurrences (alpha, beta) stay on the device; kernels read them from memory.
#pragma once

namespace amgx {

constexpr int KR_BLOCKS = 1024;              // partial sums per dot product

// partial[blockIdx.x + slot * KR_BLOCKS] = sum over this block's grid-stride share of a[i] * b[i]
__global__ __launch_bounds__(BLOCK) void kr_dot_partial_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                               double* __restrict__ partial) {
  __shared__ double red[BLOCK / WAVE];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc += a[i] * b[i];
#pragma unroll
  for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int w = 0; w < BLOCK / WAVE; ++w) s += red[w]; partial[blockIdx.x] = s; }
}
// several dot products against one vector in one pass: partial[j * KR_BLOCKS + block] = <V_j, w> share (Arnoldi: h = V^T w)
__global__ __launch_bounds__(BLOCK) void kr_multi_dot_partial_kernel(int64_t n, int m, const double* __restrict__ V, int64_t ldv,
                                                                     const double* __restrict__ w, double* __restrict__ partial) {
  __shared__ double red[BLOCK / WAVE];
  for (int j = 0; j < m; ++j) {
    const double* __restrict__ v = V + (int64_t)j * ldv;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc += v[i] * w[i];
#pragma unroll
    for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double s = 0.0; for (int q = 0; q < BLOCK / WAVE; ++q) s += red[q]; partial[(int64_t)j * KR_BLOCKS + blockIdx.x] = s; }
    __syncthreads();
  }
}
// out[j] = sum of the nb partials of product j, fixed order; one workgroup per product
__global__ __launch_bounds__(BLOCK) void kr_dot_final_kernel(int nb, const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double red[BLOCK];
  const double* p = partial + (int64_t)blockIdx.x * KR_BLOCKS;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += BLOCK) acc += p[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = BLOCK >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
// dot product in ONE launch (the CG recurrences: 7 -> 5 launches per iteration next to the cycle): every workgroup leaves its
// partial sum, takes a ticket, and the workgroup that draws the last one adds the partials exactly as kr_dot_final_kernel does
// (same order, same tree: the same bits) and re-arms the ticket counter
__global__ __launch_bounds__(BLOCK) void kr_dot_kernel(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                       double* __restrict__ partial, unsigned int* __restrict__ ticket, double* __restrict__ out) {
  __shared__ double red[BLOCK];
  __shared__ bool last;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc += a[i] * b[i];
#pragma unroll
  for (int o = WAVE >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < BLOCK / WAVE; ++w) s += red[w];
    __hip_atomic_store(partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int nb = (int)gridDim.x;
  double t = 0.0;
  for (int i = threadIdx.x; i < nb; i += BLOCK) t += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = BLOCK >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { out[0] = red[0]; *ticket = 0u; }
}
// CG update with alpha = sc[num] / sc[den] read on the device: x += alpha s, d -= alpha q
__global__ __launch_bounds__(BLOCK) void kr_cg_update_kernel(int64_t n, const double* __restrict__ sc, int num, int den,
                                                             const double* __restrict__ s, const double* __restrict__ q,
                                                             double* __restrict__ x, double* __restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double alpha = sc[num] / sc[den];
  x[i] += alpha * s[i];
  d[i] -= alpha * q[i];
}
// s = w + beta s, beta = sc[num] / sc[den]
__global__ __launch_bounds__(BLOCK) void kr_xpby_kernel(int64_t n, const double* __restrict__ sc, int num, int den,
                                                        const double* __restrict__ w, double* __restrict__ s) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  s[i] = w[i] + (sc[num] / sc[den]) * s[i];
}
// w -= sum_j h[j] V_j (Gram-Schmidt), one pass over w
__global__ __launch_bounds__(BLOCK) void kr_multi_axpy_kernel(int64_t n, int m, const double* __restrict__ V, int64_t ldv,
                                                              const double* __restrict__ h, double sign, double* __restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  double acc = w[i];
  for (int j = 0; j < m; ++j) acc += sign * h[j] * V[(int64_t)j * ldv + i];
  w[i] = acc;
}
// y = a * x   /   y += a * x with a host scalar
__global__ __launch_bounds__(BLOCK) void kr_scale_kernel(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y, int add) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) y[i] = (add ? y[i] : 0.0) + a * x[i];
}

// ---- single-reduction PCG (Chronopoulos / Gear form of the same recurrence): ONE reduction point per iteration -------------
// gamma = <r, u>, delta = <w, u> with u = C r, w = A u in one pass; partial[blk] / partial[KR_BLOCKS + blk]
__global__ __launch_bounds__(BLOCK) void kr_dot2_partial_kernel(int64_t n, const double* __restrict__ r, const double* __restrict__ u,
                                                                const double* __restrict__ w, double* __restrict__ partial) {
  __shared__ double red[2][BLOCK / WAVE];
  double a = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) { const double ui = u[i]; a += r[i] * ui; c += w[i] * ui; }
#pragma unroll
  for (int o = WAVE >> 1; o > 0; o >>= 1) { a += __shfl_xor(a, o, WAVE); c += __shfl_xor(c, o, WAVE); }
  if ((threadIdx.x & (WAVE - 1)) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int q = 0; q < BLOCK / WAVE; ++q) { s0 += red[0][q]; s1 += red[1][q]; }
    partial[blockIdx.x] = s0;
    partial[KR_BLOCKS + blockIdx.x] = s1;
  }
}
// scalar slots of the single-reduction recurrence
enum { SR_GOLD = 0, SR_GNEW = 1, SR_DELTA = 2, SR_ALPHA = 3, SR_BETA = 4, SR_FIRST = 5 };
// sc[SR_GNEW], sc[SR_DELTA] = sums of the n_local x KR_BLOCKS partials of the two products (fixed order); two workgroups
__global__ __launch_bounds__(BLOCK) void kr_sr_reduce_kernel(int n_local, const double* __restrict__ partial, double* __restrict__ sc) {
  __shared__ double red[BLOCK];
  const int j = blockIdx.x;                    // 0: gamma, 1: delta
  double acc = 0.0;
  for (int i = 0; i < n_local; ++i) {
    const double* p = partial + ((size_t)i * 2 + j) * KR_BLOCKS;
    for (int q = threadIdx.x; q < KR_BLOCKS; q += BLOCK) acc += p[q];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = BLOCK >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) sc[SR_GNEW + j] = red[0];
}
// beta = gamma_new / gamma_old (0 on the first pass), alpha = gamma_new / (delta - beta * gamma_new / alpha_old); gamma_old <- gamma_new
__global__ void kr_sr_scalars_kernel(double* __restrict__ sc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double g = sc[SR_GNEW], d = sc[SR_DELTA];
  const bool first = sc[SR_FIRST] != 0.0;
  const double beta = first ? 0.0 : g / sc[SR_GOLD];
  const double alpha = first ? g / d : g / (d - beta * g / sc[SR_ALPHA]);
  sc[SR_BETA] = beta; sc[SR_ALPHA] = alpha; sc[SR_GOLD] = g; sc[SR_FIRST] = 0.0;
}
// p = u + beta p; s = w + beta s; x += alpha p; r -= alpha s      (one pass: 6 reads, 4 writes)
__global__ __launch_bounds__(BLOCK) void kr_sr_update_kernel(int64_t n, const double* __restrict__ sc, const double* __restrict__ u,
                                                             const double* __restrict__ w, double* __restrict__ p, double* __restrict__ s,
                                                             double* __restrict__ x, double* __restrict__ r) {
  const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const double alpha = sc[SR_ALPHA], beta = sc[SR_BETA];
  const double pi = u[i] + beta * p[i], si = w[i] + beta * s[i];
  p[i] = pi; s[i] = si;
  x[i] += alpha * pi;
  r[i] -= alpha * si;
}

struct Krylov {
  Handle& h;
  int64_t n;
  DevBuf<double> partial, sc;              // partial sums; device scalars
  DevBuf<unsigned int> ticket;             // kr_dot_kernel's arrival counter (0 between launches)
  explicit Krylov(Handle& hh) : h(hh), n(hh.lev[0].len()) {
    if (hh.lev[0].n != hh.lev[0].ncols) throw Err("Krylov solvers need a square level-0 matrix (single rank)");
    partial.alloc((size_t)KR_BLOCKS * 64);
    sc.alloc(64);
    ticket.alloc(1);
    HIPCHK(hipMemsetAsync(sc.p, 0, 64 * sizeof(double), h.stream));
    HIPCHK(hipMemsetAsync(ticket.p, 0, sizeof(unsigned int), h.stream));
  }
  int nb() const { return (int)std::max<int64_t>(1, std::min<int64_t>(KR_BLOCKS, (n + BLOCK - 1) / BLOCK)); }
  void dot(const double* a, const double* b, int slot) {
    hipLaunchKernelGGL(kr_dot_kernel, dim3(nb()), dim3(BLOCK), 0, h.stream, n, a, b, partial.p, ticket.p, sc.p + slot);
    HIPCHK(hipGetLastError());
  }
  void multi_dot(int m, const double* V, const double* w, int slot0) {
    if (m > 48) throw Err("multi_dot: too many vectors");
    hipLaunchKernelGGL(kr_multi_dot_partial_kernel, dim3(nb()), dim3(BLOCK), 0, h.stream, n, m, V, n, w, partial.p);
    hipLaunchKernelGGL(kr_dot_final_kernel, dim3(m), dim3(BLOCK), 0, h.stream, nb(), partial.p, sc.p + slot0);
    HIPCHK(hipGetLastError());
  }
  double read(int slot) {
    double v = 0.0;
    HIPCHK(hipMemcpyAsync(&v, sc.p + slot, sizeof(double), hipMemcpyDeviceToHost, h.stream));
    HIPCHK(hipStreamSynchronize(h.stream));
    return v;
  }
  void read(int slot0, int m, double* out) {
    HIPCHK(hipMemcpyAsync(out, sc.p + slot0, m * sizeof(double), hipMemcpyDeviceToHost, h.stream));
    HIPCHK(hipStreamSynchronize(h.stream));
  }
  int grid() const { return Handle::grid_for(n); }
  // work vectors live in the handle (grow-only) and keep their addresses from solve to solve: the cycle's graph is keyed on the
  // (right-hand side, result) pointers, so a solver that allocated per call paid a fresh capture + instantiation -- and the
  // hipMalloc / hipFree of its vectors (GMRES(30) at cfg 2: 2.6 GB) -- on every solve
  struct WsBuf { double* p; };
  WsBuf ws(int slot, size_t count) {
    DevBuf<double>& b = h.kr_ws[slot];
    if (b.n < count) b.alloc(count);
    return WsBuf{b.p};
  }

  // x = b - A x style helpers through the handle's SpMV kernels
  void precond(const double* r, double* z, bool use_pre) {
    if (use_pre) h.run_cycle(z, r, true);
    else h.copy(z, r, n);
  }

  // preconditioned CG (NGSolve CGSolver as the reference's drivers use it: err_k = sqrt(|<C r_k, r_k>|), stop at
  // err_k <= tol * err_0; reference tests/h1/amg_utils.py:337-363).  x holds the initial guess.
  int pcg(const double* b, double* x, double tol, int maxit, bool use_pre, double* errs) {
    WsBuf d = ws(0, n), w = ws(1, n), s = ws(2, n);
    constexpr int SAS = 2;                                   // scalar slots: 0 / 1 = <w, d> of the last two iterations, 2 = <s, A s>
    h.residual(h.lev[0].A, x, b, d.p);                       // d = b - A x
    precond(d.p, w.p, use_pre);
    h.copy(s.p, w.p, n);
    int cur = 1;
    dot(w.p, d.p, cur);
    const double err0 = std::sqrt(std::fabs(read(cur)));
    if (errs) errs[0] = err0;
    if (err0 == 0.0) return 0;
    int it = 0;
    for (it = 1; it <= maxit; ++it) {
      h.mult(h.lev[0].A, s.p, w.p);                          // w = A s
      const int old = cur;
      cur = 1 - cur;
      dot(s.p, w.p, SAS);
      hipLaunchKernelGGL(kr_cg_update_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, sc.p, old, SAS, s.p, w.p, x, d.p);   // alpha = <w,d> / <s, A s>
      precond(d.p, w.p, use_pre);
      dot(w.p, d.p, cur);
      hipLaunchKernelGGL(kr_xpby_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, sc.p, cur, old, w.p, s.p);               // beta = <w,d>_new / <w,d>_old
      HIPCHK(hipGetLastError());
      const double err = std::sqrt(std::fabs(read(cur)));
      if (errs) errs[it] = err;
      if (err <= tol * err0) break;
    }
    if (it > maxit) it = maxit;
    return it;
  }

  // The same preconditioned CG with ONE reduction point per iteration (Chronopoulos / Gear): u = C r, w = A u, gamma = <r, u> and
  // delta = <w, u> in one fused pass, alpha and beta from (gamma, delta) on the device, then p, s, x, r in one fused pass -- three
  // launches per iteration beside the cycle and the level-0 product instead of five, and one device -> host scalar.  Mathematically
  // the recurrence of pcg() (alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1}) equals <C r, r> / <p, A p>); the rounding
  // differs, histories agree to ~1e-6 (tests/test_gpu_krylov.py).  err_k = sqrt(|<C r_k, r_k>|) as in pcg().
  int pcg_sr(const double* b, double* x, double tol, int maxit, double* errs) {
    WsBuf r = ws(0, n), u = ws(1, n), w = ws(2, n), p = ws(3, n), s = ws(4, n);
    h.zero(p.p, n); h.zero(s.p, n);
    const double one = 1.0;
    HIPCHK(hipMemcpyAsync(sc.p + SR_FIRST, &one, sizeof(double), hipMemcpyHostToDevice, h.stream));
    auto reduce = [&]() {
      hipLaunchKernelGGL(kr_dot2_partial_kernel, dim3(nb()), dim3(BLOCK), 0, h.stream, n, r.p, u.p, w.p, partial.p);
      hipLaunchKernelGGL(kr_sr_reduce_kernel, dim3(2), dim3(BLOCK), 0, h.stream, 1, partial.p, sc.p);
      hipLaunchKernelGGL(kr_sr_scalars_kernel, dim3(1), dim3(1), 0, h.stream, sc.p);
      HIPCHK(hipGetLastError());
    };
    h.residual(h.lev[0].A, x, b, r.p);                       // r = b - A x
    precond(r.p, u.p, true);
    h.mult(h.lev[0].A, u.p, w.p);
    HIPCHK(hipMemsetAsync(partial.p, 0, (size_t)2 * KR_BLOCKS * sizeof(double), h.stream));   // (slots a short vector never writes)
    reduce();
    const double err0 = std::sqrt(std::fabs(read(SR_GOLD)));
    if (errs) errs[0] = err0;
    if (err0 == 0.0) return 0;
    int it = 0;
    for (it = 1; it <= maxit; ++it) {
      hipLaunchKernelGGL(kr_sr_update_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, sc.p, u.p, w.p, p.p, s.p, x, r.p);
      precond(r.p, u.p, true);
      h.mult(h.lev[0].A, u.p, w.p);
      reduce();
      const double err = std::sqrt(std::fabs(read(SR_GOLD)));
      if (errs) errs[it] = err;
      if (err <= tol * err0) break;
    }
    if (it > maxit) it = maxit;
    return it;
  }

  // restarted GMRES(m), left-preconditioned: minimises |C (b - A x)|; classical Gram-Schmidt with one re-orthogonalisation
  // pass (two fused passes over the basis instead of 2 j dependent dot / axpy pairs), Givens rotations on the host.
  // err_k = |C r_k| (the recurrence value), stop at err_k <= tol * err_0.
  int gmres(const double* b, double* x, double tol, int maxit, int restart, bool use_pre, double* errs) {
    // (the basis lives in HBM: (restart + 1) vectors; multi_dot handles up to 48 of them per pass)
    if (restart > 40) throw Err("amgx_gmres: restart lengths above 40 are not supported (got " + std::to_string(restart) + ")");
    const int m = std::max(1, restart);
    WsBuf V = ws(5, (size_t)(m + 1) * n), w = ws(1, n), t = ws(2, n);
    DevBuf<double> hdev;
    hdev.alloc(64);
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), hcol(m + 1), hc2(m + 1), y(m);
    int it = 0;
    double err0 = -1.0;
    while (it < maxit) {
      h.residual(h.lev[0].A, x, b, t.p);                     // t = b - A x
      precond(t.p, V.p, use_pre);                            // v_0 = C t (not yet normalised)
      dot(V.p, V.p, 0);
      const double beta = std::sqrt(read(0));
      if (err0 < 0.0) { err0 = beta; if (errs) errs[0] = err0; }
      if (beta == 0.0 || beta <= tol * err0) break;
      hipLaunchKernelGGL(kr_scale_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, 1.0 / beta, V.p, V.p, 0);
      std::fill(g.begin(), g.end(), 0.0);
      g[0] = beta;
      int j = 0;
      bool done = false;
      for (j = 0; j < m && it < maxit; ++j) {
        ++it;
        h.mult(h.lev[0].A, V.p + (size_t)j * n, t.p);
        precond(t.p, w.p, use_pre);                          // w = C A v_j
        std::fill(hcol.begin(), hcol.end(), 0.0);
        for (int pass = 0; pass < 2; ++pass) {
          multi_dot(j + 1, V.p, w.p, 0);
          read(0, j + 1, hc2.data());
          HIPCHK(hipMemcpyAsync(hdev.p, hc2.data(), (j + 1) * sizeof(double), hipMemcpyHostToDevice, h.stream));
          hipLaunchKernelGGL(kr_multi_axpy_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, j + 1, V.p, n, hdev.p, -1.0, w.p);
          HIPCHK(hipStreamSynchronize(h.stream));            // hc2 is reused by the next pass
          for (int i = 0; i <= j; ++i) hcol[i] += hc2[i];
        }
        dot(w.p, w.p, 0);
        const double hn = std::sqrt(read(0));
        hcol[j + 1] = hn;
        if (hn > 0.0) hipLaunchKernelGGL(kr_scale_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, 1.0 / hn, w.p, V.p + (size_t)(j + 1) * n, 0);
        for (int i = 0; i < j; ++i) {                        // previous rotations
          const double a = cs[i] * hcol[i] + sn[i] * hcol[i + 1];
          hcol[i + 1] = -sn[i] * hcol[i] + cs[i] * hcol[i + 1];
          hcol[i] = a;
        }
        const double den = std::hypot(hcol[j], hcol[j + 1]);
        cs[j] = den > 0 ? hcol[j] / den : 1.0;
        sn[j] = den > 0 ? hcol[j + 1] / den : 0.0;
        hcol[j] = den;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
        for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = hcol[i];
        const double err = std::fabs(g[j + 1]);
        if (errs) errs[it] = err;
        if (err <= tol * err0 || hn == 0.0) { done = true; ++j; break; }
      }
      // y = H^-1 g (upper triangular), x += V y
      const int k = j;
      for (int i = k - 1; i >= 0; --i) {
        double sacc = g[i];
        for (int q = i + 1; q < k; ++q) sacc -= H[(size_t)i * m + q] * y[q];
        const double piv = H[(size_t)i * m + i];
        // a zero pivot = the Krylov space stopped growing with a singular projected system (breakdown without convergence,
        // e.g. a singular operator): that direction gets no update instead of an inf / nan
        y[i] = piv != 0.0 ? sacc / piv : 0.0;
      }
      if (k > 0) {
        HIPCHK(hipMemcpyAsync(hdev.p, y.data(), k * sizeof(double), hipMemcpyHostToDevice, h.stream));
        hipLaunchKernelGGL(kr_multi_axpy_kernel, dim3(grid()), dim3(BLOCK), 0, h.stream, n, k, V.p, n, hdev.p, 1.0, x);
        HIPCHK(hipStreamSynchronize(h.stream));
      }
      if (done) break;
    }
    HIPCHK(hipGetLastError());
    return it;
  }
};

}  // namespace amgx

extern "C" {

int amgx_pcg(amgx_handle hh, const double* b, double* x, double tol, int maxit, int use_precond, int flags, double* errs, int32_t* iters) {
  return guard(hh, [&](amgx::Handle& h) {
    if (!b || !x || maxit < 0) throw amgx::Err("amgx_pcg: bad arguments");
    const int64_t n = h.lev[0].len();
    Staged st(h, flags);
    const double* db = st.in(0, b, n, 0);
    double* dx = st.inout(1, x, n, true, 0);
    if (use_precond && (db == h.lev[0].x.p || dx == h.lev[0].x.p)) throw amgx::Err("amgx_pcg: vectors alias the handle's work vectors");
    amgx::Krylov K(h);
    const int it = ((flags & AMGX_PCG_SINGLE_REDUCTION) && use_precond) ? K.pcg_sr(db, dx, tol, maxit, errs) : K.pcg(db, dx, tol, maxit, use_precond != 0, errs);
    if (iters) *iters = it;
    st.out(1, x, n, 0);
    st.finish();
  });
}

int amgx_gmres(amgx_handle hh, const double* b, double* x, double tol, int maxit, int restart, int use_precond, int flags, double* errs,
               int32_t* iters) {
  return guard(hh, [&](amgx::Handle& h) {
    if (!b || !x || maxit < 0 || restart < 1) throw amgx::Err("amgx_gmres: bad arguments");
    const int64_t n = h.lev[0].len();
    Staged st(h, flags);
    const double* db = st.in(0, b, n, 0);
    double* dx = st.inout(1, x, n, true, 0);
    if (use_precond && (db == h.lev[0].x.p || dx == h.lev[0].x.p)) throw amgx::Err("amgx_gmres: vectors alias the handle's work vectors");
    amgx::Krylov K(h);
    const int it = K.gmres(db, dx, tol, maxit, restart, use_precond != 0, errs);
    if (iters) *iters = it;
    st.out(1, x, n, 0);
    st.finish();
  });
}

}  // extern "C"
