// MI355X-native AMG apply path: device hierarchy, cycle driver and the C ABI of include/amgx.h.
//
// Host-side structure mirrors the reference's solve layer (not its code):
//   Handle                 <-> AMGMatrix                 (reference src/base/solve/amg_matrix.hpp:14-87)
//   Handle::level_smooth   <-> BaseSmoother / ProxySmoother contract (src/base/smoothers/base_smoother.hpp:68-229)
//   Handle::cycle_v/w/bs   <-> SmoothV / SmoothW / SmoothBS (src/base/solve/amg_matrix.cpp:37-307)
//   Handle::smooth_v_from_level <-> SmoothVFromLevel      (amg_matrix.cpp:310-374)
// Everything between the entry point and the result stays on the GPU; one application = a fixed sequence
// of kernel launches on one HIP stream, captured once per (b, x) pair into a hipGraph and replayed.
#include "../../../include/amgx.h"
#include "kernels.hpp"
#include <dlfcn.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace amgx {

struct Err : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      throw ::amgx::Err(std::string(#call) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

// roctx ranges named like the reference's timers (AMGMatrix::SmoothV: "AMGMatrix::Mult", "level 0" .. "level 3", "rest",
// "coarse inv", src/base/solve/amg_matrix.cpp:166-178; "ProlMap::TransferF2C" / "ProlMap::TransferC2F",
// src/base/coarsening/dof_map.cpp:616-630; "GSS3<bs=N>::SmoothRHS", src/base/smoothers/gssmoother.cpp:201,266), so a
// rocprofv3 --marker-trace of the GPU path lines up with an NGSolve trace of the CPU path.  Host-side ranges around the
// enqueue sections; active only with AMGX_ROCTX=1 (libroctx64 is resolved at run time).
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  static Roctx& get() {
    static Roctx r;
    static bool init = false;
    if (!init) {
      init = true;
      if (std::getenv("AMGX_ROCTX")) {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) { r.push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); r.pop = (int (*)())dlsym(h, "roctxRangePop"); }
        if (!r.push || !r.pop) { r.push = nullptr; r.pop = nullptr; }
      }
    }
    return r;
  }
};
struct Range {
  bool on;
  explicit Range(const char* name) : on(Roctx::get().push != nullptr) { if (on) Roctx::get().push(name); }
  explicit Range(const std::string& name) : Range(name.c_str()) {}
  ~Range() { if (on) Roctx::get().pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};
static const char* level_range_name(int l) {          // lev_timers of the reference: levels >= 4 share "rest"
  static const char* n[] = {"level 0", "level 1", "level 2", "level 3", "rest"};
  return n[l < 4 ? l : 4];
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
  }
  void upload(const T* host, size_t count) {
    alloc(count);
    if (count) HIPCHK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
  }
  template <class A> void upload(const std::vector<T, A>& v) { upload(v.data(), v.size()); }
};

enum Fmt : int { FMT_CSRVEC = 0, FMT_SELL = 1, FMT_BSELL = 2, FMT_RB = 4 };     // (3 is reported for windowed SELL, see amgx_matrix_info)

// gathered-vector access of the BSELL kernels (BSellMat::xmode); AMGX_BSELL_XMODE overrides (same arithmetic in every mode).
// Same box, cfg 3 GS / cfg 5 GS / cfg 5 block-Jacobi applications per second (profiles/r04/gs_experiments.txt):
//   0 (BS 8-byte loads) 281.9 / 160.6 / 193.1    1 (16-byte loads) 277.7 / 160.4 / 189.3    2 (one load + lane exchange) 246.2 / 149.9 / 189.3
// another box: 0 270.4 / 158.5 / 194.1; column indices 2 / 3 / 4 steps ahead (3 / 5 / 4): 261.6 / 154.7 / 193.9, 267.1 / 157.2 / 186.6, 229.5 / 126.7 / 186.3
static int bsell_xmode() {
  static const int m = [] { const char* e = std::getenv("AMGX_BSELL_XMODE"); return e ? std::max(0, std::min(5, std::atoi(e))) : 0; }();
  return m;
}

struct DevMatrix {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int br = 1, bc = 1;
  int fmt = FMT_CSRVEC;
  int lanes = 1;                       // G of the CSR-vector kernels
  // CSR
  DevBuf<int32_t> rowptr, col;
  DevBuf<double> val;
  // SELL-64-pair
  int n_slices = 0;
  int64_t stored = 0;                  // stored entries incl. padding
  int64_t stream_bytes = 0;            // bytes of matrix data one SpMV reads from HBM (values + indices + pointers)
  struct Sell {
    DevBuf<int64_t> slice_ptr;
    DevBuf<int32_t> col32, cbase;
    DevBuf<uint16_t> col16;
    DevBuf<double> val;
    int rowrel = 0, diag_first = 0, wdiag = 0;
    int xcd = 0;                       // workgroup -> rows mapping of the streaming kernels: see SellMat::xcd
    // windowed form (sell_win_spmv_kernel): inside every window of `win` consecutive rows the rows are stored in order
    // of decreasing length (slice padding 1.42 -> 1.08 on Q at cfg 2); rowloc[slot] = row of the slot inside its window
    int win = 0;
    DevBuf<uint16_t> rowloc;
    SellMat view() const { return SellMat{slice_ptr.p, col32.p, col16.p, cbase.p, val.p, rowrel, diag_first, wdiag, xcd}; }
  } sell;
  struct Rb {                          // rigid-body transfer blocks (kernels.hpp, RbMat): P_ik = w_ik Q(t_ik), or its transpose
    int dim = 0, bf = 0, bc = 0;       // spatial dimension of Q (0: w I), fine / coarse block size
    bool transposed = false;           // rows = coarse vertices (P^T)
    int lanes = 16;
    DevBuf<int32_t> ptr, col;
    DevBuf<double> w, t;
    int64_t nnz = 0;
    RbMat view() const { return RbMat{ptr.p, col.p, w.p, t.p, nnz}; }
  } rb;
  struct BSell {                       // block SELL (kernels.hpp, BSellMat)
    DevBuf<int64_t> slice_ptr;
    DevBuf<int32_t> col;
    DevBuf<double> val;
    BSellMat view() const { return BSellMat{slice_ptr.p, col.p, val.p, bsell_xmode()}; }
  } bsell;
  bool empty() const { return n_rows == 0; }
};

struct DevGS {                          // colour-major data for multicolour Gauss-Seidel
  int n_colors = 0;
  // scalar: SELL copy of A in colour-major row order
  std::vector<int> color_slice_ptr;     // [n_colors+1] slice ranges
  int lanes = 1;                        // G of the colour-major SELL-G copy
  DevMatrix::Sell sell;
  // split copies for pre-smoothing from x = 0 (forward sweep needs only lower-colour couplings, the residual after it
  // only higher-colour couplings): one pass over A instead of two
  DevMatrix::Sell lower, upper;
  int n_slices_total = 0;
  bool has_split = false;
  DevBuf<int32_t> rowid;
  // block: colour-major row list over the CSR of A
  std::vector<int> color_row_ptr;       // [n_colors+1]
  DevBuf<int32_t> rowlist;
  // block, preferred: colour-major BSELL copy (bgs_bsell_color_kernel); color_slice_ptr / rowid as in the scalar form
  DevMatrix bcopy;
  bool bsell_ok = false;
  // block levels, pre-smoothing from x = 0: the couplings to lower / to higher colours only (as `lower` / `upper` above)
  DevMatrix blower, bupper;
  bool bsplit = false;
};

struct DevGSB {                         // block-hybrid Gauss-Seidel (gsb_sweep_kernel): blocks of B consecutive rows
  int B = 0, G = 1, TH = 1024;          // rows per block, lanes per row, workgroup size (B * G == TH)
  int n_blocks = 0, n_colors = 0;
  int lowin_maxw = 0;                   // widest slice of `lowin` (entries per lane): <= 5 selects the narrow sweep-from-zero kernel
  // long-row levels: local-window image of `rest` (sell_lw_pre_restrict_kernel, MODE 1) for the fused residual + restriction
  DevMatrix restLW;
  DevBuf<int32_t> lw_cptr, lw_ccol;
  // ... and of `full` for the general sweep (gsb_sweep_kernel<..., LW>): codes instead of columns, see GsbArgs
  DevMatrix::Sell fullLW;
  DevBuf<int32_t> flw_cptr, flw_ccol;
  bool has_fullLW = false;
  int full_maxw = 0;                    // widest slice of `full`: <= 11 selects the mid-width general sweep (fewer registers: two 1024-lane
                                        //   workgroups per CU instead of one, so that one block's colour phases hide behind another's loads)
  DevMatrix::Sell full, lowin;          // block-local SELL-G copies (slots colour-sorted inside a block): all entries /
                                        //   only the in-block couplings to LOWER colours (forward sweep from x = 0)
  DevBuf<int32_t> rowid;
  DevBuf<uint8_t> slotcolor;
  DevMatrix rest;                       // natural-order copy of everything `lowin` leaves out, without the diagonal
  DevBuf<double> cvec;                  // 1 / dinv - a_kk: r = c .* x - rest x right after the sweep from zero
  bool has_split = false;
  bool on() const { return B > 0; }
};

struct DevBGSB {                        // block-hybrid Gauss-Seidel on square-block levels (bgsb_sweep_kernel)
  int BB = 0;                           // block rows per workgroup
  int n_blocks = 0, n_colors = 0;
  DevMatrix off, in, upin;              // BSELL images, every entry of A in exactly one: `in` / `upin` = the in-block couplings to LOWER /
                                        //   HIGHER colours (local block columns, rows sorted by colour inside the block); `off` = everything
                                        //   else incl. the diagonal blocks (global block columns, block-list row order).  A forward sweep
                                        //   walks `in` colour by colour and streams `upin` with the sweep-start values, a backward one the reverse
  DevBuf<int32_t> off_ptr, in_ptr, in_row;   // slice ranges per block / per (block, colour); local block row of every `in` slot (-1: padding)
  DevBuf<int32_t> blk_ptr, blk_rows;         // block rows of every sweep block (runs of consecutive rows, or compact blocks: gs_block_ids)
  // pre-smoothing from x = 0 in ONE pass over A (like DevGSB::lowin / rest): the sweep from zero reads `in` only; afterwards
  // b_k - acc_k = Dmod_k x_k = fac_k A_kk x_k on every swept row, hence r = b - A x = -(R x) with R = A - L_in and the diagonal
  // blocks scaled by (1 - fac_k), stored negated in `rest` (natural order BSELL): r = rest * x
  DevMatrix rest;
  bool has_split = false;
  // block-COLOURED form (amgx_level_desc.gs_block_color, bgsb_sweep_kernel): the sweep blocks listed by block colour; a sweep =
  // one in-place launch per block colour.  `offlo` = the part of `off` a sweep from zero needs: the couplings to swept rows of blocks
  // with a LOWER block colour (the diagonal blocks and the couplings to rows that are never swept multiply zeros there); `rest` then
  // holds -(couplings to higher block colours + `upin`): after the sweep from zero r_k = -sum_{j swept after k} A_kj x_j, so sweep
  // + residual still read A once in total
  bool bc = false;
  int n_bcolors = 0;
  std::vector<int> bc_ptr;              // [n_bcolors + 1] ranges of blk_list
  DevBuf<int32_t> blk_list;
  DevMatrix offlo;
  bool on() const { return BB > 0; }
};

struct DevBGS {                         // block Gauss-Seidel over aggregate blocks (bgs_block_kernel)
  int n_colors = 0;
  int max_m = 0;                        // largest block (scalar dofs)
  std::vector<int> color_ptr;           // [n_colors+1] ranges of the colour-major block list
  DevBuf<int32_t> blocklist, block_ptr, block_rows;
  DevBuf<int64_t> dinv_ptr;
  DevBuf<double> dinv;
  DevBuf<int32_t> rowptr, col;          // CSR of A (the level matrix itself may live in a SELL format)
  DevBuf<double> val;
};

struct DevRestrict {                    // column-blocked P^T (see restrict_chunk_kernel)
  int n_chunks = 0;
  int64_t n_slots = 0;
  DevBuf<int32_t> chunk_slot, slot_ptr, optr, oidx, dest;     // dest = inverse of oidx (empty: partials stay in slot order)
  DevBuf<double> w, part;
  DevBuf<uint16_t> fi;
  int ept = 4;                          // entries of P per thread of the fused kernels (4 or 6: the fullest chunk decides)
  // compact chunks (cluster_slices): chunk c works on the 64-row slices slice_list[c * spc .. (c + 1) * spc) instead of spc
  // consecutive ones (-1: no slice); empty = consecutive slices
  DevBuf<int32_t> slice_list;
  bool empty() const { return n_chunks == 0; }
};

struct DevCsr {                         // plain CSR copy for the single-workgroup coarse tail
  DevBuf<int32_t> rowptr, col;
  DevBuf<double> val;
  void upload(const amgx_matrix& A, const double* scaled_vals = nullptr) {
    const int64_t nnz = A.rowptr[A.n_rows];
    std::vector<int32_t> rp(A.n_rows + 1);
    for (int64_t i = 0; i <= A.n_rows; ++i) rp[i] = (int32_t)A.rowptr[i];
    rowptr.upload(rp);
    col.upload(A.col, nnz);
    val.upload(scaled_vals ? scaled_vals : A.val, nnz);
  }
};

struct DevLevel {
  DevCsr tA, tApre, tP, tPT;            // only on tail levels
  DevBuf<int32_t> t_rowlist, t_cptr, t_rowcolor;    // colour-major row list (and the colour of every row) of a Gauss-Seidel tail level
  DevRestrict R;
  DevRestrict RF;                       // chunk-local P^T for sell_pre_restrict_kernel (fused pre-smoothing + restriction)
  int fused_block = 1024;               // workgroup size = rows per chunk of the fused kernel
  DevMatrix A, P, PT;
  DevMatrix Apre;                       // scalar Jacobi levels: A * diag(omega * dinv), see EP_PRE in kernels.hpp
  // long-row levels: a second image of A' for the fused down kernel only, with chunk-local 16-bit columns into the sorted list of
  // the distinct columns of every 256-row chunk (sell_lw_pre_restrict_kernel: the gathered vector is staged in LDS)
  DevMatrix ApreLW;
  DevBuf<int32_t> lw_cptr, lw_ccol;
  // the same for the folded prolongation Q (sell_lw_win_spmv_kernel): windowed SELL with window-local 16-bit columns
  DevMatrix QLW;
  DevBuf<int32_t> qlw_cptr, qlw_ccol;
  DevMatrix Q;                          // scalar Jacobi levels of the V-cycle: (I - omega*Dinv*A) P, see fold_prolongation()
  DevBuf<double> dinv;
  DevGS gs;
  DevGSB gsb;
  DevBGSB bgsb;
  DevRestrict RG;                       // chunk-local P^T for the fused residual + restriction after a block-hybrid sweep
  DevBGS bgs;
  int sm_type = AMGX_SM_JACOBI;
  double omega = 0.9;
  int sm_steps = 1, sm_symm = 0;
  int64_t n = 0;                        // block rows (= owned rows of a rank-partitioned level)
  int64_t ncols = 0;                    // block columns of A (>= n: owned + ghost columns)
  int bs = 1;
  int64_t len() const { return n * bs; }
  int64_t ext_len() const { return ncols * bs; }
  DevBuf<double> x, rhs, res, tmp;      // x_level / rhs_level / res_level (amg_matrix.cpp:19-26) + ping-pong buffer
};

// ---------------------------------------------------------------------------------------------------
// host-side format construction
// ---------------------------------------------------------------------------------------------------

// vectors whose elements are NOT zeroed on resize: the SELL builders write every slot of these arrays themselves, and a serial
// memset of 2.4 GB per image was a third of a second of amgx_create at cfg 2
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <class U, class... Args> void construct(U* p, Args&&... args) { if constexpr (sizeof...(Args) > 0) ::new ((void*)p) U(std::forward<Args>(args)...); else ::new ((void*)p) U; }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

// AMGX_SETUP_LOG=1: wall-clock time of the stages of amgx_create on stderr
struct SetupClock {
  bool on = std::getenv("AMGX_SETUP_LOG") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what, int level = -1) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[amgx_create] %-42s level %2d  %8.1f ms\n", what, level, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// Host threads for the format builders of amgx_create (cold path, but 10^8 entries at cfg 2: serial loops were 7.5 s of
// "upload"): contiguous index ranges, one per thread; f(begin, end, thread).  AMGX_SETUP_THREADS overrides the count.
static int setup_threads() {
  static int T = [] {
    int t = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    if (const char* e = std::getenv("AMGX_SETUP_THREADS")) t = std::max(1, std::atoi(e));
    return t;
  }();
  return T;
}
template <class F>
static void par_for(int64_t n, F&& f, int64_t min_per_thread = 4096) {
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / std::max<int64_t>(1, min_per_thread)));
  if (T <= 1) { f((int64_t)0, n, 0); return; }
  std::vector<std::thread> th;
  std::exception_ptr err = nullptr;
  std::vector<std::exception_ptr> errs(T);
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      try { f(n * t / T, n * (t + 1) / T, t); } catch (...) { errs[t] = std::current_exception(); }
    });
  for (auto& q : th) q.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
}

// The images of one level (A, P / P^T, smoother data, A', Q) are independent of each other: they are built by concurrent host
// tasks, so the serial pieces of one builder (prefix sums, allocations, the host-to-device copies) overlap with the threaded
// loops of the others.  AMGX_SETUP_SERIAL=1 runs them one after the other.
struct SetupTasks {
  int device;
  bool serial = std::getenv("AMGX_SETUP_SERIAL") != nullptr;
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> errs;
  std::mutex mu;
  explicit SetupTasks(int dev) : device(dev) {}
  ~SetupTasks() { for (auto& t : th) if (t.joinable()) t.join(); }
  bool log = std::getenv("AMGX_SETUP_LOG") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  template <class F>
  void timed(F& f, const char* name) {
    if (!log) { f(); return; }
    const auto a = std::chrono::steady_clock::now();
    f();
    const auto b = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[amgx_create]   task %-28s %8.1f .. %8.1f ms\n", name, std::chrono::duration<double, std::milli>(a - t0).count(),
                 std::chrono::duration<double, std::milli>(b - t0).count());
  }
  template <class F>
  void run(F f, const char* name = "") {
    if (serial) { timed(f, name); return; }
    th.emplace_back([this, f, name]() mutable {
      try { HIPCHK(hipSetDevice(device)); timed(f, name); }
      catch (...) { std::lock_guard<std::mutex> g(mu); errs.push_back(std::current_exception()); }
    });
  }
  void wait() {
    for (auto& t : th) if (t.joinable()) t.join();
    th.clear();
    if (!errs.empty()) { auto e = errs.front(); errs.clear(); std::rethrow_exception(e); }
  }
};

template <class T>
static void par_assign(RawVec<T>& v, size_t n, T value) {      // resize without the serial value-initialisation, fill on all threads
  v.resize(n);
  T* p = v.data();
  par_for((int64_t)n, [&](int64_t a, int64_t b, int) { std::fill(p + a, p + b, value); }, 1 << 20);
}

// lanes per row of the CSR-vector kernels: the widest group that still keeps >= 80 % of its lanes busy
// (a row of length L costs ceil(L/G) steps of G lanes), else the most efficient one
static int pick_lanes(double avg_len) {
  if (avg_len <= 2.0) return 2;
  int best = 2;
  double best_eff = 0.0;
  for (int g = 64; g >= 2; g >>= 1) {
    const double steps = std::ceil(avg_len / g);
    const double eff = avg_len / (steps * g);
    if (eff >= 0.8) return g;
    if (eff > best_eff) { best_eff = eff; best = g; }
  }
  return best;
}

// Long rows (coarse levels of a reference-shaped hierarchy: 50-100 entries per row, ragged): with one thread per row the 64
// lanes of a wave gather entry k of 64 DIFFERENT rows per step -- 64 unrelated cache lines, the kernel is bound by the
// address path (0.45 T gathers/s measured at the 1.24 M-row level of cfg 2, 4.5 TB/s) -- while G lanes per row walk G
// consecutive (ascending, hence neighbouring) columns of ONE row.  Lanes per row for rows of this average length:
static int sell_long_row_lanes(double avg) {
  int g = 1;        // measured at that level (same box, 1 / 2 / 4 / 8 lanes): 193 / 176 / 188 / 204 us -- no win, the default stays one lane
  if (const char* e = std::getenv("AMGX_SELL_LONG_ROW_LANES")) g = avg >= 24.0 ? std::max(1, std::atoi(e)) : 1;
  return g;
}

// SELL-64-pair image of the rows `rows[0..m)` of a scalar CSR matrix (row id < 0 => empty padding row).
// Element (lane, j) of a slice of width w sits at  base + (j/2)*128 + lane*2 + (j&1)  for j < 2*(w/2)
// and at  base + (w-1)*64 + lane  for the odd trailing column.  Padding: value 0, column = a valid index.
// Column indices are stored as 32-bit values and, for every slice where it is possible, additionally as
// 16-bit deltas  col = (rowrel ? row : 0) + cbase[column] + d  (see kernels.hpp, SellMat).
struct HostSell {
  std::vector<int64_t> slice_ptr;
  RawVec<int32_t> col32;
  std::vector<int32_t> cbase;
  RawVec<uint16_t> col16;
  RawVec<double> val;
  int64_t n_comp_slices = 0, stream_bytes = 0;
  int rowrel = 0, diag_first = 0;
};

// G = lanes per row (1, 2, 4, 8, 16).  G == 1: one thread per row, slices of 64 rows, odd widths allowed.
// G > 1 ("SELL-G"): slices of 64/G rows; entry e of a row belongs to lane g = (e/2) % G of the row's lane
// group at step p = (e/2) / G, so every step of a wave still reads one contiguous 1 KiB + 512 B (or 256 B) line
// set; the G partial sums are combined by a wave shuffle reduction in the kernel.
static int64_t sell_stored(const amgx_matrix& A, int G) {
  const int R = WAVE / G;
  const int64_t ns = (A.n_rows + R - 1) / R;
  std::vector<int64_t> part(setup_threads(), 0);
  par_for(ns, [&](int64_t s0, int64_t s1, int t) {
    int64_t stored = 0;
    for (int64_t s = s0; s < s1; ++s) {
      int mx = 0;
      for (int64_t r = s * R; r < std::min<int64_t>(A.n_rows, (s + 1) * R); ++r) mx = std::max<int>(mx, (int)(A.rowptr[r + 1] - A.rowptr[r]));
      const int w = (G == 1) ? mx : 2 * (((mx + 1) / 2 + G - 1) / G);
      stored += (int64_t)w * WAVE;
    }
    part[t] += stored;
  });
  int64_t stored = 0;
  for (int64_t v : part) stored += v;
  return stored;
}

// no16 (optional, per row of A): slices holding such a row keep the 32-bit column encoding
static void build_sell(const amgx_matrix& A, const int32_t* rows, int64_t m, bool rowrel, int G, HostSell& S, bool want_diag_first = false,
                       const std::vector<char>* no16 = nullptr) {
  const int R = WAVE / G;
  const int64_t ns = (m + R - 1) / R;
  S.rowrel = rowrel ? 1 : 0;
  // diagonal-first entry order (one thread per row, square matrix, every row has its diagonal stored): the Jacobi
  // epilogues then get the own-row value of the gathered vector from entry 0 instead of a second streaming read
  std::vector<int32_t> dpos;
  bool diag_first = want_diag_first && G == 1 && !rows && A.n_rows <= A.n_cols && !std::getenv("AMGX_NO_DIAG_FIRST");
  if (diag_first) {
    dpos.assign(A.n_rows, -1);
    std::vector<char> bad(setup_threads(), 0);
    par_for(A.n_rows, [&](int64_t i0, int64_t i1, int t) {
      for (int64_t i = i0; i < i1; ++i) {
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) if (A.col[k] == i) { dpos[i] = (int32_t)(k - A.rowptr[i]); break; }
        if (dpos[i] < 0) { bad[t] = 1; break; }
      }
    });
    for (char b : bad) if (b) diag_first = false;
  }
  S.diag_first = diag_first ? 1 : 0;
  // CSR position (relative to the row start) of entry e in the device order
  auto src = [&](int64_t r, int e) -> int { if (!diag_first) return e; const int dp = dpos[r]; return e == 0 ? dp : (e <= dp ? e - 1 : e); };
  S.slice_ptr.assign(ns + 1, 0);
  auto row_of = [&](int64_t q) -> int64_t { return (q < m) ? (rows ? rows[q] : q) : -1; };
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s) {
      int mx = 0;
      for (int r = 0; r < R; ++r) {
        const int64_t rr = row_of(s * R + r);
        if (rr < 0) continue;
        mx = std::max<int>(mx, (int)(A.rowptr[rr + 1] - A.rowptr[rr]));
      }
      const int w = (G == 1) ? mx : 2 * (((mx + 1) / 2 + G - 1) / G);
      S.slice_ptr[s + 1] = (int64_t)w * WAVE;            // widths first, offsets by the prefix sum below
    }
  });
  for (int64_t s = 0; s < ns; ++s) S.slice_ptr[s + 1] += S.slice_ptr[s];
  const int64_t stored = S.slice_ptr[ns];
  S.col32.resize(stored);            // (every slot of col32 / col16 / val is written by the fill pass below)
  S.col16.resize(stored);
  S.cbase.assign(stored / WAVE, 0);
  S.val.resize(stored);
  S.n_comp_slices = 0;
  S.stream_bytes = 8 * (ns + 1);
  std::vector<int64_t> t_comp(setup_threads(), 0), t_bytes(setup_threads(), 0);
  std::vector<uint8_t> comp_flag((size_t)ns, 0);           // (the flag goes into bit 0 of slice_ptr[s] after all slices are filled)
  par_for(ns, [&](int64_t sl0, int64_t sl1, int tid) {
  for (int64_t s = sl0; s < sl1; ++s) {
    const int64_t base = S.slice_ptr[s];
    const int w = (int)((S.slice_ptr[s + 1] - base) / WAVE);
    const int wp = w & ~1;
    auto off = [&](int l, int j) { return (j < wp) ? base + (int64_t)(j >> 1) * (2 * WAVE) + l * 2 + (j & 1) : base + (int64_t)(w - 1) * WAVE + l; };
    // (lane l, column j) -> entry index within the lane's row
    auto entry = [&](int l, int j) { return (G == 1) ? j : 2 * ((j >> 1) * G + (l % G)) + (j & 1); };
    bool comp = true;
    if (no16)
      for (int r = 0; r < R; ++r) { const int64_t rr = row_of(s * R + r); if (rr >= 0 && (*no16)[rr]) comp = false; }
    for (int j = 0; j < w; ++j) {
      // pass 1: real entries -> 32-bit columns, column base for the 16-bit form
      int64_t cb = INT64_MAX;
      for (int l = 0; l < WAVE; ++l) {
        const int64_t r = row_of(s * R + l / G);
        if (r < 0) continue;
        const int64_t rb = A.rowptr[r];
        const int len = (int)(A.rowptr[r + 1] - rb);
        const int e = entry(l, j);
        if (e < len) {
          const int64_t o = off(l, j);
          const int64_t ks = rb + src(r, e);
          S.col32[o] = A.col[ks];
          S.val[o] = A.val[ks];
          cb = std::min<int64_t>(cb, (int64_t)A.col[ks] - (rowrel ? r : 0));
        }
      }
      if (cb == INT64_MAX) cb = 0;
      if (cb < INT32_MIN / 2 || cb > INT32_MAX / 2) comp = false;
      S.cbase[base / WAVE + j] = (int32_t)cb;
      // pass 2: deltas and padding
      for (int l = 0; l < WAVE; ++l) {
        const int64_t q = s * R + l / G;
        const int64_t r = row_of(q);
        const int64_t o = off(l, j);
        const int64_t rb = r >= 0 ? A.rowptr[r] : 0;
        const int len = r >= 0 ? (int)(A.rowptr[r + 1] - rb) : 0;
        const int e = entry(l, j);
        // the row id the kernel will use for this lane: plain SELL = position, colour-major = rowid (lanes with r < 0 exit early)
        const int64_t rk = rows ? r : q;
        const int64_t rr = rowrel ? rk : 0;
        if (e < len) {
          const int64_t d = (int64_t)A.col[rb + src(r, e)] - rr - cb;
          if (d < 0 || d > 65535) comp = false; else S.col16[o] = (uint16_t)d;
        } else {
          // padding (value 0): any valid column; prefer one reachable in both encodings
          const int32_t padcol = (r >= 0 && len) ? A.col[rb] : 0;
          S.col32[o] = padcol;
          S.val[o] = 0.0;
          if (rows && r < 0) { S.col16[o] = 0; continue; }      // lane never executes
          int64_t d = (int64_t)padcol - rr - cb;
          if (d < 0 || d > 65535) {
            d = std::max<int64_t>(0, -(rr + cb));                 // smallest delta that gives a column >= 0
            if (d > 65535 || rr + cb + d >= A.n_cols) comp = false;
          }
          if (comp) S.col16[o] = (uint16_t)d;
        }
      }
    }
    if (comp && w > 0) { comp_flag[s] = 1; t_comp[tid]++; t_bytes[tid] += (int64_t)w * WAVE * 10 + 4 * w; }
    else t_bytes[tid] += (int64_t)w * WAVE * 12;
  }
  }, 64);
  for (int64_t s = 0; s < ns; ++s) if (comp_flag[s]) S.slice_ptr[s] |= 1;
  for (int64_t v : t_comp) S.n_comp_slices += v;
  for (int64_t v : t_bytes) S.stream_bytes += v;
}

// G == 1, diagonal-first SELL: overwrite the diagonal slot (entry 0) of every row (see SellMat::wdiag)
static void patch_sell_diag(HostSell& S, int64_t n_rows, const double* dval) {
  const int64_t ns = (int64_t)S.slice_ptr.size() - 1;
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s) {
      const int64_t base = S.slice_ptr[s] & ~(int64_t)63;
      const int w = (int)(((S.slice_ptr[s + 1] & ~(int64_t)63) - base) / WAVE);
      if (w == 0) continue;
      for (int l = 0; l < WAVE; ++l) {
        const int64_t row = s * WAVE + l;
        if (row >= n_rows) break;
        S.val[w >= 2 ? base + l * 2 : base + l] = dval[row];
      }
    }
  }, 64);
}

static void upload_sell(const HostSell& S, DevMatrix::Sell& D) {
  const int64_t ns = (int64_t)S.slice_ptr.size() - 1;
  D.rowrel = S.rowrel;
  D.diag_first = S.diag_first;
  D.slice_ptr.upload(S.slice_ptr);
  D.val.upload(S.val);
  if (S.n_comp_slices < ns) D.col32.upload(S.col32);               // only read by 32-bit slices
  if (S.n_comp_slices > 0) {
    D.col16.upload(S.col16);
    // slack behind the last slice: gsb_sweep_kernel reads the column bases of a slice as one unconditional group of 2*GSB_WP + 1
    std::vector<int32_t> cb(S.cbase);
    cb.resize(cb.size() + 32, 0);
    D.cbase.upload(cb);
  }
}

// block SELL image of a square-block matrix (see kernels.hpp BSellMat); returns false if the padding would exceed `max_pad`
// rows (optional): the block rows in storage order (-1 = padding slot), m of them; default = natural order
static bool build_bsell(const amgx_matrix& A, DevMatrix& D, double max_pad, const int32_t* rows = nullptr, int64_t m = -1) {
  const int bs = A.br;
  const int RB = WAVE / bs;
  const int64_t n = rows ? m : A.n_rows;
  const int64_t ns = (n + RB - 1) / RB;
  auto row_of = [&](int64_t q) -> int64_t { return q < n ? (rows ? rows[q] : q) : -1; };
  std::vector<int64_t> sp(ns + 1, 0);
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s) {
      int w = 0;
      for (int64_t q = s * RB; q < std::min<int64_t>(n, (s + 1) * RB); ++q) { const int64_t r = row_of(q); if (r >= 0) w = std::max<int>(w, (int)(A.rowptr[r + 1] - A.rowptr[r])); }
      sp[s + 1] = w;
    }
  }, 64);
  for (int64_t s = 0; s < ns; ++s) sp[s + 1] += sp[s];
  const int64_t steps = sp[ns], nnz = A.rowptr[A.n_rows];
  if (nnz == 0 || (double)steps * RB > max_pad * (double)nnz) return false;
  RawVec<int32_t> col;
  RawVec<double> val;
  par_assign(col, (size_t)steps * RB, (int32_t)0);
  par_assign(val, (size_t)steps * bs * WAVE, 0.0);
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s) {
      const int w = (int)(sp[s + 1] - sp[s]);
      for (int rb = 0; rb < RB; ++rb) {
        const int64_t r = row_of(s * RB + rb);
        const int64_t rbeg = r >= 0 ? A.rowptr[r] : 0;
        const int len = r >= 0 ? (int)(A.rowptr[r + 1] - rbeg) : 0;
        for (int k = 0; k < w; ++k) {
          const int64_t kk = sp[s] + k;
          col[kk * RB + rb] = k < len ? A.col[rbeg + k] : (int32_t)std::min<int64_t>(std::max<int64_t>(r, 0), A.n_rows - 1);   // padding: a valid block column
          if (k >= len) continue;
          const double* b = A.val + (rbeg + k) * bs * bs;
          double* vk = val.data() + kk * (bs * WAVE);
          for (int rr = 0; rr < bs; ++rr) {
            const int lane = rb * bs + rr;
            for (int c = 0; c < bs; ++c) {
              const int cp = c / 2;
              if ((bs & 1) && c == bs - 1) vk[(bs / 2) * (2 * WAVE) + lane] = b[rr * bs + c];
              else vk[cp * (2 * WAVE) + lane * 2 + (c & 1)] = b[rr * bs + c];
            }
          }
        }
      }
    }
  }, 16);
  D.fmt = FMT_BSELL;
  D.n_slices = (int)ns;
  D.stored = steps * RB;
  D.stream_bytes = steps * ((int64_t)bs * WAVE * 8 + RB * 4) + 8 * (ns + 1);
  D.bsell.slice_ptr.upload(sp); D.bsell.col.upload(col); D.bsell.val.upload(val);
  return true;
}

// Rigid-body structure of a transfer matrix (see RbMat): every block of P (bf x bc) must be w Q(t), every block of P^T (bc x bf)
// its transpose.  Returns false (and leaves D untouched) if any block deviates: the general block formats take over.
static bool try_build_rb(const amgx_matrix& M, bool transposed, DevMatrix& D) {
  if (std::getenv("AMGX_NO_RB_TRANSFER")) return false;
  const int bf = transposed ? M.bc : M.br, bc = transposed ? M.br : M.bc;
  int dim;
  if (bc == 6 && (bf == 3 || bf == 6)) dim = 3;
  else if (bc == 3 && (bf == 2 || bf == 3)) dim = 2;
  else if (bc == 2 && bf == 2) dim = 0;
  else return false;
  const int64_t nnz = M.rowptr[M.n_rows];
  if (nnz == 0 || nnz >= (int64_t)2147483647) return false;
  std::vector<double> w((size_t)nnz), t((size_t)std::max(1, dim) * nnz, 0.0);
  std::vector<char> bad(setup_threads(), 0);
  const int br = M.br, bcm = M.bc;
  par_for(nnz, [&](int64_t k0, int64_t k1, int tid) {
    for (int64_t k = k0; k < k1; ++k) {
      const double* b = M.val + k * br * bcm;
      // entry (r of the fine block row, c of the coarse block column) of the un-transposed block
      auto at = [&](int r, int c) { return transposed ? b[c * bcm + r] : b[r * bcm + c]; };
      const double wk = at(0, 0);
      double tt[3] = {0.0, 0.0, 0.0};
      double ex[6][6];
      for (int r = 0; r < bf; ++r) for (int c = 0; c < bc; ++c) ex[r][c] = 0.0;
      const int nd = dim == 0 ? bf : dim;
      for (int r = 0; r < nd; ++r) ex[r][r] = wk;
      for (int r = nd; r < bf; ++r) ex[r][r] = wk;
      if (dim == 3) {
        if (wk != 0.0) { tt[0] = at(1, 5) / wk; tt[1] = at(2, 3) / wk; tt[2] = at(0, 4) / wk; }
        ex[0][4] = wk * tt[2]; ex[0][5] = -wk * tt[1];
        ex[1][3] = -wk * tt[2]; ex[1][5] = wk * tt[0];
        ex[2][3] = wk * tt[1]; ex[2][4] = -wk * tt[0];
      } else if (dim == 2) {
        if (wk != 0.0) { tt[0] = at(1, 2) / wk; tt[1] = -at(0, 2) / wk; }
        ex[0][2] = -wk * tt[1]; ex[1][2] = wk * tt[0];
      }
      const double tol = 1e-13 * (std::fabs(wk) * (1.0 + std::fabs(tt[0]) + std::fabs(tt[1]) + std::fabs(tt[2])));
      for (int r = 0; r < bf; ++r)
        for (int c = 0; c < bc; ++c)
          if (!(std::fabs(at(r, c) - ex[r][c]) <= tol)) { bad[tid] = 1; return; }
      w[k] = wk;
      for (int q = 0; q < dim; ++q) t[(size_t)q * nnz + k] = tt[q];
    }
  }, 1 << 12);
  for (char c : bad) if (c) return false;
  std::vector<int32_t> rp((size_t)M.n_rows + 1);
  for (int64_t i = 0; i <= M.n_rows; ++i) rp[i] = (int32_t)M.rowptr[i];
  D.fmt = FMT_RB;
  D.rb.dim = dim; D.rb.bf = bf; D.rb.bc = bc; D.rb.transposed = transposed; D.rb.nnz = nnz;
  D.rb.ptr.upload(rp); D.rb.col.upload(M.col, (size_t)nnz); D.rb.w.upload(w); D.rb.t.upload(t);
  D.stored = nnz;
  D.stream_bytes = nnz * (4 + 8 + 8 * (int64_t)dim) + 4 * (M.n_rows + 1);
  return true;
}

static void check_matrix(const amgx_matrix& A, const char* what) {
  if (A.n_rows < 0 || A.n_cols < 0 || !A.rowptr) throw Err(std::string(what) + ": invalid matrix descriptor");
  if (A.br < 1 || A.br > 6 || A.bc < 1 || A.bc > 6) throw Err(std::string(what) + ": block sizes must be in 1..6");
  const int64_t nnz = A.rowptr[A.n_rows];
  if (nnz > 0 && (!A.col || !A.val)) throw Err(std::string(what) + ": col / val missing");
  if (nnz >= (int64_t)2147483647) throw Err(std::string(what) + ": more than 2^31-1 stored blocks are not supported on the device");
  if (A.n_cols >= (int64_t)2147483647 / 8) throw Err(std::string(what) + ": too many columns for int32 indices");
  par_for(A.n_rows, [&](int64_t i0, int64_t i1, int) {
    for (int64_t i = i0; i < i1; ++i)
      if (A.rowptr[i + 1] < A.rowptr[i]) throw Err(std::string(what) + ": rowptr is not monotone");
  });
  par_for(nnz, [&](int64_t k0, int64_t k1, int) {
    for (int64_t k = k0; k < k1; ++k)
      if (A.col[k] < 0 || A.col[k] >= A.n_cols) throw Err(std::string(what) + ": column index out of range");
  }, 1 << 16);
}

// rb_mode: 1 = A is a prolongation, 2 = its transpose: block matrices are first tried in the rigid-body form (try_build_rb)
static void upload_matrix(const amgx_matrix& A, DevMatrix& D, const char* what, bool allow_sell = true, bool rowrel_ok = false, bool keep_csr = false,
                          double max_pad = 1.35, int win = 0, const double* diag_override = nullptr, int rb_mode = 0) {
  check_matrix(A, what);
  D.n_rows = A.n_rows; D.n_cols = A.n_cols; D.br = A.br; D.bc = A.bc;
  D.nnz = A.rowptr[A.n_rows];
  if (rb_mode && (A.br > 1 || A.bc > 1) && try_build_rb(A, rb_mode == 2, D)) return;
  const double avg = D.n_rows ? (double)D.nnz / (double)D.n_rows : 0.0;
  D.lanes = pick_lanes(avg);
  // scalar matrices: sliced ELL with G lanes per row; G = the smallest power of two that yields >= 2^20
  // threads (enough waves to fill 256 CUs), accepted if the padding stays below 35 %
  int sellG = 0;
  if (allow_sell && A.br == 1 && A.bc == 1 && D.n_rows > 0 && D.nnz > 0) {
    int G = 1;
    while (G < 16 && D.n_rows * G < ((int64_t)1 << 20) && avg > 3.0 * G) G <<= 1;
    G = std::max(G, sell_long_row_lanes(avg));
    if (const char* e = std::getenv("AMGX_SELL_MAX_LANES")) G = std::max(1, std::min(G, std::atoi(e)));   // test hook
    for (int g = G; g >= 1; g >>= 1)
      if ((double)sell_stored(A, g) <= max_pad * (double)D.nnz) { sellG = g; break; }
  }
  // one thread per row with ragged rows: length-sorted windows instead of padding every slice to its longest row.  Short ragged
  // rows whose plain slices pad beyond max_pad (prolongations of the reference's setup: 1 ... 6 entries, 3.3 on average) take the
  // windowed form too if ITS padding is acceptable -- the CSR-vector kernel runs such a P at 2.3 TB/s (263 us at cfg 2)
  // (win < 0: the windowed form only as that fallback, never instead of an acceptable plain SELL image)
  const bool win_fallback_only = win < 0;
  if (win < 0) win = -win;
  bool windowed = win > 0 && !win_fallback_only && sellG == 1 && (double)sell_stored(A, 1) > 1.10 * (double)D.nnz && !std::getenv("AMGX_NO_SELL_WINDOW");
  const bool try_win = win > 0 && sellG == 0 && allow_sell && A.br == 1 && A.bc == 1 && D.n_rows >= 4 * win && D.nnz > 0 && avg <= 12.0 &&
                       !std::getenv("AMGX_NO_SELL_WINDOW") && !std::getenv("AMGX_NO_SELL_WINDOW_SHORT");
  std::vector<int32_t> rows;
  std::vector<uint16_t> rowloc;
  if (windowed || try_win) {
    rows.resize((size_t)A.n_rows);
    rowloc.resize((size_t)A.n_rows);
    par_for((A.n_rows + win - 1) / win, [&](int64_t q0, int64_t q1, int) {
      for (int64_t q = q0; q < q1; ++q) {
        const int64_t w0 = q * win, w1 = std::min<int64_t>(A.n_rows, w0 + win);
        for (int64_t i = w0; i < w1; ++i) rows[i] = (int32_t)i;
        std::stable_sort(rows.begin() + w0, rows.begin() + w1, [&](int32_t a, int32_t b) {
          return A.rowptr[a + 1] - A.rowptr[a] > A.rowptr[b + 1] - A.rowptr[b]; });
        for (int64_t i = w0; i < w1; ++i) rowloc[i] = (uint16_t)(rows[i] - w0);
      }
    }, 8);
    if (try_win) {
      int64_t stored = 0;
      for (int64_t sl = 0; sl * WAVE < A.n_rows; ++sl) {           // (sorted windows: the first row of a slice is its longest)
        const int32_t r = rows[sl * WAVE];
        stored += (int64_t)(A.rowptr[r + 1] - A.rowptr[r]) * WAVE;
      }
      windowed = (double)stored <= max_pad * (double)D.nnz;
    }
  }
  if (windowed) {
    HostSell S;
    build_sell(A, rows.data(), A.n_rows, false, 1, S, false);
    D.fmt = FMT_SELL;
    D.lanes = 1;
    D.n_slices = (int)(S.slice_ptr.size() - 1);
    D.stored = S.slice_ptr.back() & ~(int64_t)63;
    D.stream_bytes = S.stream_bytes + 2 * A.n_rows;
    upload_sell(S, D.sell);
    D.sell.win = win;
    D.sell.rowloc.upload(rowloc);
  } else if (sellG) {
    HostSell S;
    build_sell(A, nullptr, A.n_rows, rowrel_ok && A.n_cols >= A.n_rows, sellG, S, rowrel_ok);
    const bool wdiag = diag_override && S.diag_first && sellG == 1;
    if (wdiag) patch_sell_diag(S, A.n_rows, diag_override);
    D.fmt = FMT_SELL;
    D.lanes = sellG;
    D.n_slices = (int)(S.slice_ptr.size() - 1);
    D.stored = S.slice_ptr.back() & ~(int64_t)63;
    D.stream_bytes = S.stream_bytes;
    upload_sell(S, D.sell);
    D.sell.wdiag = wdiag ? 1 : 0;
  } else {
    D.fmt = FMT_CSRVEC;
    D.stored = D.nnz;
    D.stream_bytes = D.nnz * (8 * (int64_t)A.br * A.bc + 4) + 4 * (A.n_rows + 1);
    // square blocks with near-uniform row lengths: block SELL (one lane per scalar row); the CSR arrays are kept only
    // if a block Gauss-Seidel sweep needs them
    bool bsell = false;
    if (A.br == A.bc && (A.br == 2 || A.br == 3 || A.br == 6) && A.n_rows == A.n_cols && D.nnz > 0 && !std::getenv("AMGX_NO_BSELL"))
      bsell = build_bsell(A, D, 1.30);   // BSELL streams at ~5.6 TB/s vs ~4.0 TB/s of the CSR block kernels: worth up to ~35 % padding
    if (bsell && !keep_csr) return;
    std::vector<int32_t> rp(A.n_rows + 1);
    for (int64_t i = 0; i <= A.n_rows; ++i) rp[i] = (int32_t)A.rowptr[i];
    D.rowptr.upload(rp);
    D.col.upload(A.col, D.nnz);
    D.val.upload(A.val, (size_t)D.nnz * A.br * A.bc);
  }
}

// block Gauss-Seidel data: validated (a wrong block colouring would be a data race), blocks listed colour-major
static void build_bgs(const amgx_level_desc& d, DevLevel& L) {
  const int64_t n = d.A.n_rows;
  const int nb = d.bgs_n_blocks, nc = d.bgs_n_colors;
  if (nb < 0 || (nb > 0 && (!d.bgs_block_ptr || !d.bgs_block_rows || !d.bgs_dinv_ptr || !d.bgs_dinv || !d.bgs_color || nc <= 0)))
    throw Err("AMGX_SM_BGS needs blocks, their inverses and a block colouring (amgx_level_desc.bgs_*)");
  DevBGS& g = L.bgs;
  g.n_colors = nb ? nc : 0;
  std::vector<int32_t> blockof(n, -1);
  const int bs = d.A.br;
  for (int k = 0; k < nb; ++k) {
    const int64_t m = d.bgs_block_ptr[k + 1] - d.bgs_block_ptr[k];
    if (m < 0 || m * bs > BGS_MAX_M) throw Err("block Gauss-Seidel: block with more than 1024 scalar dofs");
    g.max_m = std::max(g.max_m, (int)(m * bs));
    if ((m * bs) * (m * bs) != d.bgs_dinv_ptr[k + 1] - d.bgs_dinv_ptr[k]) throw Err("block Gauss-Seidel: bgs_dinv_ptr does not match the block sizes");
    if (d.bgs_color[k] < 0 || d.bgs_color[k] >= nc) throw Err("block Gauss-Seidel: block colour out of range");
    for (int q = d.bgs_block_ptr[k]; q < d.bgs_block_ptr[k + 1]; ++q) {
      const int32_t i = d.bgs_block_rows[q];
      if (i < 0 || i >= n || blockof[i] >= 0) throw Err("block Gauss-Seidel: blocks must be disjoint sets of valid rows");
      blockof[i] = k;
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    const int k = blockof[i];
    if (k < 0) continue;
    for (int64_t e = d.A.rowptr[i]; e < d.A.rowptr[i + 1]; ++e) {
      const int64_t j = d.A.col[e];
      if (j >= n) continue;            // ghost column: frozen during the sweep
      const int kj = blockof[j];
      if (kj >= 0 && kj != k && d.bgs_color[kj] == d.bgs_color[k]) throw Err("invalid block colouring: two coupled blocks share a colour");
    }
  }
  g.color_ptr.assign(nc + 1, 0);
  for (int k = 0; k < nb; ++k) g.color_ptr[d.bgs_color[k] + 1]++;
  for (int c = 0; c < nc; ++c) g.color_ptr[c + 1] += g.color_ptr[c];
  std::vector<int32_t> list(nb);
  std::vector<int> pos(g.color_ptr.begin(), g.color_ptr.end() - 1);
  for (int k = 0; k < nb; ++k) list[pos[d.bgs_color[k]]++] = k;
  g.blocklist.upload(list);
  g.block_ptr.upload(d.bgs_block_ptr, (size_t)nb + 1);
  g.block_rows.upload(d.bgs_block_rows, (size_t)d.bgs_block_ptr[nb]);
  g.dinv_ptr.upload(d.bgs_dinv_ptr, (size_t)nb + 1);
  g.dinv.upload(d.bgs_dinv, (size_t)d.bgs_dinv_ptr[nb]);
  const int64_t nnz = d.A.rowptr[n];
  if (nnz >= (int64_t)2147483647) throw Err("block Gauss-Seidel: too many entries for 32-bit offsets");
  std::vector<int32_t> rp(n + 1);
  for (int64_t i = 0; i <= n; ++i) rp[i] = (int32_t)d.A.rowptr[i];
  g.rowptr.upload(rp);
  g.col.upload(d.A.col, (size_t)nnz);
  g.val.upload(d.A.val, (size_t)nnz * bs * bs);
}

// Compact chunks for the fused pre-smoothing + restriction kernels.  A chunk of 512 CONSECUTIVE rows of a lexicographically numbered
// grid is 2.4 grid lines of one plane, while the fine support of a coarse basis function spans 4 lines x 4 planes: 13 chunks hold a
// piece of every coarse row (cfg 2, reference hierarchy: 16 M partial sums for 1.24 M coarse rows, restrict_sum_kernel 53 us).  The
// matrix image keeps its natural 64-row slices (coalesced streaming, diagonal-first rows, row-relative 16-bit columns); only the
// ASSIGNMENT of slices to workgroups changes: slices that share coarse columns are clustered greedily over the slice graph (shared
// columns of P as weights), spc slices per chunk.  On the grid above a chunk becomes ~64 (x) x 3 x 3: 4-5 partial sums per coarse row.
// Returns the slice list (n_chunks * spc entries, -1 = no slice).  Independent ranges per setup thread (chunks do not cross them).
static std::vector<int32_t> cluster_slices(const amgx_matrix& P, int spc) {
  const int64_t nf = P.n_rows, nc = P.n_cols;
  const int64_t ns = (nf + WAVE - 1) / WAVE;
  // coarse columns of every slice (sorted, unique) and the slices of every coarse column
  std::vector<int64_t> sptr((size_t)ns + 1, 0);
  std::vector<std::vector<int32_t>> scols((size_t)ns);
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t sl = s0; sl < s1; ++sl) {
      std::vector<int32_t>& u = scols[sl];
      const int64_t r0 = sl * WAVE, r1 = std::min<int64_t>(nf, r0 + WAVE);
      u.assign(P.col + P.rowptr[r0], P.col + P.rowptr[r1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
    }
  }, 64);
  std::vector<int32_t> cptr((size_t)nc + 1, 0);
  for (int64_t sl = 0; sl < ns; ++sl) for (int32_t J : scols[sl]) cptr[J + 1]++;
  for (int64_t J = 0; J < nc; ++J) cptr[J + 1] += cptr[J];
  std::vector<int32_t> cs((size_t)cptr[nc]);
  { std::vector<int32_t> pos(cptr.begin(), cptr.end() - 1); for (int64_t sl = 0; sl < ns; ++sl) for (int32_t J : scols[sl]) cs[pos[J]++] = (int32_t)sl; }
  const int T = std::max(1, std::min<int>(setup_threads(), (int)(ns / (64 * spc) + 1)));
  std::vector<std::vector<int32_t>> lists((size_t)T);
  std::vector<std::thread> th;
  auto work = [&](int t) {
    const int64_t a = ns * t / T, b = ns * (t + 1) / T;
    std::vector<char> done((size_t)(b - a), 0);
    std::vector<int32_t> wgt((size_t)(b - a), 0), touched;
    std::vector<int32_t>& out = lists[t];
    for (int64_t seed = a; seed < b; ++seed) {
      if (done[seed - a]) continue;
      touched.clear();
      int64_t cur = seed;
      for (int q = 0; q < spc; ++q) {
        out.push_back((int32_t)cur);
        done[cur - a] = 1;
        if (q + 1 == spc) break;
        for (int32_t J : scols[cur])
          for (int32_t k = cptr[J]; k < cptr[J + 1]; ++k) {
            const int64_t o = cs[k];
            if (o < a || o >= b || done[o - a]) continue;
            if (wgt[o - a]++ == 0) touched.push_back((int32_t)o);
          }
        int64_t best = -1;
        int32_t bw = 0;
        for (int32_t o : touched) if (!done[o - a] && (wgt[o - a] > bw || (wgt[o - a] == bw && bw > 0 && o < best))) { bw = wgt[o - a]; best = o; }
        if (best < 0) {                                        // nothing adjacent is left: take the next free slice in order
          for (int64_t o = seed + 1; o < b; ++o) if (!done[o - a]) { best = o; break; }
          if (best < 0) { for (int r = q + 1; r < spc; ++r) out.push_back(-1); break; }
        }
        cur = best;
      }
      for (int32_t o : touched) wgt[o - a] = 0;
    }
  };
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  std::vector<int32_t> list;
  for (int t = 0; t < T; ++t) list.insert(list.end(), lists[t].begin(), lists[t].end());
  return list;
}

// CH = fine rows per chunk; threads = workgroup size of the kernel that consumes the chunk (= CH unless several lanes share a row)
// slice_list (optional, one thread per row only): chunk c = the 64-row slices slice_list[c * CH / 64 ...) (cluster_slices)
static void build_restrict(const amgx_matrix& P, DevRestrict& R, int CH = RESTRICT_CHUNK, int max_entries = RESTRICT_MAX_ENTRIES, int threads = 0,
                           const std::vector<int32_t>* slice_list = nullptr) {
  if (threads <= 0) threads = CH;
  const int64_t nf = P.n_rows, nc = P.n_cols;
  const int spc = CH / WAVE;
  if (slice_list && (CH % WAVE != 0 || slice_list->size() % spc != 0)) throw Err("build_restrict: slice list does not match the chunk size");
  const int64_t nch = slice_list ? (int64_t)slice_list->size() / spc : (nf + CH - 1) / CH;
  // pass 1 (parallel over chunks): slots (= distinct coarse columns) per chunk; a chunk's entries are the P entries of its rows
  std::vector<int32_t> chunk_slot(nch + 1, 0);
  std::vector<char> too_long(setup_threads(), 0);
  std::vector<int64_t> fullest(setup_threads(), 0);
  struct Trip { int32_t J; uint16_t i; double w; };
  auto chunk_trips = [&](int64_t c, std::vector<Trip>& t) {
    t.clear();
    if (slice_list) {
      for (int q = 0; q < spc; ++q) {
        const int64_t sl = (*slice_list)[c * spc + q];
        if (sl < 0) continue;
        const int64_t r0 = sl * WAVE, r1 = std::min<int64_t>(nf, r0 + WAVE);
        for (int64_t i = r0; i < r1; ++i)
          for (int64_t k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) t.push_back({P.col[k], (uint16_t)(q * WAVE + (i - r0)), P.val[k]});
      }
    } else {
      const int64_t r0 = c * CH, r1 = std::min<int64_t>(nf, r0 + CH);
      for (int64_t i = r0; i < r1; ++i)
        for (int64_t k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) t.push_back({P.col[k], (uint16_t)(i - r0), P.val[k]});
    }
    std::stable_sort(t.begin(), t.end(), [](const Trip& a, const Trip& b) { return a.J < b.J; });
  };
  std::vector<int64_t> chunk_ent(nch + 1, 0);            // first entry of every chunk
  par_for(nch, [&](int64_t c0, int64_t c1, int tid) {
    std::vector<Trip> t;
    for (int64_t c = c0; c < c1; ++c) {
      chunk_trips(c, t);
      if ((int64_t)t.size() > max_entries) { too_long[tid] = 1; return; }
      fullest[tid] = std::max<int64_t>(fullest[tid], (int64_t)t.size());
      int32_t nsl = 0;
      for (size_t q = 0; q < t.size(); ++q) if (q == 0 || t[q].J != t[q - 1].J) nsl++;
      chunk_slot[c + 1] = nsl;
      chunk_ent[c + 1] = (int64_t)t.size();
    }
  }, 8);
  for (char c : too_long) if (c) return;             // rows too long for the LDS product buffer: keep the P^T form
  for (int64_t c = 0; c < nch; ++c) { chunk_slot[c + 1] += chunk_slot[c]; chunk_ent[c + 1] += chunk_ent[c]; }
  const int64_t n_slots = chunk_slot[nch], n_ent = P.rowptr[nf];
  std::vector<int32_t> slot_ptr((size_t)n_slots + 1, 0), slot_col((size_t)n_slots);
  std::vector<double> w((size_t)n_ent);
  std::vector<uint16_t> fi((size_t)n_ent);
  // pass 2: fill (entries of chunk c start at chunk_ent[c])
  par_for(nch, [&](int64_t c0, int64_t c1, int) {
    std::vector<Trip> t;
    for (int64_t c = c0; c < c1; ++c) {
      chunk_trips(c, t);
      int64_t e = chunk_ent[c];
      int64_t sl = chunk_slot[c];
      for (size_t q = 0; q < t.size(); ++q) {
        if (q == 0 || t[q].J != t[q - 1].J) { slot_ptr[sl] = (int32_t)e; slot_col[sl] = t[q].J; sl++; }
        w[e] = t[q].w; fi[e] = t[q].i; e++;
      }
    }
  }, 8);
  slot_ptr[n_slots] = (int32_t)n_ent;
  const int64_t ns = (int64_t)slot_col.size();
  if ((int64_t)slot_ptr.size() != ns + 1) throw Err("build_restrict: internal slot count mismatch");
  std::vector<int32_t> optr(nc + 1, 0), oidx(ns);
  for (int64_t sidx = 0; sidx < ns; ++sidx) optr[slot_col[sidx] + 1]++;
  for (int64_t J = 0; J < nc; ++J) optr[J + 1] += optr[J];
  std::vector<int32_t> pos(optr.begin(), optr.end() - 1);
  for (int64_t sidx = 0; sidx < ns; ++sidx) oidx[pos[slot_col[sidx]]++] = (int32_t)sidx;   // ascending chunk order per row
  R.n_chunks = (int)nch; R.n_slots = ns;
  int64_t mx_chunk = 0;
  for (int64_t v : fullest) mx_chunk = std::max(mx_chunk, v);
  R.ept = mx_chunk <= (int64_t)4 * threads ? 4 : 6;
  if ((CH == 256 || CH == 128) && threads == 512 && max_entries == 4 * 512) R.ept = mx_chunk <= (int64_t)2 * threads ? 2 : 4;     // (local-window chunks: see lw_image)
  if (const char* e = std::getenv("AMGX_FUSED_EPT_MAX")) if (R.ept > std::atoi(e)) { R = DevRestrict(); return; }     // (A/B hook: keep the separate kernels instead)
  R.chunk_slot.upload(chunk_slot); R.slot_ptr.upload(slot_ptr); R.optr.upload(optr);
  // Measured NON-win (profiles/r01/restrict_fused.txt): storing the partial sums row by row (scattered stores in the
  // producer, streaming loads in restrict_sum_kernel) makes the cycle 2-4 % slower at cfg 2; off unless AMGX_RSUM_SORT=1.
  if (!std::getenv("AMGX_RSUM_SORT")) R.oidx.upload(oidx);
  else {
    std::vector<int32_t> dest(ns);
    for (int64_t k = 0; k < ns; ++k) dest[oidx[k]] = (int32_t)k;
    R.dest.upload(dest);
  }
  R.w.upload(w); R.fi.upload(fi);
  R.part.alloc((size_t)std::max<int64_t>(1, ns));
  if (slice_list) R.slice_list.upload(*slice_list);
}

// "local window" image of a long-row scalar matrix (sell_lw_pre_restrict_kernel): per chunk of LW_ROWS consecutive rows the sorted
// list of its distinct columns; the SELL-2 image stores indices into that list.  vals: the (scaled) values in CSR order.
// Returns false (nothing built) if a chunk touches more than LW_CAP distinct columns.
static bool build_sell_lw(const amgx_matrix& A, const double* vals, int G, DevMatrix& D, DevBuf<int32_t>& d_cptr, DevBuf<int32_t>& d_ccol) {
  const int64_t n = A.n_rows, nnz = A.rowptr[n];
  const int LW_ROWS = 512 / G;
  const int64_t nch = (n + LW_ROWS - 1) / LW_ROWS;
  std::vector<int32_t> cnt((size_t)nch + 1, 0);
  RawVec<int32_t> lcol;
  lcol.resize((size_t)std::max<int64_t>(1, nnz));
  std::vector<std::vector<int32_t>> lists((size_t)nch);
  std::vector<char> no16((size_t)n, 0);                 // rows of chunks whose window would not fit: global 32-bit columns, no window
  int64_t cap = LW_CAP;
  const char* tcap = std::getenv("AMGX_LW_TEST_CAP");    // (tests: a smaller capacity sends some chunks through the no-window path)
  if (tcap) cap = std::min<int64_t>(cap, std::atoll(tcap));
  std::vector<int64_t> n_over(setup_threads(), 0);
  par_for(nch, [&](int64_t c0, int64_t c1, int t) {
    std::vector<int32_t> u;
    for (int64_t c = c0; c < c1; ++c) {
      const int64_t r0 = c * LW_ROWS, r1 = std::min<int64_t>(n, r0 + LW_ROWS);
      u.assign(A.col + A.rowptr[r0], A.col + A.rowptr[r1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      if ((int64_t)u.size() > cap) {
        for (int64_t i = r0; i < r1; ++i) no16[i] = 1;
        for (int64_t k = A.rowptr[r0]; k < A.rowptr[r1]; ++k) lcol[k] = A.col[k];
        n_over[t]++;
        continue;
      }
      for (int64_t k = A.rowptr[r0]; k < A.rowptr[r1]; ++k) lcol[k] = (int32_t)(std::lower_bound(u.begin(), u.end(), A.col[k]) - u.begin());
      cnt[c + 1] = (int32_t)u.size();
      lists[c] = u;
    }
  }, 4);
  int64_t overs = 0;
  for (int64_t v : n_over) overs += v;
  if (overs * 20 > nch && !tcap) return false;          // (more than 5 % of the chunks without a window: the plain image is the better choice)
  for (int64_t c = 0; c < nch; ++c) cnt[c + 1] += cnt[c];
  std::vector<int32_t> ccol((size_t)std::max<int32_t>(1, cnt[nch]));
  par_for(nch, [&](int64_t c0, int64_t c1, int) { for (int64_t c = c0; c < c1; ++c) std::copy(lists[c].begin(), lists[c].end(), ccol.begin() + cnt[c]); }, 64);
  amgx_matrix L = A;
  L.col = lcol.data();
  L.val = vals;
  HostSell S;
  build_sell(L, nullptr, n, false, G, S, false, &no16);
  const int64_t ns = (int64_t)S.slice_ptr.size() - 1;
  // (every slice of a chunk with a window fits 16-bit deltas by construction: indices < LW_CAP; anything else would be a builder bug)
  int64_t want16 = 0;
  for (int64_t sl = 0; sl < ns; ++sl) if (!no16[std::min<int64_t>(n - 1, sl * (WAVE / G))]) ++want16;
  if (S.n_comp_slices != want16) return false;
  D.n_rows = n; D.n_cols = A.n_cols; D.br = D.bc = 1; D.nnz = nnz;
  D.fmt = FMT_SELL; D.lanes = G;
  D.n_slices = (int)ns;
  D.stored = S.slice_ptr.back() & ~(int64_t)63;
  D.stream_bytes = S.stream_bytes + 4 * (int64_t)ccol.size() + 4 * (nch + 1);
  upload_sell(S, D.sell);
  d_cptr.upload(cnt);
  d_ccol.upload(ccol);
  return true;
}

// local-window form of a WINDOWED SELL image (sell_lw_win_spmv_kernel): windows of SELL_WIN consecutive rows stored by decreasing
// length as in upload_matrix, columns = indices into the window's sorted list of distinct columns (at most QW_CAP; windows beyond
// that keep 32-bit global columns).  rowptr / col / val: host CSR.  Returns false if too many windows miss the capacity.
static bool build_sell_lw_windowed(int64_t n, int64_t n_cols, const int64_t* rowptr, const int32_t* col, const double* val, DevMatrix& D,
                                   DevBuf<int32_t>& d_cptr, DevBuf<int32_t>& d_ccol) {
  const int win = SELL_WIN;
  const int64_t nnz = rowptr[n];
  const int64_t nw = (n + win - 1) / win;
  SetupClock clk;
  std::vector<int32_t> cnt((size_t)nw + 1, 0);
  RawVec<int32_t> lcol;
  lcol.resize((size_t)std::max<int64_t>(1, nnz));
  std::vector<std::vector<int32_t>> lists((size_t)nw);
  std::vector<char> no16((size_t)n, 0);
  std::vector<int32_t> rows((size_t)n);
  std::vector<uint16_t> rowloc((size_t)n);
  std::vector<int64_t> n_over(setup_threads(), 0);
  int64_t cap = QW_CAP;
  const char* tcap = std::getenv("AMGX_LW_TEST_CAP");
  if (tcap) cap = std::min<int64_t>(cap, std::max<int64_t>(8, std::atoll(tcap) / 4));
  par_for(nw, [&](int64_t q0, int64_t q1, int t) {
    std::vector<int32_t> u;
    for (int64_t q = q0; q < q1; ++q) {
      const int64_t w0 = q * win, w1 = std::min<int64_t>(n, w0 + win);
      for (int64_t i = w0; i < w1; ++i) rows[i] = (int32_t)i;
      std::stable_sort(rows.begin() + w0, rows.begin() + w1, [&](int32_t a, int32_t b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
      for (int64_t i = w0; i < w1; ++i) rowloc[i] = (uint16_t)(rows[i] - w0);
      u.assign(col + rowptr[w0], col + rowptr[w1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      if ((int64_t)u.size() > cap) {
        for (int64_t i = w0; i < w1; ++i) no16[i] = 1;
        for (int64_t k = rowptr[w0]; k < rowptr[w1]; ++k) lcol[k] = col[k];
        n_over[t]++;
        continue;
      }
      for (int64_t k = rowptr[w0]; k < rowptr[w1]; ++k) lcol[k] = (int32_t)(std::lower_bound(u.begin(), u.end(), col[k]) - u.begin());
      cnt[q + 1] = (int32_t)u.size();
      lists[q] = u;
    }
  }, 4);
  clk.lap("  lw-win: window lists + local columns");
  int64_t overs = 0;
  for (int64_t v : n_over) overs += v;
  if (overs * 20 > nw && !tcap) return false;
  for (int64_t q = 0; q < nw; ++q) cnt[q + 1] += cnt[q];
  std::vector<int32_t> ccol((size_t)std::max<int32_t>(1, cnt[nw]));
  par_for(nw, [&](int64_t q0, int64_t q1, int) { for (int64_t q = q0; q < q1; ++q) std::copy(lists[q].begin(), lists[q].end(), ccol.begin() + cnt[q]); }, 64);
  amgx_matrix L{};
  L.n_rows = n; L.n_cols = n_cols; L.br = L.bc = 1;
  L.rowptr = rowptr; L.col = lcol.data(); L.val = val;
  clk.lap("  lw-win: list copy");
  HostSell S;
  build_sell(L, rows.data(), n, false, 1, S, false, &no16);
  clk.lap("  lw-win: host SELL builder");
  D.n_rows = n; D.n_cols = n_cols; D.br = D.bc = 1; D.nnz = nnz;
  D.fmt = FMT_SELL; D.lanes = 1;
  D.n_slices = (int)(S.slice_ptr.size() - 1);
  D.stored = S.slice_ptr.back() & ~(int64_t)63;
  D.stream_bytes = S.stream_bytes + 2 * n + 4 * (int64_t)ccol.size() + 4 * (nw + 1);
  upload_sell(S, D.sell);
  D.sell.win = win;
  D.sell.rowloc.upload(rowloc);
  d_cptr.upload(cnt);
  d_ccol.upload(ccol);
  clk.lap("  lw-win: upload");
  return true;
}

// ---------------------------------------------------------------------------------------------------
// the handle
// ---------------------------------------------------------------------------------------------------

enum { PART_ALL = 0, PART_INT = 1, PART_BND = 2 };
struct Span { int part = PART_ALL; int64_t n_int = 0; };   // see Handle::spmv_ep

struct Handle {
  int device = 0;
  std::vector<DevLevel> lev;
  int cycle = AMGX_CYCLE_V, clev = AMGX_CLEV_INV;
  int64_t coarse_n = 0;
  int64_t coarse_ld = 0;                // row stride of coarse_inv (== coarse_n for an inverse handed over by the host)
  double coarse_pivot = 0.0;            // device-side inversion: smallest pivot ratio met (dense_spd.hpp)
  DevBuf<double> coarse_inv;
  hipStream_t own_stream = nullptr, stream = nullptr;
  bool use_graph = true;
  bool skip_rsum = false;               // amgx_time_op(op 7): time sell_pre_restrict_kernel alone
  // amgx_time_op(op 8): HIP events around the fused down kernel of `probe_level` INSIDE the cycle (direct launches)
  // (op 9): the same around the backward block-hybrid Gauss-Seidel sweep of `probe_level`
  int probe_level = -1, probe_kind = 8;
  hipEvent_t probe_e0 = nullptr, probe_e1 = nullptr;
  int ep_nt = 1;                        // non-temporal epilogue operands (AMGX_NO_EP_NT=1 disables)
  int tail_level = -1;                  // first level executed by tail_kernel (-1: no fused tail)
  int tail_ops = 0;
  DevBuf<TailOp> tail_prog;
  // collapsed coarse levels (kernels.hpp, dense_op_gemv_kernel): the V-cycle on the levels >= dense_level as ONE dense
  // operator, formed at create time by the device's own sub-cycle on the unit vectors (build_dense_tail)
  int dense_level = -1;
  int dense_n = 0, dense_ld = 0;
  DevBuf<double> dense_op;
  std::string err;
  struct GraphKey { const double* b; double* x; int kind; bool operator<(const GraphKey& o) const { return std::tie(b, x, kind) < std::tie(o.b, o.x, o.kind); } };
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::vector<GraphKey> graph_age;      // capture order
  // staging for host-pointer calls
  DevBuf<double> stage[3];
  DevBuf<double> kr_ws[6];              // work vectors of amgx_pcg / amgx_gmres (krylov.hpp), kept between solves
  DevBuf<double> stage_raw[3];          // host vectors of permuted levels: raw copy before / after the renumbering
  // Gauss-Seidel levels are stored in colour-major numbering (see LevelPerm in create()); perm[l][new] = old row of the
  // caller's numbering, empty = identity.  Every C-ABI entry point translates its vectors (struct Staged).
  std::vector<DevBuf<int32_t>> perm;
  bool permuted(int l) const { return l >= 0 && l < (int)perm.size() && perm[l].n > 0; }
  void perm_gather(int l, const double* src, double* dst) {
    const int64_t len = lev[l].len();
    if (!len) return;
    hipLaunchKernelGGL(perm_gather_kernel, dim3(grid_for(len)), dim3(BLOCK), 0, stream, len, lev[l].bs, perm[l].p, src, dst);
    HIPCHK(hipGetLastError());
  }
  void perm_scatter(int l, const double* src, double* dst) {
    const int64_t len = lev[l].len();
    if (!len) return;
    hipLaunchKernelGGL(perm_scatter_kernel, dim3(grid_for(len)), dim3(BLOCK), 0, stream, len, lev[l].bs, perm[l].p, src, dst);
    HIPCHK(hipGetLastError());
  }

  ~Handle() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }

  int n_levels() const { return (int)lev.size(); }

  // ------------------------------------------------------------------ primitive ops (all async on `stream`)
  static int grid_for(int64_t threads) { return (int)std::max<int64_t>(1, (threads + BLOCK - 1) / BLOCK); }

  // Part of the rows a launch covers.  Rank-partitioned levels order their owned rows [interior | boundary] (interior =
  // no ghost column; the analogue of the reference's split_ind stages, gssmoother.cpp:664-678): the interior part runs
  // while the halo exchange is in flight, the boundary part after it (hybrid_base_smoother.cpp:501-574).  A kernel works
  // in units of u rows (slice / window / chunk); units that straddle n_int belong to the boundary part.  Formats that
  // cannot be split run completely in the boundary part (correct, no overlap).
  using Span = ::amgx::Span;
  enum { PART_ALL = ::amgx::PART_ALL, PART_INT = ::amgx::PART_INT, PART_BND = ::amgx::PART_BND };
  static void unit_range(const Span& sp, int64_t u, int64_t n_units, int64_t& a, int64_t& b) {
    const int64_t split = std::min<int64_t>(n_units, sp.n_int / u);
    if (sp.part == PART_ALL) { a = 0; b = n_units; }
    else if (sp.part == PART_INT) { a = 0; b = split; }
    else { a = split; b = n_units; }
  }

  template <int EP>
  void spmv_ep(const DevMatrix& M, const double* x, double* y, const EpArgs& ep, const Span sp = Span()) {
    if (M.n_rows == 0) return;
    if (M.fmt == FMT_RB) {
      if constexpr (EP != EP_MULT && EP != EP_AXPY) throw Err("rigid-body transfer blocks: y = M x and y = yin + s M x only");
      else {
        if (sp.part == PART_INT) return;             // (not split: everything runs in the boundary part)
        const DevMatrix::Rb& R = M.rb;
        const RbMat V = R.view();
        if (!R.transposed) {
          const int grid = grid_for(M.n_rows);
#define LAUNCH_RBP(BF, BC, DM) hipLaunchKernelGGL((rb_prolong_kernel<BF, BC, DM, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, V, x, y, ep)
          if (R.dim == 3 && R.bf == 6) LAUNCH_RBP(6, 6, 3);
          else if (R.dim == 3) LAUNCH_RBP(3, 6, 3);
          else if (R.dim == 2 && R.bf == 3) LAUNCH_RBP(3, 3, 2);
          else if (R.dim == 2) LAUNCH_RBP(2, 3, 2);
          else LAUNCH_RBP(2, 2, 0);
#undef LAUNCH_RBP
        } else {
          constexpr int G = 16;
          const int grid = grid_for(M.n_rows * G);
#define LAUNCH_RBR(BF, BC, DM) hipLaunchKernelGGL((rb_restrict_kernel<BF, BC, DM, G, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, V, x, y, ep)
          if (R.dim == 3 && R.bf == 6) LAUNCH_RBR(6, 6, 3);
          else if (R.dim == 3) LAUNCH_RBR(3, 6, 3);
          else if (R.dim == 2 && R.bf == 3) LAUNCH_RBR(3, 3, 2);
          else if (R.dim == 2) LAUNCH_RBR(2, 3, 2);
          else LAUNCH_RBR(2, 2, 0);
#undef LAUNCH_RBR
        }
        HIPCHK(hipGetLastError());
        return;
      }
    }
    if (M.fmt == FMT_SELL && M.sell.win) {
      if (M.sell.win != SELL_WIN) throw Err("windowed SELL: unexpected window size");
      int64_t a, b;
      unit_range(sp, SELL_WIN, (M.n_rows + SELL_WIN - 1) / SELL_WIN, a, b);
      if (b > a)
        hipLaunchKernelGGL((sell_win_spmv_kernel<SELL_WIN, EP>), dim3((int)(b - a)), dim3(SELL_WIN), 0, stream, M.n_rows, (int)a, M.sell.view(), M.sell.rowloc.p, x, y, ep);
    } else if (M.fmt == FMT_SELL) {
      int64_t a, b;
      unit_range(sp, WAVE / M.lanes, M.n_slices, a, b);
      if (b <= a) return;
      const int grid = (int)((b - a + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
#define LAUNCH_SELL(G) hipLaunchKernelGGL((sell_spmv_kernel<G, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, (int)a, (int)b, M.sell.view(), x, y, ep)
      switch (M.lanes) {
        case 1: LAUNCH_SELL(1); break;
        case 2: LAUNCH_SELL(2); break;
        case 4: LAUNCH_SELL(4); break;
        case 8: LAUNCH_SELL(8); break;
        default: LAUNCH_SELL(16); break;
      }
#undef LAUNCH_SELL
    } else if (M.br == 1 && M.bc == 1) {
      int64_t a, b;
      unit_range(sp, 1, M.n_rows, a, b);
      if (b <= a) return;
      const int grid = grid_for((b - a) * M.lanes);
#define LAUNCH_CSR(G) hipLaunchKernelGGL((csrvec_spmv_kernel<G, EP>), dim3(grid), dim3(BLOCK), 0, stream, a, b, M.rowptr.p, M.col.p, M.val.p, x, y, ep)
      switch (M.lanes) {
        case 2: LAUNCH_CSR(2); break;
        case 4: LAUNCH_CSR(4); break;
        case 8: LAUNCH_CSR(8); break;
        case 16: LAUNCH_CSR(16); break;
        case 32: LAUNCH_CSR(32); break;
        default: LAUNCH_CSR(64); break;
      }
#undef LAUNCH_CSR
    } else if constexpr (EP == EP_PRE) {
      throw Err("EP_PRE is only built for scalar matrices");
    } else if (M.fmt == FMT_BSELL) {
      // units of RB = 64 / bs block rows (one slice); slices that straddle n_int belong to the boundary part
      int64_t a, b;
      unit_range(sp, WAVE / M.br, M.n_slices, a, b);
      if (b <= a) return;
      const int grid = (int)((b - a + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
      if (M.br == 6) hipLaunchKernelGGL((bsell_spmv_kernel<6, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, (int)a, (int)b, M.bsell.view(), x, y, ep);
      else if (M.br == 3) hipLaunchKernelGGL((bsell_spmv_kernel<3, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, (int)a, (int)b, M.bsell.view(), x, y, ep);
      else hipLaunchKernelGGL((bsell_spmv_kernel<2, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, (int)a, (int)b, M.bsell.view(), x, y, ep);
    } else if (sp.part == PART_INT) {
      return;                                   // the CSR block formats are not split: everything runs in the boundary part
    } else if (M.br >= 2 && M.bc >= 2 && (EP != EP_JAC || M.br == M.bc) && (M.br == M.bc || M.nnz >= 6 * M.n_rows)) {
      // (short rectangular rows, i.e. prolongations with <= 4 blocks per row, stay with the lane-per-block kernel: measured)
      // row-per-lane block CSR kernel; W lane groups per block row chosen from the average row length
      const double avg = M.n_rows ? (double)M.nnz / (double)M.n_rows : 0.0;
      const int W = avg >= 48.0 ? 4 : (avg >= 20.0 ? 2 : 1);
      const int rpw = WAVE / (M.br * W);
      const int64_t waves = (M.n_rows + rpw - 1) / rpw;
      const int grid = (int)std::max<int64_t>(1, (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
#define LAUNCH_RL(BR, BC, WW) hipLaunchKernelGGL((bcsr_rowlane_kernel<BR, BC, WW, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, M.rowptr.p, M.col.p, M.val.p, x, y, ep)
#define LAUNCH_RLW(BR, BC) { if (W == 4) LAUNCH_RL(BR, BC, 4); else if (W == 2) LAUNCH_RL(BR, BC, 2); else LAUNCH_RL(BR, BC, 1); }
      const int key = M.br * 10 + M.bc;
      if constexpr (EP == EP_JAC) {
        switch (key) {
          case 22: LAUNCH_RLW(2, 2); break;
          case 33: LAUNCH_RLW(3, 3); break;
          case 66: LAUNCH_RLW(6, 6); break;
          default: throw Err("unsupported block shape");
        }
      } else {
        switch (key) {
          case 22: LAUNCH_RLW(2, 2); break;
          case 33: LAUNCH_RLW(3, 3); break;
          case 66: LAUNCH_RLW(6, 6); break;
          case 36: LAUNCH_RLW(3, 6); break;
          case 63: LAUNCH_RLW(6, 3); break;
          case 23: LAUNCH_RLW(2, 3); break;
          case 32: LAUNCH_RLW(3, 2); break;
          default: throw Err("unsupported block shape " + std::to_string(M.br) + "x" + std::to_string(M.bc));
        }
      }
#undef LAUNCH_RLW
#undef LAUNCH_RL
    } else {
      const int G = std::min(M.lanes, 16) < 2 ? 2 : std::min(M.lanes, 16);
      const int grid = grid_for(M.n_rows * G);
#define LAUNCH_B(BR, BC, GG) hipLaunchKernelGGL((bcsrvec_spmv_kernel<BR, BC, GG, EP>), dim3(grid), dim3(BLOCK), 0, stream, M.n_rows, M.rowptr.p, M.col.p, M.val.p, x, y, ep)
#define LAUNCH_BG(BR, BC)                                                                   \
  switch (G) { case 2: LAUNCH_B(BR, BC, 2); break; case 4: LAUNCH_B(BR, BC, 4); break;     \
               case 8: LAUNCH_B(BR, BC, 8); break; default: LAUNCH_B(BR, BC, 16); break; }
      const int key = M.br * 10 + M.bc;
      switch (key) {
        case 22: LAUNCH_BG(2, 2); break;
        case 33: LAUNCH_BG(3, 3); break;
        case 66: LAUNCH_BG(6, 6); break;
        case 36: LAUNCH_BG(3, 6); break;
        case 63: LAUNCH_BG(6, 3); break;
        case 23: LAUNCH_BG(2, 3); break;
        case 32: LAUNCH_BG(3, 2); break;
        case 13: LAUNCH_BG(1, 3); break;
        case 31: LAUNCH_BG(3, 1); break;
        case 16: LAUNCH_BG(1, 6); break;
        case 61: LAUNCH_BG(6, 1); break;
        case 12: LAUNCH_BG(1, 2); break;
        case 21: LAUNCH_BG(2, 1); break;
        default: throw Err("unsupported block shape " + std::to_string(M.br) + "x" + std::to_string(M.bc));
      }
#undef LAUNCH_BG
#undef LAUNCH_B
    }
    HIPCHK(hipGetLastError());
  }

  void mult(const DevMatrix& M, const double* x, double* y, const Span sp = Span()) { spmv_ep<EP_MULT>(M, x, y, EpArgs{nullptr, nullptr, nullptr, 0.0, nullptr, 0}, sp); }
  void residual(const DevMatrix& M, const double* x, const double* b, double* r, const Span sp = Span()) { spmv_ep<EP_RES>(M, x, r, EpArgs{b, nullptr, nullptr, 0.0, nullptr, ep_nt & EPF_HOIST}, sp); }
  // y = yin + s * M x
  void mult_add(const DevMatrix& M, double s, const double* x, const double* yin, double* y, const Span sp = Span()) { spmv_ep<EP_AXPY>(M, x, y, EpArgs{nullptr, yin, nullptr, s, nullptr, ep_nt & EPF_HOIST}, sp); }
  // xout = xin + omega * dinv * (b - A xin)
  void jacobi_fused(const DevLevel& L, const double* xin, const double* b, double* xout, const Span sp = Span()) {
    if (xin == xout) throw Err("jacobi_fused: in-place update is not allowed");
    spmv_ep<EP_JAC>(L.A, xin, xout, EpArgs{b, xin, L.dinv.p, L.omega, nullptr, ep_nt}, sp);
  }

  void zero(double* v, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(vec_zero_kernel, dim3(grid_for((n + 1) / 2)), dim3(BLOCK), 0, stream, n, v);
    HIPCHK(hipGetLastError());
  }
  void copy(double* dst, const double* src, int64_t n) {
    if (n <= 0 || dst == src) return;
    hipLaunchKernelGGL(vec_copy_kernel, dim3(grid_for((n + 1) / 2)), dim3(BLOCK), 0, stream, n, src, dst);
    HIPCHK(hipGetLastError());
  }

  // x (+)= omega * dinv * v
  // rows: block rows to process (default: the owned rows; a rank-partitioned level may ask for its ghost rows too)
  void diag_apply(const DevLevel& L, const double* v, double* x, bool add, int64_t rows = -1) {
    if (rows < 0) rows = L.n;
    if (rows == 0) return;
    const int grid = grid_for(rows);
#define LAUNCH_D(BS)                                                                                                   \
  if (add) hipLaunchKernelGGL((diag_apply_kernel<BS, true>), dim3(grid), dim3(BLOCK), 0, stream, rows, L.dinv.p, v, x, L.omega); \
  else hipLaunchKernelGGL((diag_apply_kernel<BS, false>), dim3(grid), dim3(BLOCK), 0, stream, rows, L.dinv.p, v, x, L.omega)
    switch (L.bs) {
      case 1: LAUNCH_D(1); break;
      case 2: LAUNCH_D(2); break;
      case 3: LAUNCH_D(3); break;
      case 6: LAUNCH_D(6); break;
      default: throw Err("unsupported block size " + std::to_string(L.bs));
    }
#undef LAUNCH_D
    HIPCHK(hipGetLastError());
  }

  void axpy(int64_t n, double s, const double* x, double* y) {
    if (!n) return;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, stream, n, s, x, y);
    HIPCHK(hipGetLastError());
  }

  // one multicolour GS sweep (RHS form), forward: colours ascending, backward: descending
  // cbeg, cend: only the colours [cbeg, cend) (cend < 0: all) -- the stages of the hybrid smoother on rank-partitioned
  // levels are colour ranges (dist.hpp)
  void gs_sweep(const DevLevel& L, int dir, double* x, const double* b, bool lower_only = false, int cbeg = 0, int cend = -1) {
    Range rg("GSS3<bs=" + std::to_string(L.bs) + ">::SmoothRHS");
    const DevGS& g = L.gs;
    const DevMatrix::Sell& copy = (lower_only && g.has_split) ? g.lower : g.sell;
    if (L.gsb.on() || L.bgsb.on()) throw Err("gs_sweep: the level uses the block-hybrid form");
    if (g.n_colors == 0 && L.n > 0) throw Err("Gauss-Seidel requested but the level has no colouring");
    if (cend < 0 || cend > g.n_colors) cend = g.n_colors;
    for (int q = cbeg; q < cend; ++q) {
      const int c = dir == 0 ? q : cend - 1 - (q - cbeg);
      if (L.bs == 1) {
        const int s0 = g.color_slice_ptr[c], s1 = g.color_slice_ptr[c + 1];
        if (s1 == s0) continue;
        const int grid = (s1 - s0 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
#define LAUNCH_GSC(GG) hipLaunchKernelGGL((gs_color_kernel<GG>), dim3(grid), dim3(BLOCK), 0, stream, s0, s1, copy.view(), g.rowid.p, L.dinv.p, b, x)
        switch (g.lanes) {
          case 1: LAUNCH_GSC(1); break;
          case 2: LAUNCH_GSC(2); break;
          case 4: LAUNCH_GSC(4); break;
          case 8: LAUNCH_GSC(8); break;
          default: LAUNCH_GSC(16); break;
        }
#undef LAUNCH_GSC
      } else if (g.bsell_ok) {
        const int s0 = g.color_slice_ptr[c], s1 = g.color_slice_ptr[c + 1];
        if (s1 == s0) continue;
        const int grid = (s1 - s0 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
        const BSellMat BM = (lower_only && g.bsplit) ? g.blower.bsell.view() : g.bcopy.bsell.view();
        if (L.bs == 6) hipLaunchKernelGGL((bgs_bsell_color_kernel<6>), dim3(grid), dim3(BLOCK), 0, stream, s0, s1, BM, g.rowid.p, L.dinv.p, b, x);
        else if (L.bs == 3) hipLaunchKernelGGL((bgs_bsell_color_kernel<3>), dim3(grid), dim3(BLOCK), 0, stream, s0, s1, BM, g.rowid.p, L.dinv.p, b, x);
        else hipLaunchKernelGGL((bgs_bsell_color_kernel<2>), dim3(grid), dim3(BLOCK), 0, stream, s0, s1, BM, g.rowid.p, L.dinv.p, b, x);
      } else {
        const int r0 = g.color_row_ptr[c], r1 = g.color_row_ptr[c + 1];
        if (r1 == r0) continue;
        const double avg = L.A.n_rows ? (double)L.A.nnz / (double)L.A.n_rows : 0.0;
        int W = avg >= 48.0 ? 4 : (avg >= 20.0 ? 2 : 1);
        if (avg >= 32.0 && r1 - r0 <= 4096) W = 8;      // short colours of long rows: latency, not bandwidth -- spread the row further
        const int rpw = WAVE / (L.bs * W);
        const int64_t waves = ((int64_t)(r1 - r0) + rpw - 1) / rpw;
        const int grid = (int)std::max<int64_t>(1, (waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
#define LAUNCH_GS(BS, WW) hipLaunchKernelGGL((bgs_color_kernel<BS, WW>), dim3(grid), dim3(BLOCK), 0, stream, r0, r1, g.rowlist.p, L.A.rowptr.p, L.A.col.p, L.A.val.p, L.dinv.p, b, x)
#define LAUNCH_GSW(BS) { if (W == 8) LAUNCH_GS(BS, 8); else if (W == 4) LAUNCH_GS(BS, 4); else if (W == 2) LAUNCH_GS(BS, 2); else LAUNCH_GS(BS, 1); }
        switch (L.bs) {
          case 2: LAUNCH_GSW(2); break;
          case 3: LAUNCH_GSW(3); break;
          case 6: LAUNCH_GSW(6); break;
          default: throw Err("unsupported block size for GS");
        }
#undef LAUNCH_GSW
#undef LAUNCH_GS
      }
      HIPCHK(hipGetLastError());
    }
  }

  // one block-hybrid Gauss-Seidel sweep (gsb_sweep_kernel): ONE launch; xin == nullptr: sweep from x = 0
  // blk0, blk1: only the blocks [blk0, blk1) (blk1 < 0: all): blocks are independent of each other within a sweep, so the
  // rank-partitioned driver sweeps its boundary and interior blocks in separate launches around the halo exchange
  void gsb_sweep(const DevLevel& L, int dir, const DevMatrix::Sell& copy, const double* xin, double* xout, const double* b,
                 int blk0 = 0, int blk1 = -1) {
    Range rg("GSS3<bs=1>::SmoothRHS");
    const DevGSB& g = L.gsb;
    if (blk1 < 0 || blk1 > g.n_blocks) blk1 = g.n_blocks;
    if (blk1 <= blk0) return;
    if (xin == xout) throw Err("block-hybrid Gauss-Seidel sweeps are out of place");
    const bool fz = xin == nullptr;
    const bool lw = !fz && &copy == &g.full && g.has_fullLW && g.G > 1;
    GsbArgs a{g.rowid.p, g.slotcolor.p, L.dinv.p, b, g.n_colors, dir, lw ? g.flw_cptr.p : nullptr, lw ? g.flw_ccol.p : nullptr};
    const SellMat M = lw ? g.fullLW.view() : copy.view();
    const bool narrow = fz && &copy == &g.lowin && g.lowin_maxw > 0 && g.lowin_maxw <= 5 && !std::getenv("AMGX_GSB_NO_NARROW");
#define LAUNCH_GSB3(TT, GG, ZZ, WW) hipLaunchKernelGGL((gsb_sweep_kernel<TT, GG, ZZ, WW>), dim3(blk1 - blk0), dim3(TT), 0, stream, L.n, blk0, M, a, xin, xout)
    const bool mid = !fz && &copy == &g.full && g.G > 1 && g.full_maxw > 0 && g.full_maxw <= 11 && !std::getenv("AMGX_GSB_NO_MID");
#define LAUNCH_GSBL(TT, GG, WW) hipLaunchKernelGGL((gsb_sweep_kernel<TT, GG, false, WW, true>), dim3(blk1 - blk0), dim3(TT), 0, stream, L.n, blk0, M, a, xin, xout)
#define LAUNCH_GSB2(TT, GG) { if (narrow) LAUNCH_GSB3(TT, GG, true, 2); else if (fz) LAUNCH_GSB3(TT, GG, true, GSB_WP); \
                              else if (lw && mid && GG > 1) LAUNCH_GSBL(TT, (GG > 1 ? GG : 2), 5); else if (lw && GG > 1) LAUNCH_GSBL(TT, (GG > 1 ? GG : 2), GSB_WP); \
                              else if (mid && GG > 1) LAUNCH_GSB3(TT, (GG > 1 ? GG : 2), false, 5); else LAUNCH_GSB3(TT, GG, false, GSB_WP); }
#define LAUNCH_GSB(TT) switch (g.G) { case 1: LAUNCH_GSB2(TT, 1); break; case 2: LAUNCH_GSB2(TT, 2); break; case 4: LAUNCH_GSB2(TT, 4); break; \
                                      case 8: LAUNCH_GSB2(TT, 8); break; default: LAUNCH_GSB2(TT, 16); break; }
    if (g.TH == 256) LAUNCH_GSB(256) else if (g.TH == 512) LAUNCH_GSB(512) else LAUNCH_GSB(1024)
#undef LAUNCH_GSB
#undef LAUNCH_GSB2
#undef LAUNCH_GSBL
#undef LAUNCH_GSB3
    HIPCHK(hipGetLastError());
  }

  // one block-hybrid Gauss-Seidel sweep on a square-block level (bgsb_sweep_kernel): ONE launch; xin == nullptr: from x = 0
  // blk0 / blk1: the sweep blocks [blk0, blk1) only (rank-partitioned levels: boundary blocks before, interior blocks beside the exchange)
  void bgsb_sweep(const DevLevel& L, int dir, const double* xin, double* xout, const double* b, bool lower_only = false, int blk0 = 0, int blk1 = -1) {
    Range rg("GSS3<bs=" + std::to_string(L.bs) + ">::SmoothRHS");
    const DevBGSB& g = L.bgsb;
    const size_t lds = (size_t)2 * g.BB * L.bs * sizeof(double) + (size_t)g.BB * sizeof(int);
    // forward: colour phases over the couplings to lower colours, the upper ones stream with the sweep-start values; backward: reversed
    const BSellMat IN = dir == 0 ? g.in.bsell.view() : g.upin.bsell.view(), OTH = dir == 0 ? g.upin.bsell.view() : g.in.bsell.view();
#define LAUNCH_BGSB(BS_, MODE_, OFF_, LIST_, B0_, NB_, XIN_) hipLaunchKernelGGL((bgsb_sweep_kernel<BS_, MODE_>), dim3(NB_), dim3(BLOCK), lds, stream, g.BB, B0_, LIST_, \
                                               g.blk_ptr.p, g.blk_rows.p, OFF_, g.off_ptr.p, IN, OTH, g.in_ptr.p, g.in_row.p, g.n_colors, dir, L.dinv.p, b, XIN_, xout)
#define LAUNCH_BGSB_BS(MODE_, OFF_, LIST_, B0_, NB_, XIN_)                                       \
    switch (L.bs) {                                                                              \
      case 2: LAUNCH_BGSB(2, MODE_, OFF_, LIST_, B0_, NB_, XIN_); break;                         \
      case 3: LAUNCH_BGSB(3, MODE_, OFF_, LIST_, B0_, NB_, XIN_); break;                         \
      case 6: LAUNCH_BGSB(6, MODE_, OFF_, LIST_, B0_, NB_, XIN_); break;                         \
      default: throw Err("block-hybrid Gauss-Seidel: unsupported block size");                   \
    }
    if (g.bc) {
      // block-coloured form: one in-place launch per block colour; from zero (xin == nullptr) the first colour reads nothing
      // outside its blocks, the later ones the finished blocks of lower colours through `offlo` (lower_only: the split exists)
      if (blk0 != 0 || (blk1 >= 0 && blk1 != g.n_blocks)) throw Err("block-coloured Gauss-Seidel sweeps cover the whole level");
      if (xin != nullptr && xin != xout) throw Err("block-coloured Gauss-Seidel sweeps work in place");
      const bool fz = xin == nullptr;
      if (fz && dir != 0) throw Err("block-coloured Gauss-Seidel: the sweep from zero is a forward sweep");
      if (fz && !(lower_only && g.has_split)) throw Err("block-coloured Gauss-Seidel from zero needs the split images (zero x and sweep in place instead)");
      const BSellMat OFFA = g.off.bsell.view(), OFFL = g.offlo.bsell.view();
      for (int q = 0; q < g.n_bcolors; ++q) {
        const int c = dir ? g.n_bcolors - 1 - q : q;
        const int b0 = g.bc_ptr[c], nb = g.bc_ptr[c + 1] - b0;
        if (nb <= 0) continue;
        if (!fz) { LAUNCH_BGSB_BS(0, OFFA, g.blk_list.p, b0, nb, (const double*)xout); }
        else if (q == 0) { LAUNCH_BGSB_BS(1, OFFL, g.blk_list.p, b0, nb, (const double*)nullptr); }
        else { LAUNCH_BGSB_BS(2, OFFL, g.blk_list.p, b0, nb, (const double*)xout); }
      }
      HIPCHK(hipGetLastError());
      return;
    }
    if (blk1 < 0 || blk1 > g.n_blocks) blk1 = g.n_blocks;
    if (blk1 <= blk0) return;
    if (xin == xout) throw Err("block-hybrid Gauss-Seidel sweeps are out of place");
    (void)lower_only;
    const BSellMat OFF = g.off.bsell.view();
    const bool fz = xin == nullptr;
    if (fz) { LAUNCH_BGSB_BS(1, OFF, (const int32_t*)nullptr, blk0, blk1 - blk0, xin); }
    else { LAUNCH_BGSB_BS(0, OFF, (const int32_t*)nullptr, blk0, blk1 - blk0, xin); }
#undef LAUNCH_BGSB_BS
#undef LAUNCH_BGSB
    HIPCHK(hipGetLastError());
  }

  // one block Gauss-Seidel sweep: colours of the block graph ascending (forward) or descending (backward)
  void bgs_sweep(const DevLevel& L, int dir, double* x, const double* b, int cbeg = 0, int cend = -1) {
    const DevBGS& g = L.bgs;
    if (g.n_colors == 0 && L.n > 0 && g.block_ptr.n > 1) throw Err("block Gauss-Seidel requested but the level has no block colouring");
    if (cend < 0 || cend > g.n_colors) cend = g.n_colors;
    for (int q = cbeg; q < cend; ++q) {
      const int c = dir == 0 ? q : cend - 1 - (q - cbeg);
      const int b0 = g.color_ptr[c], b1 = g.color_ptr[c + 1];
      if (b1 == b0) continue;
      // (TH, G): one pass over the block's rows where possible (M * G <= TH), G lanes per row ~ row length / 6
      const double avg = L.A.n_rows ? (double)L.A.nnz / (double)L.A.n_rows : 0.0;
      const int G = avg > 30.0 ? (g.max_m * 16 <= 1024 ? 16 : 8) : 4;
      const int TH = g.max_m * G <= 256 ? 256 : (g.max_m * G <= 512 ? 512 : 1024);
#define LAUNCH_BGS2(BS, TT, GG) hipLaunchKernelGGL((bgs_block_kernel<BS, TT, GG>), dim3(b1 - b0), dim3(TT), 0, stream, b0, g.blocklist.p, g.block_ptr.p, g.block_rows.p, g.rowptr.p, g.col.p, g.val.p, g.dinv_ptr.p, g.dinv.p, b, x)
#define LAUNCH_BGS(BS) { if (G == 4 && TH == 256) LAUNCH_BGS2(BS, 256, 4); else if (G == 4 && TH == 512) LAUNCH_BGS2(BS, 512, 4); else if (G == 4) LAUNCH_BGS2(BS, 1024, 4); \
                         else if (G == 8 && TH <= 512) LAUNCH_BGS2(BS, 512, 8); else if (G == 8) LAUNCH_BGS2(BS, 1024, 8); else LAUNCH_BGS2(BS, 1024, 16); }
      switch (L.bs) {
        case 1: LAUNCH_BGS(1); break;
        case 2: LAUNCH_BGS(2); break;
        case 3: LAUNCH_BGS(3); break;
        case 6: LAUNCH_BGS(6); break;
        default: throw Err("unsupported block size for block Gauss-Seidel");
      }
#undef LAUNCH_BGS2
#undef LAUNCH_BGS
      HIPCHK(hipGetLastError());
    }
  }

  void coarse_solve(const double* rhs, double* x) {
    Range rg("coarse inv");
    const DevLevel& L = lev.back();
    if (clev != AMGX_CLEV_INV || coarse_n == 0) { zero(x, L.len()); return; }   // amg_matrix.cpp:242-246
    const int grid = (int)((coarse_n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    if (coarse_ld != coarse_n) hipLaunchKernelGGL(dense_op_gemv_kernel, dim3(grid), dim3(BLOCK), 0, stream, (int)coarse_n, (int)coarse_ld, coarse_inv.p, rhs, x);
    else hipLaunchKernelGGL(dense_gemv_kernel, dim3(grid), dim3(BLOCK), 0, stream, (int)coarse_n, coarse_inv.p, rhs, x);
    HIPCHK(hipGetLastError());
  }

  void transfer_f2c(int l, const double* xf, double* xc) {                                               // dof_map.cpp:636-654
    Range rg("ProlMap::TransferF2C");
    const DevRestrict& R = lev[l].R;
    if (R.empty()) { mult(lev[l].PT, xf, xc); return; }
    hipLaunchKernelGGL(restrict_chunk_kernel, dim3(R.n_chunks), dim3(BLOCK), 0, stream, lev[l].n, R.chunk_slot.p, R.slot_ptr.p,
                       R.w.p, R.fi.p, xf, R.part.p, R.dest.p);
    hipLaunchKernelGGL(restrict_sum_kernel, dim3(grid_for((lev[l + 1].n + RSUM_R - 1) / RSUM_R * RSUM_G)), dim3(BLOCK), 0, stream, lev[l + 1].n, R.optr.p,
                       R.oidx.p, R.part.p, xc);
    HIPCHK(hipGetLastError());
  }
  void add_c2f(int l, double fac, double* xf, const double* xc) { Range rg("ProlMap::TransferC2F"); mult_add(lev[l].P, fac, xc, xf, xf); }   // dof_map.cpp:697-709

  // ------------------------------------------------------------------ smoothers (flag contract: base_smoother.hpp:68-112)
  void base_smooth(DevLevel& L, int dir, double* x, const double* b, double* res, bool res_updated, bool update_res, bool x_zero) {
    if (L.sm_type == AMGX_SM_JACOBI) {
      // RichardsonSmoother::Smooth, base_smoother.cpp:61-74 (SmoothBack identical)
      if (!res_updated && x_zero) diag_apply(L, b, x, true);
      else {
        if (!res_updated) residual(L.A, x, b, res);
        diag_apply(L, res, x, true);
      }
      if (update_res) residual(L.A, x, b, res);
    } else if (L.sm_type == AMGX_SM_BGS) {
      // BSmoother::Smooth / SmoothBack (block_gssmoother.cpp:434-498): like GSS3 the reference updates the residual by
      // row-transpose scatters when asked for it; here: gather (RHS) form + one residual SpMV, same x and res
      bgs_sweep(L, dir, x, b);
      if (update_res) residual(L.A, x, b, res);
    } else if (L.bgsb.on()) {
      if (L.bgsb.bc) bgsb_sweep(L, dir, x, x, b);           // block-coloured form: in place
      else {
        copy(L.tmp.p, x, L.ext_len());
        bgsb_sweep(L, dir, L.tmp.p, x, b);
      }
      if (update_res) residual(L.A, x, b, res);
    } else if (L.gsb.on()) {
      // block-hybrid sweep (out of place: the off-block values are those from the start of the sweep)
      copy(L.tmp.p, x, L.ext_len());
      gsb_sweep(L, dir, L.gsb.full, L.tmp.p, x, b);
      if (update_res) residual(L.A, x, b, res);
    } else {
      // GSS3::Smooth / SmoothBack (gssmoother.cpp:350-398).  The reference keeps the residual current with
      // row-transpose scatters (RES form); for symmetric A the gather (RHS) form followed by one residual
      // SpMV gives the same x and res, and it has no write conflicts on a GPU.
      gs_sweep(L, dir, x, b);
      if (update_res) residual(L.A, x, b, res);
    }
  }

  void smooth_symm(DevLevel& L, double* x, const double* b, double* res, bool ru, bool ur, bool xz) {
    base_smooth(L, 0, x, b, res, ru, ur, xz);
    base_smooth(L, 1, x, b, res, ur, ur, false);
  }

  // ProxySmoother (base_smoother.hpp:169-229), present iff sm_symm || sm_steps > 1 (amg_pc.cpp:1079-1082)
  void level_smooth(DevLevel& L, int dir, double* x, const double* b, double* res, bool ru, bool ur, bool xz) {
    const int k = std::max(1, L.sm_steps);
    if (!L.sm_symm && k == 1) { base_smooth(L, dir, x, b, res, ru, ur, xz); return; }
    if (L.sm_symm) {
      smooth_symm(L, x, b, res, ru, ur, xz);
      for (int j = 1; j < k; ++j) smooth_symm(L, x, b, res, ur, ur, false);
    } else {
      base_smooth(L, dir, x, b, res, ru, ur, xz);
      for (int j = 1; j < k; ++j) base_smooth(L, dir, x, b, res, ur, ur, false);
    }
  }

  bool plain(const DevLevel& L) const { return L.sm_steps <= 1 && !L.sm_symm; }
  // (a level without rows -- a rank that owns nothing -- is trivially folded: every kernel on it is a no-op)
  bool folded(const DevLevel& L) const { return plain(L) && L.sm_type == AMGX_SM_JACOBI && (L.n == 0 || (!L.Q.empty() && (L.bs > 1 || !L.Apre.empty()))); }

  // pre-smoothing step of the cycles: x = 0; r = b; Smooth(x, b, r, 1, 1, 1)   (amg_matrix.cpp:193-206)
  // fold (only with folded(L)): x receives z = x + omega*Dinv*r, to be completed by post_smooth(..., fold = true)
  void pre_smooth(DevLevel& L, double* x, const double* b, double* r, bool fold = false, const Span sp = Span()) {
    if (plain(L) && L.sm_type == AMGX_SM_JACOBI && !L.Apre.empty()) {
      // one pass: r = b - A' b, x = omega * Dinv * b   (A' = A * omega*Dinv built at create time)
      spmv_ep<EP_PRE>(L.Apre, b, r, EpArgs{b, nullptr, L.dinv.p, L.omega, x, ep_nt | (fold ? EPF_FOLD : 0)}, sp);
    } else if (sp.part == PART_INT) {
      return;                                  // the other forms are not split: they run completely in the boundary part
    } else if (plain(L) && L.sm_type == AMGX_SM_JACOBI && L.ncols > L.n && !fold) {
      // rank-partitioned level without a pre-smoothing image (block levels): x = omega * Dinv * b is needed on the ghost
      // rows too (their b and dinv entries came with the exchange / the setup), so it is formed in the gathered buffer
      diag_apply(L, b, L.tmp.p, false, L.ncols);
      residual(L.A, L.tmp.p, b, r);
      copy(x, L.tmp.p, L.len());
    } else if (plain(L) && L.sm_type == AMGX_SM_JACOBI) {
      if (fold && L.bs > 1) {
        // x_pre = omega * Dinv * b into tmp; then ONE pass: r = b - A x_pre and z = x_pre + omega * Dinv * r
        diag_apply(L, b, L.tmp.p, false);
        spmv_ep<EP_JAC>(L.A, L.tmp.p, x, EpArgs{b, L.tmp.p, L.dinv.p, L.omega, r, ep_nt});
      } else {
        diag_apply(L, b, x, false);          // x = omega * Dinv * b      (x was zero, res == b)
        residual(L.A, x, b, r);              // r = b - A x
        if (fold) diag_apply(L, r, x, true); // z = x + omega * Dinv * r  (folded post-smoothing, see fold_prolongation)
      }
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.bgsb.on()) {
      // forward block-hybrid sweep from x = 0 (nothing outside the workgroup's rows is read), then the residual; with the
      // split copies A is read once in total: r = rest * x (see DevBGSB)
      if (L.bgsb.has_split) {
        bgsb_sweep(L, 0, nullptr, x, b, true);
        mult(L.bgsb.rest, x, r);
      } else if (L.bgsb.bc) {
        zero(x, L.len());                      // (pseudo-inverted diagonal blocks: no split images)
        bgsb_sweep(L, 0, x, x, b);
        residual(L.A, x, b, r);
      } else {
        bgsb_sweep(L, 0, nullptr, x, b);
        residual(L.A, x, b, r);
      }
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.gsb.on()) {
      // forward block-hybrid sweep from x = 0: inside a block only couplings to lower colours contribute (all other
      // values are still 0); afterwards (b - L_in x)_k = x_k / dinv_k on every swept row, hence
      // r = b - A x = (1/dinv - a_kk) .* x - (A - L_in - D) x = c .* x - rest x
      if (L.gsb.has_split) {
        gsb_sweep(L, 0, L.gsb.lowin, nullptr, x, b);
        spmv_ep<EP_CRES>(L.gsb.rest, x, r, EpArgs{x, nullptr, L.gsb.cvec.p, 0.0, nullptr, ep_nt & EPF_HOIST});
      } else {
        gsb_sweep(L, 0, L.gsb.full, nullptr, x, b);
        residual(L.A, x, b, r);
      }
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.gs.has_split && L.bs == 1) {
      // forward sweep from x = 0: only couplings to lower colours contribute; afterwards b - L x - D x = 0 on every
      // swept row, hence r = -U x (r on non-free rows is not needed: their prolongation rows are empty)
      zero(x, L.len());
      zero(r, L.len());
      gs_sweep(L, 0, x, b, true);
      const DevGS& g = L.gs;
      const int grid = (g.n_slices_total + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
#define LAUNCH_UR(GG) hipLaunchKernelGGL((gs_upper_residual_kernel<GG>), dim3(grid), dim3(BLOCK), 0, stream, g.n_slices_total, g.upper.view(), g.rowid.p, x, r)
      switch (g.lanes) {
        case 1: LAUNCH_UR(1); break;
        case 2: LAUNCH_UR(2); break;
        case 4: LAUNCH_UR(4); break;
        case 8: LAUNCH_UR(8); break;
        default: LAUNCH_UR(16); break;
      }
#undef LAUNCH_UR
      HIPCHK(hipGetLastError());
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.gs.bsplit && L.bs > 1) {
      // block levels, same identity: forward sweep over the lower-colour couplings, then r_B = -(U x)_B in one launch
      zero(x, L.len());
      zero(r, L.len());
      gs_sweep(L, 0, x, b, true);
      const DevGS& g = L.gs;
      const int ns = g.bupper.n_slices;
      const int grid = (ns + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
      const BSellMat UM = g.bupper.bsell.view();
      if (L.bs == 6) hipLaunchKernelGGL((bgs_bsell_upper_residual_kernel<6>), dim3(grid), dim3(BLOCK), 0, stream, ns, UM, g.rowid.p, x, r);
      else if (L.bs == 3) hipLaunchKernelGGL((bgs_bsell_upper_residual_kernel<3>), dim3(grid), dim3(BLOCK), 0, stream, ns, UM, g.rowid.p, x, r);
      else hipLaunchKernelGGL((bgs_bsell_upper_residual_kernel<2>), dim3(grid), dim3(BLOCK), 0, stream, ns, UM, g.rowid.p, x, r);
      HIPCHK(hipGetLastError());
    } else if (plain(L) && L.sm_type == AMGX_SM_GS) {
      zero(x, L.len());
      gs_sweep(L, 0, x, b);
      residual(L.A, x, b, r);
    } else {
      zero(x, L.len());
      copy(r, b, L.len());
      level_smooth(L, 0, x, b, r, true, true, true);
    }
  }

  // pre-smoothing followed by the restriction of the residual (amg_matrix.cpp:193-212), fused where possible
  // sp: interior part = the row-parallel pass over the interior rows only; boundary part = the remaining rows plus
  // everything that needs all rows (partial-sum reduction / P^T gather)
  void pre_smooth_restrict(int l, double* x, const double* b, double* r, double* b_coarse, bool fold = false, const Span sp = Span()) {
    DevLevel& L = lev[l];
    if (fold && !folded(L)) throw Err("pre_smooth_restrict: level has no folded prolongation");
    const int epf = ep_nt | (fold ? EPF_FOLD : 0);
    if (plain(L) && L.sm_type == AMGX_SM_JACOBI && !L.RF.empty() && !L.ApreLW.empty()) {
      // long-row level: the local-window image (gathers from LDS), chunks of 512 / lanes rows
      const DevRestrict& R = L.RF;
      const int nch = (L.ApreLW.n_slices + (512 / WAVE) - 1) / (512 / WAVE);
      if (nch != R.n_chunks) throw Err("fused restriction (local-window image): chunk / slice mismatch");
      int64_t ca, cb;
      const int LG = L.ApreLW.lanes;
      unit_range(sp, 512 / LG, nch, ca, cb);
      const int grid = (int)(cb - ca), c0 = (int)ca;
      const bool probe = probe_level == l && probe_kind == 8 && probe_e0 && sp.part != PART_BND;
      if (probe) HIPCHK(hipEventRecord(probe_e0, stream));
      if (grid > 0) {
#define LAUNCH_LW(EPT_, G_) hipLaunchKernelGGL((sell_lw_pre_restrict_kernel<EPT_, G_, 0>), dim3(grid), dim3(512), 0, stream, L.ApreLW.n_rows, c0, L.ApreLW.n_slices, \
                           L.ApreLW.sell.view(), L.lw_cptr.p, L.lw_ccol.p, b, L.dinv.p, L.omega, epf, x, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
        if (LG == 2) { if (R.ept == 2) LAUNCH_LW(2, 2); else LAUNCH_LW(4, 2); }
        else { if (R.ept == 2) LAUNCH_LW(2, 4); else LAUNCH_LW(4, 4); }
#undef LAUNCH_LW
      }
      if (probe) HIPCHK(hipEventRecord(probe_e1, stream));
      if (!skip_rsum && sp.part != PART_INT)
        hipLaunchKernelGGL(restrict_sum_kernel, dim3(grid_for((lev[l + 1].n + RSUM_R - 1) / RSUM_R * RSUM_G)), dim3(BLOCK), 0, stream, lev[l + 1].n, R.optr.p,
                           R.oidx.p, R.part.p, b_coarse);
      HIPCHK(hipGetLastError());
      return;
    }
    if (plain(L) && L.sm_type == AMGX_SM_JACOBI && !L.RF.empty()) {
      const DevRestrict& R = L.RF;
      const int FB = L.fused_block;
      const int G = L.Apre.lanes;
      const int nch = R.slice_list.n ? R.n_chunks : (L.Apre.n_slices + (FB / WAVE) - 1) / (FB / WAVE);
      if (nch != R.n_chunks) throw Err("fused restriction: chunk / slice mismatch");
      if (R.slice_list.n && sp.part != PART_ALL) throw Err("fused restriction: compact chunks are not split into interior / boundary parts");
      int64_t ca, cb;
      unit_range(sp, FB / G, nch, ca, cb);
      const int grid = (int)(cb - ca), c0 = (int)ca;
      // (rank-partitioned level: the events go around the interior launch, which runs beside the halo exchange)
      const bool probe = probe_level == l && probe_kind == 8 && probe_e0 && sp.part != PART_BND;
      if (probe) HIPCHK(hipEventRecord(probe_e0, stream));
      if (grid > 0) {
#define LAUNCH_PRF(FB_, EPT_) hipLaunchKernelGGL((sell_pre_restrict_kernel<FB_, 0, EPT_>), dim3(grid), dim3(FB_), 0, stream, L.Apre.n_rows, c0, L.Apre.n_slices, \
                             L.Apre.sell.view(), b, L.dinv.p, L.omega, epf, x, (double*)nullptr, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p, \
                             (const int32_t*)R.slice_list.p)
#define LAUNCH_PRG(G_) hipLaunchKernelGGL((sell_pre_restrict_kernel<512, 0, 4, G_>), dim3(grid), dim3(512), 0, stream, L.Apre.n_rows, c0, L.Apre.n_slices, \
                             L.Apre.sell.view(), b, L.dinv.p, L.omega, epf, x, (double*)nullptr, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
        if (L.Apre.sell.win) {
          if (FB != SELL_WIN || G != 1) throw Err("fused restriction on a windowed image: unexpected chunk shape");
#define LAUNCH_WPR(EPT_) hipLaunchKernelGGL((sell_win_pre_restrict_kernel<SELL_WIN, EPT_>), dim3(grid), dim3(SELL_WIN), 0, stream, L.Apre.n_rows, c0, L.Apre.sell.view(), \
                             L.Apre.sell.rowloc.p, b, L.dinv.p, L.omega, epf, x, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
          if (R.ept == 4) LAUNCH_WPR(4); else LAUNCH_WPR(6);
#undef LAUNCH_WPR
        }
        else if (G > 1) {
          if (FB != 512 || R.ept != 4) throw Err("fused restriction with several lanes per row: unexpected chunk shape");
          if (G == 2) LAUNCH_PRG(2); else if (G == 4) LAUNCH_PRG(4); else LAUNCH_PRG(8);
        }
        else if (FB == 256) { if (R.ept == 4) LAUNCH_PRF(256, 4); else LAUNCH_PRF(256, 6); }
        else if (FB == 512) { if (R.ept == 4) LAUNCH_PRF(512, 4); else LAUNCH_PRF(512, 6); }
        else { if (R.ept == 4) LAUNCH_PRF(1024, 4); else LAUNCH_PRF(1024, 6); }
#undef LAUNCH_PRG
#undef LAUNCH_PRF
      }
      if (probe) HIPCHK(hipEventRecord(probe_e1, stream));
      if (!skip_rsum && sp.part != PART_INT)
        hipLaunchKernelGGL(restrict_sum_kernel, dim3(grid_for((lev[l + 1].n + RSUM_R - 1) / RSUM_R * RSUM_G)), dim3(BLOCK), 0, stream, lev[l + 1].n, R.optr.p,
                           R.oidx.p, R.part.p, b_coarse);
      HIPCHK(hipGetLastError());
      return;
    }
    if (plain(L) && L.sm_type == AMGX_SM_GS && L.gsb.on() && L.gsb.has_split && !L.RG.empty() && sp.part == PART_ALL) {
      // sweep from zero, then residual of the untouched part + chunk-local restriction in one pass (r stays in LDS)
      gsb_sweep(L, 0, L.gsb.lowin, nullptr, x, b);
      gsb_residual_restrict(l, x, r, b_coarse);
      return;
    }
    pre_smooth(L, x, b, r, fold, sp);
    if (sp.part != PART_INT) transfer_f2c(l, r, b_coarse);
  }

  // after a block-hybrid sweep from zero: r = c .* x - A_rest x (x may carry ghost entries) and b_coarse = P^T r
  void gsb_residual_restrict(int l, const double* x, double* r, double* b_coarse) {
    DevLevel& L = lev[l];
    if (!L.gsb.has_split) throw Err("gsb_residual_restrict: the level has no lower / rest split");
    if (!L.gsb.restLW.empty() && !L.RG.empty()) {
      // long-row level: local-window image of the rest part (the swept x is staged in LDS per chunk)
      const DevRestrict& R = L.RG;
      const DevMatrix& M = L.gsb.restLW;
      const int nch = (M.n_slices + (512 / WAVE) - 1) / (512 / WAVE);
      if (nch != R.n_chunks) throw Err("fused Gauss-Seidel residual (local-window image): chunk / slice mismatch");
#define LAUNCH_LWC(EPT_, G_) hipLaunchKernelGGL((sell_lw_pre_restrict_kernel<EPT_, G_, 1>), dim3(nch), dim3(512), 0, stream, M.n_rows, 0, M.n_slices, M.sell.view(), \
                             L.gsb.lw_cptr.p, L.gsb.lw_ccol.p, (const double*)x, (const double*)L.gsb.cvec.p, 0.0, 0, (double*)nullptr, \
                             R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
      if (M.lanes == 2) { if (R.ept == 2) LAUNCH_LWC(2, 2); else LAUNCH_LWC(4, 2); }
      else { if (R.ept == 2) LAUNCH_LWC(2, 4); else LAUNCH_LWC(4, 4); }
#undef LAUNCH_LWC
      hipLaunchKernelGGL(restrict_sum_kernel, dim3(grid_for((lev[l + 1].n + RSUM_R - 1) / RSUM_R * RSUM_G)), dim3(BLOCK), 0, stream, lev[l + 1].n, R.optr.p,
                         R.oidx.p, R.part.p, b_coarse);
      HIPCHK(hipGetLastError());
      return;
    }
    if (L.RG.empty()) {
      spmv_ep<EP_CRES>(L.gsb.rest, x, r, EpArgs{x, nullptr, L.gsb.cvec.p, 0.0, nullptr, ep_nt & EPF_HOIST});
      transfer_f2c(l, r, b_coarse);
      return;
    }
    {
      const DevRestrict& R = L.RG;
      const DevMatrix& M = L.gsb.rest;
      const int nch = (M.n_slices + (512 / WAVE) - 1) / (512 / WAVE);
      if (nch != R.n_chunks) throw Err("fused Gauss-Seidel residual: chunk / slice mismatch");
#define LAUNCH_WCR(EPT_) hipLaunchKernelGGL((sell_win_cres_restrict_kernel<SELL_WIN, EPT_>), dim3(nch), dim3(SELL_WIN), 0, stream, M.n_rows, M.sell.view(), M.sell.rowloc.p, \
                           (const double*)x, (const double*)L.gsb.cvec.p, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
#define LAUNCH_PCR(EPT_) hipLaunchKernelGGL((sell_pre_restrict_kernel<512, 1, EPT_>), dim3(nch), dim3(512), 0, stream, M.n_rows, 0, M.n_slices, M.sell.view(), (const double*)x, \
                         (const double*)L.gsb.cvec.p, 0.0, 0, (double*)nullptr, (double*)nullptr, R.chunk_slot.p, R.slot_ptr.p, R.w.p, R.fi.p, R.part.p, R.dest.p)
      if (M.sell.win) { if (R.ept == 4) LAUNCH_WCR(4); else LAUNCH_WCR(6); }
      else { if (R.ept == 4) LAUNCH_PCR(4); else LAUNCH_PCR(6); }
#undef LAUNCH_WCR
#undef LAUNCH_PCR
      hipLaunchKernelGGL(restrict_sum_kernel, dim3(grid_for((lev[l + 1].n + RSUM_R - 1) / RSUM_R * RSUM_G)), dim3(BLOCK), 0, stream, lev[l + 1].n, R.optr.p,
                         R.oidx.p, R.part.p, b_coarse);
      HIPCHK(hipGetLastError());
    }
  }

  // coarse-grid correction + post-smoothing: x += P x_c; SmoothBack(x, b, r, 0, 0, 0)   (amg_matrix.cpp:263-302)
  // fold: x holds z of the folded pre-smoothing pass; x' = z + Q x_c (see fold_prolongation)
  void post_smooth(int l, double* x, const double* b, double* r, const double* xc, bool fold = false, const Span sp = Span()) {
    DevLevel& L = lev[l];
    if (fold && !L.QLW.empty()) {
      // local-window form of Q: the coarse values a window needs are staged in LDS (sell_lw_win_spmv_kernel); interior / boundary
      // window ranges on rank-partitioned levels (the list of an interior window holds owned coarse columns only)
      const int64_t nw = (L.QLW.n_rows + SELL_WIN - 1) / SELL_WIN;
      int64_t wa, wb;
      unit_range(sp, SELL_WIN, nw, wa, wb);
      if (wb > wa)
        hipLaunchKernelGGL((sell_lw_win_spmv_kernel<SELL_WIN, EP_AXPY>), dim3((int)(wb - wa)), dim3(SELL_WIN), 0, stream, L.QLW.n_rows, (int)wa, L.QLW.sell.view(),
                           L.QLW.sell.rowloc.p, L.qlw_cptr.p, L.qlw_ccol.p, xc, x, EpArgs{nullptr, x, nullptr, 1.0, nullptr, ep_nt & EPF_HOIST});
      HIPCHK(hipGetLastError());
    } else if (fold) {
      mult_add(L.Q, 1.0, xc, x, x, sp);
    } else if (sp.part == PART_INT) {
      return;                                  // literal forms are driven stage by stage (Dist), not through this function
    } else if (plain(L) && L.sm_type == AMGX_SM_JACOBI) {
      mult_add(L.P, 1.0, xc, x, L.tmp.p);  // tmp = x + P x_c
      jacobi_fused(L, L.tmp.p, b, x);      // x = tmp + omega * Dinv * (b - A tmp); res is not needed afterwards
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.bgsb.on()) {
      const bool probe = probe_level == l && probe_kind == 9 && probe_e0;
      if (L.bgsb.bc) {
        mult_add(L.P, 1.0, xc, x, x);      // x += P x_c
        if (probe) HIPCHK(hipEventRecord(probe_e0, stream));
        bgsb_sweep(L, 1, x, x, b);         // backward block-coloured sweep, in place (one launch per block colour)
        if (probe) HIPCHK(hipEventRecord(probe_e1, stream));
      } else {
      mult_add(L.P, 1.0, xc, x, L.tmp.p);  // tmp = x + P x_c
      if (probe) HIPCHK(hipEventRecord(probe_e0, stream));
      bgsb_sweep(L, 1, L.tmp.p, x, b);     // backward block-hybrid sweep, tmp -> x
      if (probe) HIPCHK(hipEventRecord(probe_e1, stream));
      }
    } else if (plain(L) && L.sm_type == AMGX_SM_GS && L.gsb.on() && L.n == L.ncols) {
      mult_add(L.P, 1.0, xc, x, L.tmp.p);  // tmp = x + P x_c
      const bool probe = probe_level == l && probe_kind == 9 && probe_e0;
      if (probe) HIPCHK(hipEventRecord(probe_e0, stream));
      gsb_sweep(L, 1, L.gsb.full, L.tmp.p, x, b);   // backward block-hybrid sweep, tmp -> x
      if (probe) HIPCHK(hipEventRecord(probe_e1, stream));
    } else {
      add_c2f(l, 1.0, x, xc);
      level_smooth(L, 1, x, b, r, false, false, false);
    }
  }

  // ------------------------------------------------------------------ cycles
  // l0: first level of the (sub-)cycle; x, b are the vectors of that level (l0 > 0: build_dense_tail)
  void cycle_v(double* x, const double* b, int l0 = 0) {
    const int L = n_levels();
    if (dense_level >= 0 && dense_level == l0) {           // the whole (sub-)cycle is the dense operator: x = B b
      Range rg("rest");
      hipLaunchKernelGGL(dense_op_gemv_kernel, dim3((dense_n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK), 0, stream,
                         dense_n, dense_ld, dense_op.p, b, x);
      HIPCHK(hipGetLastError());
      return;
    }
    if (l0 == L - 1) { coarse_solve(b, x); return; }
    const bool dense = dense_level > l0;
    const bool tail = !dense && tail_level > 0 && l0 == 0;
    const int T = dense ? dense_level : (tail ? tail_level : L - 1);     // levels >= T: one dense GEMV / tail_kernel
    for (int l = l0; l < T; ++l) {
      Range rg(level_range_name(l));
      double* xl = l == l0 ? x : lev[l].x.p;
      const double* bl = l == l0 ? b : lev[l].rhs.p;
      pre_smooth_restrict(l, xl, bl, lev[l].res.p, lev[l + 1].rhs.p, folded(lev[l]));
    }
    if (dense) {
      Range rg("rest");                                    // levels >= dense_level incl. "coarse inv": x_T = B b_T
      hipLaunchKernelGGL(dense_op_gemv_kernel, dim3((dense_n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(BLOCK), 0, stream,
                         dense_n, dense_ld, dense_op.p, lev[T].rhs.p, lev[T].x.p);
      HIPCHK(hipGetLastError());
    } else if (tail) {
      Range rg("rest");                                    // the coarse tail incl. "coarse inv" in one workgroup
      hipLaunchKernelGGL(tail_kernel, dim3(1), dim3(TAIL_BLOCK), 0, stream, tail_ops, tail_prog.p);
      HIPCHK(hipGetLastError());
    } else coarse_solve(lev[L - 1].rhs.p, lev[L - 1].x.p);
    for (int l = T - 1; l >= l0; --l) {
      Range rg(level_range_name(l));
      double* xl = l == l0 ? x : lev[l].x.p;
      const double* bl = l == l0 ? b : lev[l].rhs.p;
      post_smooth(l, xl, bl, lev[l].res.p, lev[l + 1].x.p, folded(lev[l]));
    }
  }

  // plain W-cycle (the reference additionally runs and discards a V-like pass at level 0, amg_matrix.cpp:46-64)
  void w_rec(int l, double* x0, const double* b0) {
    const int L = n_levels();
    if (l + 1 < L) {
      DevLevel& V = lev[l];
      double* xl = l == 0 ? x0 : V.x.p;
      const double* bl = l == 0 ? b0 : V.rhs.p;
      double* rl = V.res.p;
      pre_smooth_restrict(l, xl, bl, rl, lev[l + 1].rhs.p);
      w_rec(l + 1, x0, b0);
      add_c2f(l, 1.0, xl, lev[l + 1].x.p);
      level_smooth(V, 1, xl, bl, rl, false, true, false);
      level_smooth(V, 0, xl, bl, rl, true, true, false);
      transfer_f2c(l, rl, lev[l + 1].rhs.p);
      w_rec(l + 1, x0, b0);
      post_smooth(l, xl, bl, rl, lev[l + 1].x.p);
    } else {
      if (L == 1) coarse_solve(b0, x0);
      else coarse_solve(lev[L - 1].rhs.p, lev[L - 1].x.p);
    }
  }

  // AMGMatrix::SmoothVFromLevel, amg_matrix.cpp:310-374
  void smooth_v_from_level(int start, double* x, const double* b, double* res, bool ru, bool ur, bool xz) {
    const int L = n_levels();
    level_smooth(lev[start], 0, x, b, res, ru, true, xz);
    transfer_f2c(start, res, lev[start + 1].rhs.p);
    if (start + 2 < L)
      for (int l = start + 1; l + 1 < L; ++l) {
        pre_smooth_restrict(l, lev[l].x.p, lev[l].rhs.p, lev[l].res.p, lev[l + 1].rhs.p);
      }
    coarse_solve(lev[L - 1].rhs.p, lev[L - 1].x.p);
    if (start + 2 < L)
      for (int l = L - 2; l > start; --l) post_smooth(l, lev[l].x.p, lev[l].rhs.p, lev[l].res.p, lev[l + 1].x.p);
    add_c2f(start, 1.0, x, lev[start + 1].x.p);
    level_smooth(lev[start], 1, x, b, res, false, ur, false);
  }

  // AMGMatrix::SmoothBS, amg_matrix.cpp:110-157
  void cycle_bs(double* x, const double* b) {
    const int L = n_levels();
    if (L == 1) { coarse_solve(b, x); return; }
    for (int l = 0; l + 1 < L; ++l) {
      double* xl = l == 0 ? x : lev[l].x.p;
      const double* bl = l == 0 ? b : lev[l].rhs.p;
      double* rl = lev[l].res.p;
      zero(xl, lev[l].len());
      copy(rl, bl, lev[l].len());
      smooth_v_from_level(l, xl, bl, rl, true, true, true);
      transfer_f2c(l, rl, lev[l + 1].rhs.p);
    }
    coarse_solve(lev[L - 1].rhs.p, lev[L - 1].x.p);
    for (int l = L - 2; l >= 0; --l) {
      double* xl = l == 0 ? x : lev[l].x.p;
      const double* bl = l == 0 ? b : lev[l].rhs.p;
      add_c2f(l, 1.0, xl, lev[l + 1].x.p);
      smooth_v_from_level(l, xl, bl, lev[l].res.p, false, false, false);
    }
  }

  void do_cycle(double* x, const double* b) {
    Range rg("AMGMatrix::Mult");
    if (cycle == AMGX_CYCLE_W) w_rec(0, x, b);
    else if (cycle == AMGX_CYCLE_BS) cycle_bs(x, b);
    else cycle_v(x, b);
  }

  // one application, optionally through a captured graph keyed by the vector addresses
  void run_cycle(double* x, const double* b, bool graph_ok) {
    // the legacy default stream cannot be captured: launch directly there
    if (!(use_graph && graph_ok) || stream == nullptr) { do_cycle(x, b); return; }
    GraphKey key{b, x, 0};
    auto it = graphs.find(key);
    if (it == graphs.end()) {
      hipGraph_t g = nullptr;
      HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      try { do_cycle(x, b); }
      catch (...) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(stream, &dead); if (dead) (void)hipGraphDestroy(dead); throw; }
      HIPCHK(hipStreamEndCapture(stream, &g));
      hipGraphExec_t ge = nullptr;
      hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) throw Err(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e));
      if (graphs.size() >= 16 && !graph_age.empty()) {       // the oldest capture goes, the hot ones stay
        auto old = graphs.find(graph_age.front());
        graph_age.erase(graph_age.begin());
        if (old != graphs.end()) { (void)hipGraphExecDestroy(old->second); graphs.erase(old); }
      }
      it = graphs.emplace(key, ge).first;
      graph_age.push_back(key);
    }
    HIPCHK(hipGraphLaunch(it->second, stream));
  }

  void drop_graphs() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    graph_age.clear();
  }
};

// ---------------------------------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------------------------------

static void build_gs(const amgx_level_desc& d, DevLevel& L) {
  const int64_t n = d.A.n_rows;
  if (!d.color || d.n_colors <= 0) { if (n > 0) throw Err("AMGX_SM_GS needs a row colouring (color / n_colors)"); return; }
  const int nc = d.n_colors;
  // validate: rows of equal colour must not be coupled (otherwise the parallel sweep would race)
  for (int64_t i = 0; i < n; ++i) {
    const int ci = d.color[i];
    if (ci >= nc) throw Err("colour index out of range");
    if (ci < 0) continue;
    for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
      const int64_t j = d.A.col[k];
      if (j >= n) continue;          // ghost column of a rank-partitioned level: frozen during the sweep (hybrid GS)
      if (j != i && d.color[j] == ci) throw Err("invalid colouring: two coupled rows share a colour");
    }
  }
  std::vector<int64_t> cnt(nc + 1, 0);
  for (int64_t i = 0; i < n; ++i) if (d.color[i] >= 0) cnt[d.color[i] + 1]++;
  DevGS& g = L.gs;
  g.n_colors = nc;
  if (d.A.br == 1) {
    // lanes per row: one thread per row only while a colour still has >= 2^17 rows; below that G grows with the row
    // length so that a row is consumed in ~2-4 steps
    const int64_t nnz = d.A.rowptr[n];
    const double avg = n ? (double)nnz / (double)n : 0.0;
    int G = 1;
    if (n / std::max(1, nc) < ((int64_t)1 << 17)) while (G < 16 && avg > 4.0 * G) G <<= 1;
    g.lanes = G;
    const int R = WAVE / G;
    // colour-major row list, each colour padded to a multiple of R rows (= whole slices)
    std::vector<int64_t> cstart(nc + 1, 0);
    for (int c = 0; c < nc; ++c) cstart[c + 1] = cstart[c] + ((cnt[c + 1] + R - 1) / R) * R;
    std::vector<int32_t> rows(cstart[nc], -1);
    std::vector<int64_t> pos(cstart.begin(), cstart.end() - 1);
    for (int64_t i = 0; i < n; ++i) if (d.color[i] >= 0) rows[pos[d.color[i]]++] = (int32_t)i;
    g.color_slice_ptr.resize(nc + 1);
    for (int c = 0; c <= nc; ++c) g.color_slice_ptr[c] = (int)(cstart[c] / R);
    HostSell S;
    // (absolute 16-bit column bases, not row-relative: the row product then does not depend on the rowid load)
    const bool gs_rowrel = std::getenv("AMGX_GS_ROWREL") != nullptr;
    build_sell(d.A, rows.data(), (int64_t)rows.size(), gs_rowrel, G, S);
    upload_sell(S, g.sell);
    g.rowid.upload(rows);
    g.n_slices_total = (int)(S.slice_ptr.size() - 1);
    // the split form relies on x_k = (b - L x)_k / a_kk, i.e. on dinv being the plain inverse diagonal
    bool plain_diag = d.dinv != nullptr && d.A.n_rows == d.A.n_cols;
    for (int64_t i = 0; i < n && plain_diag; ++i) {
      if (d.color[i] < 0) continue;
      double aii = 0.0;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) if (d.A.col[k] == i) aii = d.A.val[k];
      if (!(std::fabs(d.dinv[i] * aii - 1.0) < 1e-12)) plain_diag = false;
    }
    if (plain_diag) {
      // lower / upper parts w.r.t. the colour order; couplings to non-free columns are dropped (x is 0 there)
      for (int part = 0; part < 2; ++part) {
        std::vector<int64_t> rp(n + 1, 0);
        std::vector<int32_t> cc;
        std::vector<double> vv;
        cc.reserve(nnz / 2 + 16); vv.reserve(nnz / 2 + 16);
        for (int64_t i = 0; i < n; ++i) {
          const int ci = d.color[i];
          if (ci >= 0)
            for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
              const int cj = d.color[d.A.col[k]];
              if (cj < 0) continue;
              if ((part == 0 && cj < ci) || (part == 1 && cj > ci)) { cc.push_back(d.A.col[k]); vv.push_back(d.A.val[k]); }
            }
          rp[i + 1] = (int64_t)cc.size();
        }
        amgx_matrix F = d.A;
        F.rowptr = rp.data(); F.col = cc.data(); F.val = vv.data();
        HostSell SP;
        build_sell(F, rows.data(), (int64_t)rows.size(), gs_rowrel, G, SP);
        if ((int)(SP.slice_ptr.size() - 1) != g.n_slices_total) throw Err("build_gs: split copy has a different slice count");
        upload_sell(SP, part == 0 ? g.lower : g.upper);
      }
      g.has_split = true;
    }
  } else {
    g.color_row_ptr.assign(nc + 1, 0);
    for (int c = 0; c < nc; ++c) g.color_row_ptr[c + 1] = g.color_row_ptr[c] + (int)cnt[c + 1];
    std::vector<int32_t> rows(g.color_row_ptr[nc]);
    std::vector<int> pos(g.color_row_ptr.begin(), g.color_row_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) if (d.color[i] >= 0) rows[pos[d.color[i]]++] = (int32_t)i;
    g.rowlist.upload(rows);
    // colour-major BSELL copy: every colour padded to whole slices of RB block rows (big levels: one more copy of A,
    // 2x the sweep speed; small levels keep the CSR row-list kernel: their colours are shorter than a slice)
    const int bs = d.A.br;
    int64_t min_rows = 4096;
    if (const char* e = std::getenv("AMGX_BGS_BSELL_MIN")) min_rows = std::atoll(e);      // test hook
    if ((bs == 2 || bs == 3 || bs == 6) && n / std::max(1, nc) >= min_rows && !std::getenv("AMGX_NO_BGS_BSELL")) {
      const int RB = WAVE / bs;
      std::vector<int64_t> cstart(nc + 1, 0);
      for (int c = 0; c < nc; ++c) cstart[c + 1] = cstart[c] + ((cnt[c + 1] + RB - 1) / RB) * RB;
      std::vector<int32_t> prow(cstart[nc], -1);
      std::vector<int64_t> p2(cstart.begin(), cstart.end() - 1);
      for (int64_t i = 0; i < n; ++i) if (d.color[i] >= 0) prow[p2[d.color[i]]++] = (int32_t)i;
      if (build_bsell(d.A, g.bcopy, 1.6, prow.data(), (int64_t)prow.size())) {
        g.color_slice_ptr.resize(nc + 1);
        for (int c = 0; c <= nc; ++c) g.color_slice_ptr[c] = (int)(cstart[c] / RB);
        g.rowid.upload(prow);
        g.bsell_ok = true;
        // split copies (see the scalar case): need x_B = Dinv_B (b - L x)_B to imply (b - L x - D x)_B = 0, i.e. Dinv_B = A_BB^-1
        bool plain_diag = d.dinv != nullptr && d.A.n_rows == d.A.n_cols && !std::getenv("AMGX_NO_BGS_SPLIT");
        const int bb = bs * bs;
        for (int64_t i = 0; i < n && plain_diag; ++i) {
          if (d.color[i] < 0) continue;
          const double* aii = nullptr;
          for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) if (d.A.col[k] == i) aii = d.A.val + k * bb;
          if (!aii) { plain_diag = false; break; }
          const double* di = d.dinv + i * bb;
          for (int r = 0; r < bs && plain_diag; ++r)
            for (int c = 0; c < bs; ++c) {
              double v = 0.0;
              for (int q = 0; q < bs; ++q) v += di[r * bs + q] * aii[q * bs + c];
              if (!(std::fabs(v - (r == c ? 1.0 : 0.0)) < 1e-10)) { plain_diag = false; break; }
            }
        }
        if (plain_diag) {
          bool ok = true;
          for (int part = 0; part < 2 && ok; ++part) {
            std::vector<int64_t> rp(n + 1, 0);
            std::vector<int32_t> cc;
            std::vector<double> vv;
            for (int64_t i = 0; i < n; ++i) {
              const int ci = d.color[i];
              if (ci >= 0)
                for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
                  const int cj = d.color[d.A.col[k]];
                  if (cj < 0) continue;
                  if ((part == 0 && cj < ci) || (part == 1 && cj > ci)) {
                    cc.push_back(d.A.col[k]);
                    vv.insert(vv.end(), d.A.val + k * bb, d.A.val + (k + 1) * bb);
                  }
                }
              rp[i + 1] = (int64_t)cc.size();
            }
            amgx_matrix F = d.A;
            F.rowptr = rp.data(); F.col = cc.data(); F.val = vv.data();
            DevMatrix& T = part == 0 ? g.blower : g.bupper;
            ok = build_bsell(F, T, 1e9, prow.data(), (int64_t)prow.size()) && T.n_slices == g.bcopy.n_slices;
          }
          g.bsplit = ok;
        }
      }
    }
  }
}

}  // namespace amgx
#include "devbuild.hpp"
#include "spgemm.hpp"
namespace amgx {

// Block-hybrid Gauss-Seidel data (gsb_sweep_kernel).  Validated like the colourings above: two coupled rows of one block
// sharing a colour would be a data race.
// csr: the level matrix on the device (big levels): the three images and the split are then formed there (devbuild.hpp)
// local-window image of the colour-sorted block image `full` (gsb_sweep_kernel<..., LW>): per block of B rows the sorted list of its
// distinct off-block columns; entries carry codes (in-block: row - r0 < B; off-block: B + position in the list).  Blocks whose list
// exceeds GSB_LW_CAP keep global 32-bit columns.  rows: the slot -> row list of the block image (-1 = padding), slots = n_blocks * B.
static bool build_gsb_full_lw(const amgx_matrix& A, const std::vector<int32_t>& rows, int64_t slots, int B, int G, DevGSB& g) {
  const int64_t n = A.n_rows, nnz = A.rowptr[n];
  const int64_t nb = (n + B - 1) / B;
  std::vector<int32_t> cnt((size_t)nb + 1, 0);
  RawVec<int32_t> code;
  code.resize((size_t)std::max<int64_t>(1, nnz));
  std::vector<std::vector<int32_t>> lists((size_t)nb);
  std::vector<char> no16((size_t)n, 0);
  std::vector<int64_t> n_over(setup_threads(), 0);
  int64_t cap = GSB_LW_CAP;
  const char* tcap = std::getenv("AMGX_LW_TEST_CAP");
  if (tcap) cap = std::min<int64_t>(cap, std::max<int64_t>(8, std::atoll(tcap) / 4));
  par_for(nb, [&](int64_t b0, int64_t b1, int t) {
    std::vector<int32_t> u;
    for (int64_t kb = b0; kb < b1; ++kb) {
      const int64_t r0 = kb * B, r1 = std::min<int64_t>(n, r0 + B);
      u.clear();
      for (int64_t k = A.rowptr[r0]; k < A.rowptr[r1]; ++k) if (A.col[k] < r0 || A.col[k] >= r1) u.push_back(A.col[k]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      if ((int64_t)u.size() > cap) {
        for (int64_t i = r0; i < r1; ++i) no16[i] = 1;
        for (int64_t k = A.rowptr[r0]; k < A.rowptr[r1]; ++k) code[k] = A.col[k];
        n_over[t]++;
        continue;
      }
      for (int64_t k = A.rowptr[r0]; k < A.rowptr[r1]; ++k) {
        const int32_t j = A.col[k];
        code[k] = (j >= r0 && j < r1) ? (int32_t)(j - r0) : (int32_t)(B + (std::lower_bound(u.begin(), u.end(), j) - u.begin()));
      }
      cnt[kb + 1] = (int32_t)u.size();
      lists[kb] = u;
    }
  }, 16);
  int64_t overs = 0;
  for (int64_t v : n_over) overs += v;
  if (overs * 20 > nb && !tcap) return false;
  for (int64_t kb = 0; kb < nb; ++kb) cnt[kb + 1] += cnt[kb];
  std::vector<int32_t> ccol((size_t)std::max<int32_t>(1, cnt[nb]));
  par_for(nb, [&](int64_t b0, int64_t b1, int) { for (int64_t kb = b0; kb < b1; ++kb) std::copy(lists[kb].begin(), lists[kb].end(), ccol.begin() + cnt[kb]); }, 64);
  amgx_matrix Lm = A;
  Lm.col = code.data();
  HostSell S;
  build_sell(Lm, rows.data(), slots, false, G, S, false, &no16);
  // every slice must fit the kernel's register budget exactly like `full` (same widths: same rows, same lengths)
  upload_sell(S, g.fullLW);
  g.flw_cptr.upload(cnt);
  g.flw_ccol.upload(ccol);
  g.has_fullLW = true;
  return true;
}

static void build_gsb(const amgx_level_desc& d, DevLevel& L, const amgx_matrix* P, const DevCsrSrc* csr = nullptr) {
  const int64_t n = d.A.n_rows;
  DevGSB& g = L.gsb;
  const int B = d.gs_block_rows;
  // lanes per row from the longest row (a lane holds <= 16 entries, + 1 when G = 1); workgroup size TH = B * G
  int64_t mx = 0;
  {
    std::vector<int64_t> tmx(setup_threads(), 0);
    par_for(n, [&](int64_t i0, int64_t i1, int t) { int64_t m = 0; for (int64_t i = i0; i < i1; ++i) m = std::max<int64_t>(m, d.A.rowptr[i + 1] - d.A.rowptr[i]); tmx[t] = m; });
    for (int64_t v : tmx) mx = std::max(mx, v);
  }
  int G = 1;
  while (G < 16 && mx > 16 * G + (G == 1 ? 1 : 0)) G <<= 1;
  // the ranks of a partitioned level agree on ONE block size (the smallest any of them needs, dist.py) while their longest
  // rows may differ: a rank with shorter rows spreads them over more lanes than it must, so that B * G is a workgroup size
  while (G < 16 && B >= 16 && B * G < 256) G <<= 1;
  if (n == 0) {                     // a rank that owns nothing: the form is "on" (the cycle drivers test it) with no block to sweep
    g.B = B > 0 ? B : 256; g.G = 1; g.TH = 256; g.n_colors = 0; g.n_blocks = 0;
    return;
  }
  const int TH = B * G;
  if (B < 16 || (TH != 256 && TH != 512 && TH != 1024))
    throw Err("gs_block_rows = " + std::to_string(B) + " with " + std::to_string(G) + " lanes per row (longest row: " + std::to_string(mx) +
              " entries) does not give a workgroup of 256, 512 or 1024 lanes");
  if (d.A.br != 1 || d.A.bc != 1) throw Err("block-hybrid Gauss-Seidel: scalar levels only");
  if (!d.color || d.n_colors <= 0 || d.n_colors > 254) { if (n > 0) throw Err("block-hybrid Gauss-Seidel needs a blocked colouring with at most 254 colours"); return; }
  if (!d.dinv) throw Err("dinv missing");
  const int nc = d.n_colors;
  par_for(n, [&](int64_t i0, int64_t i1, int) {
    for (int64_t i = i0; i < i1; ++i) {
      const int ci = d.color[i];
      if (ci >= nc) throw Err("colour index out of range");
      if (ci < 0) continue;
      const int64_t b0 = (i / B) * B, b1 = b0 + B;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
        const int64_t j = d.A.col[k];
        if (j != i && j >= b0 && j < b1 && j < n && d.color[j] == ci) throw Err("invalid blocked colouring: two coupled rows of one block share a colour");
      }
    }
  });
  g.B = B; g.G = G; g.TH = TH; g.n_colors = nc;
  g.n_blocks = (int)((n + B - 1) / B);
  const int64_t slots = (int64_t)g.n_blocks * B;
  std::vector<int32_t> rows((size_t)slots, -1);
  std::vector<uint8_t> sc((size_t)slots, 255);
  par_for(g.n_blocks, [&](int64_t k0, int64_t k1, int) {
    for (int64_t kb = k0; kb < k1; ++kb) {
      const int64_t b0 = kb * B, b1 = std::min<int64_t>(n, b0 + B);
      for (int64_t i = b0; i < b1; ++i) rows[i] = (int32_t)i;
      std::stable_sort(rows.begin() + b0, rows.begin() + b1, [&](int32_t a, int32_t c) {
        return (d.color[a] < 0 ? 255 : d.color[a]) < (d.color[c] < 0 ? 255 : d.color[c]); });
      for (int64_t q = b0; q < b1; ++q) sc[q] = d.color[rows[q]] < 0 ? 255 : (uint8_t)d.color[rows[q]];
    }
  }, 16);
  auto check_width = [&](const HostSell& S, const char* what) {
    const int64_t ns = (int64_t)S.slice_ptr.size() - 1;
    for (int64_t q = 0; q < ns; ++q) {
      const int w = (int)(((S.slice_ptr[q + 1] & ~(int64_t)63) - (S.slice_ptr[q] & ~(int64_t)63)) / WAVE);
      if (w > 2 * GSB_WP + 1) throw Err(std::string("block-hybrid Gauss-Seidel (") + what + "): a row has more than " + std::to_string((2 * GSB_WP) * G + 1) +
                                        " entries for gs_block_rows = " + std::to_string(B) + " (use smaller blocks)");
    }
  };
  g.rowid.upload(rows);
  g.slotcolor.upload(sc);
  {
    // long-row square levels: local-window image of `full` for the general sweep
    const double avgA = n ? (double)d.A.rowptr[n] / (double)n : 0.0;
    int64_t lw_min_rows = 100000;
    if (const char* e = std::getenv("AMGX_LW_MIN_ROWS")) lw_min_rows = std::atoll(e);
    // Measured NON-win at cfg 2 (profiles/r04/gs_experiments.txt): level-1 backward sweep 231 us with the window against 196 us without
    // (the extra barrier and the window's registers cost more than the gathers of the off-block values, which this kernel issues
    // back to back in one burst).  Built and tested, off unless AMGX_GSB_LW=1.
    if (avgA >= 24.0 && n >= lw_min_rows && d.A.n_cols == n && G > 1 && !std::getenv("AMGX_NO_LW") && std::getenv("AMGX_GSB_LW"))
      build_gsb_full_lw(d.A, rows, slots, B, G, g);
  }
  if (csr) {
    // device builders: the colour-sorted image of A, the split by kernels, the images of its two parts
    auto max_width = [&](const DevMatrix::Sell& S, const char* what) {
      const auto sp = db_download(S.slice_ptr, S.slice_ptr.n);
      int mw = 0;
      for (size_t q = 0; q + 1 < sp.size(); ++q) mw = std::max(mw, (int)(((sp[q + 1] & ~(int64_t)63) - (sp[q] & ~(int64_t)63)) / WAVE));
      if (mw > 2 * GSB_WP + 1) throw Err(std::string("block-hybrid Gauss-Seidel (") + what + "): a row has more than " + std::to_string((2 * GSB_WP) * G + 1) +
                                         " entries for gs_block_rows = " + std::to_string(B) + " (use smaller blocks)");
      return mw;
    };
    {
      DevBuf<int64_t> sp;
      const int64_t stored = dev_slice_offsets(*csr, g.rowid.p, slots, sp, G);
      dev_build_sell(*csr, g.rowid.p, slots, G, false, false, nullptr, 0.0, nullptr, sp, stored, g.full, nullptr);
      g.full_maxw = max_width(g.full, "A");
    }
    DevBuf<int32_t> dcolor;
    dcolor.upload(d.color, (size_t)n);
    DbSplit a{n, B, csr->rowptr.p, csr->col.p, csr->val.p, dcolor.p, L.dinv.p};
    DevCsrSrc part[2];
    for (int q = 0; q < 2; ++q) { part[q].n_rows = n; part[q].n_cols = d.A.n_cols; part[q].rowptr.alloc((size_t)n + 1); }
    const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(db_split_count_kernel, dim3(grid), dim3(BLOCK), 0, 0, a, part[0].rowptr.p, part[1].rowptr.p);
    HIPCHK(hipGetLastError());
    for (int q = 0; q < 2; ++q) {
      hipLaunchKernelGGL(db_scan_kernel, dim3(1), dim3(1024), 0, 0, n, part[q].rowptr.p);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpy(&part[q].nnz, part[q].rowptr.p + n, sizeof(int64_t), hipMemcpyDeviceToHost));
      part[q].col.alloc((size_t)std::max<int64_t>(1, part[q].nnz));
      part[q].val.alloc((size_t)std::max<int64_t>(1, part[q].nnz));
    }
    g.cvec.alloc((size_t)n);
    DevBuf<int> bad;
    bad.alloc(1);
    HIPCHK(hipMemset(bad.p, 0, sizeof(int)));
    hipLaunchKernelGGL(db_split_fill_kernel, dim3(grid), dim3(BLOCK), 0, 0, a, part[0].rowptr.p, part[1].rowptr.p, part[0].col.p, part[0].val.p,
                       part[1].col.p, part[1].val.p, g.cvec.p, bad.p);
    HIPCHK(hipGetLastError());
    int hbad = 0;
    HIPCHK(hipMemcpy(&hbad, bad.p, sizeof(int), hipMemcpyDeviceToHost));
    if (hbad || std::getenv("AMGX_GSB_NO_SPLIT")) { g.cvec.release(); return; }
    {
      DevBuf<int64_t> sp;
      const int64_t stored = dev_slice_offsets(part[0], g.rowid.p, slots, sp, G);
      dev_build_sell(part[0], g.rowid.p, slots, G, false, false, nullptr, 0.0, nullptr, sp, stored, g.lowin, nullptr);
      g.lowin_maxw = max_width(g.lowin, "lower part");
    }
    if (!dev_upload_matrix(part[1], g.rest, false, 1.6, SELL_WIN, nullptr)) {
      // (the host builder's CSR fallback for a badly padded remainder: from the host copy of the part)
      std::vector<int64_t> rp = db_download(part[1].rowptr, (size_t)n + 1);
      std::vector<int32_t> cc = db_download(part[1].col, (size_t)std::max<int64_t>(1, part[1].nnz));
      std::vector<double> vv = db_download(part[1].val, (size_t)std::max<int64_t>(1, part[1].nnz));
      amgx_matrix F = d.A;
      F.rowptr = rp.data(); F.col = cc.data(); F.val = vv.data();
      upload_matrix(F, g.rest, "A (block-hybrid Gauss-Seidel: rest)", true, false, false, 1.6, SELL_WIN);
    }
    g.has_split = true;
    {
      // long-row levels: local-window image of the rest part (host builder from the downloaded part)
      const double avgA = n ? (double)csr->nnz / (double)n : 0.0;
      int64_t lw_min_rows = 100000;
      if (const char* e = std::getenv("AMGX_LW_MIN_ROWS")) lw_min_rows = std::atoll(e);
      if (P && avgA >= 24.0 && n >= lw_min_rows && d.A.n_cols == n && P->br == 1 && P->bc == 1 && P->rowptr[P->n_rows] < (int64_t)2147483647 &&
          !std::getenv("AMGX_NO_LW") && !std::getenv("AMGX_NO_FUSED_RESTRICT")) {
        // (device builder first, devbuild.hpp dev_build_lw; the host builder works from the downloaded part)
        std::vector<int64_t> rp;
        std::vector<int32_t> cc;
        std::vector<double> vv;
        amgx_matrix F = d.A;
        auto host_lw = [&](int gg, DevMatrix& M, DevBuf<int32_t>& cp, DevBuf<int32_t>& cl) {
          if (rp.empty()) {
            rp = db_download(part[1].rowptr, (size_t)n + 1);
            cc = db_download(part[1].col, (size_t)std::max<int64_t>(1, part[1].nnz));
            vv = db_download(part[1].val, (size_t)std::max<int64_t>(1, part[1].nnz));
            F.rowptr = rp.data(); F.col = cc.data(); F.val = vv.data();
          }
          return build_sell_lw(F, F.val, gg, M, cp, cl);
        };
        const bool dev_lw = !std::getenv("AMGX_HOST_LW");
        const bool verify_lw = std::getenv("AMGX_VERIFY_IMAGES") != nullptr;
        int G = 0;
        for (int gg : {2, 4}) {
          bool ok = false;
          if (dev_lw) {
            int64_t cap = LW_CAP;
            const char* tcap = std::getenv("AMGX_LW_TEST_CAP");
            if (tcap) cap = std::min<int64_t>(cap, std::atoll(tcap));
            ok = dev_build_lw(part[1], false, gg, cap, tcap != nullptr, nullptr, 0.0, g.restLW, g.lw_cptr, g.lw_ccol);
            if (ok && verify_lw) {
              DevMatrix H; DevBuf<int32_t> hp, hc;
              if (!host_lw(gg, H, hp, hc)) throw Err("AMGX_VERIFY_IMAGES: Gauss-Seidel rest (local window): the host builder declines what the device builder forms");
              verify_same_lw(g.restLW, g.lw_cptr, g.lw_ccol, H, hp, hc, "Gauss-Seidel rest (local window)");
            }
            if (!ok) { g.restLW = DevMatrix(); g.lw_cptr.release(); g.lw_ccol.release(); }
          }
          if (!ok) ok = host_lw(gg, g.restLW, g.lw_cptr, g.lw_ccol);
          if (ok) { G = gg; break; }
          g.restLW = DevMatrix();
        }
        if (G) {
          build_restrict(*P, L.RG, 512 / G, 4 * 512, 512);
          if (!L.RG.empty()) return;
          g.restLW = DevMatrix(); g.lw_cptr.release(); g.lw_ccol.release();
        }
      }
    }
    if (P && g.rest.fmt == FMT_SELL && g.rest.lanes == 1 && P->br == 1 && P->bc == 1 &&
        P->rowptr[P->n_rows] < (int64_t)2147483647 && !std::getenv("AMGX_NO_FUSED_RESTRICT"))
      build_restrict(*P, L.RG, 512, 6 * 512);
    return;
  }
  {
    HostSell S;
    build_sell(d.A, rows.data(), slots, false, G, S);
    check_width(S, "A");
    g.full_maxw = 0;
    for (size_t q = 0; q + 1 < S.slice_ptr.size(); ++q)
      g.full_maxw = std::max(g.full_maxw, (int)(((S.slice_ptr[q + 1] & ~(int64_t)63) - (S.slice_ptr[q] & ~(int64_t)63)) / WAVE));
    upload_sell(S, g.full);
  }
  // split for the pre-smoothing from zero: lowin = in-block couplings to lower colours, rest = everything else but the
  // diagonal; couplings to non-free columns are dropped (x is 0 there), non-free rows are empty (their residual is not
  // needed: their prolongation rows are empty)
  std::vector<double> cv((size_t)n, 0.0);
  std::vector<int64_t> rp[2];
  RawVec<int32_t> cc[2];
  RawVec<double> vv[2];
  for (int part = 0; part < 2; ++part) rp[part].assign(n + 1, 0);
  // part of entry (i, k): 0 = in-block coupling to a lower colour, 1 = rest, -1 = dropped (the diagonal, a non-free column)
  auto part_of = [&](int64_t i, int ci, int64_t j) -> int {
    if (j == i) return -1;
    const int cj = j < n ? d.color[j] : 0;            // ghost columns count as live
    if (cj < 0) return -1;
    const int64_t b0 = (i / B) * B, b1 = b0 + B;
    return (j >= b0 && j < b1 && j < n && cj < ci) ? 0 : 1;
  };
  par_for(n, [&](int64_t i0, int64_t i1, int) {
    for (int64_t i = i0; i < i1; ++i) {
      const int ci = d.color[i];
      if (ci < 0) continue;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
        const int pt = part_of(i, ci, d.A.col[k]);
        if (pt >= 0) rp[pt][i + 1]++;
      }
    }
  });
  for (int part = 0; part < 2; ++part) {
    for (int64_t i = 0; i < n; ++i) rp[part][i + 1] += rp[part][i];
    cc[part].resize((size_t)std::max<int64_t>(1, rp[part][n]));
    vv[part].resize((size_t)std::max<int64_t>(1, rp[part][n]));
  }
  std::vector<char> bad(setup_threads(), 0);
  par_for(n, [&](int64_t i0, int64_t i1, int t) {
    for (int64_t i = i0; i < i1; ++i) {
      const int ci = d.color[i];
      if (ci < 0) continue;
      int64_t o[2] = {rp[0][i], rp[1][i]};
      double aii = 0.0;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
        const int64_t j = d.A.col[k];
        if (j == i) { aii = d.A.val[k]; continue; }
        const int pt = part_of(i, ci, j);
        if (pt < 0) continue;
        cc[pt][o[pt]] = (int32_t)j; vv[pt][o[pt]] = d.A.val[k]; ++o[pt];
      }
      if (d.dinv[i] != 0.0) cv[i] = 1.0 / d.dinv[i] - aii; else bad[t] = 1;     // a swept row without a diagonal inverse
    }
  });
  bool ok = true;
  for (char c : bad) if (c) ok = false;
  if (ok && !std::getenv("AMGX_GSB_NO_SPLIT")) {
    amgx_matrix F = d.A;
    F.rowptr = rp[0].data(); F.col = cc[0].data(); F.val = vv[0].data();
    HostSell S;
    build_sell(F, rows.data(), slots, false, G, S);
    check_width(S, "lower part");
    upload_sell(S, g.lowin);
    g.lowin_maxw = 0;
    for (size_t q = 0; q + 1 < S.slice_ptr.size(); ++q)
      g.lowin_maxw = std::max(g.lowin_maxw, (int)(((S.slice_ptr[q + 1] & ~(int64_t)63) - (S.slice_ptr[q] & ~(int64_t)63)) / WAVE));
    F.rowptr = rp[1].data(); F.col = cc[1].data(); F.val = vv[1].data();
    upload_matrix(F, g.rest, "A (block-hybrid Gauss-Seidel: rest)", true, false, false, 1.6, SELL_WIN);
    g.cvec.upload(cv);
    g.has_split = true;
    {
      const double avgA = n ? (double)d.A.rowptr[n] / (double)n : 0.0;
      int64_t lw_min_rows = 100000;
      if (const char* e = std::getenv("AMGX_LW_MIN_ROWS")) lw_min_rows = std::atoll(e);
      if (P && avgA >= 24.0 && n >= lw_min_rows && d.A.n_cols == n && P->br == 1 && P->bc == 1 && P->rowptr[P->n_rows] < (int64_t)2147483647 &&
          !std::getenv("AMGX_NO_LW") && !std::getenv("AMGX_NO_FUSED_RESTRICT")) {
        int G = 0;
        for (int gg : {2, 4}) if (build_sell_lw(F, F.val, gg, g.restLW, g.lw_cptr, g.lw_ccol)) { G = gg; break; } else g.restLW = DevMatrix();
        if (G) {
          build_restrict(*P, L.RG, 512 / G, 4 * 512, 512);
          if (!L.RG.empty()) return;
          g.restLW = DevMatrix(); g.lw_cptr.release(); g.lw_ccol.release();
        }
      }
    }
    if (P && g.rest.fmt == FMT_SELL && g.rest.lanes == 1 && P->br == 1 && P->bc == 1 &&
        P->rowptr[P->n_rows] < (int64_t)2147483647 && !std::getenv("AMGX_NO_FUSED_RESTRICT"))
      build_restrict(*P, L.RG, 512, 6 * 512);
  }
}

// ---------------------------------------------------------------------------------------------------
// Folded post-smoothing.  In the V(1,1) Jacobi cycle the post-smoothing step (amg_matrix.cpp:263-302)
//     x' = t + omega*Dinv*(b - A t),   t = x + P x_c
// is affine in (x, x_c):  x' = [x + omega*Dinv*(b - A x)] + (I - omega*Dinv*A) P x_c = z + Q x_c,  where b - A x is the
// residual the pre-smoothing pass already has in registers.  So the pre-smoothing kernel writes z instead of x (EPF_FOLD)
// and the whole post-smoothing is ONE SpMV-AXPY with Q = (I - omega*Dinv*A) P, built once here (host, threads over row
// ranges).  Q has about half the entries of A (cfg 2: 7.8 vs 14.8 per row), and the pass over P and the round trip of
// t through HBM disappear.  Same result up to rounding; AMGX_NO_FOLD=1 runs the literal sequence.
// BSELL image of selected entries of a square-block matrix: `rows` lists block rows in storage order (-1 = padding slot),
// slices of RB = 64 / bs consecutive list entries; keep(i, j) selects entries, mapcol(i, j) gives the stored block column,
// padcol = column of padding steps (any valid index of the gathered vector).
template <class Keep, class MapCol, class Scale>
static void build_bsell_sel(const amgx_matrix& A, const std::vector<int32_t>& rows, Keep keep, MapCol mapcol, int32_t padcol, DevMatrix& D, Scale scale);
template <class Keep, class MapCol>
static void build_bsell_sel(const amgx_matrix& A, const std::vector<int32_t>& rows, Keep keep, MapCol mapcol, int32_t padcol, DevMatrix& D) {
  build_bsell_sel(A, rows, keep, mapcol, padcol, D, [](int32_t, int32_t) { return 1.0; });
}
template <class Keep, class MapCol, class Scale>
static void build_bsell_sel(const amgx_matrix& A, const std::vector<int32_t>& rows, Keep keep, MapCol mapcol, int32_t padcol, DevMatrix& D, Scale scale) {
  const int bs = A.br;
  const int RB = WAVE / bs;
  const int64_t m = (int64_t)rows.size();
  if (m % RB) throw Err("build_bsell_sel: the row list must be padded to whole slices");
  const int64_t ns = m / RB;
  std::vector<int64_t> sp(ns + 1, 0);
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s) {
      int w = 0;
      for (int rb = 0; rb < RB; ++rb) {
        const int32_t i = rows[s * RB + rb];
        if (i < 0) continue;
        int c = 0;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) if (keep(i, A.col[k])) ++c;
        w = std::max(w, c);
      }
      sp[s + 1] = w;
    }
  }, 16);
  for (int64_t s = 0; s < ns; ++s) sp[s + 1] += sp[s];
  const int64_t steps = sp[ns];
  RawVec<int32_t> col;
  RawVec<double> val;
  par_assign(col, (size_t)std::max<int64_t>(1, steps * RB), padcol);
  par_assign(val, (size_t)std::max<int64_t>(1, steps * bs * WAVE), 0.0);
  par_for(ns, [&](int64_t s0, int64_t s1, int) {
    for (int64_t s = s0; s < s1; ++s)
      for (int rb = 0; rb < RB; ++rb) {
        const int32_t i = rows[s * RB + rb];
        if (i < 0) continue;
        int64_t kk = sp[s];
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
          const int32_t j = A.col[k];
          if (!keep(i, j)) continue;
          col[kk * RB + rb] = mapcol(i, j);
          const double* blk = A.val + k * bs * bs;
          const double sc = scale(i, j);
          double* vk = val.data() + kk * (bs * WAVE);
          for (int rr = 0; rr < bs; ++rr) {
            const int lane = rb * bs + rr;
            for (int c = 0; c < bs; ++c) {
              if ((bs & 1) && c == bs - 1) vk[(bs / 2) * (2 * WAVE) + lane] = sc * blk[rr * bs + c];
              else vk[(c / 2) * (2 * WAVE) + lane * 2 + (c & 1)] = sc * blk[rr * bs + c];
            }
          }
          ++kk;
        }
      }
  }, 16);
  D.fmt = FMT_BSELL; D.br = D.bc = bs;
  D.n_rows = A.n_rows; D.n_cols = A.n_cols;
  D.n_slices = (int)ns;
  D.stored = steps * RB;
  D.stream_bytes = steps * ((int64_t)bs * WAVE * 8 + RB * 4) + 8 * (ns + 1);
  D.bsell.slice_ptr.upload(sp); D.bsell.col.upload(col); D.bsell.val.upload(val);
}

// Block-hybrid Gauss-Seidel data of a square-block level (bgsb_sweep_kernel); the blocked colouring is validated (two coupled
// rows of one workgroup block sharing a colour would be a data race)
// csr: the level matrix on the device (big levels): the images are then gathered there (devbuild.hpp, dev_build_bsell)
static void build_bgsb(const amgx_level_desc& d, DevLevel& L, const DevBcsrSrc* csr = nullptr) {
  const int64_t n = d.A.n_rows;
  const int bs = d.A.br, RB = WAVE / bs;
  DevBGSB& g = L.bgsb;
  const int BB = d.gs_block_rows;
  if (bs != 2 && bs != 3 && bs != 6) throw Err("block-hybrid Gauss-Seidel: block sizes 2, 3, 6");
  if (d.A.n_cols < n) throw Err("block-hybrid Gauss-Seidel on block levels: n_cols < n_rows");
  // (n_cols > n_rows: trailing ghost columns of a rank-partitioned level; they belong to no sweep block, so their couplings sit in
  //  `off` / `rest` with global column ids and multiply whatever the caller's exchange left behind the owned entries)
  if (BB < RB || BB > 2048 || (int64_t)2 * BB * bs * 8 > 96 * 1024) throw Err("block-hybrid Gauss-Seidel: gs_block_rows out of range for this block size");
  if (n == 0) { g.BB = BB; return; }
  if (!d.color || d.n_colors <= 0) throw Err("block-hybrid Gauss-Seidel needs a blocked colouring");
  if (!d.dinv) throw Err("dinv missing");
  const int nc = d.n_colors;
  std::vector<char> bad(setup_threads(), 0);
  par_for(n, [&](int64_t i0, int64_t i1, int t) {
    for (int64_t i = i0; i < i1; ++i) {
      const int ci = d.color[i];
      if (ci >= nc) { bad[t] = 1; return; }
      if (ci < 0) continue;
      if (d.gs_block_ids) continue;                       // (checked against the block ids below)
      const int64_t b0 = (i / BB) * BB, b1 = b0 + BB;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
        const int64_t j = d.A.col[k];
        if (j != i && j >= b0 && j < b1 && j < n && d.color[j] == ci) { bad[t] = 2; return; }
      }
    }
  });
  for (char c : bad) { if (c == 1) throw Err("colour index out of range"); if (c == 2) throw Err("invalid blocked colouring: two coupled block rows of one block share a colour"); }
  // sweep blocks: runs of BB consecutive rows, or the caller's compact blocks (gs_block_ids, at most BB rows each)
  std::vector<int32_t> blk_of((size_t)n), lpos((size_t)n), blk_ptr, blk_rows((size_t)n);
  int nblk = 0;
  if (d.gs_block_ids) {
    int32_t mx = -1;
    for (int64_t i = 0; i < n; ++i) { if (d.gs_block_ids[i] < 0) throw Err("gs_block_ids: negative block id"); mx = std::max(mx, d.gs_block_ids[i]); }
    nblk = mx + 1;
    blk_ptr.assign((size_t)nblk + 1, 0);
    for (int64_t i = 0; i < n; ++i) { blk_of[i] = d.gs_block_ids[i]; blk_ptr[blk_of[i] + 1]++; }
    for (int q = 0; q < nblk; ++q) { if (blk_ptr[q + 1] > BB) throw Err("gs_block_ids: a block has more than gs_block_rows rows"); blk_ptr[q + 1] += blk_ptr[q]; }
    std::vector<int32_t> pos(blk_ptr.begin(), blk_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) { lpos[i] = pos[blk_of[i]] - blk_ptr[blk_of[i]]; blk_rows[pos[blk_of[i]]++] = (int32_t)i; }
  } else {
    nblk = (int)((n + BB - 1) / BB);
    blk_ptr.assign((size_t)nblk + 1, 0);
    for (int q = 0; q <= nblk; ++q) blk_ptr[q] = (int32_t)std::min<int64_t>(n, (int64_t)q * BB);
    for (int64_t i = 0; i < n; ++i) { blk_of[i] = (int32_t)(i / BB); lpos[i] = (int32_t)(i % BB); blk_rows[i] = (int32_t)i; }
  }
  // (the colouring was validated against runs of consecutive rows above; with block ids it is validated here)
  if (d.gs_block_ids) {
    std::vector<char> bad2(setup_threads(), 0);
    par_for(n, [&](int64_t i0, int64_t i1, int t) {
      for (int64_t i = i0; i < i1; ++i) {
        const int ci = d.color[i];
        if (ci < 0) continue;
        for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
          const int64_t j = d.A.col[k];
          if (j != i && j < n && blk_of[j] == blk_of[i] && d.color[j] == ci) { bad2[t] = 1; return; }
        }
      }
    });
    for (char c : bad2) if (c) throw Err("invalid blocked colouring: two coupled block rows of one sweep block share a colour");
  }
  g.BB = BB; g.n_blocks = nblk; g.n_colors = nc;
  // block-coloured form: colour of every sweep block (constant inside a block, coupled blocks differ)
  std::vector<int32_t> bcol;
  if (d.gs_block_color && d.gs_n_block_colors > 0) {
    if (d.A.n_cols != n) throw Err("block-coloured Gauss-Seidel: square levels only (rank-partitioned levels sweep in the hybrid form)");
    const int nbc = d.gs_n_block_colors;
    bcol.assign((size_t)nblk, -1);
    for (int64_t i = 0; i < n; ++i) {
      const int c = d.gs_block_color[i];
      if (c < 0 || c >= nbc) throw Err("gs_block_color: colour out of range");
      if (bcol[blk_of[i]] < 0) bcol[blk_of[i]] = c;
      else if (bcol[blk_of[i]] != c) throw Err("gs_block_color: not constant inside a sweep block");
    }
    std::vector<char> bad3(setup_threads(), 0);
    par_for(n, [&](int64_t i0, int64_t i1, int t) {
      for (int64_t i = i0; i < i1; ++i) {
        if (d.color[i] < 0) continue;
        for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) {
          const int64_t j = d.A.col[k];
          if (d.color[j] >= 0 && blk_of[j] != blk_of[i] && bcol[blk_of[j]] == bcol[blk_of[i]]) { bad3[t] = 1; return; }
        }
      }
    });
    for (char c : bad3) if (c) throw Err("invalid block colouring: two coupled sweep blocks share a colour (an in-place sweep would race)");
    std::vector<int32_t> list((size_t)nblk);
    g.bc_ptr.assign((size_t)nbc + 1, 0);
    for (int q = 0; q < nblk; ++q) g.bc_ptr[bcol[q] + 1]++;
    for (int c = 0; c < nbc; ++c) g.bc_ptr[c + 1] += g.bc_ptr[c];
    std::vector<int> pos(g.bc_ptr.begin(), g.bc_ptr.end() - 1);
    for (int q = 0; q < nblk; ++q) list[pos[bcol[q]]++] = q;
    g.blk_list.upload(list);
    g.bc = true;
    g.n_bcolors = nbc;
  }
  // off: block by block in list order, every block padded to whole slices
  std::vector<int32_t> rows_off, off_ptr(nblk + 1, 0);
  rows_off.reserve((size_t)n + (size_t)nblk * RB);
  for (int blk = 0; blk < nblk; ++blk) {
    for (int32_t q = blk_ptr[blk]; q < blk_ptr[blk + 1]; ++q) rows_off.push_back(blk_rows[q]);
    while (rows_off.size() % RB) rows_off.push_back(-1);
    off_ptr[blk + 1] = (int32_t)(rows_off.size() / RB);
  }
  auto same_block = [&blk_of, n](int64_t i, int64_t j) { return j < n && blk_of[i] == blk_of[j]; };
  // Only the in-block couplings to the colours a sweep has ALREADY visited need its new values; everything else -- couplings
  // that leave the block, the diagonal block, in-block couplings to the colours still to come -- multiplies sweep-start values
  // and goes into the streaming phase 0, where all waves work.  The colour phases, which run one after the other inside a
  // workgroup, are left with ~9 % of A for line blocks (22 % for compact blocks) instead of 20 % (50 %).
  auto lower_in = [&](int32_t i, int32_t j) { return j != i && same_block(i, j) && d.color[i] >= 0 && d.color[j] >= 0 && d.color[j] < d.color[i]; };
  auto upper_in = [&](int32_t i, int32_t j) { return j != i && same_block(i, j) && d.color[i] >= 0 && d.color[j] >= 0 && d.color[j] > d.color[i]; };
  DevBuf<int32_t> d_blk_of, d_lpos, d_color, d_rows;
  DevBuf<double> d_fac;
  DbBgsbMaps maps;
  if (csr) {
    d_blk_of.upload(blk_of); d_lpos.upload(lpos); d_color.upload(d.color, (size_t)n);
    maps.blk_of = d_blk_of.p; maps.lpos = d_lpos.p; maps.color = d_color.p;
    d_rows.upload(rows_off);
    dev_build_bsell(*csr, d_rows.p, (int64_t)rows_off.size(), BB_OFF, maps, 0, 0.0, g.off);
  } else
  build_bsell_sel(d.A, rows_off, [&](int32_t i, int32_t j) { return !lower_in(i, j) && !upper_in(i, j); }, [](int32_t, int32_t j) { return j; }, 0, g.off);
  // block-coloured form: which entries of `off` couple to a sweep block that a FORWARD sweep visits later (the "high" part);
  // everything else of `off` (diagonal blocks, couplings to earlier blocks, couplings to rows that are never swept) is "low"
  auto off_high = [&](int32_t i, int32_t j) { return j != i && j < n && !same_block(i, j) && d.color[i] >= 0 && d.color[j] >= 0 && bcol[blk_of[j]] > bcol[blk_of[i]]; };
  auto off_low = [&](int32_t i, int32_t j) { return j != i && j < n && !same_block(i, j) && d.color[i] >= 0 && d.color[j] >= 0 && bcol[blk_of[j]] < bcol[blk_of[i]]; };
  DevBuf<int32_t> d_bcol;
  if (g.bc && csr) {
    std::vector<int32_t> bcr((size_t)n);
    for (int64_t i = 0; i < n; ++i) bcr[i] = bcol[blk_of[i]];
    d_bcol.upload(bcr);
    maps.bcolor = d_bcol.p;
  }
  // in: per block the swept rows by colour, every (block, colour) group padded to whole slices
  std::vector<int32_t> rows_in, in_ptr((size_t)nblk * nc + 1, 0), in_row;
  for (int blk = 0; blk < nblk; ++blk) {
    for (int c = 0; c < nc; ++c) {
      for (int32_t q = blk_ptr[blk]; q < blk_ptr[blk + 1]; ++q) if (d.color[blk_rows[q]] == c) rows_in.push_back(blk_rows[q]);
      while (rows_in.size() % RB) rows_in.push_back(-1);
      in_ptr[(size_t)blk * nc + c + 1] = (int32_t)(rows_in.size() / RB);
    }
  }
  in_row.resize(rows_in.size());
  for (size_t q = 0; q < rows_in.size(); ++q) in_row[q] = rows_in[q] < 0 ? -1 : lpos[rows_in[q]];
  if (csr) {
    d_rows.upload(rows_in);
    dev_build_bsell(*csr, d_rows.p, (int64_t)rows_in.size(), BB_IN, maps, 0, 0.0, g.in);
    dev_build_bsell(*csr, d_rows.p, (int64_t)rows_in.size(), BB_UPIN, maps, 0, 0.0, g.upin);
  } else {
    build_bsell_sel(d.A, rows_in, lower_in, [&lpos](int32_t, int32_t j) { return lpos[j]; }, 0, g.in);
    build_bsell_sel(d.A, rows_in, upper_in, [&lpos](int32_t, int32_t j) { return lpos[j]; }, 0, g.upin);
  }
  if (g.upin.n_slices != g.in.n_slices) throw Err("block-hybrid Gauss-Seidel: lower / upper slice mismatch");
  g.blk_ptr.upload(blk_ptr); g.blk_rows.upload(blk_rows);
  g.off_ptr.upload(off_ptr); g.in_ptr.upload(in_ptr); g.in_row.upload(in_row);
  // ---- one-pass pre-smoothing from zero: valid where dinv_k is a true inverse of fac_k * A_kk (not a pseudo-inverse) ----------
  if (std::getenv("AMGX_BGSB_NO_SPLIT")) return;
  std::vector<double> fac((size_t)n, 1.0);
  std::vector<char> nofac(setup_threads(), 0);
  par_for(n, [&](int64_t i0, int64_t i1, int t) {
    for (int64_t i = i0; i < i1; ++i) {
      if (d.color[i] < 0) continue;
      const double* Akk = nullptr;
      for (int64_t k = d.A.rowptr[i]; k < d.A.rowptr[i + 1]; ++k) if (d.A.col[k] == i) { Akk = d.A.val + k * bs * bs; break; }
      if (!Akk) { nofac[t] = 1; return; }
      const double* Dk = d.dinv + i * bs * bs;
      // M = Dinv_k A_kk must be (1 / fac) I with fac >= 1
      double m00 = 0.0;
      for (int c = 0; c < bs; ++c) m00 += Dk[c] * Akk[c * bs];
      if (!(m00 > 1e-12 && m00 <= 1.0 + 1e-10)) { nofac[t] = 1; return; }
      for (int r = 0; r < bs; ++r)
        for (int c = 0; c < bs; ++c) {
          double m = 0.0;
          for (int q = 0; q < bs; ++q) m += Dk[r * bs + q] * Akk[q * bs + c];
          if (std::fabs(m - (r == c ? m00 : 0.0)) > 1e-10 * m00) { nofac[t] = 1; return; }
        }
      fac[i] = 1.0 / m00;
    }
  });
  for (char c : nofac) if (c) return;
  std::vector<int32_t> rows_nat;
  for (int64_t i = 0; i < n; ++i) rows_nat.push_back((int32_t)i);
  while (rows_nat.size() % RB) rows_nat.push_back(-1);
  if (g.bc) {
    // block-coloured form: the plain inverse is expected (fac = 1 everywhere), otherwise no split (sweep + full residual)
    for (int64_t i = 0; i < n; ++i) if (d.color[i] >= 0 && std::fabs(fac[i] - 1.0) > 1e-10) return;
    if (csr) {
      d_rows.upload(rows_off);
      dev_build_bsell(*csr, d_rows.p, (int64_t)rows_off.size(), BB_OFFLO, maps, 0, 0.0, g.offlo);
      d_rows.upload(rows_nat);
      dev_build_bsell(*csr, d_rows.p, (int64_t)rows_nat.size(), BB_RESTBC, maps, 0, 0.0, g.rest);
    } else {
      build_bsell_sel(d.A, rows_off, off_low, [](int32_t, int32_t j) { return j; }, 0, g.offlo);
      build_bsell_sel(d.A, rows_nat, [&](int32_t i, int32_t j) { return upper_in(i, j) || off_high(i, j); }, [](int32_t, int32_t j) { return j; }, 0, g.rest,
                      [](int32_t, int32_t) { return -1.0; });
    }
    g.has_split = true;
    return;
  }
  if (csr) {
    d_fac.upload(fac);
    maps.fac = d_fac.p;
    d_rows.upload(rows_nat);
    dev_build_bsell(*csr, d_rows.p, (int64_t)rows_nat.size(), BB_REST, maps, 0, 0.0, g.rest);
  } else
  build_bsell_sel(d.A, rows_nat, [&](int32_t i, int32_t j) { return !lower_in(i, j); }, [](int32_t, int32_t j) { return j; }, 0, g.rest,
                  [&](int32_t i, int32_t j) { return (i == j && d.color[i] >= 0) ? fac[i] - 1.0 : -1.0; });
  g.has_split = true;
}

struct HostCsr {
  std::vector<int64_t> rowptr;
  std::vector<int32_t> col;
  std::vector<double> val;
};

// (block form: A has bs x bs blocks, P and Q bs x bc blocks, dinv bs x bs per block row)
static void fold_prolongation(const amgx_matrix& A, const amgx_matrix& P, const double* dinv, double omega, HostCsr& Q) {
  const int64_t n = A.n_rows, nc = P.n_cols;
  const int bs = A.br, bc = P.bc, bb = bs * bc;
  int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
  if (const char* e = std::getenv("OMP_NUM_THREADS")) T = std::max(1, std::min(T, std::atoi(e)));
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n / 4096 + 1));
  struct Part { std::vector<int32_t> col; std::vector<double> val; };
  std::vector<Part> parts(T);
  Q.rowptr.assign(n + 1, 0);
  auto work = [&](int t) {
    const int64_t r0 = n * t / T, r1 = n * (t + 1) / T;
    std::vector<int32_t> mark(nc, -1), cols, order;
    std::vector<double> own, acc, tmp(bb);    // own = P_i blocks, acc = sum_j A_ij P_j blocks (same slot numbering)
    Part& out = parts[t];
    for (int64_t i = r0; i < r1; ++i) {
      cols.clear(); own.clear(); acc.clear();
      auto slot = [&](int32_t c) -> int32_t {
        int32_t& m = mark[c];
        if (m < 0) { m = (int32_t)cols.size(); cols.push_back(c); own.resize(own.size() + bb, 0.0); acc.resize(acc.size() + bb, 0.0); }
        return m;
      };
      for (int64_t k = P.rowptr[i]; k < P.rowptr[i + 1]; ++k) {
        const int32_t sl = slot(P.col[k]);          // (may reallocate own: take the pointer afterwards)
        double* o = own.data() + (size_t)sl * bb;
        for (int e = 0; e < bb; ++e) o[e] += P.val[k * bb + e];
      }
      const double* d = dinv + i * bs * bs;
      bool dzero = true;
      for (int e = 0; e < bs * bs; ++e) if (d[e] != 0.0) { dzero = false; break; }
      if (!dzero)
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
          const int64_t j = A.col[k];
          if (j >= P.n_rows) continue;
          const double* a = A.val + k * bs * bs;
          for (int64_t q = P.rowptr[j]; q < P.rowptr[j + 1]; ++q) {
            const double* pv = P.val + q * bb;
            const int32_t sl = slot(P.col[q]);       // (may reallocate acc: take the pointer afterwards)
            double* o = acc.data() + (size_t)sl * bb;
            for (int r = 0; r < bs; ++r)
              for (int m = 0; m < bs; ++m) {
                const double arm = a[r * bs + m];
                for (int c = 0; c < bc; ++c) o[r * bc + c] += arm * pv[m * bc + c];
              }
          }
        }
      order.resize(cols.size());
      for (size_t q = 0; q < cols.size(); ++q) order[q] = (int32_t)q;
      std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cols[a] < cols[b]; });
      for (int32_t q : order) {
        const double* o = own.data() + (size_t)q * bb;
        const double* sa = acc.data() + (size_t)q * bb;
        // Q block = P block - omega * Dinv_i * (A P) block
        for (int r = 0; r < bs; ++r)
          for (int c = 0; c < bc; ++c) {
            double u = 0.0;
            for (int m = 0; m < bs; ++m) u += d[r * bs + m] * sa[m * bc + c];
            tmp[r * bc + c] = o[r * bc + c] - omega * u;
          }
        out.col.push_back(cols[q]);
        out.val.insert(out.val.end(), tmp.begin(), tmp.end());
        mark[cols[q]] = -1;
      }
      Q.rowptr[i + 1] = (int64_t)cols.size();
    }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int64_t i = 0; i < n; ++i) Q.rowptr[i + 1] += Q.rowptr[i];
  Q.col.resize((size_t)Q.rowptr[n]);
  Q.val.resize((size_t)Q.rowptr[n] * bb);
  for (int t = 0; t < T; ++t) {
    const int64_t o = Q.rowptr[n * t / T];
    std::copy(parts[t].col.begin(), parts[t].col.end(), Q.col.begin() + o);
    std::copy(parts[t].val.begin(), parts[t].val.end(), Q.val.begin() + o * bb);
    Part().col.swap(parts[t].col); Part().val.swap(parts[t].val);
  }
}

// ---------------------------------------------------------------------------------------------------
// Colour-major numbering of Gauss-Seidel levels.  A colour kernel touches the rows of ONE colour; in the caller's
// numbering those are every ~8th row, so x, b and dinv were read and written with 1/8 line utilisation, eight times
// per sweep (GS V-cycle at cfg 2: colour kernels at 2-3 TB/s).  With the rows of a colour stored contiguously the
// same kernels stream.  The renumbering is internal: the host matrices are permuted once here (P A P^T, rows of P,
// columns of P^T, ...) and the entry points translate vectors at the boundary (two extra passes over b and x per
// application, ~60 us at cfg 2, against ~400 us saved).  AMGX_NO_GS_PERM=1 keeps the caller's numbering.
struct LevelPerm {
  std::vector<int32_t> perm, iperm;       // perm[new] = old, iperm[old] = new; empty = identity
  HostCsr A, P, PT;
  std::vector<double> dinv;
  std::vector<uint8_t> free_dofs;
  std::vector<int32_t> color;
};

// rows reordered by rperm (new -> old, or null), columns renumbered by ciperm (old -> new, or null), rows re-sorted
static void permute_matrix(const amgx_matrix& M, const int32_t* rperm, const int32_t* ciperm, HostCsr& out) {
  const int64_t n = M.n_rows;
  const int bb = M.br * M.bc;
  out.rowptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) { const int64_t o = rperm ? rperm[i] : i; out.rowptr[i + 1] = out.rowptr[i] + (M.rowptr[o + 1] - M.rowptr[o]); }
  const int64_t nnz = out.rowptr[n];
  out.col.resize((size_t)nnz);
  out.val.resize((size_t)nnz * bb);
  int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
  if (const char* e = std::getenv("OMP_NUM_THREADS")) T = std::max(1, std::min(T, std::atoi(e)));
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n / 8192 + 1));
  auto work = [&](int t) {
    std::vector<std::pair<int32_t, int64_t>> row;
    for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
      const int64_t o = rperm ? rperm[i] : i;
      row.clear();
      for (int64_t k = M.rowptr[o]; k < M.rowptr[o + 1]; ++k) row.emplace_back(ciperm ? ciperm[M.col[k]] : M.col[k], k);
      if (ciperm) std::sort(row.begin(), row.end());
      int64_t w = out.rowptr[i];
      for (const auto& e : row) {
        out.col[w] = e.first;
        std::copy(M.val + e.second * bb, M.val + (e.second + 1) * bb, out.val.begin() + w * bb);
        ++w;
      }
    }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
}

static void use_csr(amgx_matrix& M, const HostCsr& c) { M.rowptr = c.rowptr.data(); M.col = c.col.data(); M.val = c.val.data(); }

// fills pl (mutable copies of the level descriptors) with colour-major versions of the GS levels; store owns the data
static void permute_gs_levels(const amgx_hierarchy_desc* d, std::vector<amgx_level_desc>& pl, std::vector<LevelPerm>& store) {
  const int L = d->n_levels;
  // Measured NON-win (profiles/r01/gs_perm.txt): the colour kernels gain 7-20 %, but the transfers lose far more
  // (P^T gathers its ~30 fine residuals per coarse row from 8 colour blocks: 129 -> 370 us; P: 78 -> 132 us) and the two
  // translation passes cost 125 us: cycle 2.35 -> 2.54 ms at cfg 2.  Off unless AMGX_GS_PERM=1 (kept under test).
  if (!std::getenv("AMGX_GS_PERM")) return;
  for (int l = 0; l < L; ++l) if (pl[l].A.n_rows != pl[l].A.n_cols) return;     // rank-partitioned handle: vectors carry ghosts
  bool any = false;
  for (int l = 0; l + 1 < L; ++l) {
    const amgx_level_desc& s = pl[l];
    if (s.sm_type != AMGX_SM_GS || !s.color || s.n_colors <= 0 || s.A.n_rows < 2) continue;
    const int64_t n = s.A.n_rows;
    LevelPerm& lp = store[l];
    // stable counting sort by colour, rows without a colour (non-free) last
    std::vector<int64_t> start(s.n_colors + 2, 0);
    for (int64_t i = 0; i < n; ++i) { const int c = s.color[i]; if (c >= s.n_colors) throw Err("colour index out of range"); start[(c < 0 ? s.n_colors : c) + 1]++; }
    for (int c = 0; c <= s.n_colors; ++c) start[c + 1] += start[c];
    lp.perm.resize(n); lp.iperm.resize(n);
    for (int64_t i = 0; i < n; ++i) { const int c = s.color[i] < 0 ? s.n_colors : s.color[i]; const int64_t q = start[c]++; lp.perm[q] = (int32_t)i; lp.iperm[i] = (int32_t)q; }
    any = true;
  }
  if (!any) return;
  for (int l = 0; l < L; ++l) {
    amgx_level_desc& s = pl[l];
    LevelPerm& lp = store[l];
    const bool me = !lp.perm.empty();
    const bool nxt = l + 1 < L && !store[l + 1].perm.empty();
    if (me) {
      const int64_t n = s.A.n_rows;
      const int b2 = s.A.br * s.A.br;
      permute_matrix(s.A, lp.perm.data(), lp.iperm.data(), lp.A);
      use_csr(s.A, lp.A);
      if (s.dinv) { lp.dinv.resize((size_t)n * b2); for (int64_t i = 0; i < n; ++i) std::copy(s.dinv + (int64_t)lp.perm[i] * b2, s.dinv + ((int64_t)lp.perm[i] + 1) * b2, lp.dinv.begin() + i * b2); s.dinv = lp.dinv.data(); }
      if (s.free_dofs) { lp.free_dofs.resize(n); for (int64_t i = 0; i < n; ++i) lp.free_dofs[i] = s.free_dofs[lp.perm[i]]; s.free_dofs = lp.free_dofs.data(); }
      if (s.color) { lp.color.resize(n); for (int64_t i = 0; i < n; ++i) lp.color[i] = s.color[lp.perm[i]]; s.color = lp.color.data(); }
    }
    if ((me || nxt) && l + 1 < L && s.P.rowptr) {
      permute_matrix(s.P, me ? lp.perm.data() : nullptr, nxt ? store[l + 1].iperm.data() : nullptr, lp.P);
      use_csr(s.P, lp.P);
      permute_matrix(s.PT, nxt ? store[l + 1].perm.data() : nullptr, me ? lp.iperm.data() : nullptr, lp.PT);
      use_csr(s.PT, lp.PT);
    }
  }
}

}  // namespace amgx
#include "dense_spd.hpp"
namespace amgx {

// ---- collapsed coarse levels (see dense_op_gemv_kernel) -----------------------------------------------------------
// Picks the first level l_c >= 1 from which the sub-cycle is cheaper as one dense GEMV than as its dependent launches,
// forms B column by column with the handle's own kernels (so B is exactly the operator the separate launches apply,
// whatever the smoother form) and stores it row-major.  AMGX_NO_DENSE_TAIL=1 disables, AMGX_DENSE_MAX=<n> caps n.
// first_level: 1 for a handle whose level 0 carries the caller's vectors; 0 for the replicated tail of a rank-partitioned
// hierarchy, where the whole handle may become one GEMV on the gathered vector
static void build_dense_tail(Handle& h, const amgx_hierarchy_desc* d, const amgx_level_desc* levels, int first_level = 1) {
  const int L = d->n_levels;
  if (d->cycle != AMGX_CYCLE_V || L < 2 + first_level || std::getenv("AMGX_NO_DENSE_TAIL")) return;
  int64_t cap = 8192;
  if (const char* e = std::getenv("AMGX_DENSE_MAX")) cap = std::max<int64_t>(0, std::atoll(e));
  for (int l = 0; l < L; ++l) if (h.lev[l].ncols != h.lev[l].n) return;       // rank-partitioned levels are driven stage by stage
  // dependent launches one cycle spends on level m (both directions), ~5 us each
  auto launches = [&](int m) -> double {
    const DevLevel& V = h.lev[m];
    const int k = std::max(1, V.sm_steps) * (V.sm_symm ? 2 : 1);
    if (V.sm_type == AMGX_SM_JACOBI) return h.folded(V) ? 3.0 : 2.0 + 3.0 * k;
    if (V.sm_type == AMGX_SM_BGS) return 3.0 + 2.0 * k * std::max(1, V.bgs.n_colors);
    if (V.gsb.on() || V.bgsb.on()) return 2.0 + 3.0 * k;
    return 3.0 + 2.0 * k * std::max(1, V.gs.n_colors);
  };
  int lc = -1;
  double est = 5.0;                               // the coarse solve
  std::vector<double> est_from(L, 0.0);
  for (int m = L - 2; m >= first_level; --m) { est += 5.0 * launches(m); est_from[m] = est; }
  for (int m = first_level; m <= L - 2; ++m) {
    const int64_t N = h.lev[m].len();
    if (N < 1 || N > cap) continue;
    const double dense_us = 4.0 + 8.0 * (double)N * (double)N / 4.0e6;      // ~4 TB/s on a few hundred workgroups
    if (dense_us < 0.8 * est_from[m]) { lc = m; break; }
  }
  if (lc < first_level) return;
  const int N = (int)h.lev[lc].len();
  const int ld = (N + 1) & ~1;
  DevBuf<double> Bt;
  Bt.alloc((size_t)N * ld);
  h.dense_op.alloc((size_t)N * ld);
  HIPCHK(hipMemsetAsync(Bt.p, 0, (size_t)N * ld * sizeof(double), h.stream));
  HIPCHK(hipMemsetAsync(h.dense_op.p, 0, (size_t)N * ld * sizeof(double), h.stream));
  const int saved_tail = h.tail_level;
  h.tail_level = -1;                              // the sub-cycle runs as separate launches from level lc
  DevLevel& V = h.lev[lc];
  try {
    for (int j = 0; j < N; ++j) {
      hipLaunchKernelGGL(dense_unit_kernel, dim3(Handle::grid_for(N)), dim3(BLOCK), 0, h.stream, (int64_t)N, (int64_t)j, V.rhs.p);
      h.cycle_v(V.x.p, V.rhs.p, lc);                // (dense_level is still -1: separate launches)
      HIPCHK(hipMemcpyAsync(Bt.p + (size_t)j * ld, V.x.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h.stream));
      if ((j & 255) == 255) HIPCHK(hipStreamSynchronize(h.stream));          // bound the depth of the launch queue
    }
    const int tb = (N + 15) / 16;
    hipLaunchKernelGGL(dense_transpose_kernel, dim3(tb, tb), dim3(BLOCK), 0, h.stream, N, ld, Bt.p, h.dense_op.p);
    HIPCHK(hipGetLastError());
    // leave the work vectors of the collapsed levels as create() made them
    for (int m = lc; m < L; ++m) {
      const size_t len = (size_t)std::max<int64_t>(1, h.lev[m].ext_len());
      for (double* v : {h.lev[m].x.p, h.lev[m].rhs.p, h.lev[m].res.p, h.lev[m].tmp.p}) HIPCHK(hipMemsetAsync(v, 0, len * sizeof(double), h.stream));
    }
    HIPCHK(hipStreamSynchronize(h.stream));
  } catch (...) { h.tail_level = saved_tail; throw; }
  h.tail_level = saved_tail;
  h.dense_level = lc;
  h.dense_n = N;
  h.dense_ld = ld;
}

// dense_first: first level that may be collapsed into the dense operator (see build_dense_tail); < 0: never
static Handle* create(const amgx_hierarchy_desc* d, int dense_first = 1) {
  if (!d || d->n_levels < 1 || !d->levels) throw Err("amgx_create: empty hierarchy descriptor");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) throw Err("amgx_create: no HIP device available (the apply path has no CPU fallback)");
  if (d->device < 0 || d->device >= ndev) throw Err("amgx_create: device ordinal out of range");
  HIPCHK(hipSetDevice(d->device));
  auto h = std::make_unique<Handle>();
  h->device = d->device;
  h->cycle = d->cycle;
  h->clev = d->clev;
  h->use_graph = d->use_graph != 0;
  h->ep_nt = (std::getenv("AMGX_NO_EP_NT") ? 0 : EPF_NT) | (std::getenv("AMGX_NO_EP_HOIST") ? 0 : EPF_HOIST);   // A/B: -0.4 % cycle time (profiles/r01/restrict_fused.txt)
  if (d->cycle < 0 || d->cycle > 2) throw Err("amgx_create: unknown cycle");
  HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  h->lev.resize(d->n_levels);
  std::vector<amgx_level_desc> pl(d->levels, d->levels + d->n_levels);
  std::vector<LevelPerm> pstore(d->n_levels);
  permute_gs_levels(d, pl, pstore);
  SetupClock clk;
  clk.lap("renumbering of Gauss-Seidel levels");
  h->perm.resize(d->n_levels);
  for (int l = 0; l < d->n_levels; ++l) if (!pstore[l].perm.empty()) h->perm[l].upload(pstore[l].perm);
  const amgx_level_desc* levels = pl.data();
  for (int l = 0; l < d->n_levels; ++l) {
    const amgx_level_desc& s = levels[l];
    DevLevel& L = h->lev[l];
    // n_cols > n_rows: the trailing columns are ghost entries of a rank-partitioned level (filled by the caller's
    // halo exchange before every operation that gathers from them)
    if (s.A.n_cols < s.A.n_rows || s.A.br != s.A.bc) throw Err("level matrix must have n_cols >= n_rows and square blocks");
    L.n = s.A.n_rows; L.ncols = s.A.n_cols; L.bs = s.A.br;
    L.sm_type = s.sm_type; L.omega = s.omega; L.sm_steps = s.sm_steps; L.sm_symm = s.sm_symm;
    if (s.sm_type != AMGX_SM_JACOBI && s.sm_type != AMGX_SM_GS && s.sm_type != AMGX_SM_BGS) throw Err("unknown smoother type");
    const bool last = (l + 1 == d->n_levels);
    check_matrix(s.A, "A");            // (before anything reads rowptr[n_rows]: a NULL / garbage descriptor is an error, not a crash)
    // (declared BEFORE the task pool: its worker lambdas capture these by reference, and an exception between tasks.run() and
    //  tasks.wait() must join the workers -- ~SetupTasks -- before the buffers they read are freed)
    DevCsrSrc csrA;
    DbDiagInfo diagA;
    DevBcsrSrc csrB;
    SetupTasks tasks(d->device);
    // big scalar levels: the CSR arrays go to the device once and kernels write the images of A, A' and Q there (devbuild.hpp)
    const bool dev_images = dev_images_wanted(s.A);
    const bool verify_images = dev_images && std::getenv("AMGX_VERIFY_IMAGES") != nullptr;
    if (dev_images) {
      check_matrix(s.A, "A");
      csrA.upload(s.A);
      if (s.dinv) L.dinv.upload(s.dinv, (size_t)(last ? L.n : L.ncols) * L.bs * L.bs);
      diagA = dev_diag_check(csrA, (s.dinv && s.A.n_rows <= s.A.n_cols) ? L.dinv.p : nullptr);
      clk.lap("CSR of A to the device", l);
    }
    // big square-block levels: the block-CSR arrays go to the device once, the BSELL images (A; the block-hybrid Gauss-Seidel
    // images) are gathered there
    const bool keep_csr_A = s.sm_type == AMGX_SM_GS && s.A.br > 1 && s.gs_block_rows == 0;
    const bool dev_bsell = dev_bsell_wanted(s.A) && s.A.n_rows == s.A.n_cols && !keep_csr_A;
    const bool verify_bsell = dev_bsell && std::getenv("AMGX_VERIFY_IMAGES") != nullptr;
    if (dev_bsell) {
      check_matrix(s.A, "A");
      csrB.upload(s.A);
      clk.lap("block CSR of A to the device", l);
    }
    // block GS walks the CSR arrays of A, so keep A in CSR there
    tasks.run([&] {
      if (dev_bsell) {
        if (dev_build_bsell(csrB, nullptr, 0, BB_ALL, DbBgsbMaps(), 0, 1.30, L.A)) {
          L.A.n_rows = s.A.n_rows; L.A.n_cols = s.A.n_cols; L.A.br = s.A.br; L.A.bc = s.A.bc;
          L.A.nnz = s.A.rowptr[s.A.n_rows];
          L.A.lanes = pick_lanes(L.A.n_rows ? (double)L.A.nnz / (double)L.A.n_rows : 0.0);
          if (verify_bsell) { DevMatrix H; upload_matrix(s.A, H, "A", true, true, false); if (H.lanes != L.A.lanes || H.nnz != L.A.nnz) throw Err("AMGX_VERIFY_IMAGES: A: descriptors differ"); verify_same_bsell(L.A, H, "A"); }
          return;
        }
        L.A = DevMatrix();
      }
      if (dev_images && dev_upload_matrix(csrA, L.A, true, 1.35, 0, &diagA)) {
        if (verify_images) { DevMatrix H; upload_matrix(s.A, H, "A", true, true, false); verify_same_image(L.A, H, "A"); }
        return;
      }
      if (verify_images) {
        DevMatrix H;
        upload_matrix(s.A, H, "A", true, true, false);
        if (H.fmt == FMT_SELL && H.lanes == 1) throw Err("AMGX_VERIFY_IMAGES: the device builder declined A where the host builder forms a SELL image");
      }
      upload_matrix(s.A, L.A, "A", true, true, s.sm_type == AMGX_SM_GS && s.A.br > 1 && s.gs_block_rows == 0);
    }, "A");
    if (!last) {
      const amgx_level_desc& c = levels[l + 1];
      if (s.P.n_rows != s.A.n_rows || s.P.n_cols > c.A.n_cols || s.P.n_cols < c.A.n_rows || s.P.br != s.A.br || s.P.bc != c.A.br)
        throw Err("P does not match the level matrices");
      if (s.PT.n_rows != s.P.n_cols || s.PT.n_cols != s.P.n_rows || s.PT.br != s.P.bc || s.PT.bc != s.P.br)
        throw Err("PT does not match P");
      if (!s.dinv) throw Err("dinv missing");
      tasks.run([&] {
        upload_matrix(s.P, L.P, "P", true, false, false, 1.35, s.P.br == 1 && s.P.bc == 1 ? -SELL_WIN : 0, nullptr, 1);
        upload_matrix(s.PT, L.PT, "PT", true, false, false, 1.35, 0, nullptr, 2);
        // big scalar levels restrict through the column-blocked form (the P^T gather is TA/L2-bound there)
        // Measured non-win (profiles/r01/restrict_blocked.txt): 121 + 22 us vs 134 us for the P^T gather at cfg 2,
        // so the blocked form is OFF unless AMGX_RESTRICT_MIN_ROWS asks for it (kept for the fused-residual plan).
        int64_t min_rows = INT64_MAX;
        if (const char* e = std::getenv("AMGX_RESTRICT_MIN_ROWS")) min_rows = std::atoll(e);
        if (s.P.br == 1 && s.P.bc == 1 && s.P.n_rows >= min_rows && s.P.rowptr[s.P.n_rows] < (int64_t)2147483647)
          build_restrict(s.P, L.R);
      }, "P, PT");
      tasks.run([&] {
        if (!dev_images) L.dinv.upload(s.dinv, (size_t)L.ncols * L.bs * L.bs);
        if (s.sm_type == AMGX_SM_GS && s.gs_block_rows > 0 && s.A.br > 1) {
          build_bgsb(s, L, dev_bsell ? &csrB : nullptr);
          if (verify_bsell) {
            DevLevel H;
            H.n = L.n; H.ncols = L.ncols; H.bs = L.bs;
            build_bgsb(s, H, nullptr);
            const DevBGSB &x = L.bgsb, &y = H.bgsb;
            if (x.BB != y.BB || x.n_blocks != y.n_blocks || x.n_colors != y.n_colors || x.has_split != y.has_split) throw Err("AMGX_VERIFY_IMAGES: block-hybrid Gauss-Seidel (blocks): the descriptors differ");
            verify_same_bsell(x.off, y.off, "block-hybrid Gauss-Seidel: off");
            verify_same_bsell(x.in, y.in, "block-hybrid Gauss-Seidel: in");
            verify_same_bsell(x.upin, y.upin, "block-hybrid Gauss-Seidel: upin");
            if (x.has_split) verify_same_bsell(x.rest, y.rest, "block-hybrid Gauss-Seidel: rest");
            if (x.bc != y.bc || x.n_bcolors != y.n_bcolors) throw Err("AMGX_VERIFY_IMAGES: block-coloured Gauss-Seidel: the descriptors differ");
            if (x.bc && x.has_split) verify_same_bsell(x.offlo, y.offlo, "block-coloured Gauss-Seidel: offlo");
          }
        }
        else if (s.sm_type == AMGX_SM_GS && s.gs_block_rows > 0) {
          build_gsb(s, L, &s.P, dev_images ? &csrA : nullptr);
          if (verify_images) {
            DevLevel H;
            H.n = L.n; H.ncols = L.ncols; H.bs = L.bs;
            build_gsb(s, H, &s.P, nullptr);
            const DevGSB &x = L.gsb, &y = H.gsb;
            if (x.B != y.B || x.G != y.G || x.TH != y.TH || x.n_blocks != y.n_blocks || x.n_colors != y.n_colors || x.lowin_maxw != y.lowin_maxw || x.full_maxw != y.full_maxw ||
                x.has_split != y.has_split) throw Err("AMGX_VERIFY_IMAGES: block-hybrid Gauss-Seidel: the descriptors differ");
            const size_t nsl = (size_t)((int64_t)x.n_blocks * x.B / (WAVE / x.G));
            verify_same_sell(x.full, y.full, nsl, 0, "block-hybrid Gauss-Seidel: A");
            if (x.has_split) {
              verify_same_sell(x.lowin, y.lowin, nsl, 0, "block-hybrid Gauss-Seidel: lower part");
              verify_same_image(x.rest, y.rest, "block-hybrid Gauss-Seidel: rest");
              const auto cx = db_download(x.cvec, (size_t)L.n), cy = db_download(y.cvec, (size_t)L.n);
              if (std::memcmp(cx.data(), cy.data(), (size_t)L.n * sizeof(double)) != 0) throw Err("AMGX_VERIFY_IMAGES: block-hybrid Gauss-Seidel: cvec differs");
            }
          }
        }
        else if (s.sm_type == AMGX_SM_GS) build_gs(s, L);
        if (s.sm_type == AMGX_SM_BGS) build_bgs(s, L);
      }, "smoother data");
      if (s.sm_type == AMGX_SM_JACOBI && s.A.br == 1 && s.sm_steps <= 1 && !s.sm_symm) {
        tasks.run([&] {
        // long-row levels (>= 1) of a reference-shaped hierarchy: the "local window" image for the fused down kernel
        auto lw_image = [&]() -> bool {
          const int64_t nnzA = s.A.rowptr[s.A.n_rows];
          const double avgA = s.A.n_rows ? (double)nnzA / (double)s.A.n_rows : 0.0;
          int64_t lw_min_rows = 100000;
          if (const char* e = std::getenv("AMGX_LW_MIN_ROWS")) lw_min_rows = std::atoll(e);
          // (rank-partitioned levels too: the window of an interior chunk holds owned columns only, ghost columns are just columns)
          if (l == 0 || avgA < 24.0 || s.A.n_rows < lw_min_rows || std::getenv("AMGX_NO_LW") ||
              s.P.br != 1 || s.P.bc != 1 || s.P.rowptr[s.P.n_rows] >= (int64_t)2147483647 || std::getenv("AMGX_NO_FUSED_RESTRICT")) return false;
          // two lanes per row (256-row chunks); levels whose 256-row chunks touch too many columns: four lanes (128-row chunks)
          int G = 0;
          std::unique_ptr<double[]> sv;
          auto host_lw = [&](int g, DevMatrix& M, DevBuf<int32_t>& cp, DevBuf<int32_t>& cc) {
            if (!sv) {
              sv.reset(new double[(size_t)std::max<int64_t>(1, nnzA)]);
              par_for(nnzA, [&](int64_t k0, int64_t k1, int) { for (int64_t k = k0; k < k1; ++k) sv[k] = s.A.val[k] * (s.omega * s.dinv[s.A.col[k]]); }, 1 << 16);
            }
            return build_sell_lw(s.A, sv.get(), g, M, cp, cc);
          };
          // (device: window lists by a bitmap in LDS, devbuild.hpp dev_build_lw; AMGX_HOST_LW=1 keeps the host builder)
          const bool dev_lw = dev_images && !std::getenv("AMGX_HOST_LW") && !std::getenv("AMGX_HOST_IMAGES");
          for (int g : {2, 4}) {
            bool ok = false;
            if (dev_lw) {
              int64_t cap = LW_CAP;
              const char* tcap = std::getenv("AMGX_LW_TEST_CAP");
              if (tcap) cap = std::min<int64_t>(cap, std::atoll(tcap));
              ok = dev_build_lw(csrA, false, g, cap, tcap != nullptr, L.dinv.p, s.omega, L.ApreLW, L.lw_cptr, L.lw_ccol);
              if (ok && verify_images) {
                DevMatrix H; DevBuf<int32_t> hp, hc;
                if (!host_lw(g, H, hp, hc)) throw Err("AMGX_VERIFY_IMAGES: A' (local window): the host builder declines what the device builder forms");
                verify_same_lw(L.ApreLW, L.lw_cptr, L.lw_ccol, H, hp, hc, "A' (local window)");
              }
              if (!ok) { L.ApreLW = DevMatrix(); L.lw_cptr.release(); L.lw_ccol.release(); }
            }
            if (!ok) ok = host_lw(g, L.ApreLW, L.lw_cptr, L.lw_ccol);
            if (ok) { G = g; break; }
            L.ApreLW = DevMatrix();
          }
          if (!G) return false;
          L.fused_block = 512;
          build_restrict(s.P, L.RF, 512 / G, 4 * 512, 512);
          if (L.RF.empty()) { L.ApreLW = DevMatrix(); L.lw_cptr.release(); L.lw_ccol.release(); return false; }
          return true;
        };
        auto fused_restrict = [&] {
          if (lw_image()) return;
          // fused pre-smoothing + restriction when A' is in the one-thread-per-row SELL form (big levels).
          // Same-process A/B with 4 instances per variant (profiles/r01/restrict_fused.txt): 1-3 % faster cycle than the
          // separate pre-smoothing + P^T gather kernels, and r is never written to HBM.  AMGX_NO_FUSED_RESTRICT=1 disables it.
          const int G = L.Apre.lanes;
          if (L.Apre.fmt == FMT_SELL && L.Apre.sell.win == SELL_WIN && G == 1 && SELL_WIN == 512 && s.P.br == 1 && s.P.bc == 1 &&
              s.P.rowptr[s.P.n_rows] < (int64_t)2147483647 && !std::getenv("AMGX_NO_FUSED_RESTRICT"))
          {
            L.fused_block = SELL_WIN;              // windowed A': a chunk = a window (sell_win_pre_restrict_kernel)
            build_restrict(s.P, L.RF, SELL_WIN, 6 * SELL_WIN, SELL_WIN);
          }
          else if (L.Apre.fmt == FMT_SELL && !L.Apre.sell.win && (G == 1 || G == 2 || G == 4 || G == 8) && s.P.br == 1 && s.P.bc == 1 &&
              s.P.rowptr[s.P.n_rows] < (int64_t)2147483647 && !std::getenv("AMGX_NO_FUSED_RESTRICT") && !(G > 1 && std::getenv("AMGX_NO_FUSED_RESTRICT_MULTI")))
          {
            L.fused_block = 512;       // same-process A/B: 512 < 1024 (epilogues of more, smaller workgroups overlap better)
            if (const char* e = std::getenv("AMGX_FUSED_BLOCK")) { const int v = std::atoi(e); L.fused_block = (v == 256 || v == 1024) ? v : 512; }
            if (G > 1) L.fused_block = 512;        // (several lanes per row: the chunk holds 512 / G rows)
            // big square one-thread-per-row levels: compact chunks (cluster_slices) -- fewer partial sums per coarse row
            int64_t cc_min = 200000;
            if (const char* e = std::getenv("AMGX_COMPACT_CHUNKS_MIN_ROWS")) cc_min = std::atoll(e);
            // (not on a handle that a rank-partitioned driver runs stage by stage -- dense_first < 0 --: its launches cover interior and
            //  boundary chunk RANGES, which only consecutive chunks have; a rank without ghost columns, e.g. world size 1, has a square level)
            if (G == 1 && dense_first >= 0 && s.A.n_rows == s.A.n_cols && s.A.n_rows >= cc_min && !std::getenv("AMGX_NO_COMPACT_CHUNKS")) {
              const std::vector<int32_t> sl = cluster_slices(s.P, L.fused_block / WAVE);
              build_restrict(s.P, L.RF, L.fused_block, 6 * L.fused_block, L.fused_block, &sl);
            } else
            build_restrict(s.P, L.RF, L.fused_block / G, 6 * L.fused_block, L.fused_block);
          }
        };
        // (device builder: A' = A diag(omega Dinv) from the CSR of A that is already there; the diagonal slot carries omega*Dinv_i
        //  under the same conditions as below)
        const bool dev_wdiag = s.omega != 0.0 && !std::getenv("AMGX_NO_WDIAG") && diagA.plain;
        // levels >= 1 of a reference-shaped hierarchy have ragged long rows (plain slices pad 20 % at the 1.24 M-row level of cfg 2,
        // length-sorted windows 3.6 %).  Measured NON-win (profiles/r04/l1_experiments.txt): the windowed image with its fused kernel
        // (sell_win_pre_restrict_kernel) runs that level in 256 us against 215 us -- these levels are bound by the scattered gathers
        // of x, and sorting the rows of a window by length puts unrelated rows into neighbouring lanes.  Opt-in: AMGX_APRE_WINDOW=1.
        const int apre_win = (l >= 1 && s.A.n_rows == s.A.n_cols && std::getenv("AMGX_APRE_WINDOW")) ? SELL_WIN : 0;
        if (dev_images && !verify_images && dev_upload_matrix(csrA, L.Apre, true, 1.35, apre_win, &diagA, L.dinv.p, s.omega, dev_wdiag ? L.dinv.p : nullptr)) {
          fused_restrict();
          return;
        }
        // column-scaled image for the fused pre-smoothing pass (memory for bandwidth: one more copy of A)
        const int64_t nnz = s.A.rowptr[s.A.n_rows];
        std::unique_ptr<double[]> sv(new double[(size_t)std::max<int64_t>(1, nnz)]);      // (uninitialised: every entry is written below)
        // (rank-partitioned levels: dinv must cover the ghost columns too, i.e. n_cols entries)
        par_for(nnz, [&](int64_t k0, int64_t k1, int) { for (int64_t k = k0; k < k1; ++k) sv[k] = s.A.val[k] * (s.omega * s.dinv[s.A.col[k]]); }, 1 << 16);
        amgx_matrix As = s.A;
        As.val = sv.get();
        // one-thread-per-row form: the diagonal slot carries omega*Dinv_i (SellMat::wdiag), the epilogue then needs no
        // dinv stream (80 MB per pass at cfg 2); AMGX_NO_WDIAG=1 keeps A'_ii there
        std::vector<double> wdv;
        if (s.omega != 0.0 && !std::getenv("AMGX_NO_WDIAG")) {
          // the epilogue re-inserts A'_ii b_i as omega*b_i (or 0 where dinv_i = 0): valid iff dinv is the plain inverse diagonal
          std::vector<char> notplain(setup_threads(), 0);
          par_for(s.A.n_rows, [&](int64_t i0, int64_t i1, int t) {
            for (int64_t i = i0; i < i1; ++i) {
              if (s.dinv[i] == 0.0) continue;
              double aii = 0.0;
              for (int64_t k = s.A.rowptr[i]; k < s.A.rowptr[i + 1]; ++k) if (s.A.col[k] == i) { aii = s.A.val[k]; break; }
              if (!(std::fabs(s.dinv[i] * aii - 1.0) < 1e-13)) { notplain[t] = 1; break; }
            }
          });
          bool plain = true;
          for (char cc : notplain) if (cc) plain = false;
          if (plain) {
            wdv.resize((size_t)s.A.n_rows);
            par_for(s.A.n_rows, [&](int64_t i0, int64_t i1, int) { for (int64_t i = i0; i < i1; ++i) wdv[i] = s.omega * s.dinv[i]; }, 1 << 16);
          }
        }
        if (dev_images && verify_images) {
          DevMatrix H;
          upload_matrix(As, H, "A (pre-smoothing image)", true, true, false, 1.35, apre_win, wdv.empty() ? nullptr : wdv.data());
          if (dev_upload_matrix(csrA, L.Apre, true, 1.35, apre_win, &diagA, L.dinv.p, s.omega, dev_wdiag ? L.dinv.p : nullptr)) verify_same_image(L.Apre, H, "A'");
          else if (H.fmt == FMT_SELL && H.lanes == 1) throw Err("AMGX_VERIFY_IMAGES: the device builder declined A' where the host builder forms a SELL image");
          else L.Apre = std::move(H);
        } else
          upload_matrix(As, L.Apre, "A (pre-smoothing image)", true, true, false, 1.35, apre_win, wdv.empty() ? nullptr : wdv.data());
        fused_restrict();
        }, "A' + fused restriction");
        // post-smoothing folded into the prolongation (V-cycle).  Square levels: Q is built here.  Rank-partitioned
        // levels: Q needs the P rows of the ghost vertices, so the caller supplies it (amgx_level_desc.Q) and drives
        // the level through amgx_cycle_down / amgx_cycle_up.
        auto qlw_wanted = [](int64_t rows) {
          int64_t mn = 100000;
          if (const char* e = std::getenv("AMGX_LW_MIN_ROWS")) mn = std::atoll(e);
          return rows >= mn && !std::getenv("AMGX_NO_LW") && !std::getenv("AMGX_NO_QLW");
        };
        if (d->cycle == AMGX_CYCLE_V && s.P.br == 1 && s.P.bc == 1 && !std::getenv("AMGX_NO_FOLD")) {
          double qpad = 1.6;
          if (const char* e = std::getenv("AMGX_Q_MAX_PAD")) qpad = std::atof(e);
          if (s.Q.rowptr) {
            if (s.Q.n_rows != s.A.n_rows || s.Q.br != 1 || s.Q.bc != 1 || s.Q.n_cols < c.A.n_rows || s.Q.n_cols > c.A.n_cols)
              throw Err("Q does not match the level matrices");
            if (s.Q.rowptr[s.Q.n_rows] >= (int64_t)2147483647) throw Err("Q: too many entries");
            tasks.run([&, qpad] {
              upload_matrix(s.Q, L.Q, "Q (folded post-smoothing prolongation)", true, false, false, qpad, SELL_WIN);
              // (rank-partitioned level: the caller's Q, columns [owned | ghost] of the coarse level)
              if (qlw_wanted(s.A.n_rows) &&
                  !build_sell_lw_windowed(s.Q.n_rows, s.Q.n_cols, s.Q.rowptr, s.Q.col, s.Q.val, L.QLW, L.qlw_cptr, L.qlw_ccol)) L.QLW = DevMatrix();
            });
          } else if (s.A.n_rows == s.A.n_cols && s.P.n_cols == c.A.n_rows) {
            tasks.run([&, qpad] {
              // (device: the sparse product and the windowed image of its result, devbuild.hpp)
              if (dev_images) {
                check_matrix(s.P, "P");
                DevCsrSrc csrP, csrQ;
                csrP.upload(s.P);
                if (dev_fold_prolongation(csrA, csrP, L.dinv.p, s.omega, csrQ) && dev_upload_matrix(csrQ, L.Q, false, qpad, SELL_WIN, nullptr)) {
                  bool qlw_done = false;
                  if (qlw_wanted(s.A.n_rows) && !std::getenv("AMGX_HOST_LW")) {
                    int64_t cap = QW_CAP;
                    const char* tcap = std::getenv("AMGX_LW_TEST_CAP");
                    if (tcap) cap = std::min<int64_t>(cap, std::max<int64_t>(8, std::atoll(tcap) / 4));
                    qlw_done = dev_build_lw(csrQ, true, 1, cap, tcap != nullptr, nullptr, 0.0, L.QLW, L.qlw_cptr, L.qlw_ccol);
                    if (!qlw_done) { L.QLW = DevMatrix(); L.qlw_cptr.release(); L.qlw_ccol.release(); }
                    else if (verify_images) {
                      std::vector<int64_t> rp = db_download(csrQ.rowptr, (size_t)s.A.n_rows + 1);
                      std::vector<int32_t> cc = db_download(csrQ.col, (size_t)std::max<int64_t>(1, csrQ.nnz));
                      std::vector<double> vv = db_download(csrQ.val, (size_t)std::max<int64_t>(1, csrQ.nnz));
                      DevMatrix H; DevBuf<int32_t> hp, hc;
                      if (!build_sell_lw_windowed(s.A.n_rows, csrQ.n_cols, rp.data(), cc.data(), vv.data(), H, hp, hc))
                        throw Err("AMGX_VERIFY_IMAGES: Q (local window): the host builder declines what the device builder forms");
                      verify_same_lw(L.QLW, L.qlw_cptr, L.qlw_ccol, H, hp, hc, "Q (local window)");
                    }
                  }
                  if (qlw_wanted(s.A.n_rows) && !qlw_done) {
                    SetupClock qclk;
                    std::vector<int64_t> rp = db_download(csrQ.rowptr, (size_t)s.A.n_rows + 1);
                    std::vector<int32_t> cc = db_download(csrQ.col, (size_t)std::max<int64_t>(1, csrQ.nnz));
                    std::vector<double> vv = db_download(csrQ.val, (size_t)std::max<int64_t>(1, csrQ.nnz));
                    qclk.lap("  lw-win: download of Q", l);
                    if (!build_sell_lw_windowed(s.A.n_rows, csrQ.n_cols, rp.data(), cc.data(), vv.data(), L.QLW, L.qlw_cptr, L.qlw_ccol)) L.QLW = DevMatrix();
                  }
                  if (verify_images) {
                    HostCsr q;
                    fold_prolongation(s.A, s.P, s.dinv, s.omega, q);
                    if ((int64_t)q.rowptr[s.A.n_rows] != csrQ.nnz) throw Err("AMGX_VERIFY_IMAGES: Q: different numbers of entries");
                    amgx_matrix Qm = s.P;
                    Qm.rowptr = q.rowptr.data(); Qm.col = q.col.data(); Qm.val = q.val.data();
                    DevMatrix H;
                    upload_matrix(Qm, H, "Q (folded post-smoothing prolongation)", true, false, false, qpad, SELL_WIN);
                    verify_same_image(L.Q, H, "Q");
                  }
                  return;
                }
                L.Q = DevMatrix();
                if (verify_images) std::fprintf(stderr, "[amgx_create] AMGX_VERIFY_IMAGES: level %d: Q is left to the host builder\n", l);
              }
              HostCsr q;
              fold_prolongation(s.A, s.P, s.dinv, s.omega, q);
              if (q.rowptr[s.A.n_rows] < (int64_t)2147483647) {
                amgx_matrix Qm = s.P;
                Qm.rowptr = q.rowptr.data(); Qm.col = q.col.data(); Qm.val = q.val.data();
                upload_matrix(Qm, L.Q, "Q (folded post-smoothing prolongation)", true, false, false, qpad, SELL_WIN);
                if (qlw_wanted(s.A.n_rows) &&
                    !build_sell_lw_windowed(s.A.n_rows, s.P.n_cols, q.rowptr.data(), q.col.data(), q.val.data(), L.QLW, L.qlw_cptr, L.qlw_ccol)) L.QLW = DevMatrix();
              }
            }, "Q = (I - w Dinv A) P");
          }
        }
      }
      // block Jacobi levels of the V-cycle: the same fold in block form (Q has the block shape of P)
      if (s.sm_type == AMGX_SM_JACOBI && s.A.br > 1 && s.sm_steps <= 1 && !s.sm_symm && d->cycle == AMGX_CYCLE_V &&
          s.A.n_rows == s.A.n_cols && s.P.n_cols == c.A.n_rows && s.P.br == s.A.br && !std::getenv("AMGX_NO_FOLD") &&
          !std::getenv("AMGX_NO_BLOCK_FOLD"))
      {
        tasks.run([&] {
        HostCsr q;
        fold_prolongation(s.A, s.P, s.dinv, s.omega, q);
        // Fold only where it pays: the way up then streams Q instead of A + P (+ the round trip of x + P x_c), but
        // rectangular-block Q runs through the CSR block kernels (~4.5 TB/s) while square-block A streams as BSELL
        // (~6.5 TB/s).  Measured at the cfg 3 shapes (profiles/r01/block_fold.txt): 3x3 fine level with 3x6 blocks in
        // P: Q has 10.8 blocks/row = 1.6 GB vs A + P = 1.65 GB -> literal is 55 us faster; 6x6 levels: Q = 0.36 GB
        // vs 1.04 GB -> folded is 100 us faster.
        auto bytes = [](int64_t nnz, int br, int bc) { return (double)nnz * (8.0 * br * bc + 4.0); };
        const double bq = bytes(q.rowptr[s.A.n_rows], s.P.br, s.P.bc);
        const double blit = bytes(s.A.rowptr[s.A.n_rows], s.A.br, s.A.bc) + bytes(s.P.rowptr[s.P.n_rows], s.P.br, s.P.bc);
        if (q.rowptr[s.A.n_rows] < (int64_t)2147483647 && bq < 0.7 * blit) {
          amgx_matrix Qm = s.P;
          Qm.rowptr = q.rowptr.data(); Qm.col = q.col.data(); Qm.val = q.val.data();
          upload_matrix(Qm, L.Q, "Q (folded post-smoothing prolongation)");
        }
        });
      }
      tasks.wait();
      clk.lap("level images (A, P, P^T, smoother data, A', Q: concurrent host tasks)", l);
    } else if (s.dinv) {
      tasks.wait();
      L.dinv.upload(s.dinv, (size_t)L.n * L.bs * L.bs);
      if (s.sm_type == AMGX_SM_GS && s.color && s.gs_block_rows > 0 && s.A.br > 1) build_bgsb(s, L);
      else if (s.sm_type == AMGX_SM_GS && s.color && s.gs_block_rows > 0) build_gsb(s, L, nullptr);
      else if (s.sm_type == AMGX_SM_GS && s.color) build_gs(s, L);
      if (s.sm_type == AMGX_SM_BGS && s.bgs_n_blocks > 0) build_bgs(s, L);
    }
    tasks.wait();
    {
      // workgroup -> rows mapping of the streaming kernels on this level (SellMat::xcd)
      int mode = 1;
      if (const char* e = std::getenv("AMGX_XCD")) mode = std::atoi(e);
      const double avg = s.A.n_rows ? (double)s.A.rowptr[s.A.n_rows] / (double)s.A.n_rows : 0.0;
      const int on = mode >= 2 || (mode == 1 && avg >= 24.0 && s.A.n_rows >= 200000);
      L.A.sell.xcd = L.Apre.sell.xcd = L.Q.sell.xcd = L.gsb.rest.sell.xcd = on;
      if (mode >= 3) L.P.sell.xcd = L.PT.sell.xcd = on;
    }
    const size_t len = (size_t)std::max<int64_t>(1, L.ext_len());
    L.x.alloc(len); L.rhs.alloc(len); L.res.alloc(len); L.tmp.alloc(len);
    HIPCHK(hipMemset(L.x.p, 0, len * sizeof(double)));
    HIPCHK(hipMemset(L.rhs.p, 0, len * sizeof(double)));
    HIPCHK(hipMemset(L.res.p, 0, len * sizeof(double)));
    HIPCHK(hipMemset(L.tmp.p, 0, len * sizeof(double)));
  }
  if (d->clev == AMGX_CLEV_INV) {
    const DevLevel& L = h->lev.back();
    if (d->coarse_n != L.len()) throw Err("clev = inv: coarse_n does not match the coarsest level");
    h->coarse_n = d->coarse_n;
    if (d->coarse_inv) {
      h->coarse_ld = d->coarse_n;
      h->coarse_inv.upload(d->coarse_inv, (size_t)d->coarse_n * d->coarse_n);
    } else {
      // no inverse handed over (the host setup stops at 4096 unknowns): invert the coarsest matrix on its free dofs here
      // (dense_spd.hpp: blocked Gauss-Jordan, trailing updates on the matrix cores)
      const amgx_level_desc& s = levels[d->n_levels - 1];
      int64_t cap = 16384;
      if (const char* e = std::getenv("AMGX_COARSE_DENSE_MAX")) cap = std::atoll(e);
      if (s.A.n_rows != s.A.n_cols) throw Err("clev = inv: the coarsest level of a rank-partitioned hierarchy cannot be inverted locally");
      if (d->coarse_n > cap) throw Err("clev = inv: coarsest level has " + std::to_string(d->coarse_n) + " unknowns, more than AMGX_COARSE_DENSE_MAX = " +
                                       std::to_string(cap) + " (a dense inverse would stream " + std::to_string(8 * d->coarse_n * d->coarse_n / 1000000) + " MB per application)");
      const int64_t n = d->coarse_n, npad = (n + GJ_T - 1) / GJ_T * GJ_T;
      const int bs = s.A.br;
      struct { DevBuf<int32_t> rowptr, col; DevBuf<double> val; } cA;       // (DevCsr holds scalar matrices only)
      {
        const int64_t nnzc = s.A.rowptr[s.A.n_rows];
        std::vector<int32_t> rp((size_t)s.A.n_rows + 1);
        for (int64_t i = 0; i <= s.A.n_rows; ++i) rp[i] = (int32_t)s.A.rowptr[i];
        cA.rowptr.upload(rp);
        cA.col.upload(s.A.col, (size_t)nnzc);
        cA.val.upload(s.A.val, (size_t)nnzc * bs * bs);
      }
      DevBuf<uint8_t> fr;
      if (s.free_dofs) fr.upload(s.free_dofs, (size_t)s.A.n_rows);
      h->coarse_inv.alloc((size_t)npad * npad);
      h->coarse_ld = npad;
      hipLaunchKernelGGL(gj_zero_kernel, dim3(Handle::grid_for(npad * npad)), dim3(BLOCK), 0, h->stream, npad * npad, h->coarse_inv.p);
      hipLaunchKernelGGL(gj_scatter_kernel, dim3(Handle::grid_for(s.A.n_rows)), dim3(BLOCK), 0, h->stream, s.A.n_rows, bs, cA.rowptr.p, cA.col.p, cA.val.p,
                         fr.p, npad, h->coarse_inv.p);
      hipLaunchKernelGGL(gj_fix_diag_kernel, dim3(Handle::grid_for(npad)), dim3(BLOCK), 0, h->stream, npad, n, bs, fr.p, npad, 1.0, h->coarse_inv.p);
      HIPCHK(hipGetLastError());
      h->coarse_pivot = dense_spd_inverse(h->coarse_inv.p, npad, npad, h->stream);
      if (!(h->coarse_pivot > 1e-14)) throw Err("clev = inv: the coarsest matrix is not positive definite on its free dofs (pivot ratio " +
                                                std::to_string(h->coarse_pivot) + "); use clev = none or a smaller coarsest level");
      hipLaunchKernelGGL(gj_fix_diag_kernel, dim3(Handle::grid_for(npad)), dim3(BLOCK), 0, h->stream, npad, n, bs, fr.p, npad, 0.0, h->coarse_inv.p);
      HIPCHK(hipStreamSynchronize(h->stream));
    }
  }
  // ---- single-workgroup coarse tail (V-cycle, plain scalar smoothers, exact coarse solve, square levels) ----------
  {
    const int L = d->n_levels;
    int T = -1;
    if (d->cycle == AMGX_CYCLE_V && d->clev == AMGX_CLEV_INV && L >= 2 && !std::getenv("AMGX_NO_TAIL_KERNEL")) {
      T = L - 1;
      while (T - 1 >= 1) {
        const amgx_level_desc& s = levels[T - 1];
        const int64_t cap = s.sm_type == AMGX_SM_GS ? TAIL_MAX_ROWS_GS : TAIL_MAX_ROWS;
        const bool ok = s.A.br == 1 && s.A.n_rows == s.A.n_cols && s.A.n_rows <= cap &&
                        (s.sm_type == AMGX_SM_JACOBI || (s.sm_type == AMGX_SM_GS && s.color && s.n_colors > 0 && s.gs_block_rows == 0)) &&
                        s.sm_steps <= 1 && !s.sm_symm && s.P.br == 1 && s.P.bc == 1 && h->coarse_n <= 512 && h->coarse_ld == h->coarse_n;
        if (!ok) break;
        --T;
      }
      if (T > L - 2) T = -1;                  // no smoothed level qualifies
    }
    if (T > 0) {
      std::vector<TailOp> prog;
      const EpArgs none{nullptr, nullptr, nullptr, 0.0, nullptr, 0};
      auto spmv = [&](int ep, const DevCsr& M, int n, const double* x, double* y, EpArgs a) {
        prog.push_back(TailOp{T_SPMV, ep, n, M.rowptr.p, M.col.p, M.val.p, x, y, a, nullptr, nullptr, 0, 0, 0, nullptr});
      };
      auto gs = [&](DevLevel& V, int nc, int backward, int lds_ok) {
        prog.push_back(TailOp{T_GS, 0, (int)V.n, V.tA.rowptr.p, V.tA.col.p, V.tA.val.p, nullptr, V.x.p,
                              EpArgs{V.rhs.p, nullptr, V.dinv.p, 0.0, nullptr, 0}, V.t_rowlist.p, V.t_cptr.p, nc, backward, lds_ok, V.t_rowcolor.p});
      };
      auto gs_lds_ok = [&](int l) {
        const amgx_matrix& A = levels[l].A;
        if (A.n_rows > TAIL_BLOCK / TAIL_G || std::getenv("AMGX_NO_TAIL_LDS")) return 0;
        for (int64_t i = 0; i < A.n_rows; ++i) if (A.rowptr[i + 1] - A.rowptr[i] > TAIL_G * TAIL_GS_K) return 0;
        return 1;
      };
      for (int l = T; l + 1 < L; ++l) {
        const amgx_level_desc& s = levels[l];
        DevLevel& V = h->lev[l];
        const int64_t nnz = s.A.rowptr[s.A.n_rows];
        V.tA.upload(s.A); V.tP.upload(s.P); V.tPT.upload(s.PT);
        if (s.sm_type == AMGX_SM_JACOBI) {
          std::vector<double> sv((size_t)nnz);
          for (int64_t k = 0; k < nnz; ++k) sv[k] = s.A.val[k] * (s.omega * s.dinv[s.A.col[k]]);
          V.tApre.upload(s.A, sv.data());
        } else {
          std::vector<int32_t> cptr(s.n_colors + 1, 0), rl;
          for (int64_t i = 0; i < s.A.n_rows; ++i) if (s.color[i] >= 0) cptr[s.color[i] + 1]++;
          for (int c = 0; c < s.n_colors; ++c) cptr[c + 1] += cptr[c];
          rl.resize(cptr[s.n_colors]);
          std::vector<int32_t> pos(cptr.begin(), cptr.end() - 1);
          for (int64_t i = 0; i < s.A.n_rows; ++i) if (s.color[i] >= 0) rl[pos[s.color[i]]++] = (int32_t)i;
          V.t_rowlist.upload(rl); V.t_cptr.upload(cptr);
          V.t_rowcolor.upload(s.color, (size_t)s.A.n_rows);
        }
      }
      for (int l = T; l + 1 < L; ++l) {       // down
        DevLevel& V = h->lev[l];
        if (V.sm_type == AMGX_SM_JACOBI) {     // r = b - A'b, x = omega*Dinv*b
          spmv(EP_PRE, V.tApre, (int)V.n, V.rhs.p, V.res.p, EpArgs{V.rhs.p, nullptr, V.dinv.p, V.omega, V.x.p, 0});
        } else {                               // x = 0; forward sweep; r = b - A x
          prog.push_back(TailOp{T_ZERO, 0, (int)V.n, nullptr, nullptr, nullptr, nullptr, V.x.p, none, nullptr, nullptr, 0, 0, 0, nullptr});
          gs(V, levels[l].n_colors, 0, gs_lds_ok(l));
          spmv(EP_RES, V.tA, (int)V.n, V.x.p, V.res.p, EpArgs{V.rhs.p, nullptr, nullptr, 0.0, nullptr, 0});
        }
        spmv(EP_MULT, V.tPT, (int)h->lev[l + 1].n, V.res.p, h->lev[l + 1].rhs.p, none);   // b_{l+1} = P^T r
      }
      prog.push_back(TailOp{T_DENSE, 0, (int)h->coarse_n, nullptr, nullptr, h->coarse_inv.p, h->lev[L - 1].rhs.p, h->lev[L - 1].x.p,
                            none, nullptr, nullptr, 0, 0, 0, nullptr});
      for (int l = L - 2; l >= T; --l) {      // up
        DevLevel& V = h->lev[l];
        if (V.sm_type == AMGX_SM_JACOBI) {     // tmp = x + P x_{l+1} ; x = tmp + omega*Dinv*(b - A tmp)
          spmv(EP_AXPY, V.tP, (int)V.n, h->lev[l + 1].x.p, V.tmp.p, EpArgs{nullptr, V.x.p, nullptr, 1.0, nullptr, 0});
          spmv(EP_JAC, V.tA, (int)V.n, V.tmp.p, V.x.p, EpArgs{V.rhs.p, V.tmp.p, V.dinv.p, V.omega, nullptr, 0});
        } else {                               // x += P x_{l+1} ; backward sweep
          spmv(EP_AXPY, V.tP, (int)V.n, h->lev[l + 1].x.p, V.x.p, EpArgs{nullptr, V.x.p, nullptr, 1.0, nullptr, 0});
          gs(V, levels[l].n_colors, 1, gs_lds_ok(l));
        }
      }
      h->tail_prog.upload(prog);
      h->tail_ops = (int)prog.size();
      h->tail_level = T;
    }
  }
  HIPCHK(hipDeviceSynchronize());
  clk.lap("coarse inverse, tail program");
  if (dense_first >= 0) build_dense_tail(*h, d, levels, dense_first);
  clk.lap("collapsed coarse levels (dense operator)");
  return h.release();
}

}  // namespace amgx

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------

struct amgx_handle_t { amgx::Handle* h; };

namespace {
thread_local std::string g_create_err;

template <class F>
int guard(amgx_handle hh, F&& f) {
  try {
    if (!hh || !hh->h) throw amgx::Err("null handle");
    HIPCHK(hipSetDevice(hh->h->device));
    f(*hh->h);
    return 0;
  } catch (const std::exception& e) {
    if (hh && hh->h) hh->h->err = e.what(); else g_create_err = e.what();
    return 1;
  }
}

// stage host vectors on the device when the caller passed host pointers
struct Staged {
  amgx::Handle& h;
  bool host;
  Staged(amgx::Handle& hh, int flags) : h(hh), host(!(flags & AMGX_DEVICE_PTR)) {}
  static void fit(amgx::DevBuf<double>& b, int64_t n) { if ((int64_t)b.n < n) b.alloc(n); }
  // lvl >= 0: the vector belongs to that level; if the level is stored in colour-major numbering the device copy is
  // renumbered (device pointers too: the work then runs on the staging buffers)
  const double* in(int slot, const double* p, int64_t n, int lvl = -1) {
    if (!p) return p;
    const bool pm = h.permuted(lvl);
    if (!host && !pm) return p;
    const double* src = p;
    if (host) {
      amgx::DevBuf<double>& dst = pm ? h.stage_raw[slot] : h.stage[slot];
      fit(dst, n);
      HIPCHK(hipMemcpyAsync(dst.p, p, n * sizeof(double), hipMemcpyHostToDevice, h.stream));
      if (!pm) return dst.p;
      src = dst.p;
    }
    fit(h.stage[slot], n);
    h.perm_gather(lvl, src, h.stage[slot].p);
    return h.stage[slot].p;
  }
  double* inout(int slot, double* p, int64_t n, bool load, int lvl = -1) {
    if (!p) return p;
    const bool pm = h.permuted(lvl);
    if (!host && !pm) return p;
    if (load) return const_cast<double*>(in(slot, p, n, lvl));
    fit(h.stage[slot], n);
    return h.stage[slot].p;
  }
  void out(int slot, double* p, int64_t n, int lvl = -1) {
    if (!p) return;
    const bool pm = h.permuted(lvl);
    if (!host && !pm) return;
    const double* src = h.stage[slot].p;
    if (pm) {
      double* dst = p;
      if (host) { fit(h.stage_raw[slot], n); dst = h.stage_raw[slot].p; }
      h.perm_scatter(lvl, src, dst);
      src = dst;
    }
    if (host) HIPCHK(hipMemcpyAsync(p, src, n * sizeof(double), hipMemcpyDeviceToHost, h.stream));
  }
  void finish() { if (host) HIPCHK(hipStreamSynchronize(h.stream)); }
};
}  // namespace

extern "C" {

const char* amgx_last_error(amgx_handle h) { return (h && h->h) ? h->h->err.c_str() : g_create_err.c_str(); }

int amgx_create(const amgx_hierarchy_desc* desc, amgx_handle* out) {
  try {
    if (!out) throw amgx::Err("amgx_create: null output");
    amgx::Handle* h = amgx::create(desc);
    *out = new amgx_handle_t{h};
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_destroy(amgx_handle h) {
  if (!h) return 0;
  if (h->h) { (void)hipSetDevice(h->h->device); (void)hipDeviceSynchronize(); delete h->h; }
  delete h;
  return 0;
}

int amgx_device_count(int32_t* n) {
  if (!n) return 1;
  int nd = 0;
  *n = (hipGetDeviceCount(&nd) == hipSuccess) ? nd : 0;
  return 0;
}

// ---- setup products on the device (spgemm.hpp) -------------------------------------------------------
int amgx_spgemm(const amgx_matrix* A, const amgx_matrix* B, amgx_csr_result* out, int64_t* n_rows, int64_t* nnz) {
  try {
    if (!A || !B || !out) throw amgx::Err("amgx_spgemm: null argument");
    *out = nullptr;
    if (A->n_cols != B->n_rows || A->bc != B->br) throw amgx::Err("amgx_spgemm: dimension mismatch");
    if (A->br * B->bc > 36 || A->br < 1 || A->bc < 1 || B->bc < 1) return 2;
    amgx::SpCsr a, b;
    a.upload(*A);
    b.upload(*B);
    auto r = std::make_unique<amgx::SpCsr>();
    if (!amgx::dev_spgemm(a, b, *r)) return 2;
    if (n_rows) *n_rows = r->n_rows;
    if (nnz) *nnz = r->nnz;
    *out = reinterpret_cast<amgx_csr_result>(r.release());
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_galerkin(const amgx_matrix* PT, const amgx_matrix* A, const amgx_matrix* P, amgx_csr_result* out, int64_t* n_rows, int64_t* nnz) {
  try {
    if (!PT || !A || !P || !out) throw amgx::Err("amgx_galerkin: null argument");
    *out = nullptr;
    if (PT->n_cols != A->n_rows || A->n_cols != P->n_rows || PT->bc != A->br || A->bc != P->br) throw amgx::Err("amgx_galerkin: dimension mismatch");
    for (const amgx_matrix* m : {PT, A, P}) if (m->br < 1 || m->bc < 1 || m->br * m->bc > 36) return 2;
    if (PT->br * P->bc > 36) return 2;
    amgx::SpCsr pta;
    {
      amgx::SpCsr pt, a;
      pt.upload(*PT);
      a.upload(*A);
      if (!amgx::dev_spgemm(pt, a, pta)) return 2;
    }
    amgx::SpCsr p;
    p.upload(*P);
    auto r = std::make_unique<amgx::SpCsr>();
    if (!amgx::dev_spgemm(pta, p, *r)) return 2;
    if (n_rows) *n_rows = r->n_rows;
    if (nnz) *nnz = r->nnz;
    *out = reinterpret_cast<amgx_csr_result>(r.release());
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_csr_result_fetch(amgx_csr_result res, int64_t* rowptr, int32_t* col, double* val) {
  std::unique_ptr<amgx::SpCsr> r(reinterpret_cast<amgx::SpCsr*>(res));
  try {
    if (!r) throw amgx::Err("amgx_csr_result_fetch: null result");
    if (rowptr) HIPCHK(hipMemcpy(rowptr, r->rowptr.p, (size_t)(r->n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (col && r->nnz) HIPCHK(hipMemcpy(col, r->col.p, (size_t)r->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (val && r->nnz) HIPCHK(hipMemcpy(val, r->val.p, (size_t)r->nnz * r->br * r->bc * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_set_stream(amgx_handle hh, void* s) {
  return guard(hh, [&](amgx::Handle& h) {
    hipStream_t ns = (hipStream_t)s;     // NULL = the legacy default stream (what torch uses unless told otherwise)
    if (ns != h.stream) { HIPCHK(hipStreamSynchronize(h.stream)); h.drop_graphs(); h.stream = ns; }
  });
}

int amgx_synchronize(amgx_handle hh) { return guard(hh, [&](amgx::Handle& h) { HIPCHK(hipStreamSynchronize(h.stream)); }); }

int amgx_apply(amgx_handle hh, const double* b, double* x, int b_status, int flags) {
  (void)b_status;   // single GPU: DISTRIBUTED == CUMULATED (b.Distribute() is a no-op, amg_matrix.cpp:164)
  return guard(hh, [&](amgx::Handle& h) {
    if (!b || !x) throw amgx::Err("amgx_apply: null vector");
    if (b == x) throw amgx::Err("amgx_apply: b and x must not alias");
    const int64_t n = h.lev[0].len();
    Staged st(h, flags);
    const double* db = st.in(0, b, n, 0);
    double* dx = st.inout(1, x, n, false, 0);
    h.run_cycle(dx, db, !(flags & AMGX_NO_GRAPH));
    st.out(1, x, n, 0);
    st.finish();
  });
}

int amgx_apply_add(amgx_handle hh, double s, const double* b, double* x, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (!b || !x) throw amgx::Err("amgx_apply_add: null vector");
    const int64_t n = h.lev[0].len();
    Staged st(h, flags);
    const double* db = st.in(0, b, n, 0);
    double* dx = st.inout(1, x, n, true, 0);
    h.run_cycle(h.lev[0].x.p, db, !(flags & AMGX_NO_GRAPH));     // cycle into x_level[0] (amg_matrix.cpp:385-389)
    h.axpy(n, s, h.lev[0].x.p, dx);
    st.out(1, x, n, 0);
    st.finish();
  });
}

int amgx_smooth(amgx_handle hh, int level, int dir, double* x, const double* b, double* res,
                int res_updated, int update_res, int x_zero, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_smooth: level out of range");
    amgx::DevLevel& L = h.lev[level];
    if (!L.dinv.p) throw amgx::Err("amgx_smooth: level has no smoother");
    if (!x || !b || !res) throw amgx::Err("amgx_smooth: null vector");
    const int64_t n = L.len();
    Staged st(h, flags);
    double* dx = st.inout(0, x, L.ext_len(), true, level);        // ghost entries (if any) are read, never written
    const double* db = st.in(1, b, n, level);
    double* dr = st.inout(2, res, n, true, level);
    h.level_smooth(L, dir, dx, db, dr, res_updated != 0, update_res != 0, x_zero != 0);
    st.out(0, x, n, level);
    st.out(2, res, n, level);
    st.finish();
  });
}

int amgx_smooth_v_from_level(amgx_handle hh, int level, double* x, const double* b, double* res,
                             int res_updated, int update_res, int x_zero, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_smooth_v_from_level: level out of range");
    const int64_t n = h.lev[level].len();
    Staged st(h, flags);
    double* dx = st.inout(0, x, n, true, level);
    const double* db = st.in(1, b, n, level);
    double* dr = st.inout(2, res, n, true, level);
    h.smooth_v_from_level(level, dx, db, dr, res_updated != 0, update_res != 0, x_zero != 0);
    st.out(0, x, n, level);
    st.out(2, res, n, level);
    st.finish();
  });
}

int amgx_residual(amgx_handle hh, int level, const double* x, const double* b, double* r, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_residual: level out of range");
    if (!x || !b || !r || x == r) throw amgx::Err("amgx_residual: bad vectors");
    amgx::DevLevel& L = h.lev[level];
    Staged st(h, flags);
    const double* dx = st.in(0, x, L.ext_len(), level);
    const double* db = st.in(1, b, L.len(), level);
    double* dr = st.inout(2, r, L.len(), false, level);
    h.residual(L.A, dx, db, dr);
    st.out(2, r, L.len(), level);
    st.finish();
  });
}

int amgx_jacobi_pre(amgx_handle hh, int level, const double* b, double* x, double* r, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_jacobi_pre: level out of range");
    amgx::DevLevel& L = h.lev[level];
    if (!L.dinv.p || L.sm_type != AMGX_SM_JACOBI) throw amgx::Err("amgx_jacobi_pre: level has no Jacobi smoother");
    if (!b || !x || !r) throw amgx::Err("amgx_jacobi_pre: null vector");
    Staged st(h, flags);
    const double* db = st.in(0, b, L.ext_len(), level);
    double* dx = st.inout(1, x, L.len(), false, level);
    double* dr = st.inout(2, r, L.len(), false, level);
    h.pre_smooth(L, dx, db, dr);
    st.out(1, x, L.len(), level);
    st.out(2, r, L.len(), level);
    st.finish();
  });
}

int amgx_cycle_down(amgx_handle hh, int level, const double* b, double* x, double* b_coarse, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_cycle_down: level out of range");
    amgx::DevLevel& L = h.lev[level];
    if (!h.folded(L)) throw amgx::Err("amgx_cycle_down: level has no folded prolongation Q (use amgx_jacobi_pre / amgx_transfer_f2c)");
    if (!b || !x || !b_coarse) throw amgx::Err("amgx_cycle_down: null vector");
    Staged st(h, flags);
    const double* db = st.in(0, b, L.ext_len(), level);
    double* dx = st.inout(1, x, L.len(), false, level);
    double* dc = st.inout(2, b_coarse, h.lev[level + 1].len(), false, level + 1);
    h.pre_smooth_restrict(level, dx, db, L.res.p, dc, true);
    st.out(1, x, L.len(), level);
    st.out(2, b_coarse, h.lev[level + 1].len(), level + 1);
    st.finish();
  });
}

int amgx_cycle_up(amgx_handle hh, int level, double* x, const double* x_coarse, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_cycle_up: level out of range");
    amgx::DevLevel& L = h.lev[level];
    if (!h.folded(L)) throw amgx::Err("amgx_cycle_up: level has no folded prolongation Q (use amgx_prolong / amgx_jacobi_post)");
    if (!x || !x_coarse) throw amgx::Err("amgx_cycle_up: null vector");
    Staged st(h, flags);
    double* dx = st.inout(0, x, L.len(), true, level);
    const double* dc = st.in(1, x_coarse, L.Q.n_cols, level + 1);
    h.post_smooth(level, dx, nullptr, L.res.p, dc, true);
    st.out(0, x, L.len(), level);
    st.finish();
  });
}

int amgx_jacobi_post(amgx_handle hh, int level, const double* xin, const double* b, double* xout, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_jacobi_post: level out of range");
    amgx::DevLevel& L = h.lev[level];
    if (!L.dinv.p || L.sm_type != AMGX_SM_JACOBI) throw amgx::Err("amgx_jacobi_post: level has no Jacobi smoother");
    if (!xin || !b || !xout || xin == xout) throw amgx::Err("amgx_jacobi_post: bad vectors");
    Staged st(h, flags);
    const double* dxi = st.in(0, xin, L.ext_len(), level);
    const double* db = st.in(1, b, L.len(), level);
    double* dxo = st.inout(2, xout, L.len(), false, level);
    h.jacobi_fused(L, dxi, db, dxo);
    st.out(2, xout, L.len(), level);
    st.finish();
  });
}

int amgx_prolong(amgx_handle hh, int level, double fac, const double* x_in, const double* x_coarse, double* x_out, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_prolong: level out of range");
    if (!x_in || !x_coarse || !x_out) throw amgx::Err("amgx_prolong: null vector");
    Staged st(h, flags);
    const double* di = st.in(0, x_in, h.lev[level].len(), level);
    const double* dc = st.in(1, x_coarse, h.lev[level].P.n_cols * h.lev[level].P.bc, level + 1);
    double* dout = st.inout(2, x_out, h.lev[level].len(), false, level);
    h.mult_add(h.lev[level].P, fac, dc, di, dout);
    st.out(2, x_out, h.lev[level].len(), level);
    st.finish();
  });
}

int amgx_matvec(amgx_handle hh, int level, const double* x, double* y, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_matvec: level out of range");
    if (!x || !y || x == y) throw amgx::Err("amgx_matvec: bad vectors");
    const int64_t n = h.lev[level].len();
    Staged st(h, flags);
    const double* dx = st.in(0, x, h.lev[level].ext_len(), level);
    double* dy = st.inout(1, y, n, false, level);
    h.mult(h.lev[level].A, dx, dy);
    st.out(1, y, n, level);
    st.finish();
  });
}

int amgx_transfer_f2c(amgx_handle hh, int level, const double* xf, double* xc, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_transfer_f2c: level out of range");
    Staged st(h, flags);
    const double* df = st.in(0, xf, h.lev[level].len(), level);
    double* dc = st.inout(1, xc, h.lev[level + 1].len(), false, level + 1);
    h.transfer_f2c(level, df, dc);
    st.out(1, xc, h.lev[level + 1].len(), level + 1);
    st.finish();
  });
}

int amgx_add_c2f(amgx_handle hh, int level, double fac, double* xf, const double* xc, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level + 1 >= h.n_levels()) throw amgx::Err("amgx_add_c2f: level out of range");
    Staged st(h, flags);
    double* df = st.inout(0, xf, h.lev[level].len(), true, level);
    const double* dc = st.in(1, xc, h.lev[level + 1].len(), level + 1);
    h.add_c2f(level, fac, df, dc);
    st.out(0, xf, h.lev[level].len(), level);
    st.finish();
  });
}

int amgx_coarse_solve(amgx_handle hh, const double* rhs, double* x, int flags) {
  return guard(hh, [&](amgx::Handle& h) {
    const int64_t n = h.lev.back().len();
    Staged st(h, flags);
    const double* dr = st.in(0, rhs, n);
    double* dx = st.inout(1, x, n, false);
    h.coarse_solve(dr, dx);
    st.out(1, x, n);
    st.finish();
  });
}

int amgx_n_levels(amgx_handle h) { return (h && h->h) ? h->h->n_levels() : 0; }

int amgx_level_info(amgx_handle hh, int level, int64_t* n, int32_t* bs, int64_t* nnz) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_level_info: level out of range");
    if (n) *n = h.lev[level].n;
    if (bs) *bs = h.lev[level].bs;
    if (nnz) *nnz = h.lev[level].A.nnz;
  });
}

int amgx_cycle_info(amgx_handle hh, int32_t* tail_level, int32_t* dense_level, int64_t* dense_n) {
  return guard(hh, [&](amgx::Handle& h) {
    if (tail_level) *tail_level = h.dense_level >= 0 ? -1 : h.tail_level;
    if (dense_level) *dense_level = h.dense_level;
    if (dense_n) *dense_n = h.dense_level >= 0 ? h.dense_n : 0;
  });
}

int amgx_matrix_info(amgx_handle hh, int level, int which, int32_t* fmt, int64_t* stored, int32_t* lanes) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_matrix_info: level out of range");
    if (which < 0 || which > 6) throw amgx::Err("matrix query: which must be 0..6");
    const amgx::DevLevel& LV = h.lev[level];
    const amgx::DevMatrix& M = which == 0 ? LV.A : which == 1 ? LV.P : which == 2 ? LV.PT : which == 3 ? LV.Apre : which == 4 ? LV.Q : which == 5 ? LV.ApreLW : LV.QLW;
    if (fmt) *fmt = M.empty() ? -1 : (which >= 5 ? 5 : ((M.fmt == amgx::FMT_SELL && M.sell.win) ? 3 : M.fmt));
    if (stored) *stored = M.stored;
    if (lanes) *lanes = M.lanes;
  });
}

int amgx_matrix_stream_bytes(amgx_handle hh, int level, int which, int64_t* bytes) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels() || !bytes) throw amgx::Err("amgx_matrix_stream_bytes: bad arguments");
    if (which < 0 || which > 6) throw amgx::Err("matrix query: which must be 0..6");
    const amgx::DevLevel& LV = h.lev[level];
    const amgx::DevMatrix& M = which == 0 ? LV.A : which == 1 ? LV.P : which == 2 ? LV.PT : which == 3 ? LV.Apre : which == 4 ? LV.Q : which == 5 ? LV.ApreLW : LV.QLW;
    *bytes = M.stream_bytes;
  });
}

int amgx_time_op(amgx_handle hh, int level, int op, int reps, double* avg_ms) {
  return guard(hh, [&](amgx::Handle& h) {
    if (level < 0 || level >= h.n_levels()) throw amgx::Err("amgx_time_op: level out of range");
    if (reps < 1 || !avg_ms) throw amgx::Err("amgx_time_op: bad arguments");
    amgx::DevLevel& L = h.lev[level];
    const bool has_c = level + 1 < h.n_levels();
    if ((op == 2 || op == 3 || op == 5 || op == 6 || op == 7) && !has_c) throw amgx::Err("amgx_time_op: no transfer on the coarsest level");
    if (op == 1 && (!L.dinv.p || L.sm_type != AMGX_SM_JACOBI)) throw amgx::Err("amgx_time_op: level has no Jacobi smoother");
    auto launch = [&]() {
      switch (op) {
        case 0: h.residual(L.A, L.x.p, L.rhs.p, L.res.p); break;
        case 1: h.jacobi_fused(L, L.tmp.p, L.rhs.p, L.x.p); break;
        case 2: h.transfer_f2c(level, L.res.p, h.lev[level + 1].rhs.p); break;
        case 3: h.mult_add(L.P, 1.0, h.lev[level + 1].x.p, L.x.p, L.tmp.p); break;
        case 4: h.run_cycle(h.lev[0].x.p, h.lev[0].rhs.p, true); break;
        case 5: h.pre_smooth_restrict(level, L.x.p, L.rhs.p, L.res.p, h.lev[level + 1].rhs.p, h.folded(L)); break;
        case 6: h.post_smooth(level, L.x.p, L.rhs.p, L.res.p, h.lev[level + 1].x.p, h.folded(L)); break;
        case 7:
          if (L.RF.empty()) throw amgx::Err("amgx_time_op: level has no fused pre-smoothing + restriction kernel");
          h.skip_rsum = true;
          try { h.pre_smooth_restrict(level, L.x.p, L.rhs.p, L.res.p, h.lev[level + 1].rhs.p, h.folded(L)); } catch (...) { h.skip_rsum = false; throw; }
          h.skip_rsum = false;
          break;
        default: throw amgx::Err("amgx_time_op: unknown op");
      }
    };
    if (op == 8 || op == 9) {
      // the dominant kernel timed where it runs: inside the cycle, between the previous cycle's last kernel and the
      // partial-sum reduction (cache state and clocks of the real application), averaged over `reps` cycles
      if (op == 8 && (L.RF.empty() || !(h.plain(L) && L.sm_type == AMGX_SM_JACOBI))) throw amgx::Err("amgx_time_op: level has no fused pre-smoothing + restriction kernel");
      if (op == 9 && !(has_c && h.plain(L) && L.sm_type == AMGX_SM_GS && (L.gsb.on() || L.bgsb.on()) && L.n == L.ncols && !h.folded(L)))
        throw amgx::Err("amgx_time_op: level has no block-hybrid Gauss-Seidel sweep");
      h.probe_kind = op;
      if (h.stream == nullptr) throw amgx::Err("amgx_time_op: op 8 needs a non-default stream");
      HIPCHK(hipEventCreate(&h.probe_e0));
      HIPCHK(hipEventCreate(&h.probe_e1));
      double tot = 0.0;
      if (h.lev[0].len()) hipLaunchKernelGGL(amgx::fill_kernel, dim3(amgx::Handle::grid_for(h.lev[0].len())), dim3(amgx::BLOCK), 0, h.stream, h.lev[0].len(), (uint64_t)2, h.lev[0].rhs.p);
      try {
        h.do_cycle(h.lev[0].x.p, h.lev[0].rhs.p);              // warm-up
        h.probe_level = level;
        for (int i = 0; i < reps; ++i) {
          h.do_cycle(h.lev[0].x.p, h.lev[0].rhs.p);
          HIPCHK(hipEventSynchronize(h.probe_e1));
          float ms = 0;
          HIPCHK(hipEventElapsedTime(&ms, h.probe_e0, h.probe_e1));
          tot += ms;
        }
      } catch (...) { h.probe_level = -1; (void)hipEventDestroy(h.probe_e0); (void)hipEventDestroy(h.probe_e1); h.probe_e0 = h.probe_e1 = nullptr; throw; }
      h.probe_level = -1;
      HIPCHK(hipStreamSynchronize(h.stream));
      (void)hipEventDestroy(h.probe_e0); (void)hipEventDestroy(h.probe_e1);
      h.probe_e0 = h.probe_e1 = nullptr;
      *avg_ms = tot / reps;
      return;
    }
    if (op != 4) {   // time on non-trivial data (the work vectors are otherwise zero in device-pointer mode)
      auto fill = [&](double* v, int64_t n, uint64_t seed) {
        if (n) hipLaunchKernelGGL(amgx::fill_kernel, dim3(amgx::Handle::grid_for(n)), dim3(amgx::BLOCK), 0, h.stream, n, seed, v);
      };
      fill(L.x.p, L.len(), 1); fill(L.rhs.p, L.len(), 2); fill(L.res.p, L.len(), 3); fill(L.tmp.p, L.len(), 4);
      if (has_c) fill(h.lev[level + 1].x.p, h.lev[level + 1].len(), 5);
      HIPCHK(hipGetLastError());
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    launch();                                    // warm-up (and graph capture for op 4)
    HIPCHK(hipStreamSynchronize(h.stream));
    HIPCHK(hipEventRecord(e0, h.stream));
    for (int i = 0; i < reps; ++i) launch();
    HIPCHK(hipEventRecord(e1, h.stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
  });
}

}  // extern "C"

#include "krylov.hpp"
#include "dist.hpp"
#include "gss4.hpp"
