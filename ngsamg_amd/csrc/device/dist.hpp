// Rank-partitioned levels: communicator, halo exchange and the collective V-cycle (included by amgx.hip).
//
// Reference counterparts (not its code):
//   Comm        <-> the MPI communicator of the ParallelDofs                     (src/base/linalg/dcc_map.hpp:20-90)
//   HaloTable   <-> DCCMap tables m_ex_dofs / g_ex_dofs + buffers                (dcc_map.cpp:17-65, 480-543)
//   exchange()  <-> StartCO2CU / ApplyCO2CU (owner -> ghost, overwrite) and StartDIS2CO / ApplyDIS2CO (ghost -> owner,
//                   add)                                                         (dcc_map.cpp:76-178, 249-302)
//   Dist::apply <-> AMGMatrix::SmoothV called collectively by every rank        (src/base/solve/amg_matrix.cpp:160-307)
//                   with HybridSmoother stages around the exchanges              (hybrid_base_smoother.cpp:501-574)
//
// MI355X design: one process per GPU; a rank stores the rows of the vertices it OWNS with columns [owned | ghost]
// (DESIGN.md 5.4), ghosts grouped by owner, so a peer's message lands contiguously (no unpack kernel in the owner ->
// ghost direction).  Owned rows are ordered [interior | boundary]: kernels on the interior rows run on the compute stream
// while pack kernel + ncclSend/ncclRecv run on the communication stream; the boundary part waits for the exchange event.
// Nothing here synchronises the host: one application is one burst of asynchronous launches.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace amgx {

// ---------------------------------------------------------------------------------------------------
// RCCL entry points, resolved at run time: a process that already holds a copy of librccl (torch ships one)
// must not get a second one through a link-time dependency.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;

  static Rccl& get() {
    static Rccl r;
    if (r.lib) return r;
    const char* names[] = {std::getenv("NGSAMG_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);       // the copy this process already uses, if any
    for (int i = 0; i < 4 && !h; ++i) if (names[i] && *names[i]) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) throw Err(std::string("RCCL not found (librccl.so.1): ") + (dlerror() ? dlerror() : "?") + "; set NGSAMG_RCCL_LIB");
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) throw Err(std::string("librccl: missing symbol ") + n); return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.lib = h;
    return r;
  }
};

#define NCCLCHK(call)                                                                                              \
  do {                                                                                                             \
    ncclResult_t r_ = (call);                                                                                      \
    if (r_ != ncclSuccess)                                                                                         \
      throw ::amgx::Err(std::string(#call) + " failed: " + ::amgx::Rccl::get().GetErrorString(r_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// device-to-device copy as a kernel of our own (see vec_copy_kernel)
static inline void dev_copy(double* dst, const double* src, int64_t n, hipStream_t st) {
  if (n <= 0 || dst == src) return;
  hipLaunchKernelGGL(vec_copy_kernel, dim3((unsigned)(((n + 1) / 2 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, n, src, dst);
  HIPCHK(hipGetLastError());
}

struct HaloTable {                       // one level of one rank
  int bs = 1;
  int64_t n = 0, n_ghost = 0, n_int = 0; // owned block rows, ghost block rows, interior rows (no ghost column)
  std::vector<int> peers;                // ascending ranks
  std::vector<int64_t> send_ptr, recv_ptr;   // [n_peers + 1] in block rows
  DevBuf<int32_t> send_idx;              // owned block rows packed per peer (m_ex_dofs)
  DevBuf<double> sendbuf;                // pack target / receive buffer of the add direction
  int64_t n_send() const { return send_ptr.empty() ? 0 : send_ptr.back(); }

  void build(const amgx_halo_desc& d, int64_t n_own, int64_t n_cols, int bsz, int nranks, int self) {
    bs = bsz; n = n_own; n_ghost = n_cols - n_own;
    if (d.n_peers < 0 || (d.n_peers > 0 && (!d.peer_rank || !d.send_ptr || !d.recv_ptr))) throw Err("halo table: missing arrays");
    n_int = d.n_interior;
    if (n_int < 0 || n_int > n) throw Err("halo table: n_interior out of range");
    peers.assign(d.peer_rank, d.peer_rank + d.n_peers);
    send_ptr.assign(d.n_peers + 1, 0); recv_ptr.assign(d.n_peers + 1, 0);
    for (int k = 0; k <= d.n_peers; ++k) { send_ptr[k] = d.send_ptr[k]; recv_ptr[k] = d.recv_ptr[k]; }
    for (int k = 0; k < d.n_peers; ++k) {
      if (peers[k] < 0 || peers[k] >= nranks || (k && peers[k] <= peers[k - 1])) throw Err("halo table: peers must be ascending valid ranks");
      if (peers[k] == self && nranks > 1) throw Err("halo table: a rank cannot be its own peer");
      if (send_ptr[k + 1] < send_ptr[k] || recv_ptr[k + 1] < recv_ptr[k]) throw Err("halo table: pointers must be monotone");
    }
    if (send_ptr[0] != 0 || recv_ptr[0] != 0) throw Err("halo table: pointers must start at 0");
    if (recv_ptr.back() != n_ghost) throw Err("halo table: the receive segments must cover the ghost block exactly");
    const int64_t ns = n_send();
    if (ns >= (int64_t)2147483647) throw Err("halo table: too many send entries");
    if (ns > 0 && !d.send_idx) throw Err("halo table: send_idx missing");
    for (int64_t i = 0; i < ns; ++i) if (d.send_idx[i] < 0 || d.send_idx[i] >= n) throw Err("halo table: send index out of the owned range");
    // the boundary-row contract behind the overlap: a row that READS a ghost must not be interior -- a property of the matrix,
    // checked where the matrix is known (dist_create).  A SENT row may be interior for the Jacobi stages (what is sent is
    // complete before the exchange starts); the Gauss-Seidel stages need more, also checked in dist_create
    if (ns) send_idx.upload(d.send_idx, (size_t)ns);
    sendbuf.alloc((size_t)std::max<int64_t>(1, std::max<int64_t>(ns, 1) * bs));
  }
  int peer_pos(int q) const { for (size_t k = 0; k < peers.size(); ++k) if (peers[k] == q) return (int)k; return -1; }
};

struct Dist;
}  // namespace amgx
struct amgx_dist_t { amgx::Dist* d; };
namespace amgx {

struct Comm {
  int kind = AMGX_COMM_LOCAL, nranks = 1, rank = 0, device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t own_compute = nullptr, compute = nullptr, comm_stream = nullptr;
  static constexpr int NEV = 8;
  hipEvent_t ev_ready[NEV], ev_done[NEV];
  int ev_next = 0;
  std::vector<Dist*> members;            // local ranks in creation order (RCCL: exactly one)
  // C-ABI wrappers handed out for this communicator's objects (amgx_dist_create, amgx_dist_handles): owned here, so that
  // amgx_comm_destroy can null them -- a call through a stale wrapper then returns an error instead of touching freed memory
  std::vector<amgx_dist_t*> dist_wrappers;
  std::vector<amgx_handle_t*> handle_views;
  std::string err;
  int64_t n_exchanges = 0;               // statistics: halo exchanges started
  // Cross-stream ordering.  Measured on MI355X (tools/sync_lab.hip, profiles/r02/sync_lab.txt): one exchange-shaped
  // dependency pair costs 15 us of stream time with hipEventRecord / hipStreamWaitEvent and 9 us with
  // hipStreamWriteValue64 / hipStreamWaitValue64 on a device flag; the latter is used where the device supports it.
  uint64_t* flags = nullptr;             // 2 * NEV counters, one cache line each
  uint64_t epoch = 0;
  bool use_values = false;
  // Whole-cycle graph.  At strong-scaling sizes (1.25 M rows per rank and below) one application is ~30 launches, three
  // RCCL groups and six cross-stream orderings for 100 - 200 us of GPU work: the host cannot enqueue them faster than the
  // GPU retires them.  The collective cycle -- both streams, pack kernels and ncclSend / ncclRecv / ncclAllGather
  // included -- is therefore captured ONCE per (b, x, b_status) into a hipGraph (cross-stream orderings become graph edges,
  // i.e. cost nothing at replay) and replayed with one launch.  AMGX_DIST_GRAPH=0 keeps direct launches; a capture that
  // fails (a runtime / RCCL build that cannot capture an operation) falls back to direct launches for good.
  bool graph_ok = true, capturing = false;
  struct GKey { std::vector<const void*> p; int status; bool operator<(const GKey& o) const { return status != o.status ? status < o.status : p < o.p; } };
  struct GVal { hipGraphExec_t exec; int64_t exchanges; };
  std::map<GKey, GVal> graphs;
  int64_t n_direct_runs = 0;             // applications launched directly; the first one always is (see dist_apply)
  int64_t n_graph_replays = 0;
  std::string graph_note;
  void drop_graphs() { for (auto& g : graphs) (void)hipGraphExecDestroy(g.second.exec); graphs.clear(); graph_age.clear(); }
  std::vector<GKey> graph_age;           // capture order: at 16 graphs the OLDEST one goes, not the hot ones
  void evict_oldest_graph() {
    while (!graph_age.empty()) {
      auto it = graphs.find(graph_age.front());
      graph_age.erase(graph_age.begin());
      if (it != graphs.end()) { (void)hipGraphExecDestroy(it->second.exec); graphs.erase(it); return; }
    }
    drop_graphs();
  }
  // workspace of amgx_dist_pcg / amgx_dist_gmres (DistKrylov), kept across solves so that its vector addresses -- the key of the
  // preconditioner's whole-cycle graph -- stay the same
  std::shared_ptr<void> krylov_ws;
  size_t krylov_members = 0;

  Comm() { for (int i = 0; i < NEV; ++i) { ev_ready[i] = nullptr; ev_done[i] = nullptr; } }
  ~Comm() {
    drop_graphs();
    for (int i = 0; i < NEV; ++i) { if (ev_ready[i]) (void)hipEventDestroy(ev_ready[i]); if (ev_done[i]) (void)hipEventDestroy(ev_done[i]); }
    if (flags) (void)hipFree(flags);
    if (nccl) (void)Rccl::get().CommDestroy(nccl);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    if (own_compute) (void)hipStreamDestroy(own_compute);
  }
  void init_streams() {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&own_compute, hipStreamNonBlocking));
    compute = own_compute;
    // the communication stream gets the higher priority: its small pack kernels and the RCCL kernels must not queue
    // behind the streaming kernels of the interior rows
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPCHK(hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, hi));
    for (int i = 0; i < NEV; ++i) {
      HIPCHK(hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
    }
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) can = 0;
    use_values = can && !std::getenv("AMGX_DIST_EVENTS");
    if (const char* e = std::getenv("AMGX_DIST_GRAPH")) graph_ok = std::atoi(e) != 0;
    if (use_values) {
      HIPCHK(hipMalloc((void**)&flags, 2 * NEV * 64));
      HIPCHK(hipMemset(flags, 0, 2 * NEV * 64));
    }
  }
  // everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
  // (while a graph is being captured the event form is used: it turns into a graph edge)
  void order(hipStream_t from, hipStream_t to, int slot, hipEvent_t ev) {
    if (use_values && !capturing) {
      uint64_t* f = flags + slot * 8;
      ++epoch;
      HIPCHK(hipStreamWriteValue64(from, f, epoch, 0));
      HIPCHK(hipStreamWaitValue64(to, f, epoch, hipStreamWaitValueGte, 0xffffffffffffffffull));
    } else {
      HIPCHK(hipEventRecord(ev, from));
      HIPCHK(hipStreamWaitEvent(to, ev, 0));
    }
  }
  // split form: the signal is enqueued now, the wait later (exchange_end)
  uint64_t signal(hipStream_t from, int slot, hipEvent_t ev) {
    if (use_values && !capturing) { ++epoch; HIPCHK(hipStreamWriteValue64(from, flags + slot * 8, epoch, 0)); return epoch; }
    HIPCHK(hipEventRecord(ev, from));
    return 0;
  }
  void wait(hipStream_t to, int slot, hipEvent_t ev, uint64_t value) {
    if (use_values && !capturing) HIPCHK(hipStreamWaitValue64(to, flags + slot * 8, value, hipStreamWaitValueGte, 0xffffffffffffffffull));
    else HIPCHK(hipStreamWaitEvent(to, ev, 0));
  }
  uint64_t done_value[NEV] = {0};

  // ---- halo exchange.  items[i] = (table, vector) of local member i.  Returns a ticket for exchange_end. -------------
  struct Item { HaloTable* t; double* vec; };

  // owner -> ghost, overwrite (reference CO2CU: BufferM, send, ApplyG; dcc_map.cpp:138-178, 280-302)
  int exchange_begin(const std::vector<Item>& items) {
    Range rg("DCCMap::StartCO2CU");
    const int tk = ev_next; ev_next = (ev_next + 1) % NEV;
    ++n_exchanges;
    order(compute, comm_stream, tk, ev_ready[tk]);                 // everything the vectors depend on is enqueued
    for (const Item& it : items) {
      const int64_t len = it.t->n_send() * it.t->bs;
      if (len) hipLaunchKernelGGL(halo_pack_kernel, dim3(Handle::grid_for(len)), dim3(BLOCK), 0, comm_stream, len, it.t->bs,
                                  it.t->send_idx.p, it.vec, it.t->sendbuf.p);
    }
    HIPCHK(hipGetLastError());
    if (kind == AMGX_COMM_RCCL) {
      Rccl& R = Rccl::get();
      const HaloTable& t = *items[0].t;
      double* vec = items[0].vec;
      if (!t.peers.empty()) {
        NCCLCHK(R.GroupStart());
        for (size_t k = 0; k < t.peers.size(); ++k) {
          const int64_t ns = (t.send_ptr[k + 1] - t.send_ptr[k]) * t.bs, nr = (t.recv_ptr[k + 1] - t.recv_ptr[k]) * t.bs;
          if (ns) NCCLCHK(R.Send(t.sendbuf.p + t.send_ptr[k] * t.bs, (size_t)ns, ncclDouble, t.peers[k], nccl, comm_stream));
          if (nr) NCCLCHK(R.Recv(vec + (t.n + t.recv_ptr[k]) * t.bs, (size_t)nr, ncclDouble, t.peers[k], nccl, comm_stream));
        }
        NCCLCHK(R.GroupEnd());
      }
    } else {
      for (size_t i = 0; i < items.size(); ++i) {
        const HaloTable& t = *items[i].t;
        for (size_t k = 0; k < t.peers.size(); ++k) {
          const int q = t.peers[k];
          const int64_t ns = (t.send_ptr[k + 1] - t.send_ptr[k]) * t.bs;
          if (!ns) continue;
          if (q < 0 || q >= (int)items.size()) throw Err("local exchange: peer out of range");
          const HaloTable& tq = *items[q].t;
          const int kq = tq.peer_pos((int)i);
          if (kq < 0 || (tq.recv_ptr[kq + 1] - tq.recv_ptr[kq]) * tq.bs != ns) throw Err("local exchange: send / receive sizes do not match");
          dev_copy(items[q].vec + (tq.n + tq.recv_ptr[kq]) * tq.bs, t.sendbuf.p + t.send_ptr[k] * t.bs, ns, comm_stream);
        }
      }
    }
    done_value[tk] = signal(comm_stream, NEV + tk, ev_done[tk]);
    return tk;
  }
  void exchange_end(int ticket) { Range rg("DCCMap::ApplyCO2CU"); wait(compute, NEV + ticket, ev_done[ticket], done_value[ticket]); }

  // ghost -> owner, add, ghost entries zeroed afterwards (reference DIS2CO: BufferG, send, ApplyM; dcc_map.cpp:76-136, 249-274)
  void accumulate(const std::vector<Item>& items) {
    Range rg("DCCMap::StartDIS2CO");
    const int tk = ev_next; ev_next = (ev_next + 1) % NEV;
    ++n_exchanges;
    order(compute, comm_stream, tk, ev_ready[tk]);
    if (kind == AMGX_COMM_RCCL) {
      Rccl& R = Rccl::get();
      const HaloTable& t = *items[0].t;
      double* vec = items[0].vec;
      if (!t.peers.empty()) {
        NCCLCHK(R.GroupStart());
        for (size_t k = 0; k < t.peers.size(); ++k) {
          const int64_t ns = (t.send_ptr[k + 1] - t.send_ptr[k]) * t.bs, nr = (t.recv_ptr[k + 1] - t.recv_ptr[k]) * t.bs;
          if (nr) NCCLCHK(R.Send(vec + (t.n + t.recv_ptr[k]) * t.bs, (size_t)nr, ncclDouble, t.peers[k], nccl, comm_stream));
          if (ns) NCCLCHK(R.Recv(t.sendbuf.p + t.send_ptr[k] * t.bs, (size_t)ns, ncclDouble, t.peers[k], nccl, comm_stream));
        }
        NCCLCHK(R.GroupEnd());
      }
    } else {
      for (size_t i = 0; i < items.size(); ++i) {
        const HaloTable& t = *items[i].t;
        for (size_t k = 0; k < t.peers.size(); ++k) {
          const int q = t.peers[k];
          const int64_t nr = (t.recv_ptr[k + 1] - t.recv_ptr[k]) * t.bs;
          if (!nr) continue;
          if (q < 0 || q >= (int)items.size()) throw Err("local exchange: peer out of range");
          const HaloTable& tq = *items[q].t;
          const int kq = tq.peer_pos((int)i);
          if (kq < 0 || (tq.send_ptr[kq + 1] - tq.send_ptr[kq]) * tq.bs != nr) throw Err("local exchange: send / receive sizes do not match");
          dev_copy(tq.sendbuf.p + tq.send_ptr[kq] * tq.bs, items[i].vec + (t.n + t.recv_ptr[k]) * t.bs, nr, comm_stream);
        }
      }
    }
    for (const Item& it : items) {
      const HaloTable& t = *it.t;
      const int64_t gl = t.n_ghost * t.bs;
      if (gl) hipLaunchKernelGGL(halo_zero_kernel, dim3(Handle::grid_for(gl)), dim3(BLOCK), 0, comm_stream, gl, it.vec + t.n * t.bs);
      for (size_t k = 0; k < t.peers.size(); ++k) {          // one launch per peer: two peers may both contribute to a row
        const int64_t len = (t.send_ptr[k + 1] - t.send_ptr[k]) * t.bs;
        if (len) hipLaunchKernelGGL(halo_unpack_add_kernel, dim3(Handle::grid_for(len)), dim3(BLOCK), 0, comm_stream, len, t.bs,
                                    t.send_idx.p + t.send_ptr[k], t.sendbuf.p + t.send_ptr[k] * t.bs, it.vec);
      }
    }
    HIPCHK(hipGetLastError());
    order(comm_stream, compute, NEV + tk, ev_done[tk]);
  }
};

// ---------------------------------------------------------------------------------------------------
// one rank's share of a rank-partitioned hierarchy + the replicated tail
struct Dist {
  Comm* comm = nullptr;
  int index = 0;                         // position in comm->members
  std::unique_ptr<Handle> top, tail;
  int k = 0;                             // levels 0..k-1 smoothed in rank-partitioned form, level k gathered
  int sm_type = AMGX_SM_JACOBI;
  bool fold = true;
  bool overlap = true;
  bool gsb = false;                      // Gauss-Seidel levels in the block-hybrid form
  bool generic = false;                  // sm_steps > 1 / sm_symm on a rank-partitioned level, or a W-cycle: the step-by-step driver
  int cycle = AMGX_CYCLE_V;              //   (DistCycle::generic_cycle) instead of the specialised V(1,1) stage sequences
  std::vector<HaloTable> halo;           // [k]
  std::vector<std::array<int, 4>> stage; // [k] colour ranges of the hybrid Gauss-Seidel stages: [s0,s1) first local part,
                                         //     [s1,s2) boundary ("EX") rows, [s2,s3) second local part (gssmoother.cpp:721-782)
  std::vector<char> send_early;          // [k] block-hybrid levels: every sent row lies in a boundary block, so the exchange of x
                                         //     may start while the interior blocks are still being swept
  std::vector<DevBuf<double>> bext, xext, text, rl;
  DevBuf<double> bk, bpad, bglob, xglob, xk_ext, x0;
  DevBuf<int64_t> kmap, compact;
  std::vector<int64_t> counts, offs;
  int64_t mcount = 0;                    // longest level-k piece
  bool force_allgather = false;
  amgx_handle_t* view_top = nullptr;     // see amgx_dist_handles
  amgx_handle_t* view_tail = nullptr;

  int64_t n(int l) const { return top->lev[l].len(); }
  // interior rows as the split launches see them: a level WITHOUT boundary rows (world size 1, or a rank whose piece touches no
  // other) reports "everything", so that the last partial slice / chunk is not left to an extra boundary launch (5 launches of
  // 4 ... 8 us per cycle at 108^3, profiles/r04/trace_dist_nv108_before.txt)
  int64_t n_int_span(int l) const { return halo[l].n_int >= top->lev[l].n ? (int64_t)1 << 60 : halo[l].n_int; }
  int64_t next(int l) const { return top->lev[l].ext_len(); }
};

static void dist_check_interior(const amgx_matrix& A, int64_t n_int) {
  for (int64_t i = 0; i < n_int; ++i)
    for (int64_t e = A.rowptr[i]; e < A.rowptr[i + 1]; ++e)
      if (A.col[e] >= A.n_rows) throw Err("rank-partitioned level: a row below n_interior has a ghost column");
}

static Dist* dist_create(Comm* c, const amgx_dist_desc* d) {
  if (!c || !d) throw Err("amgx_dist_create: null argument");
  if (c->kind == AMGX_COMM_RCCL && !c->members.empty()) throw Err("amgx_dist_create: an RCCL communicator carries one rank per process");
  if (c->kind == AMGX_COMM_LOCAL && (int)c->members.size() >= c->nranks) throw Err("amgx_dist_create: all local ranks exist already");
  const int self = c->kind == AMGX_COMM_RCCL ? c->rank : (int)c->members.size();
  if (d->rank != self) throw Err("amgx_dist_create: descriptor rank does not match the communicator (local ranks are created in order)");
  HIPCHK(hipSetDevice(c->device));
  auto D = std::make_unique<Dist>();
  D->comm = c;
  D->index = (int)c->members.size();
  D->k = d->top.n_levels - 1;
  if (D->k < 1) throw Err("amgx_dist_create: need at least one rank-partitioned level and the gathered level");
  if (!d->halo || !d->counts || !d->kmap) throw Err("amgx_dist_create: halo tables / level-k tables missing");
  D->sm_type = d->top.levels[0].sm_type;
  D->fold = d->fold != 0;
  D->cycle = d->top.cycle;
  if (D->cycle != AMGX_CYCLE_V && D->cycle != AMGX_CYCLE_W) throw Err("amgx_dist_create: rank-partitioned hierarchies run V and W cycles");
  if (d->tail.cycle != d->top.cycle) throw Err("amgx_dist_create: the replicated tail must run the same cycle as the rank-partitioned levels");
  if (D->cycle == AMGX_CYCLE_W) D->generic = true;
  D->overlap = !std::getenv("AMGX_DIST_NO_OVERLAP");
  amgx_hierarchy_desc td = d->top;
  td.device = c->device; td.use_graph = 0; td.clev = AMGX_CLEV_NONE; td.coarse_n = 0; td.coarse_inv = nullptr;
  td.cycle = AMGX_CYCLE_V;                 // (the top handle is driven stage by stage: its own cycle type plays no role)
  D->top.reset(create(&td, -1));           // (driven stage by stage: no collapsed coarse levels)
  amgx_hierarchy_desc ld = d->tail;
  ld.device = c->device;
  D->tail.reset(create(&ld, 0));           // the replicated tail may collapse completely: x_glob = B b_glob in one GEMV
  // both handles work on the communicator's compute stream
  for (Handle* h : {D->top.get(), D->tail.get()}) { HIPCHK(hipStreamSynchronize(h->stream)); h->stream = c->compute; }
  const int k = D->k;
  D->halo.resize(k);
  D->stage.resize(k);
  for (int l = 0; l < k; ++l) {
    const amgx_level_desc& s = d->top.levels[l];
    if (s.sm_type != D->sm_type) throw Err("amgx_dist_create: all rank-partitioned levels must use the same smoother");
    if (s.sm_steps > 1 || s.sm_symm) D->generic = true;      // ProxySmoother on a rank-partitioned level (base_smoother.hpp:169-229)
    if (s.sm_steps != d->top.levels[0].sm_steps || s.sm_symm != d->top.levels[0].sm_symm) throw Err("amgx_dist_create: all rank-partitioned levels must use the same sm_steps / sm_symm");
    D->halo[l].build(d->halo[l], s.A.n_rows, s.A.n_cols, s.A.br, c->nranks, self);
    dist_check_interior(s.A, D->halo[l].n_int);
    if (D->fold && s.Q.rowptr) {           // the way up splits the same way: interior rows of Q must not read coarse ghosts
      const int64_t nco = d->top.levels[l + 1].A.n_rows;
      for (int64_t i = 0; i < D->halo[l].n_int; ++i)
        for (int64_t e = s.Q.rowptr[i]; e < s.Q.rowptr[i + 1]; ++e)
          if (s.Q.col[e] >= nco) throw Err("rank-partitioned level: a row of Q below n_interior reads a coarse ghost");
    }
    const int nc = s.sm_type == AMGX_SM_GS ? s.n_colors : (s.sm_type == AMGX_SM_BGS ? s.bgs_n_colors : 0);
    D->stage[l] = {0, 0, nc, nc};
    if (d->gs_stage && s.sm_type == AMGX_SM_GS) {
      const int32_t* g = d->gs_stage + 4 * l;
      if (!(g[0] == 0 && g[0] <= g[1] && g[1] <= g[2] && g[2] <= g[3] && g[3] == nc)) throw Err("amgx_dist_create: gs_stage must be 0 <= s1 <= s2 <= n_colors");
      D->stage[l] = {g[0], g[1], g[2], g[3]};
      // rows of the first and third stage must not read ghosts (they run while the exchange is in flight)
      for (int64_t i = 0; i < s.A.n_rows; ++i) {
        const int ci = s.color[i];
        if (ci < 0 || (ci >= g[1] && ci < g[2])) continue;
        for (int64_t e = s.A.rowptr[i]; e < s.A.rowptr[i + 1]; ++e)
          if (s.A.col[e] >= s.A.n_rows) throw Err("amgx_dist_create: a row of a local Gauss-Seidel stage has a ghost column");
      }
      // send side of the same contract: the exchange of x starts after the colours [s0, s2) and runs beside [s2, s3), so a
      // SENT row must have been swept by then.  Structurally symmetric matrices give that for free (a sent row reads a
      // ghost, hence sits in the boundary stage); for anything else the level runs without the overlap.
      const amgx_halo_desc& hd = d->halo[l];
      const int64_t ns = hd.n_peers > 0 ? hd.send_ptr[hd.n_peers] : 0;
      for (int64_t i = 0; i < ns; ++i)
        if (s.color[hd.send_idx[i]] >= g[2]) { D->stage[l] = {g[0], g[1], nc, nc}; break; }
    }
    D->send_early.push_back(1);
    if (D->top->lev[l].gsb.on() || D->top->lev[l].bgsb.on()) {
      // block-hybrid form: the boundary blocks [n_int / B, end) are swept first, then x travels beside the interior blocks
      const amgx_halo_desc& hd = d->halo[l];
      const int64_t ns = hd.n_peers > 0 ? hd.send_ptr[hd.n_peers] : 0;
      if (D->top->lev[l].bgsb.on() && s.gs_block_ids) throw Err("amgx_dist_create: rank-partitioned block levels sweep runs of consecutive rows (no gs_block_ids)");
      if (D->top->lev[l].bgsb.bc) throw Err("amgx_dist_create: rank-partitioned block levels sweep in the hybrid form (no gs_block_color)");
      const int64_t B = D->top->lev[l].gsb.on() ? D->top->lev[l].gsb.B : D->top->lev[l].bgsb.BB;
      const int64_t first_bnd = (D->halo[l].n_int / B) * B;
      for (int64_t i = 0; i < ns; ++i)
        if (hd.send_idx[i] < first_bnd) { D->send_early[l] = 0; break; }
    }
  }
  if (D->fold && D->generic) throw Err("amgx_dist_create: fold is the V(1,1) Jacobi form (no sm_steps / sm_symm / W-cycle)");
  if (D->fold) {
    if (D->sm_type != AMGX_SM_JACOBI) throw Err("amgx_dist_create: fold needs Jacobi levels");
    for (int l = 0; l < k; ++l) if (!D->top->folded(D->top->lev[l])) throw Err("amgx_dist_create: fold requested but level " + std::to_string(l) + " has no Q");
  }
  if (D->sm_type == AMGX_SM_GS) {
    int on = 0;
    for (int l = 0; l < k; ++l) on += (D->top->lev[l].gsb.on() || D->top->lev[l].bgsb.on()) ? 1 : 0;
    if (on != 0 && on != k) throw Err("amgx_dist_create: either all or none of the rank-partitioned Gauss-Seidel levels use gs_block_rows");
    D->gsb = on == k;
  }
  D->bext.resize(k); D->xext.resize(k); D->text.resize(k); D->rl.resize(k);
  auto zalloc = [&](DevBuf<double>& b, int64_t len) { b.alloc((size_t)std::max<int64_t>(1, len)); HIPCHK(hipMemset(b.p, 0, std::max<int64_t>(1, len) * sizeof(double))); };
  for (int l = 0; l < k; ++l) {
    zalloc(D->bext[l], D->next(l)); zalloc(D->xext[l], D->next(l)); zalloc(D->rl[l], D->n(l));
    if ((!D->fold && D->sm_type == AMGX_SM_JACOBI) || D->gsb || D->generic) zalloc(D->text[l], D->next(l));
  }
  // level k: gathered in rank order
  D->counts.assign(d->counts, d->counts + c->nranks);
  D->offs.assign(c->nranks + 1, 0);
  for (int r = 0; r < c->nranks; ++r) { if (D->counts[r] < 0) throw Err("amgx_dist_create: negative count"); D->offs[r + 1] = D->offs[r] + D->counts[r]; D->mcount = std::max(D->mcount, D->counts[r]); }
  // AMGX_DIST_FORCE_ALLGATHER (tests, one-GPU rehearsals): world size 1 goes through ncclAllGather too instead of the copy
  // shortcut; "pad" additionally pretends the pieces differ in size (every slot is 5 rows longer than the longest piece), so
  // that the padded all-gather and the compaction kernel run
  if (const char* e = std::getenv("AMGX_DIST_FORCE_ALLGATHER")) {
    D->force_allgather = true;
    if (std::string(e) == "pad") D->mcount += 5;
  }
  const int bsk = D->top->lev[k].bs;
  if (D->counts[self] * bsk != D->n(k)) throw Err("amgx_dist_create: counts[rank] does not match level k");
  if (D->offs.back() * bsk != D->tail->lev[0].len()) throw Err("amgx_dist_create: the replicated tail does not match the gathered level");
  if (d->kmap_len != D->next(k)) throw Err("amgx_dist_create: kmap must cover level k [owned | ghost]");
  for (int64_t i = 0; i < d->kmap_len; ++i) if (d->kmap[i] < 0 || d->kmap[i] >= D->offs.back() * bsk) throw Err("amgx_dist_create: kmap out of range");
  D->kmap.upload(d->kmap, (size_t)d->kmap_len);
  zalloc(D->bk, D->mcount * bsk); zalloc(D->bglob, D->offs.back() * bsk); zalloc(D->xglob, D->offs.back() * bsk);
  zalloc(D->xk_ext, D->next(k)); zalloc(D->x0, D->n(0));
  bool equal = true;
  for (int r = 0; r < c->nranks; ++r) equal = equal && D->counts[r] == D->mcount;
  if (!equal && c->kind == AMGX_COMM_RCCL) {
    // ncclAllGather moves pieces of one size: pad to the longest piece, then drop the padding with one gather
    zalloc(D->bpad, D->mcount * bsk * c->nranks);
    std::vector<int64_t> ci((size_t)(D->offs.back() * bsk));
    for (int r = 0; r < c->nranks; ++r)
      for (int64_t i = 0; i < D->counts[r] * bsk; ++i) ci[D->offs[r] * bsk + i] = (int64_t)r * D->mcount * bsk + i;
    D->compact.upload(ci);
  }
  HIPCHK(hipDeviceSynchronize());
  return D.release();
}

// ---------------------------------------------------------------------------------------------------
// the collective cycle.  All members of the communicator advance stage by stage (one member under RCCL).
struct DistCycle {
  Comm& c;
  std::vector<Dist*>& M;
  std::vector<double*> x;                // level-0 solution vectors (device)
  using Span = Handle::Span;
  // replicated tail: direct launches by default -- a graph launch between directly launched kernels costs ~10 us of
  // stream time (profiles/r02/trace_dist_world1.txt), the tail's handful of kernels do not pay that back
  bool tail_graph = std::getenv("AMGX_DIST_TAIL_GRAPH") != nullptr;

  std::vector<Comm::Item> items(int l, int which) {      // which: 0 bext, 1 xext, 2 text
    std::vector<Comm::Item> it;
    for (Dist* d : M) it.push_back({&d->halo[l], which == 0 ? d->bext[l].p : which == 1 ? d->xext[l].p : d->text[l].p});
    return it;
  }
  double* xl(Dist* d, int i, int l) { return l == 0 ? x[i] : d->xext[l].p; }
  double* bnext(Dist* d, int l) { return l + 1 < d->k ? d->bext[l + 1].p : d->bk.p; }

  void gather_level_k() {
    for (size_t i = 0; i < M.size(); ++i) {
      Dist* d = M[i];
      const int bsk = d->top->lev[d->k].bs;
      if (c.kind == AMGX_COMM_RCCL) {
        Rccl& R = Rccl::get();
        if (c.nranks == 1 && !d->force_allgather) d->top->copy(d->bglob.p, d->bk.p, d->n(d->k));
        else if (d->compact.n == 0) NCCLCHK(R.AllGather(d->bk.p, d->bglob.p, (size_t)(d->mcount * bsk), ncclDouble, c.nccl, c.compute));
        else {
          NCCLCHK(R.AllGather(d->bk.p, d->bpad.p, (size_t)(d->mcount * bsk), ncclDouble, c.nccl, c.compute));
          const int64_t len = d->offs.back() * bsk;
          hipLaunchKernelGGL(index_gather_kernel, dim3(Handle::grid_for(len)), dim3(BLOCK), 0, c.compute, len, d->compact.p, d->bpad.p, d->bglob.p);
          HIPCHK(hipGetLastError());
        }
      } else {
        for (size_t q = 0; q < M.size(); ++q)
          if (d->counts[i]) dev_copy(M[q]->bglob.p + d->offs[i] * bsk, d->bk.p, d->counts[i] * bsk, c.compute);
      }
    }
  }

  void tail_and_pick() {
    for (Dist* d : M) {
      d->tail->run_cycle(d->xglob.p, d->bglob.p, tail_graph && !c.capturing);
      const int64_t len = d->next(d->k);
      if (len) hipLaunchKernelGGL(index_gather_kernel, dim3(Handle::grid_for(len)), dim3(BLOCK), 0, c.compute, len, d->kmap.p, d->xglob.p, d->xk_ext.p);
      HIPCHK(hipGetLastError());
    }
  }

  // ---- Jacobi, post-smoothing folded into the prolongation: 2k - 1 exchanges ------------------------
  void jacobi_folded() {
    const int k = M[0]->k;
    for (int l = 0; l < k; ++l) {
      const int tk = c.exchange_begin(items(l, 0));
      if (M[0]->overlap)
        for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; d->top->pre_smooth_restrict(l, xl(d, i, l), d->bext[l].p, d->rl[l].p, bnext(d, l), true, Span{Handle::PART_INT, d->n_int_span(l)}); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->pre_smooth_restrict(l, xl(d, i, l), d->bext[l].p, d->rl[l].p, bnext(d, l), true, M[0]->overlap ? Span{Handle::PART_BND, d->n_int_span(l)} : Span());
      }
    }
    gather_level_k();
    tail_and_pick();
    int tk = -1;
    for (int l = k - 1; l >= 0; --l) {
      if (tk >= 0 && M[0]->overlap)
        for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; d->top->post_smooth(l, xl(d, i, l), nullptr, d->rl[l].p, d->xext[l + 1].p, true, Span{Handle::PART_INT, d->n_int_span(l)}); }
      const bool split = tk >= 0 && M[0]->overlap;
      if (tk >= 0) c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        const double* xc = l + 1 < k ? d->xext[l + 1].p : d->xk_ext.p;
        d->top->post_smooth(l, xl(d, i, l), nullptr, d->rl[l].p, xc, true, split ? Span{Handle::PART_BND, d->n_int_span(l)} : Span());
      }
      tk = l > 0 ? c.exchange_begin(items(l, 1)) : -1;
    }
  }

  // ---- Jacobi, literal stage order (base_smoother.cpp:61-74 around dof_map.cpp:636-709): 2k exchanges ----------------
  void jacobi_literal() {
    const int k = M[0]->k;
    for (int l = 0; l < k; ++l) {
      const int tk = c.exchange_begin(items(l, 0));
      if (M[0]->overlap)
        for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; d->top->pre_smooth(d->top->lev[l], xl(d, i, l), d->bext[l].p, d->rl[l].p, false, Span{Handle::PART_INT, d->n_int_span(l)}); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->pre_smooth(d->top->lev[l], xl(d, i, l), d->bext[l].p, d->rl[l].p, false, M[0]->overlap ? Span{Handle::PART_BND, d->n_int_span(l)} : Span());
        d->top->transfer_f2c(l, d->rl[l].p, bnext(d, l));
      }
    }
    gather_level_k();
    tail_and_pick();
    for (int l = k - 1; l >= 0; --l) {
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        const double* xc = l + 1 < k ? d->xext[l + 1].p : d->xk_ext.p;       // owned part first in both
        d->top->mult_add(d->top->lev[l].P, 1.0, xc, xl(d, i, l), d->text[l].p);
      }
      const int tk = c.exchange_begin(items(l, 2));
      if (M[0]->overlap)
        for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; d->top->jacobi_fused(d->top->lev[l], d->text[l].p, d->bext[l].p, xl(d, i, l), Span{Handle::PART_INT, d->n_int_span(l)}); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->jacobi_fused(d->top->lev[l], d->text[l].p, d->bext[l].p, xl(d, i, l), M[0]->overlap ? Span{Handle::PART_BND, d->n_int_span(l)} : Span());
      }
    }
  }

  // ---- hybrid (block) Gauss-Seidel: local sweeps with the off-rank values frozen at their sweep-start values
  //      (HybridGSSmoother, gssmoother.cpp:709-861; stages LOC_1 / EX / LOC_2 = colour ranges, :721-782) ---------------
  void sweep(Dist* d, int l, int dir, double* xv, const double* b, int c0, int c1) {
    if (c1 <= c0) return;
    DevLevel& L = d->top->lev[l];
    if (d->sm_type == AMGX_SM_BGS) d->top->bgs_sweep(L, dir, xv, b, c0, c1);
    else d->top->gs_sweep(L, dir, xv, b, false, c0, c1);
  }
  // ---- Gauss-Seidel in the block-hybrid form (gsb_sweep_kernel): the blocks of a sweep are independent of each other
  //      (couplings that leave a block -- ghost columns included -- use the sweep-start vector), so the boundary blocks are
  //      swept first, the exchange of x starts, and the interior blocks run behind it; on the way up the interior blocks
  //      run while x + P x_c travels.  Two launches per sweep instead of one per colour and stage.
  void hybrid_gsb(const std::vector<const double*>& b0) {
    const int k = M[0]->k;
    auto bl = [&](Dist* d, size_t i, int l) { return l == 0 ? b0[i] : (const double*)d->bext[l].p; };
    // (a level with a sent row inside an interior block sweeps all its blocks before the exchange: see Dist::send_early)
    auto nbi = [&](Dist* d, int l) {
      const DevLevel& L = d->top->lev[l];
      return d->send_early[l] ? (int)(d->halo[l].n_int / (L.gsb.on() ? L.gsb.B : L.bgsb.BB)) : 0;
    };
    // scalar levels: gsb_sweep_kernel; square-block levels: bgsb_sweep_kernel (same block ranges, same stages)
    auto sweep_zero = [&](Dist* d, int l, double* xout, const double* b, int q0, int q1) {
      DevLevel& L = d->top->lev[l];
      if (L.bgsb.on()) d->top->bgsb_sweep(L, 0, nullptr, xout, b, L.bgsb.has_split, q0, q1);
      else d->top->gsb_sweep(L, 0, L.gsb.has_split ? L.gsb.lowin : L.gsb.full, nullptr, xout, b, q0, q1);
    };
    auto sweep_back = [&](Dist* d, int l, const double* xin, double* xout, const double* b, int q0, int q1) {
      DevLevel& L = d->top->lev[l];
      if (L.bgsb.on()) d->top->bgsb_sweep(L, 1, xin, xout, b, false, q0, q1);
      else d->top->gsb_sweep(L, 1, L.gsb.full, xin, xout, b, q0, q1);
    };
    for (int l = 0; l < k; ++l) {
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->zero(d->xext[l].p + d->n(l), d->next(l) - d->n(l));
        sweep_zero(d, l, d->xext[l].p, bl(d, i, l), nbi(d, l), -1);
      }
      const int tk = c.exchange_begin(items(l, 1));
      for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; sweep_zero(d, l, d->xext[l].p, bl(d, i, l), 0, nbi(d, l)); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        DevLevel& L = d->top->lev[l];
        if (L.gsb.on() && L.gsb.has_split) d->top->gsb_residual_restrict(l, d->xext[l].p, d->rl[l].p, bnext(d, l));
        else {
          if (L.bgsb.on() && L.bgsb.has_split) d->top->mult(L.bgsb.rest, d->xext[l].p, d->rl[l].p);      // r = rest x (see DevBGSB)
          else d->top->residual(L.A, d->xext[l].p, bl(d, i, l), d->rl[l].p);
          d->top->transfer_f2c(l, d->rl[l].p, bnext(d, l));
        }
      }
    }
    gather_level_k();
    tail_and_pick();
    for (int l = k - 1; l >= 0; --l) {
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        const double* xc = l + 1 < k ? d->xext[l + 1].p : d->xk_ext.p;
        d->top->mult_add(d->top->lev[l].P, 1.0, xc, d->xext[l].p, d->text[l].p);
      }
      const int tk = c.exchange_begin(items(l, 2));
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        // (amgx_dist_time_kernel, op 9: HIP events around the interior blocks' launch of the first local rank)
        const bool probe = i == 0 && d->top->probe_level == l && d->top->probe_kind == 9 && d->top->probe_e0;
        if (probe) HIPCHK(hipEventRecord(d->top->probe_e0, c.compute));
        sweep_back(d, l, d->text[l].p, l == 0 ? x[i] : d->xext[l].p, bl(d, i, l), 0, nbi(d, l));
        if (probe) HIPCHK(hipEventRecord(d->top->probe_e1, c.compute));
      }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; sweep_back(d, l, d->text[l].p, l == 0 ? x[i] : d->xext[l].p, bl(d, i, l), nbi(d, l), -1); }
    }
  }

  void hybrid_gs(const std::vector<const double*>& b0) {
    const int k = M[0]->k;
    auto bl = [&](Dist* d, size_t i, int l) { return l == 0 ? b0[i] : (const double*)d->bext[l].p; };
    for (int l = 0; l < k; ++l) {
      // pre: x = 0; forward sweep (all off-rank values are 0, no exchange needed before it); the owner -> ghost exchange
      // of x starts as soon as the boundary stage is done and hides behind the second local stage
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->zero(d->xext[l].p, d->next(l));
        sweep(d, l, 0, d->xext[l].p, bl(d, i, l), d->stage[l][0], d->stage[l][2]);
      }
      const int tk = c.exchange_begin(items(l, 1));
      for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; sweep(d, l, 0, d->xext[l].p, bl(d, i, l), d->stage[l][2], d->stage[l][3]); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        d->top->residual(d->top->lev[l].A, d->xext[l].p, bl(d, i, l), d->rl[l].p);
        d->top->transfer_f2c(l, d->rl[l].p, bnext(d, l));
      }
    }
    gather_level_k();
    tail_and_pick();
    for (int l = k - 1; l >= 0; --l) {
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        const double* xc = l + 1 < k ? d->xext[l + 1].p : d->xk_ext.p;
        d->top->mult_add(d->top->lev[l].P, 1.0, xc, d->xext[l].p, d->xext[l].p);
      }
      // post: backward sweep = the stages in reverse order; the exchange hides behind the (reversed) second local stage
      const int tk = c.exchange_begin(items(l, 1));
      for (size_t i = 0; i < M.size(); ++i) { Dist* d = M[i]; sweep(d, l, 1, d->xext[l].p, bl(d, i, l), d->stage[l][2], d->stage[l][3]); }
      c.exchange_end(tk);
      for (size_t i = 0; i < M.size(); ++i) {
        Dist* d = M[i];
        sweep(d, l, 1, d->xext[l].p, bl(d, i, l), d->stage[l][0], d->stage[l][2]);
        if (l == 0) d->top->copy(x[i], d->xext[0].p, d->n(0));
      }
    }
  }

  // ---- step-by-step driver: sm_steps / sm_symm (ProxySmoother, base_smoother.hpp:169-229: k x Smooth, or k x (Smooth + SmoothBack)
  //      for the pre- AND the post-smoothing) and the W-cycle (AMGMatrix::SmoothW, amg_matrix.cpp:37-107) on rank-partitioned
  //      levels.  Every smoothing step is one step of the parallel smoother (hybrid_base_smoother.cpp:246-289): owner -> ghost
  //      exchange of x, then the local step with the ghost values frozen (Jacobi: x + w Dinv (b - (M + G) x); Gauss-Seidel: the
  //      staged / block-hybrid sweep).  No interior / boundary overlap here: one exchange, one launch per step.
  struct GenState { std::vector<double*> cur, oth; };        // per level: the [owned | ghost] buffer that holds x, and the spare one
  std::vector<GenState> gx;
  std::vector<const double*> gb0;
  const double* gbl(size_t i, int l) { return l == 0 ? gb0[i] : (const double*)M[i]->bext[l].p; }
  void gen_exchange(int l) {
    std::vector<Comm::Item> it;
    for (size_t i = 0; i < M.size(); ++i) it.push_back({&M[i]->halo[l], gx[l].cur[i]});
    c.exchange_end(c.exchange_begin(it));
  }
  void gen_step(int l, int dir, bool x_zero) {
    if (!x_zero) gen_exchange(l);                              // (from x = 0 the ghost values are zeros already)
    for (size_t i = 0; i < M.size(); ++i) {
      Dist* d = M[i];
      Handle& h = *d->top;
      DevLevel& L = h.lev[l];
      double*& cur = gx[l].cur[i];
      double*& oth = gx[l].oth[i];
      const double* b = gbl(i, l);
      if (d->sm_type == AMGX_SM_JACOBI) { h.jacobi_fused(L, cur, b, oth); std::swap(cur, oth); }
      else if (L.bgsb.on()) { h.bgsb_sweep(L, dir, cur, oth, b); std::swap(cur, oth); }
      else if (L.gsb.on()) { h.gsb_sweep(L, dir, L.gsb.full, cur, oth, b); std::swap(cur, oth); }
      else if (d->sm_type == AMGX_SM_BGS) h.bgs_sweep(L, dir, cur, b);
      else h.gs_sweep(L, dir, cur, b);
    }
  }
  // k steps in direction dir, or k x (forward, backward) with sm_symm
  void gen_smooth(int l, int dir, bool x_zero) {
    const DevLevel& L0 = M[0]->top->lev[l];
    const int k = std::max(1, L0.sm_steps);
    for (int j = 0; j < k; ++j) {
      if (L0.sm_symm) { gen_step(l, 0, x_zero && j == 0); gen_step(l, 1, false); }
      else gen_step(l, dir, x_zero && j == 0);
    }
  }
  void gen_zero(int l) { for (size_t i = 0; i < M.size(); ++i) M[i]->top->zero(gx[l].cur[i], M[i]->next(l)); }
  void gen_residual_restrict(int l) {                          // r = b - (M + G) x, b_{l+1} = P^T r  (P is rank-local: no exchange)
    gen_exchange(l);
    for (size_t i = 0; i < M.size(); ++i) {
      Dist* d = M[i];
      d->top->residual(d->top->lev[l].A, gx[l].cur[i], gbl(i, l), d->rl[l].p);
      d->top->transfer_f2c(l, d->rl[l].p, bnext(d, l));
    }
  }
  void gen_prolong(int l) {                                    // x_l += P x_{l+1}
    const int k = M[0]->k;
    for (size_t i = 0; i < M.size(); ++i) {
      Dist* d = M[i];
      const double* xc = l + 1 < k ? (const double*)gx[l + 1].cur[i] : (const double*)d->xk_ext.p;
      d->top->mult_add(d->top->lev[l].P, 1.0, xc, gx[l].cur[i], gx[l].cur[i]);
    }
  }
  void gen_v(int l) {
    const int k = M[0]->k;
    if (l == k) { gather_level_k(); tail_and_pick(); return; }
    gen_zero(l);
    gen_smooth(l, 0, true);
    gen_residual_restrict(l);
    gen_v(l + 1);
    gen_prolong(l);
    gen_smooth(l, 1, false);
  }
  void gen_w(int l) {                                          // Handle::w_rec on rank-partitioned levels
    const int k = M[0]->k;
    if (l == k) { gather_level_k(); tail_and_pick(); return; }
    gen_zero(l);
    gen_smooth(l, 0, true);
    gen_residual_restrict(l);
    gen_w(l + 1);
    gen_prolong(l);
    gen_smooth(l, 1, false);
    gen_smooth(l, 0, false);
    gen_residual_restrict(l);
    gen_w(l + 1);
    gen_prolong(l);
    gen_smooth(l, 1, false);
  }
  void generic_cycle(const std::vector<const double*>& b0) {
    const int k = M[0]->k;
    gb0 = b0;
    gx.assign(k, GenState());
    for (int l = 0; l < k; ++l)
      for (Dist* d : M) { gx[l].cur.push_back(d->xext[l].p); gx[l].oth.push_back(d->text[l].p); }
    if (M[0]->cycle == AMGX_CYCLE_W) gen_w(0); else gen_v(0);
    for (size_t i = 0; i < M.size(); ++i) M[i]->top->copy(x[i], gx[0].cur[i], M[i]->n(0));
  }
};

// b_status 0 (DISTRIBUTED): b carries [owned | ghost] entries and the ghost entries are contributions to their owners
// (b.Distribute() state of the reference, amg_matrix.cpp:164); 1 (CUMULATED): the owned entries are complete.
static void dist_apply(Comm& c, const double* const* b, double* const* x, int b_status, int flags) {
  Range rg("AMGMatrix::Mult");
  std::vector<Dist*>& M = c.members;
  if (M.empty()) throw Err("amgx_dist_apply: the communicator has no rank-partitioned hierarchy");
  if (c.kind == AMGX_COMM_LOCAL && (int)M.size() != c.nranks) throw Err("amgx_dist_apply: not all local ranks have been created");
  if (!b || !x) throw Err("amgx_dist_apply: null vector list");
  const bool host = !(flags & AMGX_DEVICE_PTR);
  for (size_t i = 0; i < M.size(); ++i)
    if ((!b[i] || !x[i]) && M[i]->n(0) > 0) throw Err("amgx_dist_apply: null vector");
  // ---- replay / capture of the whole collective cycle (see Comm::graphs) ----------------------------------------------
  const bool want_graph = c.graph_ok && !host && !(flags & AMGX_NO_GRAPH) && c.compute != nullptr;
  Comm::GKey key;
  if (want_graph) {
    key.status = b_status;
    for (size_t i = 0; i < M.size(); ++i) { key.p.push_back(b[i]); key.p.push_back(x[i]); }
    auto it = c.graphs.find(key);
    if (it != c.graphs.end()) {
      HIPCHK(hipGraphLaunch(it->second.exec, c.compute));
      c.n_exchanges += it->second.exchanges;
      ++c.n_graph_replays;
      return;
    }
  }
  const int64_t ex0 = c.n_exchanges;
  // The FIRST application of a communicator is always launched directly: RCCL sets up its point-to-point and all-gather
  // connections lazily inside the first calls, which must not happen inside a stream capture; the capture starts with the
  // second application, when every connection exists.
  const bool capture_now = want_graph && c.n_direct_runs > 0;
  if (capture_now) {
    // the communication stream joins the capture through the first cross-stream ordering and is joined back by the last
    // exchange_end / accumulate, as hipStreamEndCapture requires
    c.capturing = true;
    hipError_t e = hipStreamBeginCapture(c.compute, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { c.capturing = false; c.graph_ok = false; (void)hipGetLastError(); }
  }
  auto body = [&]() {
  DistCycle cy{c, M, {}};
  std::vector<const double*> b0(M.size());
  for (size_t i = 0; i < M.size(); ++i) {
    Dist* d = M[i];
    if ((!b[i] || !x[i]) && d->n(0) > 0) throw Err("amgx_dist_apply: null vector");
    const int64_t nb = b_status == 0 ? d->next(0) : d->n(0);
    if (nb == 0) {}
    else if (host) HIPCHK(hipMemcpyAsync(d->bext[0].p, b[i], nb * sizeof(double), hipMemcpyHostToDevice, c.compute));
    else if (b[i] != d->bext[0].p) dev_copy(d->bext[0].p, b[i], nb, c.compute);
    b0[i] = d->bext[0].p;
    cy.x.push_back(host ? d->x0.p : x[i]);
  }
  if (b_status == 0) {
    std::vector<Comm::Item> it;
    for (Dist* d : M) it.push_back({&d->halo[0], d->bext[0].p});
    c.accumulate(it);
  }
  if (M[0]->generic) cy.generic_cycle(b0);
  else if (M[0]->sm_type == AMGX_SM_JACOBI) { if (M[0]->fold) cy.jacobi_folded(); else cy.jacobi_literal(); }
  else if (M[0]->gsb) cy.hybrid_gsb(b0);
  else cy.hybrid_gs(b0);
  };
  if (c.capturing) {
    hipGraph_t g = nullptr;
    bool ok = true;
    std::string why;
    try { body(); } catch (const std::exception& ex) { ok = false; why = ex.what(); }
    const hipError_t e = hipStreamEndCapture(c.compute, &g);
    c.capturing = false;
    hipGraphExec_t ge = nullptr;
    if (ok && e == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
      (void)hipGraphDestroy(g);
      if (c.graphs.size() >= 16) c.evict_oldest_graph();
      const int64_t nex = c.n_exchanges - ex0;
      c.graphs.emplace(key, Comm::GVal{ge, nex});
      c.graph_age.push_back(key);
      HIPCHK(hipGraphLaunch(ge, c.compute));
      ++c.n_graph_replays;
      return;
    }
    // the capture did not work out: forget it, never try again on this communicator, run this application directly
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    c.graph_ok = false;
    c.n_exchanges = ex0;
    c.graph_note = "whole-cycle graph capture failed (" + (why.empty() ? std::string(hipGetErrorString(e)) : why) + "): direct launches";
  }
  ++c.n_direct_runs;
  body();
  if (host) {
    for (size_t i = 0; i < M.size(); ++i)
      if (M[i]->n(0)) HIPCHK(hipMemcpyAsync(x[i], M[i]->x0.p, M[i]->n(0) * sizeof(double), hipMemcpyDeviceToHost, c.compute));
    HIPCHK(hipStreamSynchronize(c.compute));
  }
}

// ---------------------------------------------------------------------------------------------------
// Preconditioned CG over the rank-partitioned level-0 operator (SURVEY.md 8f-3 for several ranks).  On the reference side
// this is NGSolve's CGSolver on ParallelVectors (tests/h1/amg_utils.py:337-363): every rank runs the recurrences on its
// owned entries, the level-0 product needs the ghost values of the search direction (one owner -> ghost exchange, hidden
// behind the interior rows), the two inner products per iteration are sums over the ranks (MPI all-reduce there;
// ncclAllReduce of one device scalar here, deterministic local reductions), the preconditioner is the collective cycle
// (dist_apply, replayed from its graph).  The residual lives in the level-0 right-hand-side buffer of the cycle, so
// the preconditioner reads it in place.  err_k = sqrt(|<C r_k, r_k>|); stop at err_k <= tol * err_0.
// out[j] = sum over the local ranks i and their KR_BLOCKS partials of product j (partial laid out [rank][64][KR_BLOCKS]); fixed order
__global__ __launch_bounds__(BLOCK) void kr_dist_multi_final_kernel(int n_local, const double* __restrict__ partial, double* __restrict__ out) {
  __shared__ double red[BLOCK];
  const int j = blockIdx.x;
  double acc = 0.0;
  for (int i = 0; i < n_local; ++i) {
    const double* p = partial + ((size_t)i * 64 + j) * KR_BLOCKS;
    for (int q = threadIdx.x; q < KR_BLOCKS; q += BLOCK) acc += p[q];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = BLOCK >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[j] = red[0];
}

struct DistKrylov {
  Comm& c;
  std::vector<Dist*>& M;
  std::vector<DevBuf<double>> sext, w;           // search direction [owned | ghost], work vector (owned)
  DevBuf<double> partial, sc;
  explicit DistKrylov(Comm& cc) : c(cc), M(cc.members) {
    sext.resize(M.size()); w.resize(M.size());
    for (size_t i = 0; i < M.size(); ++i) {
      sext[i].alloc((size_t)std::max<int64_t>(1, M[i]->next(0)));
      w[i].alloc((size_t)std::max<int64_t>(1, M[i]->n(0)));
      HIPCHK(hipMemsetAsync(sext[i].p, 0, std::max<int64_t>(1, M[i]->next(0)) * sizeof(double), c.compute));
    }
    partial.alloc((size_t)KR_BLOCKS * M.size());
    sc.alloc(64);
    HIPCHK(hipMemsetAsync(partial.p, 0, (size_t)KR_BLOCKS * M.size() * sizeof(double), c.compute));    // slots a short rank never writes stay 0
    HIPCHK(hipMemsetAsync(sc.p, 0, 64 * sizeof(double), c.compute));
  }
  static int nb(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(KR_BLOCKS, (n + BLOCK - 1) / BLOCK)); }
  double* d(size_t i) { return M[i]->bext[0].p; }              // residual = right-hand side of the cycle
  // sc[slot] = sum over all ranks of <a_i, b_i> (owned entries)
  void dot(const std::vector<const double*>& a, const std::vector<const double*>& b, int slot) {
    for (size_t i = 0; i < M.size(); ++i) {
      const int64_t n = M[i]->n(0);
      if (n) hipLaunchKernelGGL(kr_dot_partial_kernel, dim3(nb(n)), dim3(BLOCK), 0, c.compute, n, a[i], b[i], partial.p + i * KR_BLOCKS);
    }
    hipLaunchKernelGGL(kr_dot_final_kernel, dim3(1), dim3(BLOCK), 0, c.compute, (int)(KR_BLOCKS * M.size()), partial.p, sc.p + slot);
    HIPCHK(hipGetLastError());
    if (c.kind == AMGX_COMM_RCCL && (c.nranks > 1 || M[0]->force_allgather))      // (world 1: only when the collectives are forced)
      NCCLCHK(Rccl::get().AllReduce(sc.p + slot, sc.p + slot, 1, ncclDouble, ncclSum, c.nccl, c.compute));
  }
  double read(int slot) {
    double v = 0.0;
    HIPCHK(hipMemcpyAsync(&v, sc.p + slot, sizeof(double), hipMemcpyDeviceToHost, c.compute));
    HIPCHK(hipStreamSynchronize(c.compute));
    return v;
  }
  // y_i = A_i [v_i | ghosts of v]: exchange of the ghost part of vext behind the interior rows
  template <class F>
  void with_halo(std::vector<DevBuf<double>>& vext, F&& rows) {
    std::vector<Comm::Item> it;
    for (size_t i = 0; i < M.size(); ++i) it.push_back({&M[i]->halo[0], vext[i].p});
    const int tk = c.exchange_begin(it);
    const bool ov = M[0]->overlap;
    if (ov) for (size_t i = 0; i < M.size(); ++i) rows(i, Handle::Span{Handle::PART_INT, M[i]->n_int_span(0)});
    c.exchange_end(tk);
    for (size_t i = 0; i < M.size(); ++i) rows(i, ov ? Handle::Span{Handle::PART_BND, M[i]->n_int_span(0)} : Handle::Span());
  }
  void precond(bool use_pre) {                   // w = C d
    if (!use_pre) { for (size_t i = 0; i < M.size(); ++i) M[i]->top->copy(w[i].p, d(i), M[i]->n(0)); return; }
    std::vector<const double*> bp(M.size());
    std::vector<double*> xp(M.size());
    for (size_t i = 0; i < M.size(); ++i) { bp[i] = d(i); xp[i] = w[i].p; }
    dist_apply(c, bp.data(), xp.data(), 1, AMGX_DEVICE_PTR);
  }
  int pcg(const double* const* b, double* const* x, double tol, int maxit, bool use_pre, double* errs) {
    const size_t R = M.size();
    std::vector<const double*> wv(R), dv(R), sv(R);
    for (size_t i = 0; i < R; ++i) { wv[i] = w[i].p; dv[i] = d(i); sv[i] = sext[i].p; }
    // d = b - A x
    for (size_t i = 0; i < R; ++i) M[i]->top->copy(sext[i].p, x[i], M[i]->n(0));
    with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->residual(M[i]->top->lev[0].A, sext[i].p, b[i], d(i), sp); });
    precond(use_pre);
    for (size_t i = 0; i < R; ++i) M[i]->top->copy(sext[i].p, w[i].p, M[i]->n(0));
    constexpr int SAS = 2;
    int cur = 1;
    dot(wv, dv, cur);
    const double err0 = std::sqrt(std::fabs(read(cur)));
    if (errs) errs[0] = err0;
    if (err0 == 0.0) return 0;
    int it = 0;
    for (it = 1; it <= maxit; ++it) {
      with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->mult(M[i]->top->lev[0].A, sext[i].p, w[i].p, sp); });     // w = A s
      const int old = cur;
      cur = 1 - cur;
      dot(sv, wv, SAS);
      for (size_t i = 0; i < R; ++i) {
        const int64_t n = M[i]->n(0);
        if (n) hipLaunchKernelGGL(kr_cg_update_kernel, dim3(Handle::grid_for(n)), dim3(BLOCK), 0, c.compute, n, sc.p, old, SAS, sext[i].p, w[i].p, x[i], d(i));
      }
      precond(use_pre);
      dot(wv, dv, cur);
      for (size_t i = 0; i < R; ++i) {
        const int64_t n = M[i]->n(0);
        if (n) hipLaunchKernelGGL(kr_xpby_kernel, dim3(Handle::grid_for(n)), dim3(BLOCK), 0, c.compute, n, sc.p, cur, old, w[i].p, sext[i].p);
      }
      HIPCHK(hipGetLastError());
      const double err = std::sqrt(std::fabs(read(cur)));
      if (errs) errs[it] = err;
      if (err <= tol * err0) break;
    }
    if (it > maxit) it = maxit;
    return it;
  }

  // ---- single-reduction PCG over the ranks (Krylov::pcg_sr): u = C r, w = A u, ONE ncclAllReduce of (gamma, delta) per iteration
  // instead of two all-reduces of one scalar; the residual lives in the cycle's right-hand-side buffer, u in the [owned | ghost]
  // buffer the level-0 product reads
  std::vector<DevBuf<double>> sr_p, sr_s;
  DevBuf<double> sr_partial;
  int pcg_sr(const double* const* b, double* const* x, double tol, int maxit, double* errs) {
    const size_t R = M.size();
    if (sr_p.size() != R) {
      sr_p.clear(); sr_s.clear(); sr_p.resize(R); sr_s.resize(R);
      for (size_t i = 0; i < R; ++i) { sr_p[i].alloc((size_t)std::max<int64_t>(1, M[i]->n(0))); sr_s[i].alloc((size_t)std::max<int64_t>(1, M[i]->n(0))); }
      sr_partial.alloc((size_t)2 * KR_BLOCKS * R);
      HIPCHK(hipMemsetAsync(sr_partial.p, 0, (size_t)2 * KR_BLOCKS * R * sizeof(double), c.compute));
    }
    for (size_t i = 0; i < R; ++i) { M[i]->top->zero(sr_p[i].p, M[i]->n(0)); M[i]->top->zero(sr_s[i].p, M[i]->n(0)); }
    const double one = 1.0;
    HIPCHK(hipMemcpyAsync(sc.p + SR_FIRST, &one, sizeof(double), hipMemcpyHostToDevice, c.compute));
    auto precond_u = [&]() {                               // u = C r, written into the owned part of sext
      std::vector<const double*> bp(R);
      std::vector<double*> xp(R);
      for (size_t i = 0; i < R; ++i) { bp[i] = d(i); xp[i] = sext[i].p; }
      dist_apply(c, bp.data(), xp.data(), 1, AMGX_DEVICE_PTR);
    };
    auto av = [&]() { with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->mult(M[i]->top->lev[0].A, sext[i].p, w[i].p, sp); }); };   // w = A u
    auto reduce = [&]() {
      for (size_t i = 0; i < R; ++i) {
        const int64_t n = M[i]->n(0);
        if (n) hipLaunchKernelGGL(kr_dot2_partial_kernel, dim3(nb(n)), dim3(BLOCK), 0, c.compute, n, d(i), sext[i].p, w[i].p, sr_partial.p + (size_t)i * 2 * KR_BLOCKS);
      }
      hipLaunchKernelGGL(kr_sr_reduce_kernel, dim3(2), dim3(BLOCK), 0, c.compute, (int)R, sr_partial.p, sc.p);
      HIPCHK(hipGetLastError());
      if (c.kind == AMGX_COMM_RCCL && (c.nranks > 1 || M[0]->force_allgather))
        NCCLCHK(Rccl::get().AllReduce(sc.p + SR_GNEW, sc.p + SR_GNEW, 2, ncclDouble, ncclSum, c.nccl, c.compute));
      hipLaunchKernelGGL(kr_sr_scalars_kernel, dim3(1), dim3(1), 0, c.compute, sc.p);
      HIPCHK(hipGetLastError());
    };
    // r = b - A x
    for (size_t i = 0; i < R; ++i) M[i]->top->copy(sext[i].p, x[i], M[i]->n(0));
    with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->residual(M[i]->top->lev[0].A, sext[i].p, b[i], d(i), sp); });
    precond_u();
    av();
    reduce();
    const double err0 = std::sqrt(std::fabs(read(SR_GOLD)));
    if (errs) errs[0] = err0;
    if (err0 == 0.0) return 0;
    int it = 0;
    for (it = 1; it <= maxit; ++it) {
      for (size_t i = 0; i < R; ++i) {
        const int64_t n = M[i]->n(0);
        if (n) hipLaunchKernelGGL(kr_sr_update_kernel, dim3(Handle::grid_for(n)), dim3(BLOCK), 0, c.compute, n, sc.p, sext[i].p, w[i].p, sr_p[i].p, sr_s[i].p, x[i], d(i));
      }
      precond_u();
      av();
      reduce();
      const double err = std::sqrt(std::fabs(read(SR_GOLD)));
      if (errs) errs[it] = err;
      if (err <= tol * err0) break;
    }
    if (it > maxit) it = maxit;
    return it;
  }

  // ---- restarted GMRES(m), left-preconditioned, over the ranks: Krylov::gmres with rank-local vectors.  The Arnoldi inner
  // products h = V^T w are ONE fused local pass + ONE ncclAllReduce of j + 1 scalars per Gram-Schmidt pass (the reference's
  // driver is ngsolve.krylovspace.GMRes on ParallelVectors: one MPI all-reduce per inner product); Givens rotations on the
  // host, identically on every rank (all ranks read the same reduced values).  err_k = |C r_k|, stop at err_k <= tol * err_0.
  std::vector<DevBuf<double>> gV, gt;      // per local rank: basis (m + 1) x n_owned, work vector
  DevBuf<double> hdev, gpartial;
  int g_m = 0;
  // sc[0 .. m) = sum over all ranks of <V_j, w> (owned entries)
  void multi_dot(int m, const std::vector<double*>& w) {
    if (m > 48) throw Err("multi_dot: too many vectors");
    for (size_t i = 0; i < M.size(); ++i) {
      const int64_t n = M[i]->n(0);
      if (n) hipLaunchKernelGGL(kr_multi_dot_partial_kernel, dim3(nb(n)), dim3(BLOCK), 0, c.compute, n, m, gV[i].p, n, w[i], gpartial.p + (size_t)i * 64 * KR_BLOCKS);
    }
    hipLaunchKernelGGL(kr_dist_multi_final_kernel, dim3(m), dim3(BLOCK), 0, c.compute, (int)M.size(), gpartial.p, sc.p);
    HIPCHK(hipGetLastError());
    if (c.kind == AMGX_COMM_RCCL && (c.nranks > 1 || M[0]->force_allgather))
      NCCLCHK(Rccl::get().AllReduce(sc.p, sc.p, (size_t)m, ncclDouble, ncclSum, c.nccl, c.compute));
  }
  void read_n(int m, double* out) {
    HIPCHK(hipMemcpyAsync(out, sc.p, m * sizeof(double), hipMemcpyDeviceToHost, c.compute));
    HIPCHK(hipStreamSynchronize(c.compute));
  }
  void precond_vec(const std::vector<double*>& r, const std::vector<double*>& z, bool use_pre) {      // z = C r (r is copied into the cycle's rhs buffer)
    for (size_t i = 0; i < M.size(); ++i) M[i]->top->copy(use_pre ? d(i) : z[i], r[i], M[i]->n(0));
    if (!use_pre) return;
    std::vector<const double*> bp(M.size());
    std::vector<double*> xp(M.size());
    for (size_t i = 0; i < M.size(); ++i) { bp[i] = d(i); xp[i] = z[i]; }
    dist_apply(c, bp.data(), xp.data(), 1, AMGX_DEVICE_PTR);
  }
  void matvec(const std::vector<double*>& v, const std::vector<double*>& y) {                         // y = A v (level 0, ghosts of v exchanged)
    for (size_t i = 0; i < M.size(); ++i) M[i]->top->copy(sext[i].p, v[i], M[i]->n(0));
    with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->mult(M[i]->top->lev[0].A, sext[i].p, y[i], sp); });
  }
  int gmres(const double* const* b, double* const* x, double tol, int maxit, int restart, bool use_pre, double* errs) {
    if (restart > 40) throw Err("amgx_dist_gmres: restart lengths above 40 are not supported (got " + std::to_string(restart) + ")");
    const int m = std::max(1, restart);
    const size_t R = M.size();
    if (g_m < m || gV.size() != R) {
      gV.clear(); gt.clear(); gV.resize(R); gt.resize(R);
      for (size_t i = 0; i < R; ++i) { gV[i].alloc((size_t)(m + 1) * std::max<int64_t>(1, M[i]->n(0))); gt[i].alloc((size_t)std::max<int64_t>(1, M[i]->n(0))); }
      gpartial.alloc((size_t)KR_BLOCKS * 64 * R);      // (its own buffer: slots a short rank never writes must stay 0, here and in dot())
      HIPCHK(hipMemsetAsync(gpartial.p, 0, (size_t)KR_BLOCKS * 64 * R * sizeof(double), c.compute));
      hdev.alloc(64);
      g_m = m;
    }
    auto grid = [&](size_t i) { return Handle::grid_for(M[i]->n(0)); };
    auto vj = [&](int j) { std::vector<double*> v(R); for (size_t i = 0; i < R; ++i) v[i] = gV[i].p + (size_t)j * M[i]->n(0); return v; };
    std::vector<double*> wv(R), tv(R), xv(R);
    for (size_t i = 0; i < R; ++i) { wv[i] = w[i].p; tv[i] = gt[i].p; xv[i] = x[i]; }
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), hcol(m + 1), hc2(m + 1), y(m);
    int it = 0;
    double err0 = -1.0;
    while (it < maxit) {
      // t = b - A x
      for (size_t i = 0; i < R; ++i) M[i]->top->copy(sext[i].p, x[i], M[i]->n(0));
      with_halo(sext, [&](size_t i, Handle::Span sp) { M[i]->top->residual(M[i]->top->lev[0].A, sext[i].p, b[i], tv[i], sp); });
      precond_vec(tv, vj(0), use_pre);                       // v_0 = C t (not yet normalised)
      { std::vector<const double*> a(R), bb(R); for (size_t i = 0; i < R; ++i) { a[i] = vj(0)[i]; bb[i] = a[i]; } dot(a, bb, 0); }
      const double beta = std::sqrt(read(0));
      if (err0 < 0.0) { err0 = beta; if (errs) errs[0] = err0; }
      if (beta == 0.0 || beta <= tol * err0) break;
      for (size_t i = 0; i < R; ++i) if (M[i]->n(0)) hipLaunchKernelGGL(kr_scale_kernel, dim3(grid(i)), dim3(BLOCK), 0, c.compute, M[i]->n(0), 1.0 / beta, vj(0)[i], vj(0)[i], 0);
      std::fill(g.begin(), g.end(), 0.0);
      g[0] = beta;
      int j = 0;
      bool done = false;
      for (j = 0; j < m && it < maxit; ++j) {
        ++it;
        matvec(vj(j), tv);
        precond_vec(tv, wv, use_pre);                        // w = C A v_j
        std::fill(hcol.begin(), hcol.end(), 0.0);
        for (int pass = 0; pass < 2; ++pass) {
          multi_dot(j + 1, wv);
          read_n(j + 1, hc2.data());
          HIPCHK(hipMemcpyAsync(hdev.p, hc2.data(), (j + 1) * sizeof(double), hipMemcpyHostToDevice, c.compute));
          for (size_t i = 0; i < R; ++i) if (M[i]->n(0)) hipLaunchKernelGGL(kr_multi_axpy_kernel, dim3(grid(i)), dim3(BLOCK), 0, c.compute, M[i]->n(0), j + 1, gV[i].p, M[i]->n(0), hdev.p, -1.0, wv[i]);
          HIPCHK(hipStreamSynchronize(c.compute));           // hc2 is reused by the next pass
          for (int i = 0; i <= j; ++i) hcol[i] += hc2[i];
        }
        { std::vector<const double*> a(R); for (size_t i = 0; i < R; ++i) a[i] = wv[i]; dot(a, a, 0); }
        const double hn = std::sqrt(read(0));
        hcol[j + 1] = hn;
        if (hn > 0.0) for (size_t i = 0; i < R; ++i) if (M[i]->n(0)) hipLaunchKernelGGL(kr_scale_kernel, dim3(grid(i)), dim3(BLOCK), 0, c.compute, M[i]->n(0), 1.0 / hn, wv[i], vj(j + 1)[i], 0);
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
      const int k = j;
      for (int i = k - 1; i >= 0; --i) {
        double sacc = g[i];
        for (int q = i + 1; q < k; ++q) sacc -= H[(size_t)i * m + q] * y[q];
        const double piv = H[(size_t)i * m + i];
        y[i] = piv != 0.0 ? sacc / piv : 0.0;
      }
      if (k > 0) {
        HIPCHK(hipMemcpyAsync(hdev.p, y.data(), k * sizeof(double), hipMemcpyHostToDevice, c.compute));
        for (size_t i = 0; i < R; ++i) if (M[i]->n(0)) hipLaunchKernelGGL(kr_multi_axpy_kernel, dim3(grid(i)), dim3(BLOCK), 0, c.compute, M[i]->n(0), k, gV[i].p, M[i]->n(0), hdev.p, 1.0, xv[i]);
        HIPCHK(hipStreamSynchronize(c.compute));
      }
      if (done) break;
    }
    HIPCHK(hipGetLastError());
    return it;
  }
};

}  // namespace amgx

// ---------------------------------------------------------------------------------------------------
// C ABI (include/amgx.h, "rank-partitioned hierarchies")
// ---------------------------------------------------------------------------------------------------
struct amgx_comm_t { amgx::Comm* c; };
struct amgx_halo_t { amgx::Comm* c; amgx::HaloTable t; };

namespace {
template <class F>
int cguard(amgx_comm cc, F&& f) {
  try {
    if (!cc || !cc->c) throw amgx::Err("null communicator");
    HIPCHK(hipSetDevice(cc->c->device));
    f(*cc->c);
    return 0;
  } catch (const std::exception& e) {
    if (cc && cc->c) cc->c->err = e.what(); else g_create_err = e.what();
    return 1;
  }
}
}  // namespace

extern "C" {

const char* amgx_comm_last_error(amgx_comm c) { return (c && c->c) ? c->c->err.c_str() : g_create_err.c_str(); }

int amgx_comm_unique_id(char* id128) {
  try {
    if (!id128) throw amgx::Err("amgx_comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == AMGX_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    NCCLCHK(amgx::Rccl::get().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_comm_create(int kind, int n_ranks, int rank, const char* id128, int device, amgx_comm* out) {
  try {
    if (!out) throw amgx::Err("amgx_comm_create: null output");
    if (kind != AMGX_COMM_RCCL && kind != AMGX_COMM_LOCAL) throw amgx::Err("amgx_comm_create: unknown kind");
    if (n_ranks < 1) throw amgx::Err("amgx_comm_create: n_ranks must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw amgx::Err("amgx_comm_create: no HIP device available (the apply path has no CPU fallback)");
    if (device < 0 || device >= ndev) throw amgx::Err("amgx_comm_create: device ordinal out of range");
    auto c = std::make_unique<amgx::Comm>();
    c->kind = kind; c->nranks = n_ranks; c->device = device;
    c->rank = kind == AMGX_COMM_RCCL ? rank : 0;
    c->init_streams();
    if (kind == AMGX_COMM_RCCL) {
      if (rank < 0 || rank >= n_ranks || !id128) throw amgx::Err("amgx_comm_create: RCCL needs rank in [0, n_ranks) and the unique id of rank 0");
      ncclUniqueId id;
      std::memcpy(&id, id128, sizeof(id));
      NCCLCHK(amgx::Rccl::get().CommInitRank(&c->nccl, n_ranks, id, rank));     // collective over all ranks
    }
    *out = new amgx_comm_t{c.release()};
    return 0;
  } catch (const std::exception& e) { g_create_err = e.what(); return 1; }
}

int amgx_comm_destroy(amgx_comm c) {
  if (!c) return 0;
  if (c->c) {
    (void)hipSetDevice(c->c->device);
    (void)hipDeviceSynchronize();
    // wrappers stay allocated (a caller may still hold them) but point nowhere: amgx_dist_* / amgx_* calls through them fail cleanly
    for (amgx_dist_t* w : c->c->dist_wrappers) w->d = nullptr;
    for (amgx_handle_t* v : c->c->handle_views) v->h = nullptr;
    for (amgx::Dist* d : c->c->members) delete d;
    delete c->c;
  }
  delete c;
  return 0;
}

int amgx_comm_set_stream(amgx_comm cc, void* s) {
  return cguard(cc, [&](amgx::Comm& c) {
    hipStream_t ns = s ? (hipStream_t)s : c.own_compute;       // NULL: back to the communicator's own stream
    if (ns == c.compute) return;
    HIPCHK(hipStreamSynchronize(c.compute));
    HIPCHK(hipStreamSynchronize(c.comm_stream));
    c.compute = ns;
    c.drop_graphs();
    for (amgx::Dist* d : c.members) { d->top->drop_graphs(); d->tail->drop_graphs(); d->top->stream = ns; d->tail->stream = ns; }
  });
}

int amgx_comm_synchronize(amgx_comm cc) {
  return cguard(cc, [&](amgx::Comm& c) { HIPCHK(hipStreamSynchronize(c.comm_stream)); HIPCHK(hipStreamSynchronize(c.compute)); });
}

int amgx_comm_info(amgx_comm cc, int32_t* kind, int32_t* n_ranks, int32_t* rank, int64_t* n_exchanges) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (kind) *kind = c.kind;
    if (n_ranks) *n_ranks = c.nranks;
    if (rank) *rank = c.rank;
    if (n_exchanges) *n_exchanges = c.n_exchanges;
  });
}

int amgx_comm_graph_info(amgx_comm cc, int32_t* enabled, int64_t* n_graphs, int64_t* n_replays) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (enabled) *enabled = c.graph_ok ? 1 : 0;
    if (n_graphs) *n_graphs = (int64_t)c.graphs.size();
    if (n_replays) *n_replays = c.n_graph_replays;
  });
}
const char* amgx_comm_graph_note(amgx_comm c) { return (c && c->c) ? c->c->graph_note.c_str() : ""; }

int amgx_dist_create(amgx_comm cc, const amgx_dist_desc* desc, amgx_dist* out) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (!out) throw amgx::Err("amgx_dist_create: null output");
    amgx::Dist* d = amgx::dist_create(&c, desc);
    c.members.push_back(d);
    *out = new amgx_dist_t{d};
    c.dist_wrappers.push_back(*out);
  });
}

// the communicator owns the rank objects and their wrappers (freed / nulled by amgx_comm_destroy); nothing to do here
int amgx_dist_destroy(amgx_dist) { return 0; }

int amgx_dist_rhs_buffer(amgx_dist d, double** b, int64_t* n_owned, int64_t* n_ext) {
  if (!d || !d->d) return 1;
  if (b) *b = d->d->bext[0].p;
  if (n_owned) *n_owned = d->d->n(0);
  if (n_ext) *n_ext = d->d->next(0);
  return 0;
}

int amgx_dist_handles(amgx_dist d, amgx_handle* top, amgx_handle* tail) {
  // borrowed views for queries / measurement (amgx_matrix_info, amgx_time_op): one pair per rank object, owned by the
  // communicator; after amgx_comm_destroy they are null handles (every amgx_* call on them returns an error)
  if (!d || !d->d) return 1;
  amgx::Comm* c = d->d->comm;
  if (!d->d->view_top) { d->d->view_top = new amgx_handle_t{d->d->top.get()}; c->handle_views.push_back(d->d->view_top); }
  if (!d->d->view_tail) { d->d->view_tail = new amgx_handle_t{d->d->tail.get()}; c->handle_views.push_back(d->d->view_tail); }
  if (top) *top = d->d->view_top;
  if (tail) *tail = d->d->view_tail;
  return 0;
}

int amgx_dist_apply(amgx_comm cc, const double* const* b, double* const* x, int b_status, int flags) {
  return cguard(cc, [&](amgx::Comm& c) { amgx::dist_apply(c, b, x, b_status, flags); });
}

int amgx_dist_time_kernel(amgx_comm cc, int level, int op, int reps, double* avg_ms) {
  return cguard(cc, [&](amgx::Comm& c) {
    using namespace amgx;
    if (c.members.empty() || (c.kind == AMGX_COMM_LOCAL && (int)c.members.size() != c.nranks)) throw Err("amgx_dist_time_kernel: not all ranks have a hierarchy");
    if (reps < 1 || !avg_ms || (op != 8 && op != 9)) throw Err("amgx_dist_time_kernel: bad arguments (op 8: fused Jacobi down kernel, op 9: backward block-hybrid sweep)");
    Dist* d0 = c.members[0];
    if (level < 0 || level >= d0->k) throw Err("amgx_dist_time_kernel: not a rank-partitioned level");
    Handle& h = *d0->top;
    DevLevel& L = h.lev[level];
    if (op == 8 && (L.RF.empty() || !(h.plain(L) && L.sm_type == AMGX_SM_JACOBI) || !d0->fold)) throw Err("amgx_dist_time_kernel: level has no fused pre-smoothing + restriction kernel");
    if (op == 9 && !(d0->gsb && (L.gsb.on() || L.bgsb.on()))) throw Err("amgx_dist_time_kernel: level has no block-hybrid Gauss-Seidel sweep");
    std::vector<const double*> bb;
    std::vector<double*> xx;
    for (Dist* d : c.members) {
      if (d->n(0)) hipLaunchKernelGGL(fill_kernel, dim3(Handle::grid_for(d->n(0))), dim3(BLOCK), 0, c.compute, d->n(0), (uint64_t)2, d->bext[0].p);
      bb.push_back(d->bext[0].p); xx.push_back(d->x0.p);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventCreate(&h.probe_e0));
    HIPCHK(hipEventCreate(&h.probe_e1));
    h.probe_kind = op;
    double tot = 0.0;
    try {
      dist_apply(c, bb.data(), xx.data(), 1, AMGX_DEVICE_PTR | AMGX_NO_GRAPH);        // warm-up (and RCCL's lazy connections)
      h.probe_level = level;
      for (int i = 0; i < reps; ++i) {
        dist_apply(c, bb.data(), xx.data(), 1, AMGX_DEVICE_PTR | AMGX_NO_GRAPH);
        HIPCHK(hipStreamSynchronize(c.compute));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, h.probe_e0, h.probe_e1));
        tot += ms;
      }
    } catch (...) { h.probe_level = -1; (void)hipEventDestroy(h.probe_e0); (void)hipEventDestroy(h.probe_e1); h.probe_e0 = h.probe_e1 = nullptr; throw; }
    h.probe_level = -1;
    (void)hipEventDestroy(h.probe_e0); (void)hipEventDestroy(h.probe_e1);
    h.probe_e0 = h.probe_e1 = nullptr;
    *avg_ms = tot / reps;
  });
}

int amgx_dist_pcg(amgx_comm cc, const double* const* b, double* const* x, double tol, int maxit, int use_precond, int flags, double* errs,
                  int32_t* iters) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (!b || !x || maxit < 0) throw amgx::Err("amgx_dist_pcg: bad arguments");
    if (!(flags & AMGX_DEVICE_PTR)) throw amgx::Err("amgx_dist_pcg: device vectors only (AMGX_DEVICE_PTR)");
    if (c.members.empty() || (c.kind == AMGX_COMM_LOCAL && (int)c.members.size() != c.nranks)) throw amgx::Err("amgx_dist_pcg: not all ranks have a hierarchy");
    for (size_t i = 0; i < c.members.size(); ++i) {
      amgx::Dist* d = c.members[i];
      if ((!b[i] || !x[i]) && d->n(0) > 0) throw amgx::Err("amgx_dist_pcg: null vector");
      if (b[i] == d->bext[0].p || x[i] == d->bext[0].p) throw amgx::Err("amgx_dist_pcg: b / x alias the cycle's right-hand-side buffer (it holds the residual)");
    }
    // the workspace lives with the communicator: the preconditioner's whole-cycle graph is keyed on the vector addresses, so a
    // second solve replays the graph of the first instead of capturing a new one (and leaving a stale one behind)
    if (!c.krylov_ws || c.krylov_members != c.members.size()) {
      c.krylov_ws = std::shared_ptr<void>(new amgx::DistKrylov(c), [](void* p) { delete static_cast<amgx::DistKrylov*>(p); });
      c.krylov_members = c.members.size();
    }
    amgx::DistKrylov& K = *static_cast<amgx::DistKrylov*>(c.krylov_ws.get());
    const int it = ((flags & AMGX_PCG_SINGLE_REDUCTION) && use_precond) ? K.pcg_sr(b, x, tol, maxit, errs) : K.pcg(b, x, tol, maxit, use_precond != 0, errs);
    if (iters) *iters = it;
  });
}

int amgx_dist_gmres(amgx_comm cc, const double* const* b, double* const* x, double tol, int maxit, int restart, int use_precond, int flags,
                    double* errs, int32_t* iters) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (!b || !x || maxit < 0 || restart < 1) throw amgx::Err("amgx_dist_gmres: bad arguments");
    if (!(flags & AMGX_DEVICE_PTR)) throw amgx::Err("amgx_dist_gmres: device vectors only (AMGX_DEVICE_PTR)");
    if (c.members.empty() || (c.kind == AMGX_COMM_LOCAL && (int)c.members.size() != c.nranks)) throw amgx::Err("amgx_dist_gmres: not all ranks have a hierarchy");
    for (size_t i = 0; i < c.members.size(); ++i) {
      amgx::Dist* d = c.members[i];
      if ((!b[i] || !x[i]) && d->n(0) > 0) throw amgx::Err("amgx_dist_gmres: null vector");
      if (b[i] == d->bext[0].p || x[i] == d->bext[0].p) throw amgx::Err("amgx_dist_gmres: b / x alias the cycle's right-hand-side buffer");
    }
    if (!c.krylov_ws || c.krylov_members != c.members.size()) {
      c.krylov_ws = std::shared_ptr<void>(new amgx::DistKrylov(c), [](void* p) { delete static_cast<amgx::DistKrylov*>(p); });
      c.krylov_members = c.members.size();
    }
    amgx::DistKrylov& K = *static_cast<amgx::DistKrylov*>(c.krylov_ws.get());
    const int it = K.gmres(b, x, tol, maxit, restart, use_precond != 0, errs);
    if (iters) *iters = it;
  });
}

// ---- stand-alone halo maps (the DCCMap surface: python_smoothers / tests use it without a hierarchy) -----------------
int amgx_halo_create(amgx_comm cc, const amgx_halo_desc* d, int64_t n_owned, int64_t n_ghost, int32_t bs, int32_t rank, amgx_halo* out) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (!d || !out) throw amgx::Err("amgx_halo_create: null argument");
    if (bs < 1 || bs > 6) throw amgx::Err("amgx_halo_create: block size must be in 1..6");
    auto h = std::make_unique<amgx_halo_t>();
    h->c = &c;
    // world size 1 over RCCL: a rank may list itself as peer (self send / receive) -- used to exercise the wire on one GPU
    h->t.build(*d, n_owned, n_owned + n_ghost, bs, c.nranks, c.nranks > 1 ? (c.kind == AMGX_COMM_RCCL ? c.rank : rank) : -1);
    *out = h.release();
  });
}
int amgx_halo_destroy(amgx_halo h) { delete h; return 0; }

// mode 0: owner -> ghost, overwrite (CO2CU); mode 1: ghost -> owner, add, ghosts zeroed (DIS2CO).
// halos / vecs: one per local rank (RCCL: one); device vectors of (n_owned + n_ghost) * bs entries on the communicator's stream
int amgx_halo_exchange(amgx_comm cc, int n_local, const amgx_halo* halos, double* const* vecs, int mode) {
  return cguard(cc, [&](amgx::Comm& c) {
    if (n_local < 1 || !halos || !vecs) throw amgx::Err("amgx_halo_exchange: bad arguments");
    if (c.kind == AMGX_COMM_RCCL && n_local != 1) throw amgx::Err("amgx_halo_exchange: one rank per process under RCCL");
    if (c.kind == AMGX_COMM_LOCAL && n_local != c.nranks) throw amgx::Err("amgx_halo_exchange: pass all local ranks");
    std::vector<amgx::Comm::Item> it;
    for (int i = 0; i < n_local; ++i) { if (!halos[i] || halos[i]->c != &c || !vecs[i]) throw amgx::Err("amgx_halo_exchange: bad halo / vector"); it.push_back({&halos[i]->t, vecs[i]}); }
    if (mode == 0) c.exchange_end(c.exchange_begin(it));
    else if (mode == 1) c.accumulate(it);
    else throw amgx::Err("amgx_halo_exchange: mode must be 0 or 1");
  });
}

}  // extern "C"
