// Host AMG setup (cold path).  What it restates from the reference, in simplified form:
//   * strength graph from matrix entries: edge weight |trace-like(A_ij)|
//       (src/h1/h1_impl.hpp:383-429 BuildAlgMesh_ALG_scal)
//   * successive pairwise agglomeration in rounds until the level's coarsening target is met
//       (SPW: src/base/coarsening/spw_agg.hpp:21-85; targets first_aaf / aaf: h1_impl.hpp:333-334)
//   * Dirichlet vertices are dropped from the coarse space (free_verts; amg_pc_vertex_impl.hpp)
//   * piecewise prolongation, then one smoothing step with the auxiliary (edge-weight) matrix,
//       truncated to sp_max_per_row entries (vertex_factory_impl.hpp:1601-1659, 2440-2751)
//   * elasticity: rigid-body prolongation blocks Q = [I, -skew(t); 0, I]
//       (src/elasticity/elasticity_energy.hpp:447-490), displacement-only fine level embeds with 3x6 blocks
//       (elasticity_pc_impl.hpp:668-685)
//   * Galerkin coarse matrices (P^T A) P (utils_sparseMM.hpp:93-109), explicit P^T (utils_sparseMM.cpp:54-93)
//   * stop rule: max_levels / max_coarse_size (base_factory.cpp:339-340)
//   * smoother diagonals (gssmoother.cpp:143-170) and the coarsest-level inverse (amg_pc.cpp:843-928)
// Bit-identical aggregates w.r.t. the reference are neither attainable (no NGSolve/netgen) nor needed for
// apply-path parity; the apply path only consumes the frozen arrays produced here.
#include "hierarchy.hpp"
#include <omp.h>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <numeric>
#include <sstream>

namespace amgh {

namespace {

struct Graph {
  int64_t n = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> adj;
  std::vector<double> w;
  std::vector<double> vs;      // robust_soc: per vertex the largest edge weight collapsed inside it (empty = none)
};

Graph strength_graph(const BCSR& A, const std::vector<uint8_t>& free, int dim, int energy) {
  Graph G;
  G.n = A.n_rows;
  const int bs = A.br;
  const int sub = (energy == 1) ? std::min(dim, bs) : bs;   // elasticity: displacement-displacement part only
  std::vector<int64_t> cnt(G.n + 1, 0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < G.n; i++) {
    int64_t c = 0;
    if (free[i])
      for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) { int32_t j = A.col[k]; if (j != i && free[j]) c++; }
    cnt[i + 1] = c;
  }
  G.ptr.assign(G.n + 1, 0);
  for (int64_t i = 0; i < G.n; i++) G.ptr[i + 1] = G.ptr[i] + cnt[i + 1];
  G.adj.resize(G.ptr[G.n]);
  G.w.resize(G.ptr[G.n]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < G.n; i++) {
    if (!free[i]) continue;
    int64_t p = G.ptr[i];
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      int32_t j = A.col[k];
      if (j == i || !free[j]) continue;
      const double* b = &A.val[k * bs * bs];
      double w;
      if (bs == 1) w = std::fabs(b[0]);
      else {
        double s = 0;
        for (int r = 0; r < sub; r++) for (int c = 0; c < sub; c++) s += b[r * bs + c] * b[r * bs + c];
        w = std::sqrt(s);
      }
      G.adj[p] = j; G.w[p] = w; p++;
    }
  }
  return G;
}

// Numbering of the coarse vertices along a Z-order curve through the aggregates' centroids (experiment, NGSAMG_COARSE_ORDER=morton):
// the agglomerator numbers aggregates in the order it forms them (roughly the fine numbering reversed, orphans appended), so on a
// lexicographic grid 256 consecutive coarse rows span ~2.4 grid lines and their 52-entry stencils touch ~4000 distinct columns;
// compact blocks of rows touch a third of that -- smaller gather footprints for every kernel of the coarse levels.
static bool coarse_order_morton() {
  static const bool on = [] { const char* e = std::getenv("NGSAMG_COARSE_ORDER"); return e && std::string(e) == "morton"; }();
  return on;
}
static void renumber_morton(std::vector<int32_t>& agg, int64_t nc, const std::vector<double>& coords, int dim) {
  const int64_t n = (int64_t)agg.size();
  std::vector<double> cen((size_t)nc * dim, 0.0);
  std::vector<int32_t> cnt(nc, 0);
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) { cnt[agg[i]]++; for (int d = 0; d < dim; d++) cen[(int64_t)agg[i] * dim + d] += coords[i * dim + d]; }
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t I = 0; I < nc; I++) for (int d = 0; d < dim; d++) {
    double& c = cen[I * dim + d];
    c /= std::max(1, cnt[I]);
    lo[d] = std::min(lo[d], c); hi[d] = std::max(hi[d], c);
  }
  const int bits = dim == 2 ? 30 : 20;
  std::vector<std::pair<uint64_t, int32_t>> key(nc);
#pragma omp parallel for schedule(static)
  for (int64_t I = 0; I < nc; I++) {
    uint64_t k = 0;
    uint32_t q[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) {
      const double w = hi[d] - lo[d];
      const double t = w > 0 ? (cen[I * dim + d] - lo[d]) / w : 0.0;
      q[d] = (uint32_t)std::min<double>((double)((1u << bits) - 1), std::max(0.0, t * (double)(1u << bits)));
    }
    for (int b = bits - 1; b >= 0; b--) for (int d = dim - 1; d >= 0; d--) k = (k << 1) | ((q[d] >> b) & 1u);
    key[I] = {k, (int32_t)I};
  }
  std::sort(key.begin(), key.end());
  std::vector<int32_t> perm(nc);
  for (int64_t r = 0; r < nc; r++) perm[key[r].second] = (int32_t)r;
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) agg[i] = perm[agg[i]];
}

constexpr double ROBUST_SCALE_GAP = 16.0;

// One pairwise matching round.  map[i] = new vertex id.  Returns the number of new vertices.
int64_t pairwise_round(const Graph& G, const std::vector<uint8_t>& active, double thresh, std::vector<int32_t>& map) {
  const int64_t n = G.n;
  std::vector<double> mx(n, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double m = 0;
    for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) m = std::max(m, G.w[k]);
    // the collapsed scale only speaks where it is of another order than the vertex's live connections (on a quasi-uniform
    // mesh the two differ by the spread of the element sizes, and the plain measure must not change there)
    if (!G.vs.empty() && G.vs[i] > ROBUST_SCALE_GAP * m) m = G.vs[i];
    mx[i] = m;
  }
  map.assign(n, -1);
  int64_t nn = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!active[i] || map[i] >= 0) continue;
    int32_t best = -1;
    double bs = 0;
    for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
      int32_t j = G.adj[k];
      if (map[j] >= 0 || !active[j]) continue;
      double d = mx[i] * mx[j];
      if (d <= 0) continue;
      double s = G.w[k] / std::sqrt(d);
      if (s >= thresh && s > bs) { bs = s; best = j; }
    }
    map[i] = (int32_t)nn;
    if (best >= 0) map[best] = (int32_t)nn;
    nn++;
  }
  return nn;
}

// Contract a graph along map (values in [0, nn)); vertices with map < 0 are dropped. Edge weights are summed.
Graph contract(const Graph& G, const std::vector<int32_t>& map, int64_t nn) {
  // member lists
  std::vector<int64_t> mptr(nn + 1, 0);
  for (int64_t i = 0; i < G.n; i++) if (map[i] >= 0) mptr[map[i] + 1]++;
  for (int64_t I = 0; I < nn; I++) mptr[I + 1] += mptr[I];
  std::vector<int32_t> mem(mptr[nn]);
  {
    std::vector<int64_t> pos(mptr.begin(), mptr.end() - 1);
    for (int64_t i = 0; i < G.n; i++) if (map[i] >= 0) mem[pos[map[i]]++] = (int32_t)i;
  }
  Graph C;
  C.n = nn;
  if (!G.vs.empty()) {           // scale of a merged vertex: the scales of its members and the edges that vanish inside it
    C.vs.assign(nn, 0.0);
    for (int64_t i = 0; i < G.n; i++) {
      const int32_t I = map[i];
      if (I < 0) continue;
      double m = G.vs[i];
      for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) if (map[G.adj[k]] == I) m = std::max(m, G.w[k]);
      C.vs[I] = std::max(C.vs[I], m);
    }
  }
  // two passes over the members' edges (count, then fill at the final offsets): growing per-thread result vectors instead costs
  // 5-10x the traversal itself in page faults of the reallocated buffers (1.7 s of a 2.5 s agglomeration at 2 M vertices)
  const int nt = std::min(omp_get_max_threads(), 32);   // each thread owns dense markers of size n_cols: bound the memory
  std::vector<int64_t> len(nn, 0), tstart(nt + 1);
  for (int t = 0; t <= nt; t++) tstart[t] = (nn * t) / nt;
  C.ptr.assign(nn + 1, 0);
#pragma omp parallel num_threads(nt)
  {
    int t = omp_get_thread_num();
    std::vector<int64_t> owner(nn, -1);
    std::vector<int32_t> slot(nn, 0);
    std::vector<int32_t> cols;
    std::vector<double> acc;
    std::vector<int32_t> order;
    for (int64_t I = tstart[t]; I < tstart[t + 1]; I++) {
      int64_t c = 0;
      for (int64_t m = mptr[I]; m < mptr[I + 1]; m++) {
        int32_t i = mem[m];
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          int32_t J = map[G.adj[k]];
          if (J < 0 || J == I) continue;
          if (owner[J] != I) { owner[J] = I; c++; }
        }
      }
      len[I] = c;
    }
#pragma omp barrier
#pragma omp single
    {
      for (int64_t I = 0; I < nn; I++) C.ptr[I + 1] = C.ptr[I] + len[I];
      C.adj.resize(C.ptr[nn]);
      C.w.resize(C.ptr[nn]);
    }
    std::fill(owner.begin(), owner.end(), (int64_t)-1);
    for (int64_t I = tstart[t]; I < tstart[t + 1]; I++) {
      cols.clear(); acc.clear();
      for (int64_t m = mptr[I]; m < mptr[I + 1]; m++) {
        int32_t i = mem[m];
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          int32_t J = map[G.adj[k]];
          if (J < 0 || J == I) continue;
          if (owner[J] != I) { owner[J] = I; slot[J] = (int32_t)cols.size(); cols.push_back(J); acc.push_back(G.w[k]); }
          else acc[slot[J]] += G.w[k];
        }
      }
      order.resize(cols.size());
      std::iota(order.begin(), order.end(), 0);
      std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cols[a] < cols[b]; });
      int64_t p = C.ptr[I];
      for (auto q : order) { C.adj[p] = cols[q]; C.w[p] = acc[q]; p++; }
    }
  }
  return C;
}

// Aggregation for one level: repeated pairwise rounds until n_agg <= target * n_free.
// Returns agg (fine vertex -> aggregate or -1) and the number of aggregates.
// out_scale (robust_soc, G0.vs set): per aggregate the largest edge weight that vanishes inside it (and the members' scales)
int64_t aggregate(const Graph& G0, const std::vector<uint8_t>& free, double target, const Options& o,
                  std::vector<int32_t>& agg, int& rounds_done, std::vector<double>* out_scale = nullptr) {
  const int64_t n = G0.n;
  int64_t nfree = 0;
  for (int64_t i = 0; i < n; i++) nfree += free[i] ? 1 : 0;
  agg.assign(n, -1);
  if (nfree == 0) return 0;
  // round 0 works on the fine graph restricted to free vertices
  std::vector<int32_t> map;
  Graph cur;              // current contracted graph (empty => use G0)
  const Graph* g = &G0;
  std::vector<uint8_t> active(free.begin(), free.end());
  std::vector<int32_t> size;   // fine vertices per current vertex
  int64_t ncur = nfree;
  bool first = true;
  rounds_done = 0;
  for (int round = 0; round < o.max_rounds; round++) {
    if (!first && (double)ncur <= target * (double)nfree) break;
    int64_t nn = pairwise_round(*g, active, o.soc_thresh, map);
    if (!first && nn > 0.97 * ncur) { break; }   // stuck: no more viable partners
    // compose
    if (first) {
      for (int64_t i = 0; i < n; i++) agg[i] = free[i] ? map[i] : -1;
    } else {
      for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) agg[i] = map[agg[i]];
    }
    Graph next = contract(*g, map, nn);
    cur = std::move(next);
    g = &cur;
    active.assign(nn, 1);
    ncur = nn;
    first = false;
    rounds_done++;
  }
  if (first) return 0;
  // orphan round: aggregates made of a single fine vertex join their strongest neighbour
  size.assign(ncur, 0);
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) size[agg[i]]++;
  std::vector<int32_t> remap(ncur);
  std::iota(remap.begin(), remap.end(), 0);
  std::vector<double> cmx;
  std::vector<uint8_t> speaks;
  if (!cur.vs.empty()) {         // robust_soc: an orphan joins a neighbour only over a connection that counts on both ends' scales
    cmx.assign(ncur, 0.0);
    speaks.assign(ncur, 0);
    for (int64_t I = 0; I < ncur; I++) {
      for (int64_t k = cur.ptr[I]; k < cur.ptr[I + 1]; k++) cmx[I] = std::max(cmx[I], cur.w[k]);
      if (cur.vs[I] > ROBUST_SCALE_GAP * cmx[I]) { cmx[I] = cur.vs[I]; speaks[I] = 1; }
    }
  }
  for (int64_t I = 0; I < ncur; I++) {
    if (size[I] != 1 || remap[I] != I) continue;
    int32_t best = -1;
    double bw = 0;
    for (int64_t k = cur.ptr[I]; k < cur.ptr[I + 1]; k++) {
      const int32_t J0 = cur.adj[k];
      int32_t J = remap[J0];
      if (J == I) continue;
      if (!cmx.empty() && (speaks[I] || speaks[J0])) {       // (elsewhere the orphan joins its strongest neighbour as before)
        const double dd = cmx[I] * cmx[J0];
        if (!(dd > 0) || cur.w[k] / std::sqrt(dd) < o.soc_thresh) continue;
      }
      if (cur.w[k] > bw) { bw = cur.w[k]; best = J; }
    }
    if (best >= 0) { remap[I] = best; size[best] += 1; size[I] = 0; }
  }
  // path-compress and renumber
  std::vector<int32_t> newid(ncur, -1);
  int64_t nn = 0;
  for (int64_t I = 0; I < ncur; I++) {
    int32_t r = remap[I];
    while (remap[r] != r) r = remap[r];
    remap[I] = r;
  }
  for (int64_t I = 0; I < ncur; I++) if (remap[I] == I) newid[I] = (int32_t)nn++;
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) agg[i] = newid[remap[agg[i]]];
  if (out_scale && !G0.vs.empty()) {
    out_scale->assign(nn, 0.0);
    for (int64_t i = 0; i < n; i++) {
      const int32_t I = agg[i];
      if (I < 0) continue;
      double m = G0.vs[i];
      for (int64_t k = G0.ptr[i]; k < G0.ptr[i + 1]; k++) if (agg[G0.adj[k]] == I) m = std::max(m, G0.w[k]);
      (*out_scale)[I] = std::max((*out_scale)[I], m);
    }
  }
  return nn;
}

// ---------------------------------------------------------------------------------------------------------------------
// Elasticity with the energy's edge matrices (Options::edge_mats): the matrix-valued form of the reference's smoothed
// prolongation.  State of a vertex = (displacement, rotation), BS = dim + dim (dim - 1) / 2 numbers; an edge (i, j) carries a
// symmetric BS x BS matrix E in the frame of its midpoint m, its energy is |Q(m - x_i) u_i - Q(m - x_j) u_j|_E^2 with the
// rigid-body transformation Q(t) = [I, S(t); 0, I] (state at x + t of the rigid motion given at x;
// src/elasticity/elasticity_energy.hpp:28-118 with rot_scaling = 1, which is what BuildAlgMesh_ALG_blk sets and
// AttachedEVD::map_data forwards, elasticity_pc_impl.hpp:479-480, elasticity_impl.hpp:153-155).
constexpr int EM_MAX = 6;

static inline int em_bs(int dim) { return dim + (dim * (dim - 1)) / 2; }

static inline void rb_Q(int dim, const double* t, double* Q) {
  const int BS = em_bs(dim);
  for (int q = 0; q < BS * BS; q++) Q[q] = 0.0;
  for (int r = 0; r < BS; r++) Q[r * BS + r] = 1.0;
  if (dim == 2) { Q[0 * 3 + 2] = -t[1]; Q[1 * 3 + 2] = t[0]; }
  else {
    Q[0 * 6 + 4] = t[2];  Q[0 * 6 + 5] = -t[1];
    Q[1 * 6 + 3] = -t[2]; Q[1 * 6 + 5] = t[0];
    Q[2 * 6 + 3] = t[1];  Q[2 * 6 + 4] = -t[0];
  }
}
static inline void em_mm(int n, const double* A, const double* B, double* C) {        // C = A B
  for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) { double s = 0; for (int k = 0; k < n; k++) s += A[r * n + k] * B[k * n + c]; C[r * n + c] = s; }
}
static inline void em_mtm(int n, const double* A, const double* B, double* C) {       // C = A^T B
  for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) { double s = 0; for (int k = 0; k < n; k++) s += A[k * n + r] * B[k * n + c]; C[r * n + c] = s; }
}

Graph contract_edge_mats(const Graph& g, const std::vector<double>& E, int dim, const std::vector<int32_t>& agg, int64_t nc,
                         const std::vector<double>& xf, const std::vector<double>& xc, std::vector<double>& Ec);

// ---------------------------------------------------------------------------------------------------------------------
// SPW agglomeration with the scalar strength of connection, as the reference runs it for H1 (and for elasticity with
// crs_robust = false): src/base/coarsening/spw_agg_impl.hpp
//   FormAgglomerates_impl (:1436-1830): numRounds (3) pairing rounds on successively contracted graphs + one orphan round
//   PairingIteration / IterateVertsRev (:943-1014, 1072-1263): vertices in REVERSE index order; in the first round the
//       unhandled neighbours of lower degree are visited before the vertex itself (fewer orphans)
//   FindNeib3Step (:637-775) with the scalar weights of FindNeighborToMatch (:778-867) and CalcApproxSOC
//       (agglomerator_utils.hpp:243-266): soc_ij = w_ij / sqrt(maxTrOD_i maxTrOD_j) (GEOM average), computed for ALL
//       neighbours, partner = an unhandled neighbour with soc_ij >= 0.25 max_k soc_ik.  For scalar data the reference's
//       "robust" weights are all 1, i.e. any neighbour that passes the filter is acceptable; the strongest one is taken here.
//   SPWAggData::Map (:424-628): coarse edge weight = sum of the fine edge weights between two aggregates; maxTrOD of a coarse
//       vertex = max over its coarse edges AND over the maxTrOD of all its members -- a stiff region that has collapsed into
//       one vertex keeps the scale of what it swallowed, so its remaining soft connections stay weak
//   JoiningIteration / FindNeighborToJoin (:870-940, 1265-1365; CalcApproxJoinSOC agglomerator_utils.hpp:1011-1032):
//       aggregates that still consist of ONE base vertex ("orphans") join a neighbouring real aggregate over the connection
//       with soc = w_Oj / maxTrOD_O >= 0.25 of the orphan's largest
// maxTrOD is computed afresh from the edge weights of every level (VertexAgglomerator::InitializeAggData,
// agglomerator_impl.hpp:66-200); not restated: fixed aggregates, L2-dominant vertex collapse (vert_thresh = 0 by default),
// the MPI equivalence-class rules, the energy (robust) SOC of matrix-valued vertex data.
constexpr double SPW_REL_THRESH = 0.25;     // cfg.scalRelThresh (spw_agg_impl.hpp:1410)

// ngs_amg_spw_pick_avg (spw_agg.hpp:22, 62-65; Average, agglomerator_utils.hpp:213-226): how the two vertices' maxTrOD enter the scalar
// strength soc = w / avg: 0 min, 1 geom (default), 2 harm, 3 alg, 4 max
static inline double spw_avg(int type, double a, double b) {
  switch (type) {
    case 0: return std::min(a, b);
    case 2: return (a + b) > 0.0 ? 2.0 * (a * b) / (a + b) : 0.0;
    case 3: return 0.5 * (a + b);
    case 4: return std::max(a, b);
    default: return std::sqrt(a * b);
  }
}

static int32_t spw_find_partner(const Graph& g, const std::vector<double>& mt, const std::vector<uint8_t>& handled, int64_t v, bool join,
                                const std::vector<uint8_t>* joinable, int avg = 1) {
  double mx = 0.0;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
    const int32_t j = g.adj[k];
    const double den = join ? mt[v] : spw_avg(avg, mt[v], mt[j]);
    if (den > 0.0) mx = std::max(mx, g.w[k] / den);
  }
  if (!(mx > 0.0)) return -1;
  const double th = SPW_REL_THRESH * mx;
  int32_t best = -1;
  double bw = -1.0;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
    const int32_t j = g.adj[k];
    if (join ? !(*joinable)[j] : handled[j]) continue;
    const double den = join ? mt[v] : spw_avg(avg, mt[v], mt[j]);
    if (!(den > 0.0)) continue;
    const double soc = g.w[k] / den;
    if (soc >= th && soc > bw) { bw = soc; best = j; }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------------------------------
// Energy-based ("robust") strength of connection of the reference's SPW agglomerator for matrix-valued data (crs_robust;
// needs the edge matrices):
//   PrepRobSOC / CalcRobSOC (agglomerator_utils.hpp:845-927): E = the edge's matrix + the neighbour boost (AddNeibBoost, :598-653:
//       for every common neighbour n of i and j the parallel sum E_in (E_in + E_jn)^+ E_jn, formed in n's frame and moved to the
//       edge's midpoint); d_i, d_j = the replacement-matrix diagonals ("aux diagonals": sum of Q^T E Q over a vertex's edges) moved
//       to the midpoint; C = d_i (d_i + d_j)^+ d_j
//   CalcRobustPairSOC (:763-841): the smallest eigenvalue of E v = lambda C v on the complement of ker C (eigenvalues of C below
//       1e-10 of its largest count as kernel)
//   FindNeib3Step with robustPick (spw_agg_impl.hpp:637-775): the neighbours that pass the scalar filter are re-weighted with this
//       number, the strongest is taken if it reaches min(0.25 max scalar soc, edge_thresh = 0.025)
constexpr double ROB_EDGE_THRESH = 0.025;       // agglomerator.hpp:16
constexpr double ROB_ZERO_EV = 1e2 * 1e-12;     // 1e2 RelZeroTol (agglomerator_utils.hpp:923)

static inline void em_qtmq(int n, const double* Q, const double* M, double* out) {   // out = Q^T M Q
  double T[EM_MAX * EM_MAX];
  em_mm(n, M, Q, T);
  em_mtm(n, Q, T, out);
}

// A (A + B)^+ B
static inline void em_parallel_sum(int n, const double* A, const double* B, double* out) {
  double S[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX];
  for (int x = 0; x < n * n; x++) S[x] = A[x] + B[x];
  pseudo_inverse_with_tol(S, n);
  em_mm(n, S, B, T);
  em_mm(n, A, T, out);
}

static double robust_pair_soc(int n, const double* C, const double* E) {
  double w[EM_MAX * EM_MAX], ev[EM_MAX], V[EM_MAX * EM_MAX];
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) w[i * n + j] = 0.5 * (C[i * n + j] + C[j * n + i]);
  sym_eig(w, n, ev, V);                         // columns of V = eigenvectors
  double lmax = 0;
  for (int k = 0; k < n; k++) lmax = std::max(lmax, ev[k]);
  const double th = ROB_ZERO_EV * lmax;
  int idx[EM_MAX], m = 0;
  for (int k = 0; k < n; k++) if (ev[k] > th) idx[m++] = k;
  if (m == 0) return 0.0;
  // S = L^-1/2 V_r^T E V_r L^-1/2
  double S[EM_MAX * EM_MAX], sv[EM_MAX], SV[EM_MAX * EM_MAX];
  for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) {
    double s = 0;
    for (int p = 0; p < n; p++) { double t = 0; for (int q = 0; q < n; q++) t += 0.5 * (E[p * n + q] + E[q * n + p]) * V[q * n + idx[b]]; s += V[p * n + idx[a]] * t; }
    S[a * m + b] = s / std::sqrt(ev[idx[a]] * ev[idx[b]]);
  }
  sym_eig(S, m, sv, SV);
  double lmin = sv[0];
  for (int k = 1; k < m; k++) lmin = std::min(lmin, sv[k]);
  return std::max(0.0, lmin);
}

struct RobustData {
  int dim = 3;
  const std::vector<double>* E = nullptr;   // per entry of the round's graph
  const std::vector<double>* x = nullptr;   // positions of the round's vertices
  std::vector<double> aux;                  // aux diagonal per vertex, in the vertex's frame
  bool neib_boost = true;                   // cfg.neibBoost   (ngs_amg_spw_neib_boost, spw_agg.hpp:27, 56)
  bool pick_robust = true;                  // cfg.robustPick  (ngs_amg_spw_pick_robust, spw_agg.hpp:26, 55)
  int pick_avg = 1;                         // cfg.avgTypeScal (ngs_amg_spw_pick_avg)
  double in_agg_edge_factor = -1.0;         // -2 (1 - diagStabBoost) (ngs_amg_spw_diag_stab_boost = 0.5)
};

static void robust_aux_diags(const Graph& g, RobustData& R) {
  const int dim = R.dim, BS = em_bs(dim), BB = BS * BS;
  R.aux.assign((size_t)g.n * BB, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < g.n; v++) {
    double Q[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX];
    for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
      const int32_t j = g.adj[k];
      double t[3] = {0, 0, 0};
      for (int d = 0; d < dim; d++) t[d] = 0.5 * ((*R.x)[(int64_t)j * dim + d] - (*R.x)[v * dim + d]);
      rb_Q(dim, t, Q);
      em_qtmq(BS, Q, &(*R.E)[(size_t)k * BB], T);
      for (int x = 0; x < BB; x++) R.aux[(size_t)v * BB + x] += T[x];
    }
  }
}

// the edge's matrix with the neighbour boost, in the frame of the edge's midpoint
static void robust_boosted_edge(const Graph& g, const RobustData& R, int64_t i, int64_t k, double* E, double* mid) {
  const int dim = R.dim, BS = em_bs(dim), BB = BS * BS;
  const int32_t j = g.adj[k];
  const std::vector<double>& X = *R.x;
  double Q[EM_MAX * EM_MAX], Ein[EM_MAX * EM_MAX], Ejn[EM_MAX * EM_MAX], H[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX];
  std::copy(&(*R.E)[(size_t)k * BB], &(*R.E)[(size_t)k * BB] + BB, E);
  for (int d = 0; d < 3; d++) mid[d] = 0.0;
  for (int d = 0; d < dim; d++) mid[d] = 0.5 * (X[i * dim + d] + X[(int64_t)j * dim + d]);
  // neighbour boost over the common neighbours
  if (R.neib_boost) for (int64_t ki = g.ptr[i]; ki < g.ptr[i + 1]; ki++) {
    const int32_t nb = g.adj[ki];
    if (nb == j) continue;
    int64_t kj = -1;
    for (int64_t q = g.ptr[j]; q < g.ptr[j + 1]; q++) if (g.adj[q] == nb) { kj = q; break; }
    if (kj < 0) continue;
    double t[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) t[d] = 0.5 * (X[i * dim + d] - X[(int64_t)nb * dim + d]);      // n -> midpoint of (n, i)
    rb_Q(dim, t, Q);
    em_qtmq(BS, Q, &(*R.E)[(size_t)ki * BB], Ein);
    for (int d = 0; d < dim; d++) t[d] = 0.5 * (X[(int64_t)j * dim + d] - X[(int64_t)nb * dim + d]);
    rb_Q(dim, t, Q);
    em_qtmq(BS, Q, &(*R.E)[(size_t)kj * BB], Ejn);
    em_parallel_sum(BS, Ein, Ejn, H);
    for (int d = 0; d < dim; d++) t[d] = X[(int64_t)nb * dim + d] - mid[d];                       // edge midpoint -> n
    rb_Q(dim, t, Q);
    em_qtmq(BS, Q, H, T);
    for (int x = 0; x < BB; x++) E[x] += T[x];
  }
}

static double robust_soc(const Graph& g, const RobustData& R, int64_t i, int64_t k) {
  const int dim = R.dim, BS = em_bs(dim), BB = BS * BS;
  const int32_t j = g.adj[k];
  const std::vector<double>& X = *R.x;
  double E[EM_MAX * EM_MAX], Q[EM_MAX * EM_MAX], mid[3];
  robust_boosted_edge(g, R, i, k, E, mid);
  double di[EM_MAX * EM_MAX], dj[EM_MAX * EM_MAX], C[EM_MAX * EM_MAX];
  double t[3] = {0, 0, 0};
  for (int d = 0; d < dim; d++) t[d] = X[i * dim + d] - mid[d];
  rb_Q(dim, t, Q);
  em_qtmq(BS, Q, &R.aux[(size_t)i * BB], di);
  for (int d = 0; d < dim; d++) t[d] = X[(int64_t)j * dim + d] - mid[d];
  rb_Q(dim, t, Q);
  em_qtmq(BS, Q, &R.aux[(size_t)j * BB], dj);
  em_parallel_sum(BS, di, dj, C);
  return robust_pair_soc(BS, C, E);
}

// Aggregate-wide stability check (checkBigSOC: AggregateWideStabilityCheck, agglomerator_utils.hpp:392-539), used from the second
// pairing round on when ngs_amg_spw_cbs is set (spw_agg.hpp:31, 57; off by default): for the union of the BASE-level members of two
// round vertices, A = the replacement matrix of the base-level edges inside the union, M = the block diagonal of the base-level aux
// diagonals, P = the rigid-body modes of the union; the pair is viable if A - rho (M - M P (P^T M P)^+ P^T M) is positive
// semi-definite (the reference tests that with a pivoted Cholesky factorisation, here with the smallest eigenvalue; its assembly of
// the second block row reuses Q_i^T E where the symmetric replacement matrix has Q_j^T E -- the symmetric matrix is assembled here).
struct BigSocData {
  const Graph* g0 = nullptr;
  const std::vector<double>* E0 = nullptr;
  const std::vector<double>* x0 = nullptr;
  std::vector<double> aux0;                 // base-level aux diagonals
  std::vector<int64_t> mptr;                // members (base vertices, ascending) of the current round's vertices
  std::vector<int32_t> mem;
  int dim = 3;
};

static bool big_soc_ok(const BigSocData& B, int32_t vi, int32_t vj, double rho) {
  const int dim = B.dim, BS = em_bs(dim), BB = BS * BS;
  std::vector<int32_t> mems(B.mem.begin() + B.mptr[vi], B.mem.begin() + B.mptr[vi + 1]);
  mems.insert(mems.end(), B.mem.begin() + B.mptr[vj], B.mem.begin() + B.mptr[vj + 1]);
  std::sort(mems.begin(), mems.end());
  const int n = (int)mems.size();
  if (n < 3) return true;
  const int N = BS * n;
  const Graph& g = *B.g0;
  const std::vector<double>& X = *B.x0;
  std::vector<double> A((size_t)N * N, 0.0), M((size_t)N * N, 0.0), P((size_t)N * BS, 0.0);
  double Qi[EM_MAX * EM_MAX], Qj[EM_MAX * EM_MAX], QiE[EM_MAX * EM_MAX], QjE[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX];
  auto add = [&](int a, int b, double sgn, const double* blk) { for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) A[(size_t)(a * BS + r) * N + b * BS + c] += sgn * blk[r * BS + c]; };
  for (int a = 0; a < n; a++) {
    const int32_t vK = mems[a];
    for (int64_t k = g.ptr[vK]; k < g.ptr[vK + 1]; k++) {
      const int32_t vJ = g.adj[k];
      if (vJ >= vK) continue;
      auto it = std::lower_bound(mems.begin(), mems.end(), vJ);
      if (it == mems.end() || *it != vJ) continue;
      const int b = (int)(it - mems.begin());
      double ti[3] = {0, 0, 0}, tj[3] = {0, 0, 0};
      for (int d = 0; d < dim; d++) { const double mid = 0.5 * (X[(int64_t)vK * dim + d] + X[(int64_t)vJ * dim + d]); ti[d] = mid - X[(int64_t)vK * dim + d]; tj[d] = mid - X[(int64_t)vJ * dim + d]; }
      rb_Q(dim, ti, Qi);
      rb_Q(dim, tj, Qj);
      em_mtm(BS, Qi, &(*B.E0)[(size_t)k * BB], QiE);
      em_mtm(BS, Qj, &(*B.E0)[(size_t)k * BB], QjE);
      em_mm(BS, QiE, Qi, T); add(a, a, 1.0, T);
      em_mm(BS, QiE, Qj, T); add(a, b, -1.0, T);
      em_mm(BS, QjE, Qi, T); add(b, a, -1.0, T);
      em_mm(BS, QjE, Qj, T); add(b, b, 1.0, T);
    }
  }
  for (int a = 0; a < n; a++) {
    const double* d = &B.aux0[(size_t)mems[a] * BB];
    for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) M[(size_t)(a * BS + r) * N + a * BS + c] = d[r * BS + c];
    double t[3] = {0, 0, 0};
    for (int dd = 0; dd < dim; dd++) t[dd] = X[(int64_t)mems[a] * dim + dd] - X[(int64_t)mems[0] * dim + dd];
    rb_Q(dim, t, Qi);
    for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) P[(size_t)(a * BS + r) * BS + c] = Qi[r * BS + c];
  }
  // PTM = P^T M (BS x N), PTMP = PTM P
  std::vector<double> PTM((size_t)BS * N, 0.0), W((size_t)BS * N, 0.0);
  for (int r = 0; r < BS; r++) for (int c = 0; c < N; c++) { double sm = 0; for (int q = 0; q < N; q++) sm += P[(size_t)q * BS + r] * M[(size_t)q * N + c]; PTM[(size_t)r * N + c] = sm; }
  double PTMP[EM_MAX * EM_MAX];
  for (int r = 0; r < BS; r++) for (int c = 0; c < BS; c++) { double sm = 0; for (int q = 0; q < N; q++) sm += PTM[(size_t)r * N + q] * P[(size_t)q * BS + c]; PTMP[r * BS + c] = sm; }
  pseudo_inverse_with_tol(PTMP, BS);
  for (int r = 0; r < BS; r++) for (int c = 0; c < N; c++) { double sm = 0; for (int q = 0; q < BS; q++) sm += PTMP[r * BS + q] * PTM[(size_t)q * N + c]; W[(size_t)r * N + c] = sm; }
  double maxd = 0;
  for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) {
    double sm = 0;
    for (int q = 0; q < BS; q++) sm += PTM[(size_t)q * N + r] * W[(size_t)q * N + c];
    A[(size_t)r * N + c] -= rho * (M[(size_t)r * N + c] - sm);
  }
  for (int r = 0; r < N; r++) for (int c = r + 1; c < N; c++) { const double m = 0.5 * (A[(size_t)r * N + c] + A[(size_t)c * N + r]); A[(size_t)r * N + c] = A[(size_t)c * N + r] = m; }
  for (int r = 0; r < N; r++) maxd = std::max(maxd, std::fabs(A[(size_t)r * N + r]));
  std::vector<double> ev(N), V((size_t)N * N);
  sym_eig(A.data(), N, ev.data(), V.data());
  double lmin = ev[0];
  for (int r = 1; r < N; r++) lmin = std::min(lmin, ev[r]);
  return lmin >= -1e-10 * maxd;
}

static int32_t spw_find_partner_robust(const Graph& g, const RobustData& R, const std::vector<double>& mt, const std::vector<uint8_t>& handled, int64_t v,
                                       const BigSocData* big = nullptr) {
  double mx = 0.0;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
    const double den = spw_avg(R.pick_avg, mt[v], mt[g.adj[k]]);
    if (den > 0.0) mx = std::max(mx, g.w[k] / den);
  }
  if (!(mx > 0.0)) return -1;
  const double th = SPW_REL_THRESH * mx;
  int32_t best = -1;
  double bw = -1.0;
  std::vector<std::pair<double, int32_t>> cand;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
    const int32_t j = g.adj[k];
    if (handled[j]) continue;
    const double den = spw_avg(R.pick_avg, mt[v], mt[j]);
    if (!(den > 0.0) || g.w[k] / den < th) continue;
    if (!R.pick_robust) { cand.push_back({g.w[k] / den, (int32_t)(k - g.ptr[v])}); continue; }
    const double w = robust_soc(g, R, v, k);
    if (big) cand.push_back({w, j});
    if (w > bw) { bw = w; best = j; }
  }
  const double wth = std::min(th, ROB_EDGE_THRESH);
  if (!R.pick_robust) {
    // robustPick = false (FindNeib3Step, spw_agg_impl.hpp:722-765): the scalar order decides, the robust number only vetoes: the
    // strongest scalar candidate whose robust strength reaches min(scalar threshold, edge_thresh) (and passes the aggregate-wide check)
    std::stable_sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    for (const auto& c : cand) {
      const int64_t k = g.ptr[v] + c.second;
      if (robust_soc(g, R, v, k) < wth) continue;
      if (big && !big_soc_ok(*big, (int32_t)v, g.adj[k], wth)) continue;
      return g.adj[k];
    }
    return -1;
  }
  if (!big) return (best >= 0 && bw >= wth) ? best : -1;
  // strongest first, the first one that also passes the aggregate-wide check (rho = min(robust threshold, absBigThresh = edge_thresh))
  std::stable_sort(cand.begin(), cand.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  for (const auto& c : cand) {
    if (c.first < wth) break;
    if (big_soc_ok(*big, (int32_t)v, c.second, wth)) return c.second;
  }
  return -1;
}

// orphan round (FindNeighborToJoin with robustPick, spw_agg_impl.hpp:870-940): scalar filter w_Oj / maxTrOD_O >= 0.25 max, survivors
// re-weighted with CalcRobJoinSOC (agglomerator_utils.hpp:1086-1126): the boosted edge matrix against the ORPHAN's aux diagonal
// alone (smallest generalised eigenvalue; the reference takes the diagonal in the orphan's own frame -- restated as written)
static int32_t spw_find_join_robust(const Graph& g, const RobustData& R, const std::vector<double>& mt, const std::vector<uint8_t>& joinable, int64_t v) {
  const int BS = em_bs(R.dim), BB = BS * BS;
  if (!(mt[v] > 0.0)) return -1;
  double mx = 0.0;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) mx = std::max(mx, g.w[k] / mt[v]);
  if (!(mx > 0.0)) return -1;
  const double th = SPW_REL_THRESH * mx;
  int32_t best = -1;
  double bw = -1.0;
  for (int64_t k = g.ptr[v]; k < g.ptr[v + 1]; k++) {
    const int32_t j = g.adj[k];
    if (!joinable[j] || g.w[k] / mt[v] < th) continue;
    double E[EM_MAX * EM_MAX], mid[3];
    robust_boosted_edge(g, R, v, k, E, mid);
    const double w = robust_pair_soc(BS, &R.aux[(size_t)v * BB], E);
    if (w > bw) { bw = w; best = j; }
  }
  return (best >= 0 && bw >= std::min(th, ROB_EDGE_THRESH)) ? best : -1;
}

int64_t aggregate_spw(const Graph& G0, const std::vector<uint8_t>& free, const Options& o, std::vector<int32_t>& agg, int& rounds_done,
                      const std::vector<double>* E0 = nullptr, const std::vector<double>* x0 = nullptr) {
  const int64_t n = G0.n;
  agg.assign(n, -1);
  rounds_done = 0;
  int64_t nfree = 0;
  for (int64_t i = 0; i < n; i++) nfree += free[i] ? 1 : 0;
  if (nfree == 0) return 0;
  const int num_rounds = std::max(1, o.spw_rounds);
  Graph cur;
  const Graph* g = &G0;
  std::vector<double> mt(n, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) { double m = 0; for (int64_t k = G0.ptr[i]; k < G0.ptr[i + 1]; k++) m = std::max(m, G0.w[k]); mt[i] = m; }
  std::vector<int32_t> size;               // base vertices per current vertex
  std::vector<int32_t> map;
  int64_t ncur = 0;
  const bool tlog = std::getenv("NGSAMG_SETUP_LOG") != nullptr;
  double tl = omp_get_wtime();
  auto lap = [&](const char* what, int round) {
    if (!tlog) return;
    const double t = omp_get_wtime();
    std::fprintf(stderr, "[setup_levels]     spw round %d  %-22s %8.1f ms\n", round, what, 1e3 * (t - tl));
    tl = t;
  };
  // crs_robust: the rounds carry the edge matrices and the positions of their vertices (SPWAggData::Map, spw_agg_impl.hpp:424-628:
  // a pair sits at the midpoint of its two members, :451-460; edge matrices are moved to the new midpoints, :536-537)
  const bool robust = o.crs_robust && E0 && x0;
  RobustData R;
  R.dim = o.dim;
  std::vector<double> curE, curx;
  if (robust) { R.E = E0; R.x = x0; R.neib_boost = o.spw_neib_boost != 0; R.pick_robust = o.spw_pick_robust != 0; }
  R.pick_avg = o.spw_pick_avg;
  R.in_agg_edge_factor = -2.0 * (1.0 - std::min(1.0, std::max(0.0, o.spw_diag_stab_boost)));      // (spw_agg_impl.hpp:516, 1386-1387)
  const bool cbs = robust && o.spw_cbs;
  BigSocData big;
  for (int round = 0; round < num_rounds; round++) {
    const int64_t m = g->n;
    if (robust && round == 0) {
      // base level: aux diagonals from the edges; maxTrOD = the largest average trace of an edge's contribution IN THE VERTEX'S FRAME
      // (VertexAgglomerator::InitializeAggData, agglomerator_impl.hpp:124-147).  Later rounds carry the members' diagonals (below).
      robust_aux_diags(*g, R);
      const int BS = em_bs(o.dim), BB = BS * BS;
#pragma omp parallel for schedule(static)
      for (int64_t v = 0; v < m; v++) {
        double Q[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX], mv = 0.0;
        for (int64_t k = g->ptr[v]; k < g->ptr[v + 1]; k++) {
          double t[3] = {0, 0, 0};
          for (int d = 0; d < o.dim; d++) t[d] = 0.5 * ((*R.x)[(int64_t)g->adj[k] * o.dim + d] - (*R.x)[v * o.dim + d]);
          rb_Q(o.dim, t, Q);
          em_qtmq(BS, Q, &(*R.E)[(size_t)k * BB], T);
          double tr = 0;
          for (int r = 0; r < BS; r++) tr += T[r * BS + r];
          mv = std::max(mv, tr / BS);
        }
        mt[v] = mv;
      }
    }
    if (cbs && round == 0) { big.g0 = &G0; big.E0 = E0; big.x0 = x0; big.dim = o.dim; big.aux0 = R.aux; }
    if (cbs && round > 0) {        // base-level members of this round's vertices
      big.mptr.assign(m + 1, 0);
      for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) big.mptr[agg[i] + 1]++;
      for (int64_t I = 0; I < m; I++) big.mptr[I + 1] += big.mptr[I];
      big.mem.resize(big.mptr[m]);
      std::vector<int64_t> pos(big.mptr.begin(), big.mptr.end() - 1);
      for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) big.mem[pos[agg[i]]++] = (int32_t)i;
    }
    const BigSocData* bigp = (cbs && round > 0) ? &big : nullptr;
    std::vector<uint8_t> handled(m, 0);
    if (round == 0) for (int64_t i = 0; i < m; i++) handled[i] = free[i] ? 0 : 1;
    map.assign(m, -1);
    int64_t nn = 0;
    auto make_pair = [&](int64_t v) {
      const int32_t nb = robust ? spw_find_partner_robust(*g, R, mt, handled, v, bigp) : spw_find_partner(*g, mt, handled, v, false, nullptr, o.spw_pick_avg);
      const int32_t cv = (int32_t)nn++;
      if (nb >= 0) { map[nb] = cv; handled[nb] = 1; }
      map[v] = cv;
      handled[v] = 1;
    };
    std::vector<int32_t> ld;
    for (int64_t v = m - 1; v >= 0; v--) {
      if (handled[v]) continue;
      if (round == 0) {
        // neighbours of lower degree first, by ascending degree (IterateVertsRev<PREFER_LDEG = true>)
        const int64_t deg = g->ptr[v + 1] - g->ptr[v];
        ld.clear();
        for (int64_t k = g->ptr[v]; k < g->ptr[v + 1]; k++) {
          const int32_t j = g->adj[k];
          if (!handled[j] && g->ptr[j + 1] - g->ptr[j] < deg) ld.push_back(j);
        }
        std::stable_sort(ld.begin(), ld.end(), [&](int32_t a, int32_t b) { return g->ptr[a + 1] - g->ptr[a] < g->ptr[b + 1] - g->ptr[b]; });
        for (int32_t j : ld) if (!handled[j]) make_pair(j);
      }
      if (!handled[v]) make_pair(v);
    }
    lap("pairing", round);
    if (nn == 0) break;
    // compose with the base-level map, contract the graph, carry the scales
    if (round == 0) { for (int64_t i = 0; i < n; i++) agg[i] = free[i] ? map[i] : -1; }
    else for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) agg[i] = map[agg[i]];
    Graph next;
    if (robust) {
      const int dim = o.dim;
      std::vector<double> nx((size_t)nn * dim, 0.0), nE;
      std::vector<int32_t> cnt(nn, 0);
      for (int64_t i = 0; i < m; i++) if (map[i] >= 0) { cnt[map[i]]++; for (int d = 0; d < dim; d++) nx[(int64_t)map[i] * dim + d] += (*R.x)[i * dim + d]; }
      for (int64_t I = 0; I < nn; I++) for (int d = 0; d < dim; d++) nx[I * dim + d] /= std::max(1, cnt[I]);
      next = contract_edge_mats(*g, *R.E, dim, map, nn, *R.x, nx, nE);
      // inside the rounds the scalar weight of a contracted edge is the SUM of its fine edges' weights (cEdgeTrace, spw_agg_impl.hpp:542),
      // not the trace of the transformed matrix (that is the rule from level to level)
      {
        Graph sw = contract(*g, map, nn);
        if (sw.adj != next.adj) throw Error("aggregate_spw: contracted graphs differ");
        next.w = std::move(sw.w);
      }
      // aux diagonal of a merged vertex = its members' diagonals moved to its position (SPWAggData::Map, spw_agg_impl.hpp:462-486)
      // minus (1 - diagStabBoost) of what the edges that vanished inside it had put there (:516, 545-566; diagStabBoost = 0.5,
      // spw_agg.hpp:42: every such edge sits in both members' diagonals, one of the two copies is taken out)
      {
        const int BS = em_bs(dim), BB = BS * BS;
        std::vector<double> naux((size_t)nn * BB, 0.0);
        double Q[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX];
        for (int64_t i = 0; i < m; i++) {
          const int32_t I = map[i];
          if (I < 0) continue;
          double t[3] = {0, 0, 0};
          for (int d = 0; d < dim; d++) t[d] = (*R.x)[i * dim + d] - nx[(int64_t)I * dim + d];
          rb_Q(dim, t, Q);
          em_qtmq(BS, Q, &R.aux[(size_t)i * BB], T);
          for (int x = 0; x < BB; x++) naux[(size_t)I * BB + x] += T[x];
          for (int64_t k = g->ptr[i]; k < g->ptr[i + 1]; k++) {
            const int32_t j = g->adj[k];
            if (j <= i || map[j] != I) continue;            // every in-aggregate edge once
            for (int d = 0; d < dim; d++) t[d] = 0.5 * ((*R.x)[i * dim + d] + (*R.x)[(int64_t)j * dim + d]) - nx[(int64_t)I * dim + d];
            rb_Q(dim, t, Q);
            em_qtmq(BS, Q, &(*R.E)[(size_t)k * BB], T);
            for (int x = 0; x < BB; x++) naux[(size_t)I * BB + x] += R.in_agg_edge_factor * T[x];
          }
        }
        R.aux = std::move(naux);
      }
      curE = std::move(nE); curx = std::move(nx);
      R.E = &curE; R.x = &curx;
    } else next = contract(*g, map, nn);
    lap("contract", round);
    std::vector<double> nmt(nn, 0.0);
    for (int64_t I = 0; I < nn; I++) for (int64_t k = next.ptr[I]; k < next.ptr[I + 1]; k++) nmt[I] = std::max(nmt[I], next.w[k]);
    for (int64_t i = 0; i < m; i++) if (map[i] >= 0) nmt[map[i]] = std::max(nmt[map[i]], mt[i]);
    mt = std::move(nmt);
    cur = std::move(next);
    g = &cur;
    ncur = nn;
    rounds_done++;
  }
  if (rounds_done == 0) return 0;
  // orphan round
  size.assign(ncur, 0);
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) size[agg[i]]++;
  std::vector<int32_t> fin(ncur, -1);
  int64_t nn = 0;
  bool any_orphan = false;
  std::vector<uint8_t> joinable(ncur, 0);
  for (int64_t I = 0; I < ncur; I++) { if (size[I] > 1) { joinable[I] = 1; fin[I] = (int32_t)nn++; } else any_orphan = true; }
  if (o.spw_orphan_round && any_orphan) {
    for (int64_t I = ncur - 1; I >= 0; I--) {
      if (joinable[I]) continue;
      const int32_t J = robust ? spw_find_join_robust(cur, R, mt, joinable, I) : spw_find_partner(cur, mt, joinable, I, true, &joinable);
      fin[I] = J >= 0 ? fin[J] : (int32_t)nn++;
    }
  } else {
    for (int64_t I = 0; I < ncur; I++) if (fin[I] < 0) fin[I] = (int32_t)nn++;
  }
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) agg[i] = fin[agg[i]];
  return nn;
}

// scalar prolongation weights (n_f x n_c CSR): aux-matrix smoothed piecewise prolongation
BCSR prolongation_weights(const Graph& G0, const std::vector<int32_t>& agg, int64_t nc, const Options& o) {
  const int64_t n = G0.n;
  BCSR W;
  W.n_rows = n; W.n_cols = nc; W.br = W.bc = 1;
  const int maxr = std::max(1, o.sp_max_per_row);
  std::vector<int32_t> cols((size_t)n * maxr);
  std::vector<double> vals((size_t)n * maxr);
  std::vector<int32_t> len(n, 0);
#pragma omp parallel
  {
    std::vector<std::pair<int32_t, double>> cand;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const int32_t I = agg[i];
      if (I < 0) continue;
      cand.clear();
      double total = 0;
      if (o.enable_sp)
        for (int64_t k = G0.ptr[i]; k < G0.ptr[i + 1]; k++) {
          int32_t J = agg[G0.adj[k]];
          if (J < 0) continue;
          total += G0.w[k];
          bool found = false;
          for (auto& c : cand) if (c.first == J) { c.second += G0.w[k]; found = true; break; }
          if (!found) cand.push_back({J, G0.w[k]});
        }
      int32_t* oc = &cols[(size_t)i * maxr];
      double* ov = &vals[(size_t)i * maxr];
      if (total <= 0) { oc[0] = I; ov[0] = 1.0; len[i] = 1; continue; }
      for (auto& c : cand) c.second *= o.sp_omega / total;
      if (o.sp_omega != 1.0) {
        bool found = false;
        for (auto& c : cand) if (c.first == I) { c.second += 1.0 - o.sp_omega; found = true; break; }
        if (!found) cand.push_back({I, 1.0 - o.sp_omega});
      }
      // The vertex's own aggregate is always kept ("hierarchic" prolongation): otherwise an aggregate whose
      // members all prefer their neighbours would end up with an empty column of P, i.e. a zero row in P^T A P.
      {
        bool found = false;
        for (auto& c : cand) if (c.first == I) { found = true; break; }
        if (!found) cand.push_back({I, 0.0});
        for (auto& c : cand) if (c.first == I) c.second = std::max(c.second, o.sp_min_frac);
      }
      std::sort(cand.begin(), cand.end(), [I](auto& a, auto& b) {
        if ((a.first == I) != (b.first == I)) return a.first == I;      // own aggregate first
        return a.second > b.second || (a.second == b.second && a.first < b.first);
      });
      int keep = 0;
      double s = 0;
      for (auto& c : cand) {
        if (keep >= maxr) break;
        if (keep > 0 && c.second < o.sp_min_frac) break;
        keep++; s += c.second;
      }
      std::sort(cand.begin(), cand.begin() + keep, [](auto& a, auto& b) { return a.first < b.first; });
      for (int q = 0; q < keep; q++) { oc[q] = cand[q].first; ov[q] = cand[q].second / s; }
      len[i] = keep;
    }
  }
  W.rowptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; i++) W.rowptr[i + 1] = W.rowptr[i] + len[i];
  W.col.resize(W.rowptr[n]);
  W.val.resize(W.rowptr[n]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int q = 0; q < len[i]; q++) { W.col[W.rowptr[i] + q] = cols[(size_t)i * maxr + q]; W.val[W.rowptr[i] + q] = vals[(size_t)i * maxr + q]; }
  return W;
}

// Smoothed prolongation of the reference for scalar levels (VertexAMGFactory::SemiAuxSProlMap, vertex_factory_impl.hpp:1836-2290,
// and its aux-only form): per fine vertex i in aggregate I
//   * no edge neighbour in I: the piecewise row (1 at I);
//   * "classic" (semi_aux only, :1919-1950, :2060-2135): if every algebraic neighbour of row i of the level matrix is in an
//     aggregate and they cover at most sp_max_per_row_classic aggregates, the row of (I - omega D^-1 A) P_pw;
//   * else "aux" (:1952-2017, :2137-2262): columns = I plus the aggregates of the edge neighbours in order of decreasing summed
//     edge weight while weight > sp_min_frac * (0.2 in-weight + weights so far) and >= sp_min_frac * (largest edge weight at i),
//     at most sp_max_per_row; values = the row of (I - omega Dr^-1 R) P_pw with the replacement matrix R of the edges to the
//     used neighbours (R_ij = -w_ij, R_ii = their sum).
// The reference sorts the candidate columns with an unstable sort; ties are broken here by first appearance.
BCSR prolongation_weights_ref(const BCSR* A, const Graph& G, const std::vector<int32_t>& agg, int64_t nc, const Options& o) {
  const int64_t n = G.n;
  const int maxr = std::max(1, o.sp_max_per_row), maxc = std::max(1, o.sp_max_per_row_classic), cap = std::max(maxr, maxc);
  const double minf = o.sp_min_frac, omega = o.sp_omega;
  BCSR W;
  W.n_rows = n; W.n_cols = nc; W.br = W.bc = 1;
  std::vector<int32_t> cols((size_t)n * cap);
  std::vector<double> vals((size_t)n * cap);
  std::vector<int32_t> len(n, 0);
#pragma omp parallel
  {
    std::vector<std::pair<int32_t, double>> trow;
    std::vector<int32_t> cc;
    std::vector<double> vv;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const int32_t I = agg[i];
      if (I < 0) continue;
      int32_t* oc = &cols[(size_t)i * cap];
      double* ov = &vals[(size_t)i * cap];
      int nniscv = 0;
      for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) if (agg[G.adj[k]] == I) nniscv++;
      if (nniscv == 0 || !o.enable_sp) { oc[0] = I; ov[0] = 1.0; len[i] = 1; continue; }
      cc.clear();
      bool classic = false;
      if (A) {
        classic = true;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
          const int32_t cj = agg[A->col[k]];
          if (cj < 0) { classic = false; break; }
          auto it = std::lower_bound(cc.begin(), cc.end(), cj);
          if (it == cc.end() || *it != cj) cc.insert(it, cj);
        }
        classic = classic && (int)cc.size() <= maxc;
      }
      if (classic) {
        if (cc.size() == 1) { oc[0] = cc[0]; ov[0] = 1.0; len[i] = 1; continue; }
        vv.assign(cc.size(), 0.0);
        double aii = 0.0;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) if (A->col[k] == i) { aii = A->val[k]; break; }
        const double d = 1.0 / aii;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
          const int32_t j = A->col[k];
          const size_t ci = std::lower_bound(cc.begin(), cc.end(), agg[j]) - cc.begin();
          if (j == i) vv[ci] += 1.0;
          vv[ci] -= omega * d * A->val[k];
        }
      } else {
        trow.clear();
        double in_wt = 0.0, dgwt = 0.0;
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          dgwt = std::max(dgwt, G.w[k]);
          const int32_t J = agg[G.adj[k]];
          if (J < 0) continue;
          if (J == I) { in_wt += G.w[k]; continue; }
          bool found = false;
          for (auto& t : trow) if (t.first == J) { t.second += G.w[k]; found = true; break; }
          if (!found) trow.push_back({J, G.w[k]});
        }
        std::stable_sort(trow.begin(), trow.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
        double cw_sum = 0.2 * in_wt;
        cc.assign(1, I);
        const size_t max_adds = std::min<size_t>((size_t)(maxr - 1), trow.size());
        for (size_t j = 0; j < max_adds; j++) {
          cw_sum += trow[j].second;
          if (!(trow[j].second > minf * cw_sum) || trow[j].second < minf * dgwt) break;
          cc.push_back(trow[j].first);
        }
        std::sort(cc.begin(), cc.end());
        if (cc.size() == 1) { oc[0] = I; ov[0] = 1.0; len[i] = 1; continue; }
        vv.assign(cc.size(), 0.0);
        double rii = 0.0;
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          const int32_t J = agg[G.adj[k]];
          if (J >= 0 && std::binary_search(cc.begin(), cc.end(), J)) rii += G.w[k];
        }
        const double d = 1.0 / rii;
        const size_t cI = std::lower_bound(cc.begin(), cc.end(), I) - cc.begin();
        vv[cI] += 1.0;
        vv[cI] -= omega * d * rii;
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          const int32_t J = agg[G.adj[k]];
          if (J < 0 || !std::binary_search(cc.begin(), cc.end(), J)) continue;
          vv[std::lower_bound(cc.begin(), cc.end(), J) - cc.begin()] -= omega * d * (-G.w[k]);
        }
      }
      for (size_t q = 0; q < cc.size(); q++) { oc[q] = cc[q]; ov[q] = vv[q]; }
      len[i] = (int32_t)cc.size();
    }
  }
  W.rowptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; i++) W.rowptr[i + 1] = W.rowptr[i] + len[i];
  W.col.resize(W.rowptr[n]);
  W.val.resize(W.rowptr[n]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int q = 0; q < len[i]; q++) { W.col[W.rowptr[i] + q] = cols[(size_t)i * cap + q]; W.val[W.rowptr[i] + q] = vals[(size_t)i * cap + q]; }
  return W;
}

// block prolongation from scalar weights
BCSR block_prolongation(const BCSR& W, int bs_f, int bs_c, int dim, int energy,
                        const std::vector<double>& xf, const std::vector<double>& xc) {
  if (bs_f == 1 && bs_c == 1) return W;
  BCSR P;
  P.n_rows = W.n_rows; P.n_cols = W.n_cols; P.br = bs_f; P.bc = bs_c;
  P.rowptr = W.rowptr; P.col = W.col;
  P.val.assign((size_t)W.nnz() * bs_f * bs_c, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < W.n_rows; i++)
    for (int64_t k = W.rowptr[i]; k < W.rowptr[i + 1]; k++) {
      double* b = &P.val[(size_t)k * bs_f * bs_c];
      const double w = W.val[k];
      if (energy == 0) {
        for (int r = 0; r < std::min(bs_f, bs_c); r++) b[r * bs_c + r] = w;
        continue;
      }
      // rigid body block: displacement rows [I, R(t)], rotation rows [0, I]; u = u_c + w_c x t
      const int32_t J = W.col[k];
      double t[3] = {0, 0, 0};
      for (int d = 0; d < dim; d++) t[d] = xf[i * dim + d] - xc[(int64_t)J * dim + d];
      for (int r = 0; r < dim; r++) b[r * bs_c + r] = w;
      if (dim == 2) {
        if (bs_c > 2) { b[0 * bs_c + 2] = -w * t[1]; b[1 * bs_c + 2] = w * t[0]; }
        if (bs_f > 2) b[2 * bs_c + 2] = w;
      } else {
        if (bs_c > 3) {
          b[0 * bs_c + 4] = w * t[2];  b[0 * bs_c + 5] = -w * t[1];
          b[1 * bs_c + 3] = -w * t[2]; b[1 * bs_c + 5] = w * t[0];
          b[2 * bs_c + 3] = w * t[1];  b[2 * bs_c + 4] = -w * t[0];
        }
        for (int r = dim; r < bs_f; r++) b[r * bs_c + r] = w;
      }
    }
  return P;
}


// Edge matrices of the finest level from the assembled matrix (VertexAMGPC::BuildAlgMesh_ALG_blk, elasticity_pc_impl.hpp:446-488):
//   x = (sum_r |a_rr| + 2 sum_{r<c} |a_rc| over the block A_ij) / (bs^2 sqrt(sum_i sum_j)), sum_v = trace of the diagonal block of
//   v.  As written, the reference's second loop reads `MAT(dis[j], dis[j])`, i.e. takes BOTH traces from the edge's first vertex; with
//   that an edge from a soft vertex to a stiff one is as strong as the soft vertex's other edges whenever the soft vertex has the
//   lower number, and the material-jump problems of its own tests (tests/elasticity/mdim/jump, budget 50) need 56-68 iterations
//   here instead of 15-18: the geometric mean of the two vertices' traces, which the formula's sqrt(sum_i sum_j) spells, is what
//   this setup uses (NGSAMG_EMAT_NORM=literal: the line as written);
//   displacement-only vertices: E = x t t^T on the displacement part (a spring along the edge), vertices with rotations: E = x I.
// The graph holds the edges between free vertices (the reference gives edges at Dirichlet vertices a dummy value that no formula
// reads: those vertices are not mapped); its scalar weight is ENERGY::GetApproxWeight = trace(E) / BS (elasticity_energy.hpp:690-696).
Graph fine_edge_mats(const BCSR& A, const std::vector<uint8_t>& free, const std::vector<double>& xf, int dim, std::vector<double>& E) {
  const int BS = em_bs(dim), bs = A.br, BB = BS * BS;
  Graph G = strength_graph(A, free, dim, 1);
  const int64_t n = G.n;
  std::vector<double> tr(n, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) { double s = 0; for (int r = 0; r < bs; r++) s += A.val[(k * bs + r) * bs + r]; tr[i] = s; break; }
  E.assign((size_t)G.ptr[n] * BB, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    if (!free[i]) continue;
    int64_t p = G.ptr[i];
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      const int32_t j = A.col[k];
      if (j == i || !free[j]) continue;
      const double* b = &A.val[(size_t)k * bs * bs];
      const double* bu = b;
      // the block as the reference reads it: rows of the edge's first vertex, columns of its second (the transposed block seen from j)
      double x = 0;
      for (int r = 0; r < bs; r++) {
        x += std::fabs(bu[r * bs + r]);
        for (int c = r + 1; c < bs; c++) x += 2.0 * std::fabs(j > i ? bu[r * bs + c] : bu[c * bs + r]);
      }
      static const bool literal = [] { const char* e = std::getenv("NGSAMG_EMAT_NORM"); return e && std::string(e) == "literal"; }();
      const double s0 = tr[std::min<int64_t>(i, j)];
      const double nrm = (double)(bs * bs) * (literal ? std::sqrt(s0 * s0) : std::sqrt(tr[i] * tr[j]));
      x = nrm > 0.0 ? x / nrm : 0.0;
      double* e = &E[(size_t)p * BB];
      if (bs == dim) {
        double t[3] = {0, 0, 0}, len = 0;
        for (int d = 0; d < dim; d++) { t[d] = xf[(int64_t)j * dim + d] - xf[i * dim + d]; len += t[d] * t[d]; }
        len = std::sqrt(len);
        if (len > 0.0) { for (int r = 0; r < dim; r++) for (int c = 0; c < dim; c++) e[r * BS + c] = x * (t[r] / len) * (t[c] / len); }
        else for (int r = 0; r < dim; r++) e[r * BS + r] = x;        // coincident vertices: no direction to project on
      } else {
        for (int r = 0; r < BS; r++) e[r * BS + r] = x;
      }
      double trE = 0;
      for (int r = 0; r < BS; r++) trE += e[r * BS + r];
      G.w[p] = trE / BS;
      p++;
    }
  }
  return G;
}

// Edge matrices of the next level (AttachedEED::map_data, elasticity_impl.hpp:23-78): E_IJ = sum over the fine edges between the
// aggregates I and J of Q^T E_ij Q with Q = Q(m_ij - m_IJ), the transformation from the coarse edge's midpoint to the fine one's.
// Edges to unmapped vertices go into the reference's vertex weights (read only by its L2-dominance rule, off by default): dropped.
Graph contract_edge_mats(const Graph& g, const std::vector<double>& E, int dim, const std::vector<int32_t>& agg, int64_t nc,
                         const std::vector<double>& xf, const std::vector<double>& xc, std::vector<double>& Ec) {
  const int BS = em_bs(dim), BB = BS * BS;
  const int64_t n = g.n;
  std::vector<int64_t> mptr(nc + 1, 0);
  for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) mptr[agg[i] + 1]++;
  for (int64_t I = 0; I < nc; I++) mptr[I + 1] += mptr[I];
  std::vector<int32_t> mem(mptr[nc]);
  { std::vector<int64_t> pos(mptr.begin(), mptr.end() - 1); for (int64_t i = 0; i < n; i++) if (agg[i] >= 0) mem[pos[agg[i]]++] = (int32_t)i; }
  Graph C;
  C.n = nc;
  std::vector<int64_t> cnt(nc + 1, 0);
#pragma omp parallel
  {
    std::vector<int32_t> nb;
#pragma omp for schedule(static)
    for (int64_t I = 0; I < nc; I++) {
      nb.clear();
      for (int64_t q = mptr[I]; q < mptr[I + 1]; q++) { const int32_t i = mem[q]; for (int64_t k = g.ptr[i]; k < g.ptr[i + 1]; k++) { const int32_t J = agg[g.adj[k]]; if (J >= 0 && J != I) nb.push_back(J); } }
      std::sort(nb.begin(), nb.end());
      cnt[I + 1] = std::unique(nb.begin(), nb.end()) - nb.begin();
    }
  }
  C.ptr.assign(nc + 1, 0);
  for (int64_t I = 0; I < nc; I++) C.ptr[I + 1] = C.ptr[I] + cnt[I + 1];
  C.adj.resize(C.ptr[nc]);
  C.w.assign(C.ptr[nc], 0.0);
  Ec.assign((size_t)C.ptr[nc] * BB, 0.0);
#pragma omp parallel
  {
    std::vector<int32_t> nb;
    double Q[EM_MAX * EM_MAX], EQ[EM_MAX * EM_MAX], QEQ[EM_MAX * EM_MAX];
#pragma omp for schedule(static)
    for (int64_t I = 0; I < nc; I++) {
      nb.clear();
      for (int64_t q = mptr[I]; q < mptr[I + 1]; q++) { const int32_t i = mem[q]; for (int64_t k = g.ptr[i]; k < g.ptr[i + 1]; k++) { const int32_t J = agg[g.adj[k]]; if (J >= 0 && J != I) nb.push_back(J); } }
      std::sort(nb.begin(), nb.end());
      nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
      int32_t* adj = &C.adj[C.ptr[I]];
      std::copy(nb.begin(), nb.end(), adj);
      for (int64_t q = mptr[I]; q < mptr[I + 1]; q++) {
        const int32_t i = mem[q];
        for (int64_t k = g.ptr[i]; k < g.ptr[i + 1]; k++) {
          const int32_t j = g.adj[k], J = agg[j];
          if (J < 0 || J == I) continue;
          const int64_t p = C.ptr[I] + (std::lower_bound(nb.begin(), nb.end(), J) - nb.begin());
          double t[3] = {0, 0, 0};
          for (int d = 0; d < dim; d++) t[d] = 0.5 * (xf[(int64_t)i * dim + d] + xf[(int64_t)j * dim + d]) - 0.5 * (xc[I * dim + d] + xc[(int64_t)J * dim + d]);
          rb_Q(dim, t, Q);
          em_mm(BS, &E[(size_t)k * BB], Q, EQ);
          em_mtm(BS, Q, EQ, QEQ);
          double* e = &Ec[(size_t)p * BB];
          for (int x = 0; x < BB; x++) e[x] += QEQ[x];
        }
      }
      for (int64_t p = C.ptr[I]; p < C.ptr[I + 1]; p++) { double tr = 0; for (int r = 0; r < BS; r++) tr += Ec[(size_t)p * BB + r * BS + r]; C.w[p] = tr / BS; }
    }
  }
  return C;
}

// Matrix-valued smoothed prolongation (VertexAMGFactory::SemiAuxSProlMap for TM = Mat<BS,BS>, vertex_factory_impl.hpp:1836-2290).
// Column selection as in prolongation_weights_ref (the scalar weights are the edges' approximate weights); values:
//   piecewise block  pw(j, J) = Q(x_j - X_J)                                   (ENERGY::CalcQHh, elasticity_energy_impl.hpp:71-77)
//   classic row      P_i = pw_i - omega D^+ sum_j A_ij pw_j,  D = A_ii scaled to trace BS, pseudo-inverse, scaled back (:2088-2137)
//   aux row          replacement-matrix row of the edges to the used neighbours: R_ii = sum_j Qij^T E Qij, R_ij = -Qij^T E Qji with
//                    Qij = Q(m - x_i), Qji = Q(m - x_j); scaled by BS / trace(R_ii), R_ii^+ by CalcPseudoInverseWithTol;
//                    P_i = pw_i - omega R_ii^+ (R_ii pw_i + sum_j R_ij pw_j)                                      (:2140-2262)
// Rigid-body modes are reproduced exactly by every aux row (E acts on differences of rigid-body states at the edge midpoint).
// A: the level matrix with BS x BS blocks, or nullptr (aux only: prol_type 1, or a displacement-only finest level, whose matrix
// the reference's factory does not hold in TM form).  bs_f < BS: the displacement rows of every block (the finest level's embedding,
// elasticity_pc_impl.hpp:668-685).
BCSR prolongation_edge_mats(const BCSR* A, const Graph& G, const std::vector<double>& E, int dim, int bs_f, const std::vector<int32_t>& agg,
                            int64_t nc, const std::vector<double>& xf, const std::vector<double>& xc, const Options& o) {
  const int BS = em_bs(dim), BB = BS * BS;
  const int64_t n = G.n;
  const int maxr = std::max(1, o.sp_max_per_row), maxc = std::max(1, o.sp_max_per_row_classic), cap = std::max(maxr, maxc);
  const double minf = o.sp_min_frac, omega = o.sp_omega;
  std::vector<int32_t> cols((size_t)n * cap), len(n, 0);
  std::vector<uint8_t> kind(n, 0);        // 0 piecewise, 1 classic, 2 aux
#pragma omp parallel
  {
    std::vector<std::pair<int32_t, double>> trow;
    std::vector<int32_t> cc;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const int32_t I = agg[i];
      if (I < 0) continue;
      int32_t* oc = &cols[(size_t)i * cap];
      int nniscv = 0;
      for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) if (agg[G.adj[k]] == I) nniscv++;
      if (nniscv == 0 || !o.enable_sp) { oc[0] = I; len[i] = 1; continue; }
      cc.clear();
      bool classic = false;
      if (A) {
        classic = true;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
          const int32_t cj = agg[A->col[k]];
          if (cj < 0) { classic = false; break; }
          auto it = std::lower_bound(cc.begin(), cc.end(), cj);
          if (it == cc.end() || *it != cj) cc.insert(it, cj);
        }
        classic = classic && (int)cc.size() <= maxc;
      }
      if (!classic) {
        trow.clear();
        double in_wt = 0.0, dgwt = 0.0;
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) {
          dgwt = std::max(dgwt, G.w[k]);
          const int32_t J = agg[G.adj[k]];
          if (J < 0) continue;
          if (J == I) { in_wt += G.w[k]; continue; }
          bool found = false;
          for (auto& t : trow) if (t.first == J) { t.second += G.w[k]; found = true; break; }
          if (!found) trow.push_back({J, G.w[k]});
        }
        std::stable_sort(trow.begin(), trow.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
        double cw_sum = 0.2 * in_wt;
        cc.assign(1, I);
        const size_t max_adds = std::min<size_t>((size_t)(maxr - 1), trow.size());
        for (size_t j = 0; j < max_adds; j++) {
          cw_sum += trow[j].second;
          if (!(trow[j].second > minf * cw_sum) || trow[j].second < minf * dgwt) break;
          cc.push_back(trow[j].first);
        }
        std::sort(cc.begin(), cc.end());
      }
      for (size_t q = 0; q < cc.size(); q++) oc[q] = cc[q];
      len[i] = (int32_t)cc.size();
      kind[i] = cc.size() == 1 ? 0 : classic ? 1 : 2;
    }
  }
  BCSR P;
  P.n_rows = n; P.n_cols = nc; P.br = bs_f; P.bc = BS;
  P.rowptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; i++) P.rowptr[i + 1] = P.rowptr[i] + len[i];
  P.col.resize(P.rowptr[n]);
  P.val.assign((size_t)P.rowptr[n] * bs_f * BS, 0.0);
  auto pw = [&](int64_t j, int32_t J, double* Q) {
    double t[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) t[d] = xf[j * dim + d] - xc[(int64_t)J * dim + d];
    rb_Q(dim, t, Q);
  };
#pragma omp parallel
  {
    std::vector<double> vals((size_t)cap * BB), rm;
    std::vector<int32_t> unb;
    double Qij[EM_MAX * EM_MAX], Qji[EM_MAX * EM_MAX], QM[EM_MAX * EM_MAX], T[EM_MAX * EM_MAX], T2[EM_MAX * EM_MAX], d[EM_MAX * EM_MAX], ds[EM_MAX * EM_MAX], PW[EM_MAX * EM_MAX];
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const int m = len[i];
      if (m == 0) continue;
      const int32_t I = agg[i];
      const int32_t* cc = &cols[(size_t)i * cap];
      std::fill(vals.begin(), vals.begin() + (size_t)m * BB, 0.0);
      auto colidx = [&](int32_t J) -> int { const int32_t* it = std::lower_bound(cc, cc + m, J); return (it != cc + m && *it == J) ? (int)(it - cc) : -1; };
      auto sub_omega_d = [&](int ci, const double* od) {        // vals[ci] -= omega d od
        em_mm(BS, d, od, T2);
        double* v = &vals[(size_t)ci * BB];
        for (int x = 0; x < BB; x++) v[x] -= omega * T2[x];
      };
      if (kind[i] == 0) {
        pw(i, cc[0], &vals[0]);
      } else if (kind[i] == 1) {
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) if (A->col[k] == i) { std::copy(&A->val[(size_t)k * BB], &A->val[(size_t)k * BB] + BB, d); break; }
        double tr = 0;
        for (int r = 0; r < BS; r++) tr += d[r * BS + r];
        const double trinv = (double)BS / tr;
        for (int x = 0; x < BB; x++) d[x] *= trinv;
        pseudo_inverse_with_tol(d, BS);
        for (int x = 0; x < BB; x++) d[x] *= trinv;
        for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
          const int32_t j = A->col[k];
          const int ci = colidx(agg[j]);
          if (ci < 0) continue;
          pw(j, agg[j], PW);
          if (j == i) { double* v = &vals[(size_t)ci * BB]; for (int x = 0; x < BB; x++) v[x] += PW[x]; }
          em_mm(BS, &A->val[(size_t)k * BB], PW, T);
          sub_omega_d(ci, T);
        }
      } else {
        unb.clear();
        for (int64_t k = G.ptr[i]; k < G.ptr[i + 1]; k++) { const int32_t J = agg[G.adj[k]]; if (J >= 0 && colidx(J) >= 0) unb.push_back((int32_t)(k - G.ptr[i])); }
        rm.assign(unb.size() * BB, 0.0);
        for (int x = 0; x < BB; x++) ds[x] = 0.0;
        for (size_t q = 0; q < unb.size(); q++) {
          const int64_t k = G.ptr[i] + unb[q];
          const int32_t j = G.adj[k];
          double ti[3] = {0, 0, 0}, tj[3] = {0, 0, 0};
          for (int dd = 0; dd < dim; dd++) { const double mid = 0.5 * (xf[i * dim + dd] + xf[(int64_t)j * dim + dd]); ti[dd] = mid - xf[i * dim + dd]; tj[dd] = mid - xf[(int64_t)j * dim + dd]; }
          rb_Q(dim, ti, Qij);
          rb_Q(dim, tj, Qji);
          em_mtm(BS, Qij, &E[(size_t)k * BB], QM);      // Qij^T E
          em_mm(BS, QM, Qji, T);
          for (int x = 0; x < BB; x++) rm[q * BB + x] = -T[x];
          em_mm(BS, QM, Qij, T);
          for (int x = 0; x < BB; x++) ds[x] += T[x];
        }
        double tr = 0;
        for (int r = 0; r < BS; r++) tr += ds[r * BS + r];
        if (!(tr > 0.0)) {        // structurally present but numerically empty edges: the piecewise row
          const int cI0 = colidx(I);
          pw(i, I, &vals[(size_t)cI0 * BB]);
          for (int q = 0; q < m; q++) {
            const int64_t p = P.rowptr[i] + q;
            P.col[p] = cc[q];
            for (int r = 0; r < bs_f; r++) for (int c = 0; c < BS; c++) P.val[((size_t)p * bs_f + r) * BS + c] = vals[(size_t)q * BB + r * BS + c];
          }
          continue;
        }
        const double trinv = (double)BS / tr;
        for (auto& v : rm) v *= trinv;
        for (int x = 0; x < BB; x++) { ds[x] *= trinv; d[x] = ds[x]; }
        pseudo_inverse_with_tol(d, BS);
        const int cI = colidx(I);
        pw(i, I, PW);
        { double* v = &vals[(size_t)cI * BB]; for (int x = 0; x < BB; x++) v[x] += PW[x]; }
        em_mm(BS, ds, PW, T);
        sub_omega_d(cI, T);
        for (size_t q = 0; q < unb.size(); q++) {
          const int32_t j = G.adj[G.ptr[i] + unb[q]];
          pw(j, agg[j], PW);
          em_mm(BS, &rm[q * BB], PW, T);
          sub_omega_d(colidx(agg[j]), T);
        }
      }
      for (int q = 0; q < m; q++) {
        const int64_t p = P.rowptr[i] + q;
        P.col[p] = cc[q];
        for (int r = 0; r < bs_f; r++) for (int c = 0; c < BS; c++) P.val[((size_t)p * bs_f + r) * BS + c] = vals[(size_t)q * BB + r * BS + c];
      }
    }
  }
  return P;
}


// sp_improve_its (ngs_amg_sp_improve_its, off by default; SemiAuxSProlMap's last stage with ImproveSProlRow,
// vertex_factory_impl.hpp:1745-1831, 2350-2420): further smoothing steps on the prolongation WITHOUT growing its graph.  Per step
// AP = A P; a row with more than one entry becomes P_i - omega D^+ (AP)_i E with D = A_ii (pseudo-inverse with tolerance for
// blocks) and the coarse extension E: an entry of (AP)_i in a column c of the row's own pattern stays where it is, an entry outside
// the pattern is moved to the row's own aggregate I through the rigid-body transformation Q(X_c - X_I) (identity for H1).
void improve_prolongation(const BCSR& A, BCSR& P, const std::vector<int32_t>& agg, const std::vector<double>& xc, int dim, int energy,
                          double omega, int its) {
  const int br = P.br, bc = P.bc, bb = br * bc;
  if (A.br != br || A.bc != br) throw Error("improve_prolongation: block shapes do not match");
  const bool rb = energy == 1 && bc == em_bs(dim);
  for (int it = 0; it < its; it++) {
    const BCSR AP = matmul(A, P);
#pragma omp parallel
    {
      std::vector<double> up, d(br * br), T(bb), T2(bb);
      double Q[EM_MAX * EM_MAX];
#pragma omp for schedule(static)
      for (int64_t i = 0; i < P.n_rows; i++) {
        const int64_t p0 = P.rowptr[i], len = P.rowptr[i + 1] - p0;
        if (len < 2) continue;
        const int32_t I = agg[i];
        const int32_t* pc = &P.col[p0];
        const int32_t* itI = std::lower_bound(pc, pc + len, I);
        if (itI == pc + len || *itI != I) continue;
        const int64_t posI = itI - pc;
        bool have_d = false;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) { std::copy(&A.val[(size_t)k * br * br], &A.val[(size_t)k * br * br] + br * br, d.begin()); have_d = true; break; }
        if (!have_d) continue;
        if (br == 1) { if (d[0] == 0.0) continue; d[0] = 1.0 / d[0]; }
        else pseudo_inverse_with_tol(d.data(), br);
        up.assign((size_t)len * bb, 0.0);
        for (int64_t k = AP.rowptr[i]; k < AP.rowptr[i + 1]; k++) {
          const int32_t c = AP.col[k];
          const double* v = &AP.val[(size_t)k * bb];
          for (int r = 0; r < br; r++) for (int q = 0; q < bc; q++) { double sm = 0; for (int x = 0; x < br; x++) sm += d[r * br + x] * v[x * bc + q]; T[r * bc + q] = sm; }
          const int32_t* itc = std::lower_bound(pc, pc + len, c);
          if (itc != pc + len && *itc == c) {
            double* u = &up[(size_t)(itc - pc) * bb];
            for (int x = 0; x < bb; x++) u[x] -= omega * T[x];
          } else {
            double* u = &up[(size_t)posI * bb];
            if (rb) {
              double t[3] = {0, 0, 0};
              for (int dd = 0; dd < dim; dd++) t[dd] = xc[(int64_t)c * dim + dd] - xc[(int64_t)I * dim + dd];
              rb_Q(dim, t, Q);
              for (int r = 0; r < br; r++) for (int q = 0; q < bc; q++) { double sm = 0; for (int x = 0; x < bc; x++) sm += T[r * bc + x] * Q[x * bc + q]; T2[r * bc + q] = sm; }
              for (int x = 0; x < bb; x++) u[x] -= omega * T2[x];
            } else {
              for (int x = 0; x < bb; x++) u[x] -= omega * T[x];
            }
          }
        }
        double* pv = &P.val[(size_t)p0 * bb];
        for (size_t x = 0; x < (size_t)len * bb; x++) pv[x] += up[x];
      }
    }
  }
}

}  // namespace

double robust_pair_soc_of(int n, const double* C, const double* E) {
  if (n < 1 || n > EM_MAX) throw Error("robust_pair_soc: block size must be 1 ... 6");
  return robust_pair_soc(n, C, E);
}

template <class Mat>
static void calc_dinv_t(const Mat& A, const uint8_t* free, bool pinv, double* dinv) {
  const int bs = A.br, bb = bs * bs;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.n_rows; i++) {
    double* d = &dinv[i * bb];
    for (int q = 0; q < bb; q++) d[q] = 0.0;
    if (free && !free[i]) continue;
    const double* blk = nullptr;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) { blk = &A.val[k * bb]; break; }
    if (!blk) continue;
    for (int q = 0; q < bb; q++) d[q] = blk[q];
    if (pinv) pseudo_inverse_try_normal(d, bs);
    else if (bs == 1) d[0] = 1.0 / d[0];
    else if (!dense_inverse(d, bs)) pseudo_inverse_try_normal(d, bs);
  }
}

void calc_dinv(const BCSR& A, const uint8_t* free, bool pinv, double* dinv) { calc_dinv_t(A, free, pinv, dinv); }

int greedy_coloring(const BCSR& A, const uint8_t* free, int32_t* color) {
  const int64_t n = A.n_rows;
  int ncol = 0;
  std::vector<int64_t> mark(64, -1);
  for (int64_t i = 0; i < n; i++) {
    color[i] = -1;
    if (free && !free[i]) continue;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      int32_t j = A.col[k];
      if (j >= i) continue;          // only already-coloured (lower) neighbours matter
      int32_t c = color[j];
      if (c >= 0) { if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1); mark[c] = i; }
    }
    int c = 0;
    while (c < (int)mark.size() && mark[c] == i) c++;
    if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1);
    color[i] = c;
    ncol = std::max(ncol, c + 1);
  }
  return ncol;
}

// greedy colouring that only sees couplings INSIDE blocks of `block_rows` consecutive rows (block-hybrid Gauss-Seidel:
// couplings that cross a block boundary are frozen during a sweep, so they put no constraint on the order)
template <class Mat>
static int greedy_coloring_blocked_t(const Mat& A, const uint8_t* free, int64_t block_rows, int32_t* color) {
  const int64_t n = A.n_rows;
  int ncol = 0;
#pragma omp parallel
  {
    std::vector<int64_t> mark(64, -1);
    int loc = 0;
#pragma omp for schedule(dynamic, 16)
    for (int64_t b0 = 0; b0 < n; b0 += block_rows) {
      const int64_t b1 = std::min<int64_t>(n, b0 + block_rows);
      for (int64_t i = b0; i < b1; i++) {
        color[i] = -1;
        if (free && !free[i]) continue;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
          const int64_t j = A.col[k];
          if (j >= i || j < b0) continue;
          const int32_t c = color[j];
          if (c >= 0) { if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1); mark[c] = i; }
        }
        int c = 0;
        while (c < (int)mark.size() && mark[c] == i) c++;
        if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1);
        color[i] = c;
        loc = std::max(loc, c + 1);
      }
    }
#pragma omp critical
    ncol = std::max(ncol, loc);
  }
  return ncol;
}

int greedy_coloring_blocked(const BCSR& A, const uint8_t* free, int64_t block_rows, int32_t* color) { return greedy_coloring_blocked_t(A, free, block_rows, color); }
int greedy_coloring_blocked(const CsrView& A, const uint8_t* free, int64_t block_rows, int32_t* color) { return greedy_coloring_blocked_t(A, free, block_rows, color); }

// inverse of the l1-type modified diagonal of the hybrid smoother (reference CalcModDiag, hybrid_smoother_utils.hpp:35-142):
//   ad_k = sum over the couplings of row k that leave its block of |a_kj| / sqrt(a_kk a_jj);  md_k = max(1, 0.51 (1 + ad_k)) a_kk
// scalar matrices; non-free rows get 0
template <class Mat>
static void hybrid_mod_dinv_t(const Mat& A, const uint8_t* free, int64_t block_rows, double* dinv, const double* ghost_diag) {
  const int64_t n = A.n_rows;
  std::vector<double> d((size_t)A.n_cols, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) { d[i] = A.val[k]; break; }
  if (ghost_diag) for (int64_t j = n; j < A.n_cols; j++) d[j] = ghost_diag[j - n];      // rank-partitioned level: owners' diagonals
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    dinv[i] = 0.0;
    if ((free && !free[i]) || d[i] == 0.0) continue;
    const int64_t b0 = (i / block_rows) * block_rows, b1 = b0 + block_rows;
    double ad = 0.0;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      const int64_t j = A.col[k];
      if (j >= b0 && j < b1) continue;
      const double dj = d[j];
      if (d[i] > 0.0 && dj > 0.0) ad += std::fabs(A.val[k]) / std::sqrt(d[i] * dj);
    }
    dinv[i] = 1.0 / (std::max(1.0, 0.51 * (1.0 + ad)) * d[i]);
  }
}

void hybrid_mod_dinv(const BCSR& A, const uint8_t* free, int64_t block_rows, double* dinv, const double* ghost_diag) { hybrid_mod_dinv_t(A, free, block_rows, dinv, ghost_diag); }
void hybrid_mod_dinv(const CsrView& A, const uint8_t* free, int64_t block_rows, double* dinv, const double* ghost_diag) { hybrid_mod_dinv_t(A, free, block_rows, dinv, ghost_diag); }

// the same for square-block matrices (reference hybrid_smoother_utils.hpp:55-68, 86-98, 128-141): per block row k and scalar
// row l, ad_k(l) = sum over the couplings that leave the block of rows of sum_m |a_kj(l, m)| / sqrt(d_k(l, l) d_j(m, m)); the
// block diagonal is scaled by max(1, max_l 0.51 (1 + ad_k(l))), i.e. dinv_k = (pseudo-)inverse(A_kk) / that factor
template <class Mat>
static void hybrid_mod_dinv_block_t(const Mat& A, const uint8_t* free, int64_t block_rows, bool pinv, double* dinv, const int32_t* block_of_row) {
  const int64_t n = A.n_rows;
  const int bs = A.br, bb = bs * bs;
  std::vector<double> d((size_t)n * bs, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) { for (int r = 0; r < bs; r++) d[i * bs + r] = A.val[k * bb + r * bs + r]; break; }
  calc_dinv_t(A, free, pinv, dinv);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    if (free && !free[i]) continue;
    const int64_t b0 = (i / block_rows) * block_rows, b1 = b0 + block_rows;
    double fac = 1.0;
    for (int l = 0; l < bs; l++) {
      const double dl = d[i * bs + l];
      if (!(dl > 0.0)) continue;
      double ad = 0.0;
      for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
        const int64_t j = A.col[k];
        if (j >= n) continue;                 // (ghost columns: rank-partitioned levels use the distributed setup's own routine)
        if (block_of_row ? block_of_row[j] == block_of_row[i] : (j >= b0 && j < b1)) continue;
        for (int m = 0; m < bs; m++) {
          const double dm = d[j * bs + m];
          if (dm > 0.0) ad += std::fabs(A.val[k * bb + l * bs + m]) / std::sqrt(dl * dm);
        }
      }
      fac = std::max(fac, 0.51 * (1.0 + ad));
    }
    if (fac != 1.0) for (int q = 0; q < bb; q++) dinv[i * bb + q] /= fac;
  }
}

void hybrid_mod_dinv_block(const BCSR& A, const uint8_t* free, int64_t block_rows, bool pinv, double* dinv, const int32_t* block_of_row) {
  hybrid_mod_dinv_block_t(A, free, block_rows, pinv, dinv, block_of_row);
}
void hybrid_mod_dinv_block(const CsrView& A, const uint8_t* free, int64_t block_rows, bool pinv, double* dinv, const int32_t* block_of_row) {
  hybrid_mod_dinv_block_t(A, free, block_rows, pinv, dinv, block_of_row);
}

// ---------------------------------------------------------------------------------------------------------------------
// Compact sweep blocks for the block-hybrid Gauss-Seidel of block levels.  The reference's hybrid smoother freezes the
// couplings between MPI subdomains, which a mesh partitioner makes compact; blocks of consecutive rows of a lexicographically
// numbered grid are grid LINES (2 of 14 neighbours inside the block), and the frozen fraction costs iterations on the
// elasticity levels (oracle probe, thin beam 10 x 10 x 126 with rotations, PCG 1e-8: sequential 42, multicolour 45,
// line blocks 50, compact blocks 46).  Greedy graph growing: a block starts at the first unassigned free vertex and grows
// breadth-first over unassigned neighbours up to `target` rows; fragments smaller than target / 4 join a neighbouring block
// that still has room (<= max_rows).  Deterministic, O(nnz).  Non-free vertices get blocks of their own at the end (they are
// never swept).  Returns the number of blocks.
int64_t compact_blocks(const BCSR& A, const uint8_t* free, int target, int max_rows, int32_t* block_of_row) {
  const int64_t n = A.n_rows;
  std::vector<int32_t> size;
  std::vector<int32_t> queue;
  for (int64_t i = 0; i < n; i++) block_of_row[i] = -1;
  for (int64_t seed = 0; seed < n; seed++) {
    if (block_of_row[seed] >= 0 || (free && !free[seed])) continue;
    const int32_t b = (int32_t)size.size();
    queue.clear();
    queue.push_back((int32_t)seed);
    block_of_row[seed] = b;
    size_t head = 0;
    int cnt = 1;
    while (head < queue.size() && cnt < target) {
      const int32_t v = queue[head++];
      for (int64_t k = A.rowptr[v]; k < A.rowptr[v + 1] && cnt < target; k++) {
        const int32_t j = A.col[k];
        if (j >= n || block_of_row[j] >= 0 || (free && !free[j])) continue;
        block_of_row[j] = b;
        queue.push_back(j);
        cnt++;
      }
    }
    size.push_back(cnt);
  }
  // small fragments join a neighbouring block with room
  const int64_t nb0 = (int64_t)size.size();
  std::vector<int32_t> remap(nb0);
  std::iota(remap.begin(), remap.end(), 0);
  for (int64_t i = 0; i < n; i++) {
    const int32_t b = block_of_row[i];
    if (b < 0 || remap[b] != b || size[b] >= std::max(1, target / 4)) continue;
    int32_t best = -1;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      const int32_t j = A.col[k];
      if (j >= n || block_of_row[j] < 0) continue;
      int32_t c = block_of_row[j];
      while (remap[c] != c) c = remap[c];
      if (c != b && size[c] + size[b] <= max_rows && (best < 0 || size[c] < size[best])) best = c;
    }
    if (best >= 0) { remap[b] = best; size[best] += size[b]; size[b] = 0; }
  }
  std::vector<int32_t> newid(nb0, -1);
  int64_t nb = 0;
  for (int64_t b = 0; b < nb0; b++) {
    int32_t c = (int32_t)b;
    while (remap[c] != c) c = remap[c];
    remap[b] = c;
  }
  for (int64_t b = 0; b < nb0; b++) if (remap[b] == b) newid[b] = (int32_t)nb++;
  for (int64_t i = 0; i < n; i++) {
    if (block_of_row[i] >= 0) block_of_row[i] = newid[remap[block_of_row[i]]];
    else block_of_row[i] = (int32_t)nb++;
  }
  return nb;
}

// greedy colouring that only separates coupled rows of the SAME block (arbitrary block ids)
int greedy_coloring_blockids(const BCSR& A, const uint8_t* free, const int32_t* block_of_row, int32_t* color) {
  const int64_t n = A.n_rows;
  int ncol = 0;
  std::vector<int64_t> mark(64, -1);
  for (int64_t i = 0; i < n; i++) {
    color[i] = -1;
    if (free && !free[i]) continue;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      const int64_t j = A.col[k];
      if (j >= i || block_of_row[j] != block_of_row[i]) continue;
      const int32_t c = color[j];
      if (c >= 0) { if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1); mark[c] = i; }
    }
    int c = 0;
    while (c < (int)mark.size() && mark[c] == i) c++;
    if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1);
    color[i] = c;
    ncol = std::max(ncol, c + 1);
  }
  return ncol;
}

Hierarchy* setup_levels(const BCSR& A0, const uint8_t* free0, const double* coords0, const Options& o) {
  auto H = new Hierarchy();
  H->opts = o;
  std::ostringstream log;
  const int dim = o.dim;
  const int nrot = (dim * (dim - 1)) / 2;
  H->levels.emplace_back();
  {
    Level& L = H->levels.back();
    L.A = A0;
    L.free.assign(A0.n_rows, 1);
    if (free0) std::copy(free0, free0 + A0.n_rows, L.free.begin());
    if (coords0) L.coords.assign(coords0, coords0 + A0.n_rows * dim);
    else if (o.energy == 1) throw Error("elasticity setup needs vertex coordinates");
  }
  // edge_mats: the alg-mesh (edges with the energy's matrices) is carried from level to level beside the matrices
  const bool emats = o.edge_mats && o.energy == 1;
  if (emats && (!o.spw || o.robust_soc)) throw Error("edge_mats needs the SPW agglomerator (spw = 1, robust_soc = 0)");
  const bool carry = o.carry_mesh && !emats;
  if (carry && (!o.spw || o.robust_soc)) throw Error("carry_mesh needs the SPW agglomerator (spw = 1, robust_soc = 0)");
  Graph mesh;
  std::vector<double> meshE;
  while (true) {
    const int lev = (int)H->levels.size() - 1;
    Level& F = H->levels[lev];
    const int64_t nf = F.A.n_rows;
    int64_t nfree = 0;
    for (auto f : F.free) nfree += f;
    log << "level " << lev << ": n=" << nf << " bs=" << F.A.br << " nnz=" << F.A.nnz() << " free=" << nfree << "\n";
    if (lev + 1 >= o.max_levels) break;
    if (lev > 0 && nf <= o.max_coarse_size) break;
    if (lev == 0 && nfree <= o.max_coarse_size && nfree == nf) break;
    double t0 = omp_get_wtime();
    const double target = (lev == 0) ? o.first_aaf : o.aaf;
    const int bs_f = F.A.br;
    const int bs_c = (o.energy == 1) ? dim + nrot : bs_f;
    // One coarsening step = pairwise rounds + smoothed prolongation + Galerkin product.  enable_multistep (reference
    // base_factory: "multistep", H1 default, h1_impl.hpp:331): while the level is still larger than its target, another
    // step is taken on the intermediate matrix and the prolongations are CONCATENATED (P = P_1 P_2 ...), the intermediate
    // levels are not smoothed on.  Every sub-step coarsens by ~aaf with its own smoothed prolongation, so the concatenated
    // P interpolates from a wider coarse neighbourhood than one smoothing step over the big final aggregates could.
    Level C;
    std::vector<int32_t> agg;
    std::vector<double> xc;
    int rounds = 0, substeps = 0;
    int64_t nc = 0;
    {
      const BCSR* curA = &F.A;
      BCSR tmpA;
      std::vector<uint8_t> cur_free = F.free;
      std::vector<double> cur_coords = F.coords;
      std::vector<double> cur_vs = F.vscale;
      BCSR Ptot, PTlast;
      bool failed = false;
      const bool tlog = std::getenv("NGSAMG_SETUP_LOG") != nullptr;
      double tl = omp_get_wtime();
      auto lap = [&](const char* what) {
        if (!tlog) return;
        const double t = omp_get_wtime();
        std::fprintf(stderr, "[setup_levels] level %d  %-28s %8.1f ms\n", lev, what, 1e3 * (t - tl));
        tl = t;
      };
      while (true) {
        if (emats && lev == 0 && substeps == 0) mesh = fine_edge_mats(*curA, cur_free, cur_coords, dim, meshE);
        // carry_mesh: the alg-mesh of a coarse level is the contracted mesh of the level above (edges between aggregates, weights
        // summed: BlockTM coarse maps, H1EData::map_data), not the graph of the Galerkin matrix, whose stencil the smoothed
        // prolongation has widened
        const bool carried = carry && !(lev == 0 && substeps == 0);
        Graph G = (emats || carried) ? std::move(mesh) : strength_graph(*curA, cur_free, dim, o.energy);
        lap(emats ? "alg-mesh (edge matrices)" : carried ? "alg-mesh (carried)" : "strength graph");
        if (o.robust_soc) { G.vs = cur_vs; G.vs.resize(G.n, 0.0); }
        const double step_target = (o.enable_multistep && target < o.aaf) ? std::max(target, o.aaf) : target;
        std::vector<int32_t> sagg;
        std::vector<double> next_vs;
        int r = 0;
        int64_t cur_free_n = 0;
        for (auto f : cur_free) cur_free_n += f;
        const int64_t snc = o.spw ? aggregate_spw(G, cur_free, o, sagg, r, emats ? &meshE : nullptr, emats ? &cur_coords : nullptr)
                                  : aggregate(G, cur_free, step_target, o, sagg, r, o.robust_soc ? &next_vs : nullptr);
        lap("agglomeration");
        if (snc == 0 || snc >= cur_free_n) { failed = substeps == 0; break; }
        if (coarse_order_morton() && !cur_coords.empty() && !o.robust_soc) { renumber_morton(sagg, snc, cur_coords, dim); lap("coarse numbering"); }
        // robust_soc: what is left are vertices that must not be merged; a level that is barely smaller than its parent costs a
        // smoother and buys nothing (the coarsest-level inverse takes over)
        if (o.robust_soc && lev > 0 && (double)snc > 0.8 * (double)cur_free_n) { failed = substeps == 0; break; }
        rounds += r;
        // prolongation rule (amgh.h: prol_type); spw = 0 hierarchies keep the weight rule of the earlier rounds
        const int ptype = o.prol_type >= 0 ? o.prol_type : (o.spw ? 2 : 3);
        Options op = o;
        if (ptype == 0) op.enable_sp = 0;
        // block (elasticity) levels: the aux rule on the scalar edge weights for both smoothed types -- column selection and
        // replacement-matrix weights as in the reference, the blocks of P stay rigid-body transformations w Q(t); its matrix-valued
        // classic / aux formulas need the energy's edge matrices, which this setup does not carry
        BCSR W;
        if (!emats) W = ptype == 3 ? prolongation_weights(G, sagg, snc, op)
                                   : prolongation_weights_ref((ptype == 2 && curA->br == 1) ? curA : nullptr, G, sagg, snc, op);
        lap("prolongation weights");
        const int sbf = curA->br;
        std::vector<double> sxc;
        if (!cur_coords.empty()) {
          sxc.assign(snc * dim, 0.0);
          std::vector<int32_t> cnt(snc, 0);
          const int64_t cn = curA->n_rows;
          for (int64_t i = 0; i < cn; i++) if (sagg[i] >= 0) { cnt[sagg[i]]++; for (int d = 0; d < dim; d++) sxc[(int64_t)sagg[i] * dim + d] += cur_coords[i * dim + d]; }
          for (int64_t I = 0; I < snc; I++) for (int d = 0; d < dim; d++) sxc[I * dim + d] /= std::max(1, cnt[I]);
        }
        // edge_mats: the matrix-valued rule; `own` (3) has no such form and takes the reference's default (2)
        BCSR Pk = emats ? prolongation_edge_mats((ptype != 1 && curA->br == bs_c) ? curA : nullptr, G, meshE, dim, sbf, sagg, snc, cur_coords, sxc, op)
                        : block_prolongation(W, sbf, bs_c, dim, o.energy, cur_coords, sxc);
        if (emats) {
          std::vector<double> nextE;
          mesh = contract_edge_mats(G, meshE, dim, sagg, snc, cur_coords, sxc, nextE);
          meshE = std::move(nextE);
        } else if (carry) {
          G.vs.clear();
          mesh = contract(G, sagg, snc);
        }
        if (o.sp_improve_its > 0 && op.enable_sp) {
          improve_prolongation(*curA, Pk, sagg, sxc, dim, o.energy, o.sp_omega, o.sp_improve_its);
          lap("prolongation improve steps");
        }
        if (o.prol_only) {       // one step, P only (amgh.h): no transpose, no Galerkin product; the coarse level is a placeholder
          Ptot = std::move(Pk);
          agg = sagg;
          tmpA = BCSR();
          tmpA.n_rows = tmpA.n_cols = snc; tmpA.br = tmpA.bc = bs_c;
          tmpA.rowptr.assign(snc + 1, 0);
          curA = &tmpA;
          cur_coords = std::move(sxc);
          nc = snc;
          substeps = 1;
          break;
        }
        BCSR PkT = transpose(Pk);
        lap("block prolongation, P^T");
        BCSR nextA = restrict_matrix(PkT, *curA, Pk);
        lap("Galerkin product");
        if (substeps == 0) { Ptot = std::move(Pk); PTlast = std::move(PkT); agg = sagg; }
        else {
          Ptot = matmul(Ptot, Pk);
          for (auto& a : agg) if (a >= 0) a = sagg[a];
        }
        tmpA = std::move(nextA);
        curA = &tmpA;
        cur_free.assign(snc, 1);
        cur_coords = std::move(sxc);
        cur_vs = std::move(next_vs);
        nc = snc;
        substeps++;
        if (!o.enable_multistep || (double)nc <= 1.3 * target * (double)nfree || substeps >= 4) break;
      }
      if (failed || substeps == 0) { log << "  coarsening stuck (nc=" << nc << ")\n"; break; }
      F.P = std::move(Ptot);
      if (!o.prol_only) F.PT = substeps == 1 ? std::move(PTlast) : transpose(F.P);       // (one step: the transpose the Galerkin product used)
      F.agg = agg;
      C.A = std::move(tmpA);
      xc = std::move(cur_coords);
      C.vscale = std::move(cur_vs);
    }
    double t6 = omp_get_wtime();
    log << "  time: coarsening step(s) " << t6 - t0 << " (" << substeps << " sub-step" << (substeps == 1 ? "" : "s") << ")\n";
    C.free.assign(nc, 1);
    C.coords = std::move(xc);
    log << "  rounds=" << rounds << " nc=" << nc << " P nnz=" << F.P.nnz() << "\n";
    H->levels.push_back(std::move(C));
    if (o.prol_only) break;
  }
  if (o.prol_only) {       // placeholders of the right sizes: the caller reads P, agg and the coarse coordinates only
    for (auto& L : H->levels) {
      L.dinv.assign((size_t)L.A.n_rows * L.A.br * L.A.br, 0.0);
      L.color.assign(L.A.n_rows, 0);
      L.n_colors = 0;
    }
    H->coarse_n = 0;
    H->log = log.str();
    return H;
  }
  // smoother data per level
  {
    const bool tlog = std::getenv("NGSAMG_SETUP_LOG") != nullptr;
    double td = 0, tc = 0;
    for (auto& L : H->levels) {
      const double t0 = omp_get_wtime();
      L.dinv.resize((size_t)L.A.n_rows * L.A.br * L.A.br);
      calc_dinv(L.A, L.free.data(), o.regularize_cmats != 0, L.dinv.data());
      const double t1 = omp_get_wtime();
      L.color.resize(L.A.n_rows);
      L.n_colors = greedy_coloring(L.A, L.free.data(), L.color.data());
      td += t1 - t0; tc += omp_get_wtime() - t1;
    }
    if (tlog) std::fprintf(stderr, "[setup_levels] smoother data: inverted diagonals %8.1f ms, greedy colouring %8.1f ms\n", 1e3 * td, 1e3 * tc);
  }
  // coarsest-level inverse on the free dofs (zero rows/cols elsewhere), dense
  {
    Level& L = H->levels.back();
    const int bs = L.A.br;
    const int64_t N = L.A.n_rows * bs;
    // beyond this size the inverse is left to the device (amgx_create with coarse_inv = NULL: blocked Gauss-Jordan on the
    // matrix cores, csrc/device/dense_spd.hpp); NGSAMG_HOST_COARSE_MAX moves the limit (tests force the device path with it)
    int64_t host_max = 4096;
    if (const char* e = std::getenv("NGSAMG_HOST_COARSE_MAX")) host_max = std::atoll(e);
    if (N <= host_max) {
      std::vector<int64_t> fidx;
      for (int64_t i = 0; i < L.A.n_rows; i++) if (L.free[i]) for (int c = 0; c < bs; c++) fidx.push_back(i * bs + c);
      const int64_t nfr = (int64_t)fidx.size();
      std::vector<int64_t> pos(N, -1);
      for (int64_t q = 0; q < nfr; q++) pos[fidx[q]] = q;
      std::vector<double> D((size_t)nfr * nfr, 0.0);
      for (int64_t i = 0; i < L.A.n_rows; i++)
        for (int64_t k = L.A.rowptr[i]; k < L.A.rowptr[i + 1]; k++) {
          int64_t j = L.A.col[k];
          for (int r = 0; r < bs; r++) for (int c = 0; c < bs; c++) {
            int64_t pr = pos[i * bs + r], pc = pos[j * bs + c];
            if (pr >= 0 && pc >= 0) D[pr * nfr + pc] = L.A.val[(k * bs + r) * bs + c];
          }
        }
      // regularize_cmats on an elasticity hierarchy: the reference regularises the DIAGONAL BLOCKS of the coarsest matrix before it
      // inverts it (CoarseLevelInv -> RegularizeMatrix, amg_pc.cpp:861-862; elasticity_pc_impl.hpp:710-764): 3D RegTM<0,6,6>
      // (utils_denseLA.hpp:1198-1233): the smallest non-zero eigenvalue of the block is added along its kernel (identity if the
      // block is zero); 2D: a rotation diagonal below 1e-8 becomes 1.  A block without kernel is left alone, so coarsest matrices
      // that are positive definite -- every hierarchy the oracle accepts -- invert exactly as before.
      if (o.regularize_cmats && o.energy == 1 && bs == dim + nrot) {
        int64_t nreg = 0;
        for (int64_t i = 0; i < L.A.n_rows; i++) {
          if (!L.free[i]) continue;
          const int64_t p0 = pos[i * bs];
          auto at = [&](int r, int c) -> double& { return D[(size_t)(p0 + r) * nfr + (p0 + c)]; };     // free rows are contiguous per block
          if (dim == 2) {
            if (std::fabs(at(2, 2)) < 1e-8) { at(2, 2) = 1.0; nreg++; }
            continue;
          }
          double M[36], ev[6], V[36];
          for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) M[r * 6 + c] = 0.5 * (at(r, c) + at(c, r));
          sym_eig(M, 6, ev, V);
          double emax = ev[0];
          for (int q = 1; q < 6; q++) emax = std::max(emax, ev[q]);
          const double eps = std::max(1e-15, 1e-12 * emax);
          double min_nz = 0.0;
          int nzero = 0;
          for (int q = 0; q < 6; q++) { if (ev[q] > eps) min_nz = (min_nz == 0.0) ? ev[q] : std::min(min_nz, ev[q]); else nzero++; }
          if (nzero == 0) continue;
          nreg++;
          if (nzero < 6) {
            for (int q = 0; q < 6; q++) if (!(ev[q] > eps))
              for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) at(r, c) += min_nz * V[r * 6 + q] * V[c * 6 + q];
          } else {
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) at(r, c) = r == c ? 1.0 : 0.0;
          }
        }
        if (nreg) log << "  coarsest level: " << nreg << " diagonal block(s) regularised\n";
      }
      bool chol = nfr > 0 ? spd_inverse(D.data(), (int)nfr) : true;
      if (!chol) log << "  coarse matrix not SPD: pseudo-inverse used\n";
      H->coarse_n = N;
      H->coarse_inv.assign((size_t)N * N, 0.0);
      for (int64_t a = 0; a < nfr; a++) for (int64_t b = 0; b < nfr; b++) H->coarse_inv[fidx[a] * N + fidx[b]] = D[a * nfr + b];
    } else {
      H->coarse_n = 0;   // left to the device
      log << "  coarsest level (" << N << " unknowns): dense inverse left to the device\n";
    }
  }
  H->log = log.str();
  return H;
}

}  // namespace amgh
