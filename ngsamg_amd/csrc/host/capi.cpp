// C ABI of the host setup library, see include/amgh.h.
#include "../../../include/amgh.h"
#include "hierarchy.hpp"
#include <cstring>
#include <memory>

struct amgh_hierarchy { std::unique_ptr<amgh::Hierarchy> h; };

namespace amgh {
void kuhn_pattern(int dim, const int64_t* shape, int64_t* rowptr);
void kuhn_assemble(int dim, const int64_t* shape, const double* coords, int kind, int bs, double mu, double lam,
                   const double* cell_coef, const int64_t* rowptr, int32_t* col, double* val, double* load);
}

namespace {
thread_local std::string g_err;
thread_local std::unique_ptr<amgh::BCSR> g_mm_cache;

amgh::BCSR to_bcsr(const amgh_matrix* m) {
  amgh::BCSR A;
  A.n_rows = m->n_rows; A.n_cols = m->n_cols; A.br = m->br; A.bc = m->bc;
  A.rowptr.assign(m->rowptr, m->rowptr + m->n_rows + 1);
  const int64_t nnz = A.rowptr.back();
  A.col.assign(m->col, m->col + nnz);
  A.val.assign(m->val, m->val + nnz * m->br * m->bc);
  return A;
}

// pattern only (no 8.5 GB copy of the values of a cfg-5 level for routines that look at the graph)
amgh::BCSR to_bcsr_pattern(const amgh_matrix* m) {
  amgh::BCSR A;
  A.n_rows = m->n_rows; A.n_cols = m->n_cols; A.br = m->br; A.bc = m->bc;
  A.rowptr.assign(m->rowptr, m->rowptr + m->n_rows + 1);
  A.col.assign(m->col, m->col + A.rowptr.back());
  return A;
}

amgh::CsrView as_view(const amgh_matrix* m) {
  amgh::CsrView A;
  A.n_rows = m->n_rows; A.n_cols = m->n_cols; A.br = m->br; A.bc = m->bc;
  A.rowptr = m->rowptr; A.col = m->col; A.val = m->val;
  return A;
}

void view(const amgh::BCSR& A, amgh_matrix* m) {
  m->n_rows = A.n_rows; m->n_cols = A.n_cols; m->br = A.br; m->bc = A.bc;
  m->rowptr = A.rowptr.data(); m->col = A.col.data(); m->val = A.val.data();
}

template <class F>
int guard(F&& f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return 1; }
  catch (...) { g_err = "unknown error"; return 2; }
}

void check_matrix(const amgh_matrix* m) {
  if (!m || !m->rowptr || m->n_rows < 0 || m->n_cols < 0) throw amgh::Error("invalid matrix descriptor");
  if (m->br < 1 || m->bc < 1 || m->br > 6 || m->bc > 6) throw amgh::Error("unsupported block size (1..6)");
  if (m->rowptr[m->n_rows] > 0 && (!m->col || !m->val)) throw amgh::Error("matrix descriptor lacks col/val");
}
}  // namespace

extern "C" {

const char* amgh_last_error(void) { return g_err.c_str(); }

void amgh_default_options(amgh_options* o, int dim, int energy) {
  amgh::Options d;
  o->max_levels = d.max_levels;
  o->max_coarse_size = d.max_coarse_size;
  // elasticity: the reference defaults to first_aaf = 0.025 / 0.05 (elasticity_pc_impl.hpp:71) on top of its
  // energy-based SPW agglomeration; with this build's simpler pairwise aggregation that is too aggressive
  // (64 vs 15 CG iterations on the 10x1x1 beam), so the first level coarsens by ~1/10 instead
  if (energy == 1) { o->first_aaf = dim == 3 ? 0.1 : 0.15; o->sp_max_per_row = 1 + dim; }
  else { o->first_aaf = dim == 3 ? 0.05 : 0.1; o->sp_max_per_row = 3; }
  o->aaf = dim == 3 ? 0.125 : 0.25;
  o->enable_sp = 1;
  o->sp_omega = 1.0;
  o->sp_min_frac = dim == 3 ? 0.08 : 0.15;
  o->soc_thresh = d.soc_thresh;
  o->max_rounds = d.max_rounds;
  o->regularize_cmats = 0;
  o->dim = dim;
  o->energy = energy;
  o->log_level = 0;
  o->enable_multistep = 0;
  o->robust_soc = 0;
  o->spw = d.spw;
  o->spw_rounds = d.spw_rounds;
  o->spw_orphan_round = d.spw_orphan_round;
  o->prol_type = d.prol_type;
  o->sp_max_per_row_classic = d.sp_max_per_row_classic;
  o->edge_mats = 0;
  o->crs_robust = 0;
  o->spw_cbs = 0;
  o->sp_improve_its = 0;
  o->spw_pick_robust = 1;
  o->spw_neib_boost = 1;
  o->spw_pick_avg = 1;
  o->carry_mesh = 0;
  o->spw_diag_stab_boost = 0.5;
  o->prol_only = 0;
}

int amgh_setup(const amgh_matrix* A, const uint8_t* free_or_null, const double* coords_or_null,
               const amgh_options* opts, amgh_hierarchy** out) {
  return guard([&] {
    check_matrix(A);
    if (A->n_rows != A->n_cols || A->br != A->bc) throw amgh::Error("amgh_setup: matrix must be square with square blocks");
    if (!opts || !out) throw amgh::Error("amgh_setup: null argument");
    amgh::Options o;
    o.max_levels = opts->max_levels; o.max_coarse_size = opts->max_coarse_size;
    o.first_aaf = opts->first_aaf; o.aaf = opts->aaf; o.enable_sp = opts->enable_sp;
    o.sp_omega = opts->sp_omega; o.sp_max_per_row = opts->sp_max_per_row; o.sp_min_frac = opts->sp_min_frac;
    o.soc_thresh = opts->soc_thresh; o.max_rounds = opts->max_rounds; o.regularize_cmats = opts->regularize_cmats;
    o.dim = opts->dim; o.energy = opts->energy; o.log_level = opts->log_level; o.enable_multistep = opts->enable_multistep; o.robust_soc = opts->robust_soc;
    o.spw = opts->spw; o.spw_rounds = opts->spw_rounds; o.spw_orphan_round = opts->spw_orphan_round;
    o.prol_type = opts->prol_type; o.sp_max_per_row_classic = opts->sp_max_per_row_classic;
    o.edge_mats = opts->edge_mats; o.crs_robust = opts->crs_robust;
    o.spw_pick_robust = opts->spw_pick_robust; o.spw_neib_boost = opts->spw_neib_boost;
    o.spw_pick_avg = opts->spw_pick_avg; o.spw_diag_stab_boost = opts->spw_diag_stab_boost; o.carry_mesh = opts->carry_mesh;
    if (o.carry_mesh && (!o.spw || o.enable_multistep)) throw amgh::Error("amgh_setup: carry_mesh needs spw = 1 and enable_multistep = 0");
    if (o.spw_pick_avg < 0 || o.spw_pick_avg > 4) throw amgh::Error("amgh_setup: spw_pick_avg must be 0 (min), 1 (geom), 2 (harm), 3 (alg) or 4 (max)");
    o.spw_cbs = opts->spw_cbs; o.sp_improve_its = opts->sp_improve_its; o.prol_only = opts->prol_only;
    if (o.prol_only && o.enable_multistep) throw amgh::Error("amgh_setup: prol_only takes one step (enable_multistep = 0)");
    if (o.sp_improve_its < 0 || o.sp_improve_its > 100) throw amgh::Error("amgh_setup: sp_improve_its out of range");
    if (o.sp_improve_its && o.enable_multistep) throw amgh::Error("amgh_setup: sp_improve_its needs enable_multistep = 0");
    if (o.spw_cbs && !o.crs_robust) throw amgh::Error("amgh_setup: spw_cbs (aggregate-wide check) belongs to the energy-based strength of connection (crs_robust = 1)");
    if (o.crs_robust && !o.edge_mats) throw amgh::Error("amgh_setup: crs_robust works on the energy's edge matrices (edge_mats = 1)");
    if (o.edge_mats && o.energy != 1) throw amgh::Error("amgh_setup: edge_mats belongs to the elasticity energy (energy = 1)");
    if (o.edge_mats && !coords_or_null) throw amgh::Error("amgh_setup: edge_mats needs vertex coordinates");
    if (o.edge_mats && (!o.spw || o.enable_multistep)) throw amgh::Error("amgh_setup: edge_mats needs spw = 1 and enable_multistep = 0");
    if (o.prol_type < -1 || o.prol_type > 3) throw amgh::Error("amgh_setup: prol_type must be -1 (default), 0, 1, 2 or 3");
    // robust_soc belongs to the target-driven pairwise rounds (spw = 0): the SPW agglomerator carries maxTrOD through its rounds
    // instead and neither reads nor forwards the vertex scales, so the option (and its "barely smaller" stop rule) would act on
    // half of its data there
    if (o.robust_soc && o.spw) throw amgh::Error("amgh_setup: robust_soc needs spw = 0 (with the SPW agglomerator the accumulated maxTrOD plays that role)");
    if (o.max_levels < 1) throw amgh::Error("amgh_setup: max_levels must be >= 1");
    if (o.dim != 2 && o.dim != 3) throw amgh::Error("amgh_setup: dim must be 2 or 3");
    amgh::BCSR A0 = to_bcsr(A);
    auto h = new amgh_hierarchy();
    h->h.reset(amgh::setup_levels(A0, free_or_null, coords_or_null, o));
    *out = h;
  });
}

int amgh_n_levels(const amgh_hierarchy* h) { return h ? (int)h->h->levels.size() : 0; }

int amgh_level_get(const amgh_hierarchy* h, int level, amgh_level* out) {
  return guard([&] {
    if (!h || !out || level < 0 || level >= (int)h->h->levels.size()) throw amgh::Error("amgh_level_get: bad level");
    const amgh::Level& L = h->h->levels[level];
    std::memset(out, 0, sizeof(*out));
    view(L.A, &out->A);
    if (L.P.n_rows > 0) { view(L.P, &out->P); view(L.PT, &out->PT); }
    out->free = L.free.data();
    out->dinv = L.dinv.data();
    out->coords = L.coords.empty() ? nullptr : L.coords.data();
    out->color = L.color.data();
    out->n_colors = L.n_colors;
    out->agg = L.agg.empty() ? nullptr : L.agg.data();
  });
}

int amgh_coarse_inverse(const amgh_hierarchy* h, int64_t* n, const double** inv) {
  return guard([&] {
    if (!h || !n || !inv) throw amgh::Error("amgh_coarse_inverse: null argument");
    *n = h->h->coarse_n;
    *inv = h->h->coarse_n ? h->h->coarse_inv.data() : nullptr;
  });
}

const char* amgh_log(const amgh_hierarchy* h) { return h ? h->h->log.c_str() : ""; }

void amgh_destroy(amgh_hierarchy* h) { delete h; }

int amgh_calc_dinv(const amgh_matrix* A, const uint8_t* free_or_null, int pinv, double* dinv_out) {
  return guard([&] {
    check_matrix(A);
    if (A->br != A->bc) throw amgh::Error("amgh_calc_dinv: square blocks required");
    amgh::BCSR M = to_bcsr(A);
    amgh::calc_dinv(M, free_or_null, pinv != 0, dinv_out);
  });
}

int amgh_robust_pair_soc(int32_t n, const double* C, const double* E, double* soc_out) {
  return guard([&] {
    if (!C || !E || !soc_out) throw amgh::Error("amgh_robust_pair_soc: null argument");
    *soc_out = amgh::robust_pair_soc_of(n, C, E);
  });
}

int amgh_coloring(const amgh_matrix* A, const uint8_t* free_or_null, int32_t* color_out, int32_t* n_colors) {
  return guard([&] {
    check_matrix(A);
    amgh::BCSR M = to_bcsr(A);
    *n_colors = amgh::greedy_coloring(M, free_or_null, color_out);
  });
}

int amgh_coloring_blocked(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, int32_t* color_out, int32_t* n_colors) {
  return guard([&] {
    check_matrix(A);
    if (block_rows < 1 || !color_out || !n_colors) throw amgh::Error("amgh_coloring_blocked: bad arguments");
    *n_colors = amgh::greedy_coloring_blocked(as_view(A), free_or_null, block_rows, color_out);
  });
}

int amgh_hybrid_dinv(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, double* dinv_out) {
  return amgh_hybrid_dinv_ext(A, free_or_null, block_rows, nullptr, dinv_out);
}

int amgh_hybrid_dinv_ext(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, const double* ghost_diag_or_null,
                         double* dinv_out) {
  return guard([&] {
    check_matrix(A);
    if (A->br != 1 || A->bc != 1) throw amgh::Error("amgh_hybrid_dinv: scalar matrices only");
    if (block_rows < 1 || !dinv_out) throw amgh::Error("amgh_hybrid_dinv: bad arguments");
    amgh::hybrid_mod_dinv(as_view(A), free_or_null, block_rows, dinv_out, ghost_diag_or_null);
  });
}

int amgh_hybrid_dinv_block(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, int pinv, double* dinv_out) {
  return guard([&] {
    check_matrix(A);
    if (A->br != A->bc || A->n_rows != A->n_cols) throw amgh::Error("amgh_hybrid_dinv_block: square matrix with square blocks expected");
    if (block_rows < 1 || !dinv_out) throw amgh::Error("amgh_hybrid_dinv_block: bad arguments");
    amgh::hybrid_mod_dinv_block(as_view(A), free_or_null, block_rows, pinv != 0, dinv_out);
  });
}

int amgh_compact_blocks(const amgh_matrix* A, const uint8_t* free_or_null, int32_t target_rows, int32_t max_rows, int32_t* block_of_row_out,
                        int64_t* n_blocks_out) {
  return guard([&] {
    check_matrix(A);
    if (A->n_rows != A->n_cols || target_rows < 1 || max_rows < target_rows || !block_of_row_out) throw amgh::Error("amgh_compact_blocks: bad arguments");
    amgh::BCSR M = to_bcsr_pattern(A);
    const int64_t nb = amgh::compact_blocks(M, free_or_null, target_rows, max_rows, block_of_row_out);
    if (n_blocks_out) *n_blocks_out = nb;
  });
}

int amgh_coloring_blockids(const amgh_matrix* A, const uint8_t* free_or_null, const int32_t* block_of_row, int32_t* color_out, int32_t* n_colors) {
  return guard([&] {
    check_matrix(A);
    if (!block_of_row || !color_out) throw amgh::Error("amgh_coloring_blockids: bad arguments");
    amgh::BCSR M = to_bcsr_pattern(A);
    const int nc = amgh::greedy_coloring_blockids(M, free_or_null, block_of_row, color_out);
    if (n_colors) *n_colors = nc;
  });
}

int amgh_hybrid_dinv_block_ids(const amgh_matrix* A, const uint8_t* free_or_null, const int32_t* block_of_row, int pinv, double* dinv_out) {
  return guard([&] {
    check_matrix(A);
    if (A->br != A->bc || A->n_rows != A->n_cols || !block_of_row || !dinv_out) throw amgh::Error("amgh_hybrid_dinv_block_ids: bad arguments");
    amgh::hybrid_mod_dinv_block(as_view(A), free_or_null, 1, pinv != 0, dinv_out, block_of_row);
  });
}

int amgh_bgs_dinv(const amgh_matrix* A, int32_t n_blocks, const int32_t* block_ptr, const int32_t* block_rows, int pinv,
                  const int64_t* dinv_ptr, double* dinv_out) {
  return guard([&] {
    check_matrix(A);
    if (A->br != A->bc || A->n_rows != A->n_cols) throw amgh::Error("amgh_bgs_dinv: square matrix with square blocks expected");
    if (n_blocks < 0 || !block_ptr || (!block_rows && block_ptr[n_blocks] > 0) || !dinv_ptr || !dinv_out) throw amgh::Error("amgh_bgs_dinv: bad arguments");
    const int bs = A->br;
    const int64_t n = A->n_rows;
    for (int32_t q = 0; q < block_ptr[n_blocks]; q++) if (block_rows[q] < 0 || block_rows[q] >= n) throw amgh::Error("amgh_bgs_dinv: block row out of range");
    int bad = 0;
#pragma omp parallel
    {
      std::vector<int32_t> loc(n, -1);           // global row -> position inside the current block
      std::vector<double> D;
#pragma omp for schedule(dynamic, 64)
      for (int32_t k = 0; k < n_blocks; k++) {
        const int32_t p0 = block_ptr[k], m = block_ptr[k + 1] - p0;
        const int M = m * bs;
        if ((int64_t)M * M != dinv_ptr[k + 1] - dinv_ptr[k]) {
#pragma omp atomic write
          bad = 1;
          continue;
        }
        if (!m) continue;
        for (int32_t q = 0; q < m; q++) loc[block_rows[p0 + q]] = q;
        D.assign((size_t)M * M, 0.0);
        for (int32_t q = 0; q < m; q++) {
          const int64_t i = block_rows[p0 + q];
          for (int64_t e = A->rowptr[i]; e < A->rowptr[i + 1]; e++) {
            const int32_t lj = loc[A->col[e]];
            if (lj < 0) continue;
            for (int r = 0; r < bs; r++)
              for (int c = 0; c < bs; c++) D[(size_t)(q * bs + r) * M + (lj * bs + c)] = A->val[(e * bs + r) * bs + c];
          }
        }
        for (int32_t q = 0; q < m; q++) loc[block_rows[p0 + q]] = -1;
        if (pinv) amgh::pseudo_inverse_try_normal(D.data(), M);
        else if (!amgh::dense_inverse(D.data(), M)) {
#pragma omp atomic write
          bad = 2;
        }
        double* out = dinv_out + dinv_ptr[k];
        for (int r = 0; r < M; r++)
          for (int c = 0; c < M; c++) out[(size_t)c * M + r] = D[(size_t)r * M + c];      // column-major
      }
    }
    if (bad == 1) throw amgh::Error("amgh_bgs_dinv: dinv_ptr does not match the block sizes");
    if (bad == 2) throw amgh::Error("amgh_bgs_dinv: singular diagonal block (use pinv)");
  });
}

int amgh_bgs_coloring(const amgh_matrix* A, int32_t n_blocks, const int32_t* block_ptr, const int32_t* block_rows,
                      int32_t* color_out, int32_t* n_colors) {
  return guard([&] {
    check_matrix(A);
    if (n_blocks < 0 || !block_ptr || !color_out || !n_colors) throw amgh::Error("amgh_bgs_coloring: bad arguments");
    const int64_t n = A->n_rows;
    std::vector<int32_t> blockof(n, -1);
    for (int32_t k = 0; k < n_blocks; k++)
      for (int32_t q = block_ptr[k]; q < block_ptr[k + 1]; q++) {
        if (block_rows[q] < 0 || block_rows[q] >= n || blockof[block_rows[q]] >= 0) throw amgh::Error("amgh_bgs_coloring: blocks must be disjoint sets of valid rows");
        blockof[block_rows[q]] = k;
      }
    std::vector<int32_t> mark(64, -1);
    int nc = 0;
    for (int32_t k = 0; k < n_blocks; k++) {
      color_out[k] = -1;
      for (int32_t q = block_ptr[k]; q < block_ptr[k + 1]; q++) {
        const int64_t i = block_rows[q];
        for (int64_t e = A->rowptr[i]; e < A->rowptr[i + 1]; e++) {
          const int64_t j = A->col[e];
          if (j >= n) continue;
          const int32_t kb = blockof[j];
          if (kb < 0 || kb >= k) continue;          // only already-coloured (lower) blocks matter
          const int32_t c = color_out[kb];
          if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1);
          mark[c] = k;
        }
      }
      int c = 0;
      while (c < (int)mark.size() && mark[c] == k) c++;
      if (c >= (int)mark.size()) mark.resize(2 * c + 2, -1);
      color_out[k] = c;
      nc = std::max(nc, c + 1);
    }
    *n_colors = nc;
  });
}

int amgh_transpose_count(const amgh_matrix* A, int64_t* rowptr_out) {
  return guard([&] {
    check_matrix(A);
    for (int64_t i = 0; i <= A->n_cols; i++) rowptr_out[i] = 0;
    const int64_t nnz = A->rowptr[A->n_rows];
    for (int64_t k = 0; k < nnz; k++) rowptr_out[A->col[k] + 1]++;
    for (int64_t i = 0; i < A->n_cols; i++) rowptr_out[i + 1] += rowptr_out[i];
  });
}

int amgh_transpose_fill(const amgh_matrix* A, const int64_t* rowptr_T, int32_t* col_out, double* val_out) {
  return guard([&] {
    check_matrix(A);
    amgh::BCSR T = amgh::transpose(to_bcsr(A));
    if (T.rowptr.back() != rowptr_T[A->n_cols]) throw amgh::Error("amgh_transpose_fill: rowptr mismatch");
    std::memcpy(col_out, T.col.data(), T.col.size() * sizeof(int32_t));
    std::memcpy(val_out, T.val.data(), T.val.size() * sizeof(double));
  });
}

int amgh_matmul(const amgh_matrix* A, const amgh_matrix* B, int64_t* rowptr_out, int32_t* col_out, double* val_out) {
  return guard([&] {
    check_matrix(A); check_matrix(B);
    if (!col_out) {
      g_mm_cache.reset(new amgh::BCSR(amgh::matmul(to_bcsr(A), to_bcsr(B))));
      std::memcpy(rowptr_out, g_mm_cache->rowptr.data(), g_mm_cache->rowptr.size() * sizeof(int64_t));
    } else {
      if (!g_mm_cache) throw amgh::Error("amgh_matmul: fill call without a preceding count call");
      std::memcpy(col_out, g_mm_cache->col.data(), g_mm_cache->col.size() * sizeof(int32_t));
      std::memcpy(val_out, g_mm_cache->val.data(), g_mm_cache->val.size() * sizeof(double));
      g_mm_cache.reset();
    }
  });
}

int amgh_set_galerkin_hook(amgh_galerkin_fn run, amgh_galerkin_fetch_fn fetch, int64_t min_rows) {
  return guard([&] {
    if ((run == nullptr) != (fetch == nullptr)) throw amgh::Error("amgh_set_galerkin_hook: run and fetch come as a pair");
    amgh::GalerkinHook& h = amgh::galerkin_hook();
    h.run = reinterpret_cast<decltype(h.run)>(run);
    h.fetch = fetch;
    h.min_rows = min_rows;
  });
}

int amgh_kuhn_pattern(int dim, const int64_t* shape, int64_t* rowptr_out) {
  return guard([&] {
    if (dim != 2 && dim != 3) throw amgh::Error("kuhn: dim must be 2 or 3");
    for (int d = 0; d < dim; d++) if (shape[d] < 2) throw amgh::Error("kuhn: need at least 2 vertices per direction");
    amgh::kuhn_pattern(dim, shape, rowptr_out);
  });
}

int amgh_kuhn_assemble(int dim, const int64_t* shape, const double* coords, int kind, int bs, double mu, double lam,
                       const double* cell_coef_or_null, const int64_t* rowptr, int32_t* col_out, double* val_out,
                       double* load_out_or_null) {
  return guard([&] {
    if (dim != 2 && dim != 3) throw amgh::Error("kuhn: dim must be 2 or 3");
    const int nrot = dim * (dim - 1) / 2;
    if ((kind == 0 && bs != 1) || (kind == 1 && bs != dim) || (kind == 2 && bs != dim + nrot) || kind < 0 || kind > 2)
      throw amgh::Error("kuhn: kind / block size mismatch");
    amgh::kuhn_assemble(dim, shape, coords, kind, bs, mu, lam, cell_coef_or_null, rowptr, col_out, val_out, load_out_or_null);
  });
}

}  // extern "C"
