// Host AMG setup: produces the frozen hierarchy (level matrices, P, P^T, smoother diagonals, colours,
// coarse inverse) that is uploaded once to the GPU.  Minimal restatement of the reference's setup layer
// (SURVEY.md section 7, step 2): same formats and block shapes, own (simpler) aggregation.
#pragma once
#include "bcsr.hpp"

namespace amgh {

struct Options {
  // level control (reference src/base/factory/base_factory.hpp:92-152, base_factory.cpp:23-59)
  int max_levels = 10;
  int64_t max_coarse_size = 50;
  double first_aaf = 0.05;       // target n_1/n_0      (h1_impl.hpp:333)
  double aaf = 0.125;            // target n_{l+1}/n_l  (h1_impl.hpp:334)
  int robust_soc = 0;            // vertex scales in the strength of connection (amgh.h)
  int enable_multistep = 0;      // concatenate several coarsening steps until a level reaches its target (h1_impl.hpp:331)
  // smoothed prolongation (h1_impl.hpp:320-324, elasticity_pc_impl.hpp:58-62)
  int enable_sp = 1;
  double sp_omega = 1.0;
  int sp_max_per_row = 3;
  double sp_min_frac = 0.08;
  // agglomeration: spw = 1: the reference's SPW rule with scalar strength (spw_agg_impl.hpp; hierarchy.cpp aggregate_spw):
  // spw_rounds pairing rounds per coarsening step + orphan round; 0: target-driven pairwise rounds (rounds 1 - 2 of this build)
  int spw = 1;
  int spw_rounds = 3;            // ngs_amg_spw_rounds (spw_agg.hpp:28)
  int spw_orphan_round = 1;      // ngs_amg_spw_orphan_treatment (spw_agg.hpp:32)
  int prol_type = -1;            // ngs_amg_prol_type: 0 piecewise, 1 aux_smoothed, 2 semi_aux_smoothed (reference default), 3 own rule; -1: 2 with spw, else 3
  int sp_max_per_row_classic = 5;   // vertex_factory_impl.hpp:71
  int crs_robust = 0;            // ngs_amg_crs_robust: energy-based strength of connection in the SPW rounds (needs edge_mats; agglomerator.hpp:18)
  int spw_cbs = 0;               // ngs_amg_spw_cbs: aggregate-wide stability check from the second pairing round on (needs crs_robust; spw_agg.hpp:31)
  int sp_improve_its = 0;        // ngs_amg_sp_improve_its: smoothing steps on the prolongation inside its graph (vertex_factory_impl.hpp:2350-2420)
  int prol_only = 0;             // ONE coarsening step, P / aggregates / coarse coordinates only (amgh.h)
  int spw_pick_robust = 1;       // ngs_amg_spw_pick_robust (spw_agg.hpp:26): crs_robust picks by the robust number (1) or only vetoes with it (0)
  int spw_neib_boost = 1;        // ngs_amg_spw_neib_boost (spw_agg.hpp:27): neighbour boost of the robust edge matrix
  int spw_pick_avg = 1;          // ngs_amg_spw_pick_avg: 0 min, 1 geom, 2 harm, 3 alg, 4 max (spw_agg.hpp:22, 62-65)
  double spw_diag_stab_boost = 0.5;   // ngs_amg_spw_diag_stab_boost (spw_agg.hpp:42): crs_robust, share of the in-aggregate edges kept in the carried aux diagonals
  int carry_mesh = 0;            // coarse alg-meshes by contraction of the finer one instead of the Galerkin matrix's graph (amgh.h)
  int edge_mats = 0;             // elasticity: carry the energy's edge matrices, matrix-valued smoothed prolongation (amgh.h)
  double soc_thresh = 0.25;      // relative strength threshold for a viable partner
  int max_rounds = 8;            // hard cap of pairwise rounds per level
  // smoother diagonals
  int regularize_cmats = 0;      // => pseudo-inverse dinv (gssmoother.cpp:161-164)
  // problem class
  int dim = 3;
  int energy = 0;                // 0: H1 (P = w * I_bs), 1: elasticity (rigid body blocks, coarse bs = dim + nrot)
  int log_level = 0;
};

struct Level {
  BCSR A;
  BCSR P, PT;                    // to the next coarser level (empty on the coarsest)
  std::vector<uint8_t> free;     // per block row
  std::vector<double> dinv;      // n * bs * bs, zero for non-free rows
  std::vector<double> coords;    // n * dim
  std::vector<int32_t> color;    // greedy multicolouring of the graph of A (free rows), -1 for non-free
  int n_colors = 0;
  std::vector<int32_t> agg;      // fine vertex -> coarse vertex (or -1); kept for block smoothers / debugging
  std::vector<double> vscale;    // robust_soc: per vertex the largest edge weight collapsed inside it on the finer levels
};

struct Hierarchy {
  std::vector<Level> levels;
  std::vector<double> coarse_inv;   // dense (n_L*bs)^2 inverse of the coarsest matrix on its free dofs
  int64_t coarse_n = 0;             // scalar size of coarse_inv
  Options opts;
  std::string log;
};

Hierarchy* setup_levels(const BCSR& A0, const uint8_t* free0, const double* coords0, const Options& o);

// smoother data for one matrix (also used stand-alone by CreateJacobiSmoother / CreateHybridGSS mirrors)
void calc_dinv(const BCSR& A, const uint8_t* free, bool pinv, double* dinv);
int greedy_coloring(const BCSR& A, const uint8_t* free, int32_t* color);
int greedy_coloring_blocked(const BCSR& A, const uint8_t* free, int64_t block_rows, int32_t* color);
int greedy_coloring_blocked(const CsrView& A, const uint8_t* free, int64_t block_rows, int32_t* color);
void hybrid_mod_dinv(const BCSR& A, const uint8_t* free, int64_t block_rows, double* dinv, const double* ghost_diag = nullptr);
void hybrid_mod_dinv(const CsrView& A, const uint8_t* free, int64_t block_rows, double* dinv, const double* ghost_diag = nullptr);
void hybrid_mod_dinv_block(const BCSR& A, const uint8_t* free, int64_t block_rows, bool pinv, double* dinv, const int32_t* block_of_row = nullptr);
void hybrid_mod_dinv_block(const CsrView& A, const uint8_t* free, int64_t block_rows, bool pinv, double* dinv, const int32_t* block_of_row = nullptr);
// CalcRobustPairSOC (agglomerator_utils.hpp:763-841): smallest eigenvalue of E v = lambda C v off the kernel of C (n <= 6)
double robust_pair_soc_of(int n, const double* C, const double* E);
int64_t compact_blocks(const BCSR& A, const uint8_t* free, int target, int max_rows, int32_t* block_of_row);
int greedy_coloring_blockids(const BCSR& A, const uint8_t* free, const int32_t* block_of_row, int32_t* color);

}  // namespace amgh
