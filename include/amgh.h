/* amgh.h -- C ABI of the HOST setup library (libngsamg_host.so).
 *
 * Cold path: builds the frozen AMG hierarchy on the host, in the reference's CSR / block-CSR formats,
 * which amgx_create (amgx.h) then uploads once.  It stands where the reference's setup layer stands:
 *   BaseAMGPC::FinalizeLevel -> BuildAMGMat -> BaseAMGFactory::SetUpLevels
 *   (reference src/base/precond/amg_pc.cpp:420-434, 565-736; src/base/factory/base_factory.cpp:219-525).
 * All arrays handed out by amgh_level_get are owned by the hierarchy handle and stay valid until
 * amgh_destroy; arrays passed in are borrowed for the duration of the call only.
 *
 * Every function returns 0 on success, non-zero on error (message via amgh_last_error()); the Python
 * shim rethrows, like the reference's ngcore::Exception -> RuntimeError path.
 */
#ifndef NGSAMG_AMGH_H
#define NGSAMG_AMGH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* (block-)CSR matrix view; layout of NGSolve's SparseMatrix<Mat<br,bc>> */
typedef struct amgh_matrix {
  int64_t n_rows, n_cols;
  int32_t br, bc;            /* block height / width                                  */
  const int64_t* rowptr;     /* [n_rows+1]                                            */
  const int32_t* col;        /* [nnz] ascending per row                               */
  const double* val;         /* [nnz*br*bc] row-major blocks                          */
} amgh_matrix;

/* options = the ngs_amg_* flags that shape the hierarchy (reference amg_pc.cpp:270-339,
 * base_factory.cpp:23-59, h1_impl.hpp:300-370, elasticity_pc_impl.hpp:36-142) */
typedef struct amgh_options {
  int32_t max_levels;        /* ngs_amg_max_levels        (10)                        */
  int64_t max_coarse_size;   /* ngs_amg_max_coarse_size   (50)                        */
  double first_aaf;          /* ngs_amg_first_aaf         (0.05 3D / 0.1 2D; elasticity 0.1 / 0.15, own) */
  double aaf;                /* ngs_amg_aaf               (2^-dim)                    */
  int32_t enable_sp;         /* ngs_amg_enable_sp         (1)                         */
  double sp_omega;           /* ngs_amg_sp_omega          (1.0)                       */
  int32_t sp_max_per_row;    /* ngs_amg_sp_max_per_row    (3; 1+dim elasticity)       */
  double sp_min_frac;        /* ngs_amg_sp_min_frac       (0.08 3D / 0.15 2D)         */
  double soc_thresh;         /* pairwise strength threshold (own; 0.25)               */
  int32_t max_rounds;        /* cap of pairwise rounds per level (own; 8)             */
  int32_t regularize_cmats;  /* ngs_amg_regularize_cmats  => pseudo-inverse diagonals */
  int32_t dim;               /* spatial dimension                                     */
  int32_t energy;            /* 0 = H1 (h1_scal/h1_2d/h1_3d), 1 = elasticity          */
  int32_t log_level;
  int32_t enable_multistep;  /* ngs_amg_enable_multistep: reach first_aaf by several concatenated coarsening steps of ~aaf each,   */
                             /*   P = P_1 P_2 ... (reference default for H1: true, h1_impl.hpp:331; default HERE: 0, because the   */
                             /*   concatenated P has ~3x the entries per row: fewer iterations, slower application)    */
  int32_t robust_soc;        /* own, default 0: every vertex carries the largest edge weight collapsed inside it (through the      */
                             /*   pairwise rounds AND from level to level) and a connection is judged against that scale too, so a  */
                             /*   (where it exceeds the vertex's live connections 16-fold: quasi-uniform meshes keep their aggregates), so a */
                             /*   stiff inclusion that has become one vertex does not absorb its soft neighbours (the role of the   */
                             /*   accumulated vertex weights in the reference's strength of connection, spw_agg_impl.hpp)           */
  int32_t spw;               /* 1 (default): the reference's SPW agglomeration with scalar strength of connection (reference            */
                             /*   spw_agg_impl.hpp:637-775, 943-1263, 424-628): spw_rounds pairing rounds per coarsening step on contracted */
                             /*   graphs, partner filter soc_ij >= 0.25 max_k soc_ik with soc = w / sqrt(maxTrOD_i maxTrOD_j), maxTrOD      */
                             /*   carried through the rounds, orphan round.  0: the target-driven pairwise rounds of earlier builds       */
  int32_t spw_rounds;        /* ngs_amg_spw_rounds           (3, spw_agg.hpp:28)                                                        */
  int32_t spw_orphan_round;  /* ngs_amg_spw_orphan_treatment (1, spw_agg.hpp:32)                                                        */
  int32_t prol_type;         /* ngs_amg_prol_type (vertex_factory_impl.hpp:63-69, 849-853): 0 piecewise, 1 aux_smoothed, 2 semi_aux_smoothed   */
                             /*   (the reference's default: rows whose algebraic neighbours map to <= sp_max_per_row_classic coarse vertices   */
                             /*   are smoothed with the level matrix, the others with the replacement matrix of the edge weights,             */
                             /*   vertex_factory_impl.hpp:1836-2290); 3: the weight rule of rounds 1-2 of this build.  Block levels take the   */
                             /*   aux rule on their scalar edge weights for 1 and 2 (rigid-body blocks w Q(t)).  Default: 2 with spw = 1, else 3 */
  int32_t sp_max_per_row_classic;  /* ngs_amg_sp_max_per_row_classic (5, vertex_factory_impl.hpp:71)                                        */
  int32_t edge_mats;         /* elasticity only, default 0.  1: the setup carries the energy's edge matrices from level to level (finest     */
                             /*   level from the assembled matrix as BuildAlgMesh_ALG_blk does, elasticity_pc_impl.hpp:409-505; coarse ones by   */
                             /*   AttachedEED::map_data, elasticity_impl.hpp:23-78) and builds the MATRIX-VALUED smoothed prolongation of the      */
                             /*   reference (SemiAuxSProlMap with TM = Mat<BS,BS>, vertex_factory_impl.hpp:1836-2290): general BS x BS blocks     */
                             /*   instead of w Q(t); the strength of connection reads the edges' approximate weights trace(E) / BS.              */
  int32_t crs_robust;        /* ngs_amg_crs_robust (agglomerator.hpp:18; the reference's elasticity preconditioner sets it to false,           */
                             /*   elasticity_pc_impl.hpp:55), default 0, needs edge_mats: the SPW pairing rounds pick the partner by the          */
                             /*   energy-based strength of connection (CalcRobSOC with neighbour boost, agglomerator_utils.hpp:598-927;          */
                             /*   FindNeib3Step with robustPick, spw_agg_impl.hpp:637-775)                                                        */
  int32_t spw_cbs;           /* ngs_amg_spw_cbs (checkBigSOC, spw_agg.hpp:31, 57; default 0), needs crs_robust: from the second pairing round on a   */
                             /*   partner must also pass the aggregate-wide stability check of the two vertices' base-level members                */
                             /*   (AggregateWideStabilityCheck, agglomerator_utils.hpp:392-539)                                                    */
  int32_t sp_improve_its;    /* ngs_amg_sp_improve_its (0; vertex_factory_impl.hpp:1745-1831, 2350-2420): smoothing steps on the smoothed             */
                             /*   prolongation that keep its graph: P_i -= omega D^+ (A P)_i with the entries outside the row's pattern moved to the   */
                             /*   row's own aggregate (through the rigid-body transformation for elasticity)                                          */
  int32_t prol_only;         /* own, default 0.  1: ONE coarsening step that returns the prolongation only: level 0 carries P and the aggregates,     */
                             /*   level 1 its size, block size and coordinates; no P^T, no Galerkin product, no smoother data, no coarse inverse (the    */
                             /*   arrays of amgh_level keep their sizes, zero-filled; level 1's A has no entries).  For callers that form the coarse     */
                             /*   operator themselves -- the rank-partitioned setup (ngsamg_amd/dist.py), whose product needs the halo rows of P and     */
                             /*   where the unread dense inverse of a <= 4096-unknown coarse level alone cost more than the step                          */
  int32_t spw_pick_robust;   /* ngs_amg_spw_pick_robust (1, spw_agg.hpp:26, 55), with crs_robust: 1 = the robust strength orders the candidates,       */
                             /*   0 = the scalar strength orders them and the robust one only vetoes (FindNeib3Step, spw_agg_impl.hpp:722-765)       */
  int32_t spw_neib_boost;    /* ngs_amg_spw_neib_boost (1, spw_agg.hpp:27, 56): the neighbour boost of the robust edge matrix (AddNeibBoost)            */
  int32_t spw_pick_avg;      /* ngs_amg_spw_pick_avg (geom; spw_agg.hpp:22, 62-65): average of the two vertices' maxTrOD in the scalar strength         */
                             /*   soc = w / avg: 0 min, 1 geom, 2 harm, 3 alg, 4 max                                                                  */
  double spw_diag_stab_boost; /* ngs_amg_spw_diag_stab_boost (0.5; spw_agg.hpp:36-42), with crs_robust: share of the edges that vanish inside a pair  */
                             /*   that stays in the pair's aux diagonal (0: all removed, most matches; 1: all kept, most stable)                        */
  int32_t carry_mesh;        /* own, default 0.  1: the alg-mesh (strength graph) of a coarse level is the CONTRACTED mesh of the level above -- an     */
                             /*   edge between two aggregates that a fine edge connects, weight = the sum of those fine weights -- which is how the     */
                             /*   reference's meshes descend (coarse maps of BlockTM, H1EData / AttachedEED map_data), instead of the graph of the     */
                             /*   Galerkin matrix, whose rows the smoothed prolongation has widened (50-80 entries from level 1 on at cfg 2).          */
                             /*   Changes aggregates and aux rows from level 1 on; edge_mats implies it for elasticity.                                */
} amgh_options;

typedef struct amgh_level {
  amgh_matrix A, P, PT;      /* P / PT have n_rows == 0 on the coarsest level          */
  const uint8_t* free;       /* [n] 1 = free block row                                 */
  const double* dinv;        /* [n*bs*bs] inverted (block) diagonal, 0 for non-free    */
  const double* coords;      /* [n*dim] or NULL                                        */
  const int32_t* color;      /* [n] greedy colour of free rows, -1 otherwise           */
  int32_t n_colors;
  const int32_t* agg;        /* [n] vertex -> coarse vertex or -1 (NULL on coarsest)   */
} amgh_level;

typedef struct amgh_hierarchy amgh_hierarchy;

const char* amgh_last_error(void);
/* Fills EVERY field with the library default for the problem class.  Callers start from it and then override single fields: the
 * struct grows at its end from round to round (zero is not the default of every field: spw, spw_rounds, spw_pick_robust,
 * spw_neib_boost, spw_pick_avg = 1 (geom), spw_diag_stab_boost = 0.5, prol_type = -1, ...). */
void amgh_default_options(amgh_options* o, int dim, int energy);

int amgh_setup(const amgh_matrix* A, const uint8_t* free_or_null, const double* coords_or_null,
               const amgh_options* opts, amgh_hierarchy** out);
int amgh_n_levels(const amgh_hierarchy* h);
int amgh_level_get(const amgh_hierarchy* h, int level, amgh_level* out);
/* dense inverse of the coarsest matrix restricted to free dofs; n = scalar size (0 if unavailable) */
int amgh_coarse_inverse(const amgh_hierarchy* h, int64_t* n, const double** inv);
const char* amgh_log(const amgh_hierarchy* h);
void amgh_destroy(amgh_hierarchy* h);

/* stand-alone smoother data (CreateJacobiSmoother / CreateHybridGSS mirrors,
 * reference src/base/smoothers/python_smoothers.cpp:144-387) */
int amgh_calc_dinv(const amgh_matrix* A, const uint8_t* free_or_null, int pinv, double* dinv_out);
int amgh_coloring(const amgh_matrix* A, const uint8_t* free_or_null, int32_t* color_out, int32_t* n_colors);
/* The pair strength of connection of the energy-based agglomeration (options.crs_robust; reference CalcRobustPairSOC,
 * src/base/coarsening/agglomerator_utils.hpp:763-841): the smallest eigenvalue of E v = lambda C v on the complement of ker C
 * (eigenvalues of C below 1e-10 of its largest count as kernel), clipped at 0.  C, E: symmetric n x n, row-major, n <= 6. */
int amgh_robust_pair_soc(int32_t n, const double* C, const double* E, double* soc_out);
/* Block-hybrid Gauss-Seidel (amgx_level_desc.gs_block_rows): blocks of block_rows consecutive rows are swept like the
 * ranks of the reference's HybridGSSmoother (gssmoother.cpp:709-861) -- Gauss-Seidel inside a block, couplings that leave
 * the block frozen at their sweep-start values.  amgh_coloring_blocked: greedy colouring that only sees couplings inside a
 * block; amgh_hybrid_dinv: inverse of the l1-type modified diagonal md = max(1, 0.51 (1 + ad)) d with ad_k = sum over the
 * couplings leaving the block of |a_kj| / sqrt(d_k d_j)  (CalcModDiag, hybrid_smoother_utils.hpp:35-142); scalar matrices */
int amgh_coloring_blocked(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, int32_t* color_out, int32_t* n_colors);
int amgh_hybrid_dinv(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, double* dinv_out);
/* the same for a rank-partitioned level (A: owned rows x [owned | ghost] columns): ghost_diag = the diagonal entries of the
 * n_cols - n_rows ghost rows as their owners hold them (the reference sums the shares of all ranks, hybrid_smoother_utils.hpp:35-110) */
int amgh_hybrid_dinv_ext(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, const double* ghost_diag_or_null,
                         double* dinv_out);
/* square-block matrices (hybrid_smoother_utils.hpp:55-68, 86-98, 128-141): dinv_k = (pseudo-)inverse(A_kk) / max(1, max_l 0.51 (1 + ad_k(l))),
 * ad_k(l) = sum over the couplings leaving the block of rows of sum_m |a_kj(l, m)| / sqrt(d_k(l, l) d_j(m, m)); dinv_out: [n * bs * bs] */
int amgh_hybrid_dinv_block(const amgh_matrix* A, const uint8_t* free_or_null, int64_t block_rows, int pinv, double* dinv_out);
/* Compact sweep blocks instead of runs of consecutive rows (amgx_level_desc.gs_block_ids): the reference's hybrid smoother freezes
 * the couplings between mesh-partitioner subdomains, which are compact; amgh_compact_blocks grows blocks of ~target_rows (<= max_rows)
 * vertices breadth-first over the matrix graph.  amgh_coloring_blockids / amgh_hybrid_dinv_block_ids: the colouring and the
 * l1-modified block diagonal for such blocks (same rules as above with "same block" decided by the ids) */
int amgh_compact_blocks(const amgh_matrix* A, const uint8_t* free_or_null, int32_t target_rows, int32_t max_rows, int32_t* block_of_row_out,
                        int64_t* n_blocks_out);
int amgh_coloring_blockids(const amgh_matrix* A, const uint8_t* free_or_null, const int32_t* block_of_row, int32_t* color_out, int32_t* n_colors);
int amgh_hybrid_dinv_block_ids(const amgh_matrix* A, const uint8_t* free_or_null, const int32_t* block_of_row, int pinv, double* dinv_out);

/* Block Gauss-Seidel data (reference BSmoother, src/base/smoothers/block_gssmoother.cpp:17-150).  Blocks are sets of
 * block rows -- the aggregates of the level, as GetGSBlocks builds them (amg_pc_vertex_impl.hpp:1171-1269); block k owns
 * block_rows[block_ptr[k] .. block_ptr[k+1]) (ascending).
 * amgh_bgs_dinv: per block the inverse (pinv != 0: pseudo-inverse, utils_denseLA.hpp:1460-1570) of A restricted to the
 *   block's scalar dofs: M_k x M_k, M_k = bs * |block k|, column-major, written at dinv_out + dinv_ptr[k]
 *   (caller: dinv_ptr[k+1] = dinv_ptr[k] + M_k^2).
 * amgh_bgs_coloring: greedy colouring of the block graph (blocks k, k' are coupled if A has an entry between their
 *   rows): blocks of one colour can be relaxed in parallel (the reference colours for its shared-memory sweep too,
 *   block_gssmoother.cpp:152-213). */
int amgh_bgs_dinv(const amgh_matrix* A, int32_t n_blocks, const int32_t* block_ptr, const int32_t* block_rows, int pinv,
                  const int64_t* dinv_ptr, double* dinv_out);
int amgh_bgs_coloring(const amgh_matrix* A, int32_t n_blocks, const int32_t* block_ptr, const int32_t* block_rows,
                      int32_t* color_out, int32_t* n_colors);

/* sparse helpers exposed for tests and the Python utils mirror (SparseMM, reference python_utils.cpp:30-193) */
int amgh_transpose_count(const amgh_matrix* A, int64_t* rowptr_out /*[n_cols+1]*/);
int amgh_transpose_fill(const amgh_matrix* A, const int64_t* rowptr_T, int32_t* col_out, double* val_out);
/* C = A*B in two calls: first with col_out == NULL to obtain rowptr (size n_rows+1), then fill */
int amgh_matmul(const amgh_matrix* A, const amgh_matrix* B, int64_t* rowptr_out, int32_t* col_out, double* val_out);

/* Galerkin product on an accelerator: when a pair is installed, amgh_setup hands the products (P^T A) P of levels
 * with at least `min_rows` fine (block) rows to `run` (which returns 0 = done with *n_rows / *nnz set, 2 = "not for me": the host
 * product runs, anything else = error) and reads the arrays back with `fetch` (which also releases the result).  The device
 * library's amgx_galerkin / amgx_csr_result_fetch (amgx.h) are that pair; its result equals the host product bit for bit.
 * run = NULL removes the hook.  Not thread-safe against a running amgh_setup. */
typedef int (*amgh_galerkin_fn)(const amgh_matrix* PT, const amgh_matrix* A, const amgh_matrix* P, void** result, int64_t* n_rows, int64_t* nnz);
typedef int (*amgh_galerkin_fetch_fn)(void* result, int64_t* rowptr, int32_t* col, double* val);
int amgh_set_galerkin_hook(amgh_galerkin_fn run, amgh_galerkin_fetch_fn fetch, int64_t min_rows);

/* synthetic P1 problems on Kuhn grids (stand-in for the calling FEM package) */
int amgh_kuhn_pattern(int dim, const int64_t* shape, int64_t* rowptr_out);
int amgh_kuhn_assemble(int dim, const int64_t* shape, const double* coords, int kind, int bs, double mu, double lam,
                       const double* cell_coef_or_null, const int64_t* rowptr, int32_t* col_out, double* val_out,
                       double* load_out_or_null);

#ifdef __cplusplus
}
#endif
#endif
