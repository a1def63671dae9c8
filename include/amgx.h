/* amgx.h -- C ABI of the MI355X-native AMG apply path (libngsamg_hip.so).
 *
 * Drop-in boundary for the hot path of LukasKogler/NgsAMG: one preconditioner application
 *     BaseAMGPC::Mult(b, x)            reference src/base/precond/amg_pc.cpp:467-470
 *       -> AMGMatrix::Mult -> SmoothV  reference src/base/solve/amg_matrix.cpp:377-378, 160-307
 * and the pieces reachable through the reference's Python surface (GetSmoother(level).Smooth,
 * GetAMGMatrix().GetMatrix(level), DOFMap transfers).  An NGSolve-side subclass of ngcomp::Preconditioner
 * (or the stand-alone Python module ngsamg_amd.NgsAMG) forwards to these entry points; INTEGRATION.md shows
 * the binding.  Plain pointers and sizes only -- no torch / HIP types in the signatures (streams travel
 * as void*).
 *
 * Lifetime: amgx_create copies every host array of the descriptor to the GPU once (the descriptor is
 * borrowed during the call only).  A handle is not re-entrant (like AMGMatrix::Mult, which mutates its
 * work vectors) but distinct handles are independent.
 *
 * Errors: every function returns 0 on success, non-zero otherwise; amgx_last_error(handle) (handle may
 * be NULL for create-time errors) returns the message.  The shim rethrows it (reference: ngcore::Exception).
 *
 * All vectors are fp64, AoS block vectors (entry = bs*dof + comp; reference amg_matrix.cpp:494).
 */
#ifndef NGSAMG_AMGX_H
#define NGSAMG_AMGX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* host (block-)CSR view: layout of NGSolve SparseMatrix<Mat<br,bc>> */
typedef struct amgx_matrix {
  int64_t n_rows, n_cols;     /* block rows / block columns                           */
  int32_t br, bc;             /* block height / width (1, 2, 3, 6; P: 3x6, PT: 6x3)    */
  const int64_t* rowptr;      /* [n_rows+1] (size_t firsti in the reference)          */
  const int32_t* col;         /* [nnz] ascending per row                              */
  const double* val;          /* [nnz*br*bc] row-major blocks                         */
} amgx_matrix;

/* smoother kinds = ngs_amg_sm_type (reference src/base/precond/amg_pc.cpp:1033-1138) */
enum {
  AMGX_SM_JACOBI = 0,         /* JacobiSmoother<TM>, base_smoother.cpp:61-114         */
  AMGX_SM_GS = 1,             /* Gauss-Seidel (GSS3, gssmoother.cpp:196-398) executed as multicolour GS */
  AMGX_SM_BGS = 2             /* block Gauss-Seidel over aggregate blocks (BSmoother, block_gssmoother.cpp:17-498), */
                              /*   blocks of one colour relaxed in parallel (the reference's sm_shm sweep does too) */
};
enum { AMGX_CYCLE_V = 0, AMGX_CYCLE_W = 1, AMGX_CYCLE_BS = 2 };   /* ngs_amg_mg_cycle, amg_matrix.hpp:37-43 */
enum { AMGX_CLEV_NONE = 0, AMGX_CLEV_INV = 1 };                   /* ngs_amg_clev,     amg_matrix.cpp:217-247 */

/* flags for the vector arguments of the calls below */
enum {
  AMGX_HOST_PTR = 0,          /* vectors are host arrays: copied H2D / D2H inside the call            */
  AMGX_DEVICE_PTR = 1,        /* vectors are device arrays on the handle's GPU (no copies)            */
  AMGX_NO_GRAPH = 2,          /* launch the kernels directly instead of replaying the captured graph  */
  AMGX_PCG_SINGLE_REDUCTION = 16  /* amgx_pcg / amgx_dist_pcg with a preconditioner: the Chronopoulos / Gear form of the   */
                              /* same recurrence -- ONE reduction point (one all-reduce of two scalars) per iteration and  */
                              /* three launches beside the cycle and the level-0 product instead of five; histories agree   */
                              /* with the classical form to ~1e-6                                                            */
};

typedef struct amgx_level_desc {
  amgx_matrix A;              /* level matrix (smoothers[l]->GetAMatrix())                           */
  amgx_matrix P, PT;          /* ProlMap<TM>: P and explicit P^T (dof_map.hpp:252-334); empty on the coarsest */
  const double* dinv;         /* [n*bs*bs] inverted block diagonal, 0 on non-free rows (gssmoother.cpp:143-170) */
  const uint8_t* free_dofs;   /* [n] 1 = free block row, or NULL (all free)                           */
  int32_t sm_type;            /* AMGX_SM_*                                                           */
  double omega;               /* Jacobi damping, reference default 0.9 (base_smoother.hpp:279)       */
  int32_t sm_steps;           /* ngs_amg_sm_steps (ProxySmoother, base_smoother.hpp:169-229)         */
  int32_t sm_symm;            /* ngs_amg_sm_symm                                                     */
  const int32_t* color;       /* [n] colour of each free row (required for AMGX_SM_GS), -1 otherwise  */
  int32_t n_colors;
  /* AMGX_SM_BGS only (amgh_bgs_dinv / amgh_bgs_coloring produce these): */
  int32_t bgs_n_blocks;
  const int32_t* bgs_block_ptr;   /* [n_blocks+1]; block k owns bgs_block_rows[ptr[k] .. ptr[k+1]) (disjoint sets)   */
  const int32_t* bgs_block_rows;
  const int64_t* bgs_dinv_ptr;    /* [n_blocks+1] offsets into bgs_dinv                                             */
  const double* bgs_dinv;         /* per block the dense inverse of its diagonal block, M x M column-major, M = bs*size */
  const int32_t* bgs_color;       /* [n_blocks] colour of each block; coupled blocks must differ                    */
  int32_t bgs_n_colors;
  amgx_matrix Q;              /* optional (rowptr == NULL: none).  Folded post-smoothing prolongation                */
                              /*   Q = (I - omega*Dinv*A) P  of a Jacobi level of the V-cycle, n_rows x (coarse n_cols). */
                              /*   Square levels: the library builds it itself.  Rank-partitioned levels (A has ghost  */
                              /*   columns): the caller, who holds the P rows of the ghost vertices, supplies it; its    */
                              /*   columns index the coarse level's [owned | ghost] vector (see amgx_cycle_up).          */
  int32_t gs_block_rows;      /* AMGX_SM_GS, scalar levels: 0 = multicolour Gauss-Seidel over the whole level (one launch per  */
                              /*   colour); B > 0 = BLOCK-HYBRID Gauss-Seidel, one launch per sweep: blocks of B consecutive  */
                              /*   rows (B = 1024, 512, ..., 64) are swept like the ranks of the reference's hybrid smoother   */
                              /*   (gssmoother.cpp:709-861): GS inside a block in colour order, couplings that leave the block */
                              /*   frozen at their sweep-start values.  `color` then only has to separate coupled rows of the  */
                              /*   SAME block (amgh_coloring_blocked) and `dinv` should be the inverse of the l1-modified      */
                              /*   diagonal (amgh_hybrid_dinv).  Rows may have at most 16 * (1024 / B) + 1 entries.            */
                              /*   Square-block levels (2x2, 3x3, 6x6): B = block rows per workgroup (amgh_hybrid_dinv_block).  */
  const int32_t* gs_block_ids;/* optional, square-block levels with gs_block_rows = B > 0: [n] sweep block of every block row   */
                              /*   (ids 0 .. n_blocks-1, at most B rows per block) instead of runs of B consecutive rows --     */
                              /*   compact blocks (amgh_compact_blocks) freeze fewer couplings, like the mesh-partitioner        */
                              /*   subdomains of the reference's hybrid smoother; colours from amgh_coloring_blockids, dinv     */
                              /*   from amgh_hybrid_dinv_block_ids                                                              */
  const int32_t* gs_block_color;/* optional, square-block levels with gs_block_rows = B > 0: [n] colour of the SWEEP BLOCK of every  */
                              /*   block row (a colouring of the block graph, amgh_bgs_coloring: coupled blocks differ; constant     */
                              /*   inside a block).  A sweep is then one launch per block colour, in place: exact Gauss-Seidel in    */
                              /*   the order (block colour, block, in-block colour) -- GSS3's loop (gssmoother.cpp:196-257) in a      */
                              /*   parallel order -- instead of the hybrid form's frozen couplings between blocks; `dinv` is then     */
                              /*   the plain (pseudo-)inverse of the diagonal blocks (amgh_calc_dinv), no l1 modification.            */
  int32_t gs_n_block_colors;  /*   number of block colours (0: none, hybrid form)                                                    */
} amgx_level_desc;

typedef struct amgx_hierarchy_desc {
  int32_t n_levels;
  const amgx_level_desc* levels;
  int32_t cycle;              /* AMGX_CYCLE_*                                                        */
  int32_t clev;               /* AMGX_CLEV_*                                                         */
  int64_t coarse_n;           /* scalar size of the coarsest level                                   */
  const double* coarse_inv;   /* dense [coarse_n^2] inverse on the free dofs (crs_inv, amg_pc.cpp:843-928); NULL with clev = INV:    */
                              /*   amgx_create inverts the coarsest level matrix on its free dofs itself, on the device (blocked     */
                              /*   Gauss-Jordan, trailing updates as f64 MFMA tiles; SPD required; up to AMGX_COARSE_DENSE_MAX =     */
                              /*   16384 unknowns) -- what the host setup leaves to the device beyond 4096 unknowns                  */
  int32_t device;             /* HIP device ordinal                                                  */
  int32_t use_graph;          /* 1: capture each distinct (b, x) cycle into a hipGraph and replay it */
} amgx_hierarchy_desc;

typedef struct amgx_handle_t* amgx_handle;

const char* amgx_last_error(amgx_handle h);

int amgx_create(const amgx_hierarchy_desc* desc, amgx_handle* out);
int amgx_destroy(amgx_handle h);

/* All work of the handle is enqueued on ONE stream.  A new handle owns a private non-blocking stream; this
 * call switches it to the caller's hipStream_t (NULL = the legacy default stream), so that the caller's own
 * kernels and events on that stream order with the preconditioner without host synchronisation. */
int amgx_set_stream(amgx_handle h, void* hip_stream);
int amgx_synchronize(amgx_handle h);

/* x = C b : BaseAMGPC::Mult / AMGMatrix::Mult (amg_pc.cpp:467-470, amg_matrix.cpp:377-378).
 * b_status: 0 = DISTRIBUTED, 1 = CUMULATED (single GPU: both mean the same). */
int amgx_apply(amgx_handle h, const double* b, double* x, int b_status, int flags);
/* x += s * C b : AMGMatrix::MultAdd (amg_matrix.cpp:385-389) */
int amgx_apply_add(amgx_handle h, double s, const double* b, double* x, int flags);

/* smoothers[level]->Smooth (dir = 0) / SmoothBack (dir = 1) with the reference's flag contract
 * (base_smoother.hpp:68-112); goes through the ProxySmoother when sm_steps > 1 or sm_symm. */
int amgx_smooth(amgx_handle h, int level, int dir, double* x, const double* b, double* res,
                int res_updated, int update_res, int x_zero, int flags);
/* AMGMatrix::SmoothVFromLevel (amg_matrix.cpp:310-374) */
int amgx_smooth_v_from_level(amgx_handle h, int level, double* x, const double* b, double* res,
                             int res_updated, int update_res, int x_zero, int flags);
/* Stage entry points for rank-partitioned levels: a level matrix may have n_cols > n_rows, the
 * trailing columns being ghost entries owned by other ranks.  The caller fills the ghost part of the gathered vector
 * (halo exchange, e.g. torch.distributed over RCCL) and drives the cycle stage by stage:
 *   amgx_jacobi_pre : x = omega*Dinv*b, r = b - A x     b: n_cols entries (ghosts valid), x, r: n_rows
 *                     (RichardsonSmoother::Smooth with res_updated = update_res = x_zero = 1, base_smoother.cpp:61-74)
 *   amgx_prolong    : x_out = x_in + fac * P x_coarse   (ProlMap::AddC2F, dof_map.cpp:697-709, out of place)
 *   amgx_jacobi_post: x_out = x_in + omega*Dinv*(b - A x_in)   x_in: n_cols entries (ghosts valid), x_out != x_in */
int amgx_jacobi_pre(amgx_handle h, int level, const double* b, double* x, double* r, int flags);
/* The same two stages in the form the single-GPU V-cycle runs them (fused pre-smoothing + restriction, post-smoothing
 * folded into the prolongation); only for Jacobi levels that have Q (amgx_matrix_info(which = 4) reports it):
 *   amgx_cycle_down : z = S(S0(b)) stored to x (two Jacobi steps from zero WITHOUT coarse correction), and
 *                     b_coarse = P^T (b - A omega*Dinv*b)         b: n_cols entries (ghosts valid); x: n_rows;
 *                     b_coarse: owned rows of the coarse level     (amg_matrix.cpp:193-212)
 *   amgx_cycle_up   : x += Q x_coarse  => x = result of  x_pre + P x_c  followed by the Jacobi post-smoothing step
 *                     x_coarse: Q.n_cols entries (coarse [owned | ghost], ghosts valid)   (amg_matrix.cpp:263-302)
 * One halo exchange per stage (b before down, the coarse x before up) instead of two per level. */
int amgx_cycle_down(amgx_handle h, int level, const double* b, double* x, double* b_coarse, int flags);
int amgx_cycle_up(amgx_handle h, int level, double* x, const double* x_coarse, int flags);
/* r = b - A_level x   (BaseSmoother::CalcResiduum, base_smoother.hpp:132-142); x: n_cols entries, b, r: n_rows.
 * With amgx_smooth (whose x may carry ghost entries too: they are read, never written) this gives the stages of the
 * hybrid Gauss-Seidel smoother of rank-partitioned levels: local sweep on the owned rows with the off-rank values
 * frozen (reference HybridGSSmoother, gssmoother.cpp:709-861, with dinv = inverse of the modified diagonal). */
int amgx_residual(amgx_handle h, int level, const double* x, const double* b, double* r, int flags);
int amgx_jacobi_post(amgx_handle h, int level, const double* x_in, const double* b, double* x_out, int flags);
int amgx_prolong(amgx_handle h, int level, double fac, const double* x_in, const double* x_coarse, double* x_out, int flags);

/* y = A_level x  (GetMatrix(level).Mult); x has n_cols entries */
int amgx_matvec(amgx_handle h, int level, const double* x, double* y, int flags);
/* DOFMap::TransferF2C (x_coarse = P^T x_fine) and AddC2F (x_fine += fac * P x_coarse), dof_map.cpp:636-709 */
int amgx_transfer_f2c(amgx_handle h, int level, const double* x_fine, double* x_coarse, int flags);
int amgx_add_c2f(amgx_handle h, int level, double fac, double* x_fine, const double* x_coarse, int flags);
/* crs_inv->Mult on the coarsest level */
int amgx_coarse_solve(amgx_handle h, const double* rhs, double* x, int flags);

/* GetNLevels / GetNDof / GetBlockSize (python_amg.hpp:15-103) */
int amgx_n_levels(amgx_handle h);
int amgx_level_info(amgx_handle h, int level, int64_t* n, int32_t* bs, int64_t* nnz);
/* how the V-cycle of this handle is launched (no reference counterpart; the reference's cycle is a host loop,
 * amg_matrix.cpp:183-302):  tail_level = first level run inside the single-workgroup tail kernel (-1: none);
 * dense_level = first level of the COLLAPSED coarse levels (-1: none): the sub-cycle on the levels >= dense_level is a
 * fixed linear operator, formed once at amgx_create by running the device's own sub-cycle on the unit vectors and applied
 * as one dense GEMV of dense_n x dense_n doubles (same operator, summation order differs: rounding-level differences).
 * Environment of amgx_create: AMGX_NO_DENSE_TAIL=1 disables, AMGX_DENSE_MAX=<n> caps dense_n (default 8192). */
int amgx_cycle_info(amgx_handle h, int32_t* tail_level, int32_t* dense_level, int64_t* dense_n);
/* device-format report per level matrix: which = 0 A, 1 P, 2 PT, 3 A' = A*omega*Dinv (pre-smoothing image),
 * 4 Q = (I - omega*Dinv*A) P (post-smoothing folded into the prolongation), 5 the "local window" image of A' that the fused down
 * kernel of a long-row level reads (chunk-local 16-bit columns, gathered vector staged in LDS), 6 the local-window image of Q
 * (window-local columns, the coarse values of a 512-row window staged in LDS); fmt: -1 not built, 0 CSR-vector,
 * 1 sliced-ELL, 2 block sliced-ELL, 3 sliced-ELL with length-sorted row windows, 5 local-window sliced-ELL, 4 rigid-body transfer blocks (P_ik = w_ik Q(t_ik)
 * stored as (column, w, t): detected block by block at amgx_create, elasticity_energy.hpp:447-490); stored_entries counts padding
 * (for the traffic model in DESIGN.md) */
int amgx_matrix_info(amgx_handle h, int level, int which, int32_t* fmt, int64_t* stored_entries, int32_t* lanes_per_row);

/* bytes of matrix data (values, indices, pointers, in the device encoding) that one SpMV with this matrix
 * streams from HBM -- the model value behind "traffic" in DESIGN.md */
int amgx_matrix_stream_bytes(amgx_handle h, int level, int which, int64_t* bytes);

/* measurement hook for bench.py: launches one hot-path kernel `reps` times on the handle's stream,
 * bracketed by HIP events, and returns the average duration in milliseconds.
 *   op = 0: residual SpMV  r = b - A_level x      (the dominant kernel of the Jacobi V-cycle)
 *   op = 1: fused Jacobi post-smooth  x' = x + omega*dinv*(b - A_level x)
 *   op = 2: restriction  b_c = P^T r      op = 3: prolongation  x += P x_c
 *   op = 4: one whole cycle (amgx_apply on internal vectors)
 *   op = 5: pre-smoothing + restriction as the V-cycle runs it on this level (fused / folded where built)
 *   op = 6: coarse-grid correction + post-smoothing as the V-cycle runs it on this level
 *   op = 7: sell_pre_restrict_kernel alone (op 5 without the small restrict_sum_kernel); error if the level has none
 *   op = 8: the same kernel timed INSIDE the cycle: `reps` whole cycles are launched directly (no graph) with HIP events
 *           around that one kernel; the average is what rocprofv3 --kernel-trace reports for it (roofline.achieved)
 *   op = 9: like 8 for the backward block-hybrid Gauss-Seidel sweep of the level (gsb_sweep_kernel, the dominant kernel
 *           of a Gauss-Seidel cycle); error if the level has no such sweep */
int amgx_time_op(amgx_handle h, int level, int op, int reps, double* avg_ms);

/* ---- GSS4: Gauss-Seidel on a subset of the rows, on a compressed device copy ------------------------------------------
 * Reference: GSS4<TM> (src/base/smoothers/gssmoother.hpp:99-143, gssmoother.cpp:407-583), the smoother of the EX stage of
 * HybridGSSmoother (gssmoother.cpp:697-698).  Only the rows of the subset are copied to the device ("xdofs" and the
 * compressed matrix cA of GSS4::SetUp, :456-507) together with the transposed couplings the RES form needs.
 *   amgx_gss4_smooth      Smooth (dir 0) / SmoothBack (dir 1):        x_k += dinv_k (b_k - A_k: x)               (:565-583)
 *   amgx_gss4_smooth_res  SmoothRES / SmoothBackRES:  w = -dinv_k res_k; res += A_k:^T w; x_k -= w               (:543-561)
 *   amgx_gss4_mult_add    MultAdd:                                    x_k += s dinv_k b_k  for k in the subset   (:531-539)
 * Rows of equal colour are relaxed in parallel, colours ascending (dir 0) or descending (dir 1); the reference's
 * one-row-at-a-time order is the special case of one row per colour.  Vectors: x and b have A.n_rows * bs entries, except
 * that the gathered vector (x of amgx_gss4_smooth, res of amgx_gss4_smooth_res) has A.n_cols * bs. */
typedef struct amgx_gss4_desc {
  amgx_matrix A;              /* square blocks 1, 2, 3, 6; n_cols >= n_rows (ghost columns are read, never updated)          */
  const uint8_t* subset;      /* [n_rows] 1 = row is smoothed, or NULL (all rows)                                            */
  const double* dinv;         /* [n_rows*bs*bs] inverse of the diagonal block or of its replacement (mod_diag), read on the   */
                              /*   subset only: the caller applies CalcInverse / CalcPseudoInverseTryNormal (:417-438)        */
  const int32_t* color;       /* [n_rows] colour of the rows of the subset (coupled rows differ), -1 elsewhere               */
  int32_t n_colors;
  int32_t device;
} amgx_gss4_desc;
typedef struct amgx_gss4_t* amgx_gss4;

int amgx_gss4_create(const amgx_gss4_desc* desc, amgx_gss4* out);
int amgx_gss4_destroy(amgx_gss4 g);
const char* amgx_gss4_last_error(amgx_gss4 g);            /* g may be NULL for create-time errors */
int amgx_gss4_set_stream(amgx_gss4 g, void* hip_stream);
int amgx_gss4_synchronize(amgx_gss4 g);
/* rows of the subset, rows the subset couples to (= rows of res the RES form updates), blocks stored */
int amgx_gss4_info(amgx_gss4 g, int64_t* n_rows, int64_t* n_touched, int64_t* nnz);
int amgx_gss4_smooth(amgx_gss4 g, int dir, double* x, const double* b, int flags);
int amgx_gss4_smooth_res(amgx_gss4 g, int dir, double* x, double* res, int flags);
int amgx_gss4_mult_add(amgx_gss4 g, double s, const double* b, double* x, int flags);

/* Krylov solvers with all vectors resident on the GPU: the callers of the preconditioner on the
 * reference side are NGSolve's CGSolver / GMRes (tests/h1/amg_utils.py:346).  Operator = the level-0 matrix of the handle,
 * preconditioner = the handle's cycle (use_precond = 0: none).  x holds the initial guess and receives the solution.
 *   amgx_pcg  : err_k = sqrt(|<C r_k, r_k>|), stops at err_k <= tol * err_0 (CGSolver's criterion)
 *   amgx_gmres: restarted GMRES(restart), left-preconditioned, err_k = |C r_k|
 * errs (optional, maxit + 1 entries) receives err_0 ... err_iters; *iters the iteration count.  BLAS-1 work runs in
 * hand-written kernels with deterministic reductions; the host reads one scalar per iteration. */
int amgx_pcg(amgx_handle h, const double* b, double* x, double tol, int maxit, int use_precond, int flags, double* errs, int32_t* iters);
int amgx_gmres(amgx_handle h, const double* b, double* x, double tol, int maxit, int restart, int use_precond, int flags, double* errs,
               int32_t* iters);

/* ------------------------------------------------------------------------------------------------------------------
 * Rank-partitioned hierarchies (one process per GPU; SURVEY.md 8b "halo tables + RCCL communicator", 8e).
 *
 * What the reference does with MPI -- every rank of the communicator calls AMGMatrix::Mult collectively
 * (src/base/solve/amg_matrix.cpp:160-307), smoothers exchange halos through DCCMap (src/base/linalg/dcc_map.cpp:17-302)
 * around their local stages (src/base/smoothers/hybrid_base_smoother.cpp:501-574) -- is one call here: amgx_dist_apply.
 * Layout: a rank stores the rows of the vertices it owns, columns [owned | ghost], ghosts grouped by owning rank
 * (ascending), owned rows ordered [interior | boundary] (interior = no ghost column).  Pack kernels + ncclSend / ncclRecv
 * run on a communication stream while the interior rows are processed; nothing synchronises the host.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct amgx_comm_t* amgx_comm;
typedef struct amgx_dist_t* amgx_dist;
typedef struct amgx_halo_t* amgx_halo;
enum {
  AMGX_COMM_RCCL = 0,         /* one rank per process, ncclCommInitRank over xGMI (replaces the MPI communicator)          */
  AMGX_COMM_LOCAL = 1         /* all n_ranks ranks live in this process on one GPU ("virtual ranks": tests, rehearsal);     */
                              /*   same kernels, streams and events, device copies instead of ncclSend / ncclRecv           */
};
#define AMGX_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: called by rank 0, the caller distributes the 128 bytes (e.g. torch.distributed / MPI_Bcast) */
int amgx_comm_unique_id(char* id128);
/* RCCL: collective over all ranks (ncclCommInitRank); LOCAL: rank and id128 are ignored */
int amgx_comm_create(int kind, int n_ranks, int rank, const char* id128, int device, amgx_comm* out);
int amgx_comm_destroy(amgx_comm c);                      /* also destroys the hierarchies created on it */
const char* amgx_comm_last_error(amgx_comm c);           /* c may be NULL for create-time errors        */
/* compute stream of the communicator (and of every hierarchy on it); NULL = its own stream */
int amgx_comm_set_stream(amgx_comm c, void* hip_stream);
int amgx_comm_synchronize(amgx_comm c);
int amgx_comm_info(amgx_comm c, int32_t* kind, int32_t* n_ranks, int32_t* rank, int64_t* n_exchanges);
/* how amgx_dist_apply launches (no reference counterpart: the reference's cycle is a host loop around MPI calls,
 * amg_matrix.cpp:160-307 / dcc_map.cpp:76-178): with device vectors the whole collective cycle -- both streams, pack kernels,
 * ncclSend / ncclRecv / ncclAllGather -- is captured once per (b, x, b_status) into a hipGraph and replayed.  enabled = 0
 * after AMGX_DIST_GRAPH=0 or a failed capture (note: why; direct launches from then on); n_graphs = captured cycles held,
 * n_replays = applications served by a graph launch. */
int amgx_comm_graph_info(amgx_comm c, int32_t* enabled, int64_t* n_graphs, int64_t* n_replays);
const char* amgx_comm_graph_note(amgx_comm c);

/* halo tables of one level = DCCMap's m_ex_dofs / g_ex_dofs (dcc_map.cpp:480-543) in owner-row form */
typedef struct amgx_halo_desc {
  int32_t n_peers;
  const int32_t* peer_rank;   /* [n_peers] ascending (GetDistantProcs)                                              */
  const int64_t* send_ptr;    /* [n_peers+1]                                                                        */
  const int32_t* send_idx;    /* owned block rows whose values peer k ghosts (m_ex_dofs[k]), in the order of the     */
                              /*   peer's ghost segment                                                              */
  const int64_t* recv_ptr;    /* [n_peers+1] ghost block rows [recv_ptr[k], recv_ptr[k+1]) belong to peer k           */
                              /*   (g_ex_dofs[k]; contiguous: ghosts are grouped by owner)                           */
  int64_t n_interior;         /* owned rows [0, n_interior) have no ghost column (split_ind analogue,                 */
                              /*   gssmoother.cpp:664-678); 0 = no overlap of communication and computation          */
} amgx_halo_desc;

typedef struct amgx_dist_desc {
  amgx_hierarchy_desc top;    /* levels 0..k of this rank: A = owned rows x [owned | ghost]; P, PT rank-local; Jacobi   */
                              /*   levels with fold: Q (columns = coarse [owned | ghost]); dinv / colours / blocks as    */
                              /*   usual (dinv over n_cols entries for Jacobi); level k: only A (its shape is used)      */
  const amgx_halo_desc* halo; /* [k] halo tables of the levels 0..k-1                                                    */
  amgx_hierarchy_desc tail;   /* replicated coarse hierarchy; its level 0 = level k gathered in rank order (CtrMap,      */
                              /*   src/base/coarsening/dof_contract.cpp:49-223, without the hop back)                    */
  const int64_t* counts;      /* [n_ranks] owned block rows of level k per rank                                          */
  const int64_t* kmap;        /* [kmap_len] level-k [owned | ghost] scalar entries -> scalar index in the gathered vector */
  int64_t kmap_len;
  int32_t rank;               /* this rank (LOCAL communicators: ranks are created in order 0, 1, ...)                   */
  int32_t fold;               /* Jacobi: 1 = post-smoothing folded into the prolongation (2k-1 exchanges), 0 = literal   */
  const int32_t* gs_stage;    /* optional [4*k]: colour ranges of the hybrid Gauss-Seidel stages per level,              */
                              /*   [s0,s1) first local part, [s1,s2) rows that read ghosts ("EX"), [s2,s3) second local   */
                              /*   part (gssmoother.cpp:721-782); NULL: one stage                                        */
} amgx_dist_desc;

int amgx_dist_create(amgx_comm c, const amgx_dist_desc* desc, amgx_dist* out);
int amgx_dist_destroy(amgx_dist d);
/* x = C b collectively.  b, x: one pointer per local rank (RCCL: one).  b_status 1 (CUMULATED): b holds the owned entries;
 * 0 (DISTRIBUTED): b holds [owned | ghost] entries whose ghost part are contributions to the owners, added first
 * (b.Distribute() state, amg_matrix.cpp:164; DCCMap DIS2CO).  x: owned entries (CUMULATED on the owners).
 * flags: AMGX_HOST_PTR / AMGX_DEVICE_PTR. */
int amgx_dist_apply(amgx_comm c, const double* const* b, double* const* x, int b_status, int flags);
/* Preconditioned CG on the rank-partitioned level-0 operator, collectively (the reference's driver: NGSolve CGSolver on
 * ParallelVectors, tests/h1/amg_utils.py:337-363, whose inner products are MPI all-reduces): level-0 product with one owner ->
 * ghost exchange behind the interior rows, preconditioner = amgx_dist_apply (replayed from its graph), the two inner products
 * per iteration are deterministic local reductions + one ncclAllReduce of a device scalar each, recurrence scalars stay on the
 * device, every rank reads the same error value and takes the same decision.  b, x: device pointers to the OWNED entries, one
 * per local rank; x holds the initial guess.  err_k, tol, errs, iters as amgx_pcg. */
int amgx_dist_pcg(amgx_comm c, const double* const* b, double* const* x, double tol, int maxit, int use_precond, int flags, double* errs,
                  int32_t* iters);
/* Restarted GMRES(restart), left-preconditioned, on the rank-partitioned level-0 operator, collectively (reference driver:
 * ngsolve.krylovspace.GMRes on ParallelVectors): Arnoldi inner products = one fused local pass + one ncclAllReduce of j + 1
 * scalars per Gram-Schmidt pass, Givens rotations on every rank alike.  Arguments as amgx_dist_pcg / amgx_gmres. */
int amgx_dist_gmres(amgx_comm c, const double* const* b, double* const* x, double tol, int maxit, int restart, int use_precond, int flags,
                    double* errs, int32_t* iters);
/* Measurement hook, collective (every rank calls it with the same arguments): `reps` whole cycles with direct launches, HIP
 * events around ONE kernel of the first local rank's level `level` -- op 8: the fused Jacobi pre-smoothing + residual +
 * restriction kernel over the INTERIOR rows (the launch that runs beside the halo exchange); op 9: the backward block-hybrid
 * Gauss-Seidel sweep over the interior blocks.  The counterpart of amgx_time_op(op 8 / 9) for rank-partitioned hierarchies:
 * the kernel is timed where it runs, with the exchange in flight next to it. */
int amgx_dist_time_kernel(amgx_comm c, int level, int op, int reps, double* avg_ms);
/* the level-0 right-hand-side buffer of a rank ([owned | ghost], device): filling it in place saves the copy of b */
int amgx_dist_rhs_buffer(amgx_dist d, double** b, int64_t* n_owned, int64_t* n_ext);
/* borrowed handles of the rank-partitioned levels and of the replicated tail, for queries / measurement only */
int amgx_dist_handles(amgx_dist d, amgx_handle* top, amgx_handle* tail);

/* ---- setup products on the device (cold path; host arrays in, host arrays out) ---------------------------------------------
 * C = A B (MatMultABImpl, src/base/linalg/utils_sparseMM.cpp:107-238) and the Galerkin product A_c = (P^T A) P
 * (RestrictMatrix, utils_sparseMM.hpp:93-109; the intermediate P^T A stays on the device) for (block-)CSR matrices (result blocks
 * of at most 36 entries: 1x1 ... 6x6), columns ascending per row.  Entry (i, j) is accumulated as c = fma(a_ik, b_kj, c) over k
 * ascending from c = 0 (blocks: c[r][s] = fma(a[r][q], b[q][s], c[r][s]), q ascending inside every block product) -- the order and
 * the fused multiply-add of the host library's product (csrc/host/sparse.cpp), so the result is the same bit for bit.
 * Returns 0 = done (*out holds the result on the device, *n_rows / *nnz its size in block rows / blocks: allocate and call
 * amgx_csr_result_fetch, which copies the arrays out and releases the result), 2 = not supported (larger blocks, a row with more
 * than 8192 products or, block matrices, more distinct columns than one wave's table holds): the caller keeps its own product,
 * 1 = error (amgx_last_error(NULL)).
 * The host setup library takes the pair (amgx_galerkin, amgx_csr_result_fetch) through amgh_set_galerkin_hook (amgh.h). */
typedef struct amgx_csr_result_t* amgx_csr_result;
int amgx_device_count(int32_t* n);      /* visible HIP devices (0 without a GPU or driver; never an error) */
int amgx_spgemm(const amgx_matrix* A, const amgx_matrix* B, amgx_csr_result* out, int64_t* n_rows, int64_t* nnz);
int amgx_galerkin(const amgx_matrix* PT, const amgx_matrix* A, const amgx_matrix* P, amgx_csr_result* out, int64_t* n_rows, int64_t* nnz);
int amgx_csr_result_fetch(amgx_csr_result res, int64_t* rowptr, int32_t* col, double* val);

/* stand-alone halo map = DCCMap (dcc_map.hpp:20-90): mode 0 owner -> ghost overwrite (StartCO2CU / ApplyCO2CU,
 * dcc_map.cpp:138-178), mode 1 ghost -> owner add with the ghost entries zeroed (StartDIS2CO / ApplyDIS2CO, :76-136).
 * vecs: device vectors of (n_owned + n_ghost) * bs entries, ordered on the communicator's stream. */
int amgx_halo_create(amgx_comm c, const amgx_halo_desc* d, int64_t n_owned, int64_t n_ghost, int32_t bs, int32_t rank, amgx_halo* out);
int amgx_halo_destroy(amgx_halo h);
int amgx_halo_exchange(amgx_comm c, int n_local, const amgx_halo* halos, double* const* vecs, int mode);

#ifdef __cplusplus
}
#endif
#endif
