/* oracle.h -- CPU oracle for the NgsAMG V-cycle apply path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ngsamg_amd/, include/) may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker
 * or the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (LukasKogler/NgsAMG) cannot be built or run here (it needs NGSolve,
 * which is absent: SURVEY.md section 8c) and its tests hold no golden vectors for this path.  This file
 * is therefore a literal restatement of the reference's algorithm, pinned only by (i) structural
 * invariants the reference itself asserts and (ii) the iteration budgets of its pytest suite.
 *
 * Restated reference code (file:line relative to the reference tree):
 *   cycle            src/base/solve/amg_matrix.cpp:37-107 (W), 110-157 (BS), 160-307 (V), 310-374 (FromLevel),
 *                    377-393 (Mult / MultAdd)
 *   smoother flags   src/base/smoothers/base_smoother.hpp:68-112 (Smooth, SmoothK, SmoothSymm), 169-229 (Proxy)
 *   Jacobi           src/base/smoothers/base_smoother.cpp:61-114
 *   Gauss-Seidel     src/base/smoothers/gssmoother.cpp:111-139 (free range), 196-257 (RHS), 261-315 (RES),
 *                    350-398 (Smooth / SmoothBack)
 *   block GS         src/base/smoothers/block_gssmoother.cpp:216-283 (block order), 287-328 (RHS form), 356-392 (RES
 *                    form), 434-498 (Smooth / SmoothBack flag logic)
 *   transfers        src/base/coarsening/dof_map.cpp:636-709
 *   coarse solve     src/base/precond/amg_pc.cpp:843-928 (exact inverse on the free dofs)
 *   GSS4             src/base/smoothers/gssmoother.cpp:407-583 (gss4.c)
 */
#ifndef NGSAMG_ORACLE_H
#define NGSAMG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_matrix {
  int64_t n_rows, n_cols;
  int32_t br, bc;
  const int64_t* rowptr;
  const int32_t* col;
  const double* val;
} orc_matrix;

enum { ORC_SM_JACOBI = 0, ORC_SM_GS = 1, ORC_SM_BGS = 2 };
enum { ORC_CYCLE_V = 0, ORC_CYCLE_W = 1, ORC_CYCLE_BS = 2 };

typedef struct orc_level {
  orc_matrix A, P, PT;        /* P, PT unused on the coarsest level                                */
  const uint8_t* free;        /* [n] or NULL (all free)                                            */
  const double* dinv;         /* [n*bs*bs]                                                         */
  int32_t sm_type;            /* ORC_SM_*                                                          */
  double omega;               /* Jacobi damping (reference default 0.9)                            */
  int32_t sm_steps;           /* ProxySmoother nsteps (1 = no proxy unless sm_symm)                */
  int32_t sm_symm;            /* ProxySmoother symm                                                */
  const int32_t* gs_order;    /* NULL = natural row order (reference-exact); else forward visiting */
  int64_t gs_order_len;       /*   order of the rows (e.g. colour-major), backward = reversed      */
  const int32_t* gs_block;    /* NULL, or the owner ("rank") of every row: HYBRID Gauss-Seidel -- couplings to rows of  */
                              /*   another block use the values from the start of the sweep (reference              */
                              /*   HybridGSSmoother: local GS on M, G x_old moved to the right-hand side,           */
                              /*   hybrid_base_smoother.cpp:296-495, gssmoother.cpp:709-861); dinv then holds the   */
                              /*   inverse of the modified diagonal (hybrid_smoother_utils.hpp:111-142)             */
  /* ORC_SM_BGS: block Gauss-Seidel (reference BSmoother, block_gssmoother.cpp:216-470): blocks of block rows with    */
  /*   dense inverses of the diagonal blocks; natural block order forward, reversed backward (:266-283)               */
  int32_t n_blocks;
  const int32_t* block_ptr;   /* [n_blocks+1]                                                                      */
  const int32_t* block_rows;  /* block rows grouped by block                                                        */
  const int64_t* bdinv_ptr;   /* [n_blocks+1] offsets into bdinv                                                    */
  const double* bdinv;        /* per block M x M column-major, M = bs * block size                                  */
  const int32_t* block_order; /* NULL = natural (reference, sequential); else forward visiting order of the blocks  */
                              /*   (colour-major = what the GPU kernel does)                                        */
} orc_level;

typedef struct orc_desc {
  int32_t n_levels;
  const orc_level* levels;
  int32_t cycle;              /* ORC_CYCLE_*                                                       */
  int32_t clev_inv;           /* 1: exact coarse solve (clev = inv), 0: x_L = 0                    */
} orc_desc;

typedef struct orc_handle orc_handle;

int orc_create(const orc_desc* d, orc_handle** out);
void orc_destroy(orc_handle* h);
const char* orc_last_error(void);

/* AMGMatrix::Mult / MultAdd */
int orc_apply(orc_handle* h, const double* b, double* x);
int orc_apply_add(orc_handle* h, double s, const double* b, double* x);
/* AMGMatrix::SmoothVFromLevel */
int orc_smooth_v_from_level(orc_handle* h, int level, double* x, const double* b, double* res,
                            int res_updated, int update_res, int x_zero);
/* smoothers[level]->Smooth (dir = 0) / SmoothBack (dir = 1), through the ProxySmoother if configured */
int orc_smooth(orc_handle* h, int level, int dir, double* x, const double* b, double* res,
               int res_updated, int update_res, int x_zero);
/* DOFMap::TransferF2C / AddC2F on one level */
int orc_transfer_f2c(orc_handle* h, int level, const double* x_fine, double* x_coarse);
int orc_add_c2f(orc_handle* h, int level, double fac, double* x_fine, const double* x_coarse);
/* crs_inv->Mult */
int orc_coarse_solve(orc_handle* h, const double* rhs, double* x);
/* y = A_level x */
int orc_matvec(orc_handle* h, int level, const double* x, double* y);
/* preconditioned CG with the handle as preconditioner (NGSolve CGSolver stand-in):
 *   err_k = sqrt(<C r_k, r_k>); stops when err_k <= tol * err_0.  errs has room for maxit+1 entries.
 *   Returns the iteration count in *iters.  h may be NULL (no preconditioner) if A is given. */
int orc_pcg(orc_handle* h, const orc_matrix* A, const double* b, double* x, double tol, int maxit,
            double* errs, int* iters);
/* restarted GMRES(restart), left-preconditioned, modified Gram-Schmidt (NGSolve GMRes stand-in): err_k = |C r_k|, stops at
 * err_k <= tol * err_0; errs has room for maxit+1 entries */
int orc_gmres(orc_handle* h, const orc_matrix* A, const double* b, double* x, double tol, int maxit, int restart,
              double* errs, int* iters);
/* number of OpenMP threads used for Jacobi / SpMV / transfers (GS stays sequential) */
void orc_set_threads(int n);
/* multi-threaded CPU baseline only: replace the borrowed level arrays by private copies whose pages are first written
 * by the thread that streams them (NUMA placement); call after orc_set_threads / orc_create */
int orc_first_touch(orc_handle* h);

/* GSS4 (gss4.c): Gauss-Seidel on a subset of the rows; dinv_full = [n*bs*bs] inverted (replacement) diagonal blocks, read on
 * the subset.  order (optional): visiting order of the compressed rows 0..m-1 (forward), NULL = ascending (reference). */
typedef struct orc_gss4 orc_gss4;
int orc_gss4_create(const orc_matrix* A, const uint8_t* subset, const double* dinv_full, orc_gss4** out);
void orc_gss4_destroy(orc_gss4* g);
int64_t orc_gss4_rows(const orc_gss4* g);
int64_t orc_gss4_nnz(const orc_gss4* g);
int orc_gss4_set_order(orc_gss4* g, const int32_t* order, int64_t len);
int orc_gss4_smooth(const orc_gss4* g, int backwards, double* x, const double* b);          /* Smooth / SmoothBack       */
int orc_gss4_smooth_res(const orc_gss4* g, int backwards, double* x, double* res);          /* SmoothRES / SmoothBackRES */
int orc_gss4_mult_add(const orc_gss4* g, double s, const double* b, double* x);

#ifdef __cplusplus
}
#endif
#endif
