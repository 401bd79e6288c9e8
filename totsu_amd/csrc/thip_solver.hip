// thip_solver.hip -- the conic iteration resident on the GPU.
//
// Native restatement of totsu_core/src/solver/solver.rs:340-657 (SolverCore::{solve, calc_norms, init_vecs,
// calc_precond, update_vecs, criteria_conv, criteria_inf}) and of SelfDualEmbed::{op, trans_op, abssum}
// (solver.rs:109-183) for operators that are dense column-major matrices (MatOp, matop.rs) and a product
// cone of zero / nonneg / second-order / rotated second-order / PSD blocks (cone_*.rs).
//
// Differences from the reference are of SCHEDULE only:
//   * state vectors, tau, kappa, the dots and norms and the termination test stay on the device; the host
//     enqueues iterations back to back and polls a status word (the reference reads >= 8 scalars per
//     iteration through SliceLike::get, solver.rs:551-567,599-608).  Once the device has decided to stop,
//     every later kernel returns immediately, so the result is that of stopping at exactly that iteration;
//   * THIP_SCHED_FUSED: the A x and A^T y products of one stage come from one read of A (dual GEMV);
//   * THIP_SCHED_CARRIED: K*rx is obtained from the criteria products by linearity, rx = x_k - 2 x_{k+1}
//     (solver.rs:555) => A rx_x = (A x_k) - 2 (A x_{k+1}); both right-hand products are recomputed from the
//     iterate every iteration (no recursion, no drift);
//   * THIP_SCHED_REFERENCE issues the reference's six single GEMVs.
// Row-sharded A (one process per GPU): every m-length vector is sharded like the rows, every n-length
// vector is replicated; the only exchange is a sum-all-reduce of the n-vector A_g^T y_g (with the sharded
// scalars riding in its tail) per transposed product.
#include "thip_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

using namespace thip;

// The translation unit in its parts (split in round 6; one TU, so the kernels and the solver object stay file-local):
#include "thip_solver_kernels.inc"
#include "thip_solver_state.inc"
#include "thip_solver_passes.inc"
#include "thip_solver_sweep.inc"
#include "thip_solver_recovery.inc"
#include "thip_solver_api.inc"
