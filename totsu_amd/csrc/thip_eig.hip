// thip_eig.hip -- LinAlgEx::map_eig (totsu_core/src/linalg_ex.rs:44-65) and ConePSD::proj
// (totsu_core/src/cone_psd.rs:56-79) on the device.
//
// Contract (F64LAPACK recipe, totsu_f64lapack/src/f64lapack.rs:172-255; CUDA: f32cuda.rs:196-370):
//   packed upper (by columns) --vec_to_mat, diag*scale--> M ; M = Z diag(w) Z^T ; R = sum_i map(w_i) z_i z_i^T ;
//   R --diag/scale, pack upper--> packed.
//
// Two engines:
//  (1) eigen-decomposition by parallel one-sided Jacobi (Hestenes) on the definite shift M + sigma I,
//      sigma > ||M||_F: for a positive definite matrix the right singular vectors ARE the eigenvectors and no
//      +/-lambda pair can mix.  Round-robin ordering gives n/2 independent column-pair rotations per step, one
//      wavefront per pair (3 width-64 shuffle-tree dots, then a rotation of two G and two V columns).  n <= 64
//      runs as one workgroup with G and V in LDS; larger n as one launch per step.  This engine serves the
//      arbitrary-closure contract (thip_eig_decompose / thip_eig_rebuild) and map_kind 1 (sqrt).
//  (2) PSD projection without eigenvectors: P = (M + M sign(M)) / 2, sign(M) by an odd matrix polynomial
//      iteration -- a chain of n x n x n f32 GEMMs on v_mfma_f32_32x32x2_f32.  This is where the matrix cores
//      are a real dense contraction; a Householder/QR chain at k = 500 is O(k) dependent latency-bound
//      steps while this is 50 GEMMs.  Used by ConePSD::proj for n > 20 (measured cross-over with (1)).
#include "thip_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace thip;

// The translation unit in its parts (split in round 6, after the round-4 chain had gone):
#include "thip_eig_jacobi.inc"
#include "thip_eig_gemm.inc"
#include "thip_eig_tridiag.inc"
#include "thip_eig_chain.inc"
