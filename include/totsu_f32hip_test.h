/* totsu_f32hip_test.h -- test hooks and timing probes of libtotsu_f32hip.so.
 *
 * Everything here is exported by the same shared library as include/totsu_f32hip.h but is NOT part of the interface a
 * binding wraps (the Rust crate declares them behind its `test-hooks` feature): fault injection into the persistent
 * kernel's recovery paths, switches that force an engine, and entry points that run one kernel alone so that
 * tests/ can compare it with numpy / the oracle and tools/ can time it.  None of them is called by the product path. */
#ifndef TOTSU_F32HIP_TEST_H
#define TOTSU_F32HIP_TEST_H

#include "totsu_f32hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test switch: 0 = the library's choice, 1 = the QL engine, 2 = the device engine with its certificate forced to fail
 * (exercises the hand-over); + 4 = the Householder reduction as ONE persistent launch over the whole device (granule
 * all-gather per reflector over the fabric), + 8 = the same on the workgroups of one XCD (through its L2: the default up
 * to order 1024), + 12 = one launch per reflector (the default above); + 16 = the one-XCD launch is started with a role
 * missing, so that its bounded spins run out (the time-out path: the launches must take over).  Every call also forgets
 * that a persistent launch ever gave up.  DESIGN.md 4.5 */
int thip_test_eig_force(int engine);

/* TEST HOOK: installs a stand-in "collective" that does no arithmetic (the sum over ONE rank) and only takes time: a
 * device spin of latency_us microseconds on the stream the hook is given.  Lets a 1-GPU box measure what each overlap
 * mode hides of a collective's latency (tests/test_gpu_sharded.py). */
int thip_test_spin_allreduce(thip_solver *s, int latency_us);

/* TEST HOOK: kind 1 = the placement census of the next plan (thip_solver_init) reports a bad placement; kind 2 = one
 * workgroup of the after_sweeps-th regular sweep from now withholds its partial dots (once: a transient -- thip_solver_run
 * restores, re-arms and goes on with the one-pass schedule); kind 7 = the same in EVERY sweep from then on (a placement that
 * stays wrong: the second failure hands over to the 2-pass schedule); spin_max > 0 shortens the
 * bound of the gathers' polling loops (default ~2 s) so that a test does not wait for it.  kind 0 clears.
 * kind 3 / 4 (no fault): run the two m-kernels of a step as two launches also where the cones are all element-wise
 * (the form problems with block cones always take) / merged again -- lets a test compare the two forms.
 * kind 5 / 6 (no fault): launch the termination test after every sweep / fold it into the next step's m-kernel again
 * (the default where the m-tail is one of the merged forms: it is launched only when the host is about to look). */
int thip_test_sweep_fault(thip_solver *s, int kind, int64_t after_sweeps, int spin_max);

/* test entry point of the one-pass kernel (thip_sweep.hip): one sweep over the m x n matrix A (device, column-major),
 *   gT = A^T v ; g3 = A^T xy ; u <- u + Su o (-(gP - 2 g3) - c rtau) unless `first` ; gP <- g3 ;
 *   xx_out = xx_in + Tx o (gT + c kappa) ; hN = A u ; h3 = A xx_out            (Kahan terms ku / kx_* may be NULL)
 * `reps` launches are timed with HIP events (reps > 1 only with first != 0, which is idempotent); host_info (8 ints):
 * [0] = the kernel's error word (0 = ok), [1] = members per group, [2] = groups, [3] = panels per group, [4] = 16-byte
 * slots per streaming thread, [5] = polls of all gathers that found a granule missing (summed over workgroups and launches),
 * [6] = the most polls any one gather needed.  force_members > 0: that many workgroups per column group instead of the planner's choice
 * (a power of two the rows fit); pub_agent != 0: partial dots published with agent-scope (sc1) stores. */
typedef struct thip_sweep_test {
    size_t m, n, lda;
    const float *mat_a, *v, *xy, *c, *su, *tx;
    float *u, *ku;
    const float *xx_in, *kx_in;
    float *xx_out, *kx_out, *gp, *hn, *h3;
    float kappa, rtau;
    int32_t first, reps;
    int32_t force_members, pub_agent;
    int32_t variant, elem;          /* variant: columns per panel -- f32: 0 = one, 12 = two; 16-bit elem: 1, 2, 4 (0 = the planner's
                                     * preference).  elem: THIP_A_F32, or THIP_A_BF16 / THIP_A_F16: mat_a then points at 16-bit entries, lda in entries */
    const float *inv_s;             /* THIP_A_F16: 1 / scale per column (thip_to_f16), else NULL */
    float *host_sums;               /* optional, 4 floats on the HOST: the sums over n the sweep leaves for the criteria and the scalar
                                     * updates (tau taken as 1): ||c + A^T xy||^2, c.xx_in, c.u, c.(xx_in - 2 xx_out) */
} thip_sweep_test;
int thip_test_sweep(const thip_sweep_test *t, float *host_ms, int *host_info);

/* test entry point for the matrix-core GEMM of the PSD projection chain: C = alpha * A * B + beta * D + gamma * I_n
 * with A symmetric and B arbitrary, all ld x ld column-major, ld a multiple of 64, zero padded beyond n */
int thip_test_gemm_sym(int n, int ld, float alpha, const float *A, const float *B, float beta, const float *D,
                       float gamma, float *C);
/* the same for either shape and either kernel of the chain, nb matrices per launch (item i at offset i * ld * ld of every
 * operand).  shape 0: C = alpha * X^T * Y + beta * D + gamma * I_n, for operands whose result is symmetric (X = Y, or
 * both symmetric and commuting) -- only the lower triangle of tiles is computed, the rest mirrored; shape 1 =
 * thip_test_gemm_sym (X symmetric, Y general).  kernel 0: the library's choice; 1: one 32 x 32 tile per workgroup;
 * 2: 32 x 64 blocks (what the library picks when a launch has more tiles than the device has CUs) */
int thip_test_gemm_chain(int shape, int kernel, int n, int ld, int nb, float alpha, const float *X, const float *Y,
                         float beta, const float *D, float gamma, float *C);
/* test entry point for the all-symmetric products of the round-5 chain: O_p = alpha_p * A * B_p + beta_p * B_p + gamma_p * I_n
 * for A, B_p symmetric ld x ld (ld a multiple of 64 up to 512, nb items ld * ld apart), computed on the lower triangle of
 * 32 x 32 tiles and mirrored; with dsym_p != 0 the diagonal tiles are averaged with their transpose.  coef = { alpha0, beta0,
 * gamma0, dsym0, alpha1, beta1, gamma1, dsym1 }; B1 == NULL: one product.  kernel 0: the two-product kernel with the
 * library's tiles-per-workgroup; 1..3: that number forced; 4 / 5: the one-tile / 32 x 64 block kernels (one product) */
int thip_test_gemm_dual(int kernel, int n, int ld, int nb, const float *A, const float *B0, const float *B1, const float *coef,
                        float *O0, float *O1);
/* timing probe of the chain's launch shapes (tools/psd_chain_probe.py): `reps` dependent launches of ld x ld products,
 * *host_us = microseconds per launch.  mode 0 / 1: 32 x 64 blocks, batch of two, symmetric / general result; 2: one tile
 * per workgroup, symmetric, batch of two; 3: the same, one item; 4: mode 3 on two streams at once (per launch PAIR);
 * 5: one tile per workgroup, general, one item; 6-8: the two-product kernel (two products / one with averaged diagonal tiles /
 * one with the packed output); 9: mode 0 with averaged diagonal tiles */
int thip_test_chain_probe(int mode, int ld, int reps, float *host_us);
/* timing probe of the tiled sparse products (tools/sptile_rate.py): `reps` launches each of A^T [y0 y1] and A [x0 x1] on `mat`
 * (vectors of ones, two right-hand sides as in the one-pass recurrence); host_ms[0] / [1] = best milliseconds per launch of the
 * T / N product, host_ms[2] / [3] their averages */
int thip_test_sptile_time(thip_sptile *mat, int reps, float *host_ms);

#ifdef __cplusplus
}
#endif
#endif /* TOTSU_F32HIP_TEST_H */
