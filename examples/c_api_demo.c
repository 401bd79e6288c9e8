/* c_api_demo.c -- a plain C host over include/totsu_f32hip.h (no Python, no torch): solves the LP of
 * examples/nostd_cortex-m/src/main.rs:57-99 (reference) twice -- through the LinAlg primitives (a few calls, to
 * show the trait-level entry points) and through the device-resident fused loop -- and prints the answer.
 *   gcc -O2 -Iinclude examples/c_api_demo.c -Ltotsu_amd/lib -ltotsu_f32hip -Wl,-rpath,$PWD/totsu_amd/lib -o examples/c_api_demo
 */
#include <stdio.h>
#include <stdlib.h>
#include "totsu_f32hip.h"

#define CHK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, thip_last_error()); return 2; } } while (0)

int main(void)
{
    int ndev = 0;
    thip_device_count(&ndev);
    if (ndev == 0) { fprintf(stderr, "no GPU: %s has no CPU fallback\n", thip_version()); return 3; }
    CHK(thip_init(0));

    /* column-major A (3 x 2), b, c */
    const float a_h[6] = { 4.f, -1.f, -1.f, -1.f, 4.f, -1.f };
    const float b_h[3] = { 6.f, 6.f, 1.f };
    const float c_h[2] = { -1.f, 0.f };
    float *a, *b, *c, *y;
    CHK(thip_alloc(6, &a)); CHK(thip_alloc(3, &b)); CHK(thip_alloc(2, &c)); CHK(thip_alloc_zeroed(3, &y));
    CHK(thip_h2d(a, a_h, 6)); CHK(thip_h2d(b, b_h, 3)); CHK(thip_h2d(c, c_h, 2));

    /* trait-level primitives: y = A c, ||y|| */
    float nrm = 0.f;
    CHK(thip_transform_ge(0, 3, 2, 1.0f, a, c, 0.0f, y));
    CHK(thip_norm(3, y, &nrm));
    printf("||A c|| = %.6f (expect %.6f)\n", nrm, 4.242641f);

    /* fused loop */
    const int32_t seg_type[1] = { THIP_CONE_RPOS };
    const int64_t seg_len[1] = { 3 };
    thip_problem prob = { 2, 3, a, b, c, NULL, 1, seg_type, seg_len };
    thip_param par = { 100000, 1e-5f, 1e-6f, 1e-12f, 0, THIP_STATE_COMPENSATED, 0 };
    thip_solver *s = NULL;
    thip_status st;
    float x[2], yy[3];
    CHK(thip_solver_create(&prob, &par, THIP_SCHED_CARRIED, &s));
    CHK(thip_solver_init(s));
    CHK(thip_solver_run(s, -1, 32, &st));
    CHK(thip_solver_solution(s, x, yy));
    printf("state %d after %lld iterations: x = [%.5f, %.5f] (expect [2, 2])\n", st.state, (long long)st.iter, x[0], x[1]);
    CHK(thip_solver_destroy(s));

    /* the same LP with A as a sparse operator held once on the device (thip_sptile_*: the matrix by columns, host arrays), through the
     * one-pass recurrence in three launches */
    const int64_t colptr[3] = { 0, 3, 6 };
    const int32_t rowidx[6] = { 0, 1, 2, 0, 1, 2 };
    thip_sptile *sp = NULL;
    thip_status st2;
    float xs[2], ys[3], nrm2 = 0.f;
    CHK(thip_sptile_create(3, 2, 6, colptr, rowidx, a_h, &sp));
    CHK(thip_sptile_mv(sp, 0, 1.0f, c, 0.0f, y, 0));                 /* y = A c through the sparse copy */
    CHK(thip_norm(3, y, &nrm2));
    thip_problem prob2 = { 2, 3, NULL, b, c, NULL, 1, seg_type, seg_len };
    CHK(thip_solver_create(&prob2, &par, THIP_SCHED_SWEEP, &s));
    CHK(thip_solver_set_sptile(s, sp));
    CHK(thip_solver_init(s));
    CHK(thip_solver_run(s, -1, 32, &st2));
    CHK(thip_solver_solution(s, xs, ys));
    printf("sparse: ||A c|| = %.6f, state %d after %lld iterations: x = [%.5f, %.5f] (expect [2, 2])\n", nrm2, st2.state,
           (long long)st2.iter, xs[0], xs[1]);
    CHK(thip_solver_destroy(s));
    CHK(thip_sptile_destroy(sp));
    CHK(thip_free(a)); CHK(thip_free(b)); CHK(thip_free(c)); CHK(thip_free(y));
    CHK(thip_shutdown());
    return (st.state == THIP_ST_OK && x[0] > 1.999f && x[0] < 2.001f && x[1] > 1.999f && x[1] < 2.001f
            && st2.state == THIP_ST_OK && xs[0] > 1.999f && xs[0] < 2.001f && xs[1] > 1.999f && xs[1] < 2.001f
            && nrm2 > 4.2426f && nrm2 < 4.2427f) ? 0 : 1;
}
