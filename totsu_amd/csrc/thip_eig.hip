// thip_eig.hip -- LinAlgEx::map_eig (totsu_core/src/linalg_ex.rs:44-65) on the device.  PLACEHOLDER: filled in
// after the LP / SOCP path is measured.
#include "thip_common.h"

using namespace thip;

namespace thip {
int eig_psd_project(hipStream_t, size_t, float *, int, float, float, float *, size_t, int, const int *)
{
    return fail(THIP_E_INVALID, "map_eig: not implemented yet", __FILE__, __LINE__);
}
}  // namespace thip

extern "C" {
size_t thip_map_eig_worklen(size_t n) { return 3 * n * n + 4 * n + 64; }
int thip_map_eig(size_t n, float *mat, int has_scale, float scale_diag, float eps_zero, float *work, size_t worklen, int map_kind)
{
    THIP_NEED_INIT();
    if (worklen < thip_map_eig_worklen(n)) return fail(THIP_E_WORK, "map_eig work too short", __FILE__, __LINE__);
    return eig_psd_project(ctx().stream, n, mat, has_scale, scale_diag, eps_zero, work, worklen, map_kind, nullptr);
}
int thip_eig_decompose(size_t, float *, int, float, float, float *, size_t, float *) { return fail(THIP_E_INVALID, "not implemented", __FILE__, __LINE__); }
int thip_eig_rebuild(size_t, float *, int, float, float *, size_t, const float *, const uint8_t *) { return fail(THIP_E_INVALID, "not implemented", __FILE__, __LINE__); }
int thip_proj_psd(size_t sn, float *x, float eps_zero, float *work, size_t worklen)
{
    THIP_NEED_INIT();
    const size_t n = (size_t)((__builtin_sqrt((double)(8 * sn + 1)) - 1.0) / 2.0 + 0.5);
    if (n * (n + 1) / 2 != sn) return fail(THIP_E_INVALID, "not a triangular number", __FILE__, __LINE__);
    if (worklen < thip_map_eig_worklen(n)) return fail(THIP_E_WORK, "ConePSD work shortage", __FILE__, __LINE__);
    return eig_psd_project(ctx().stream, n, x, 1, 1.41421356237f, eps_zero, work, worklen, 0, nullptr);
}
}
