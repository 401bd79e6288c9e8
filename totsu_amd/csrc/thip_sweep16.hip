// thip_sweep16.hip -- the one-pass kernel (thip_sweep_kernel.h, described in thip_sweep.hip) for a 16-bit-STORED A: bf16, or
// f16 with one power-of-two scale per column (thip_solver_set_a_storage).  Half the bytes per sweep,
// f32 arithmetic.  A 16-byte slot is eight rows, so a streaming thread keeps twice the rows per slot in registers (v, x_y and
// the two accumulators: 32 VGPRs per slot) and the ring gets what is left: at most 4 slots per thread with one column per
// panel, 3 with two, 2 with four -- the planner (sweep_plan_one) picks the group size to fit, and prefers two or four columns
// per panel: a 16-bit column is half the bytes, and what the service wave's chain needs is TIME per panel.
// The problem solved is the one with the rounded matrix; the f32 finish is thip_solver_resume's.
#include "thip_sweep_kernel.h"

namespace thip {

template <int ELEM>
static int sweep_launch16_t(hipStream_t st, const SweepGeom &g, const SweepArgs &a)
{
    // <slots, columns per panel, LAGL, DLAG, LS, ELEM>: 3 register stages; LDS panels by what 150 KB holds
    if (g.w == 1) {
        switch (g.nslot) {
        case 1: return sweep_go<1, 1, 2, 1, 3, ELEM>(st, a);
        case 2: return sweep_go<2, 1, 2, 1, 3, ELEM>(st, a);
        case 3: return sweep_go<3, 1, 2, 1, 3, ELEM>(st, a);
        default: return sweep_go<4, 1, 2, 1, 3, ELEM>(st, a);
        }
    }
    if (g.w == 2) {
        switch (g.nslot) {
        case 1: return sweep_go<1, 2, 2, 1, 3, ELEM>(st, a);
        case 2: return sweep_go<2, 2, 2, 1, 3, ELEM>(st, a);
        default: return sweep_go<3, 2, 2, 1, 3, ELEM>(st, a);      // (4 slots x 2 columns: 8 / 71 VGPRs spilled for bf16 / f16 -- the build guard refuses it)
        }
    }
    if (g.nslot == 1) return sweep_go<1, 4, 2, 1, 3, ELEM>(st, a);
    return sweep_go<2, 4, 2, 1, 2, ELEM>(st, a);
}

int sweep_launch16(hipStream_t st, const SweepGeom &g, const SweepArgs &a)
{
    return g.elem == THIP_A_F16 ? sweep_launch16_t<2>(st, g, a) : sweep_launch16_t<1>(st, g, a);
}

}  // namespace thip
