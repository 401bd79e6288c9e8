// thip_sweep.hip -- ONE pass over A per iteration of the conic loop (THIP_SCHED_SWEEP; dense f32 A on one GPU).
//
// The carried schedule reads A twice per iteration because the second pair of products takes the vectors the first pair
// produced (solver.rs:538-560: x_{k+1} needs K^T y_k, y_{k+1} needs K (x_k - 2 x_{k+1})).  But WHICH entries a product
// needs, and when they exist, is finer than that:
//   * x_x has no cone (solver.rs:546-549 project x_y and x_s only), so x_x_{k+1}[j] = x_x_k[j] + T_x[j] ((A^T v_k)[j] + c[j] kappa_k)
//     is known as soon as COLUMN j has been multiplied with v_k -- and column j is then still in registers for the
//     N product A x_x_{k+1} that the y update and the criteria want (solver.rs:125, 594);
//   * u has no projection either (solver.rs:562-567 clamp kappa only): u_k[j] = u_{k-1}[j] + S_u[j] (-(A^T rx_y_{k-1})[j] - c[j] rtau_{k-1})
//     needs column j times the m-vector x_y_k (carried form: A^T rx_y = A^T x_y_{k-1} - 2 A^T x_y_k), and then feeds
//     the N product A u_k of the next x update (solver.rs:149).
// So one sweep over the columns does, per column j: two dots (with v_k and with x_y_k), two scalar updates (u_k[j],
// x_x_{k+1}[j]), two axpys (A u_k, A x_x_{k+1}).  Everything else of the iteration is O(n + m) work between sweeps
// (thip_solver.hip: sw_* kernels).  The recurrences are the reference's, evaluated in a skewed order; no value is
// obtained differently from the carried schedule except for the order of the floating-point sums.
//
// The kernel.  A column (m floats, 400 KB at BASELINE configs[2]) does not fit one CU, and the dot must be complete
// before the axpy can start.  So G workgroups on one XCD form a GROUP (G = 8 or 16 at configs[2]; 1 .. 32, chosen by
// sweep_plan_one / timed by thip_solver.hip): a member owns a fixed range of rows -- its slices of v and x_y and its N
// accumulators live in registers for the whole sweep -- the group walks its share of the columns in PANELS of W = 1 or
// 2 columns, and the only thing that crosses CUs is an all-gather of the members' 2 W partial dots per panel through the
// L2 the group shares (8-byte {value, tag} granules: plain stores stay in that L2, sc1 loads are served by it; the
// hand-off thip_eig.hip's one-XCD Householder reduction uses).  A gather takes microseconds under load, a panel
// 1 - 2 us of HBM stream, so a panel stays ON THE CHIP from its loads to its axpy: a ring of LAGL + 1 register stages
// (statically indexed: the loop is unrolled by the ring length), then LS more panels parked in LDS by the thread that
// loaded them.  7 waves of a workgroup stream; the 8th is a service wave (publishes the workgroup's partial dots, gathers
// the group's granules, does the scalar updates, prefetches the per-column data): a polling wave has to drain its own
// loads (vmcnt is in order), so it must not be one that prefetches A.  What a geometry fixes is the BYTES a workgroup
// stages per panel (W x rows x 4): fewer members per group (more rows each, one column per panel) buy longer intervals
// for the service wave's dependent chain and fewer workgroups that can hold a group up -- 32 members with 2-column
// panels ran at 0.65 of the HBM peak, 8 members with 1-column panels at 0.85 - 0.89 (DESIGN.md 4.7 has the steps).
// Every spin is bounded; a workgroup that gives up raises the error word and the host reports a failed run.  A census at
// kernel entry (XCC_ID + tickets) checks that exactly 32 workgroups sit on every XCD -- i.e. one per CU, all resident --
// before anything is written; thip_solver.hip runs it once as a dry run and falls back to the carried schedule if the
// placement is not the expected one.
#include "thip_common.h"

#include <cstdio>
#include <cstdlib>

using namespace thip;

namespace thip {

constexpr int SW_THREADS = 512;            // 7 streaming waves + 1 service wave
constexpr int SW_CW = 7;
constexpr int SW_CT = SW_CW * 64;          // streaming threads
constexpr int SW_RING = 32;                // granule slots per group (> 2 (LAGL - DLAG) - 1: see the header of sweep_k)
constexpr int SW_CR = 16;                  // slots of the per-column LDS ring (> LAGL - DLAG)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long sw_pack(float v, unsigned tag)
{
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope fence, which on this part
// waits for EVERY outstanding global load of the wave (vmcnt(0)) -- it would drain the ring of prefetched panels at
// every interval
__device__ __forceinline__ void sw_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void sw_barrier_dbg(int dbg)
{
    if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else sw_barrier();
}

// x + inc with an optional Kahan term (the arithmetic of thip_solver.hip's comp_add)
__device__ __forceinline__ float sw_comp_add(float x, float inc, bool comp, float &k)
{
    if (!comp) return x + inc;
    const float y = inc - k;
    const float t = x + y;
    k = (t - x) - y;
    return t;
}

template <int K, int O>
__device__ __forceinline__ void sw_halve(float *v, int lane)
{
    const bool hi = (lane & O) != 0;
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {
        const float send = hi ? v[i] : v[i + K / 2];
        const float keep = hi ? v[i + K / 2] : v[i];
        v[i] = keep + __shfl_xor(send, O, 64);
    }
}

// K per-lane values on 64 lanes -> the K wave sums; the lanes with (lane >> (6 - log2 K)) == c hold sum c
template <int K>
__device__ __forceinline__ float sw_reduce(float *v, int lane)
{
    static_assert(K == 2 || K == 4 || K == 8, "1, 2 or 4 columns per panel");
    if constexpr (K == 2) {
        sw_halve<2, 32>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 16, 64); r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    } else if constexpr (K == 8) {
        sw_halve<8, 32>(v, lane); sw_halve<4, 16>(v, lane); sw_halve<2, 8>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    } else {
        sw_halve<4, 32>(v, lane); sw_halve<2, 16>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    }
}

// census.  Every workgroup counts itself on its XCD: the ticket it draws is its place (group, member) among the 32 of that
// XCD.  sweep_census_k (the dry run of thip_solver_init and of thip_sweep_probe) ALSO waits until all 256 have counted
// themselves and checks that every XCD holds exactly 32 -- the placement the kernel needs.  The sweeps themselves do not wait
// (round 3 did: device-clock stamps put the wait at 15 us per launch, a tenth of a short sweep): a workgroup that draws a
// ticket >= 32 raises the error word at once, and a group that is short of a member runs out of its bounded spins and
// raises it too -- thip_solver_run then restores its snapshot of the iterate and goes on with the 2-pass schedule.
__device__ __forceinline__ int sw_ticket(unsigned *census, unsigned seq, int G, int *group, int *member)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;        // HW_REG_XCC_ID, bits 3:0
    if (xcc >= 8u) { atomicExch(census + 9, 1u); return 0; }
    const unsigned idx = atomicAdd(census + xcc, 1u) - seq * 32u;
    if (idx >= 32u) { atomicExch(census + 9, 2u); return 0; }
    *group = (int)xcc * (32 / G) + (int)idx / G;
    *member = (int)idx % G;
    return 1;
}

__device__ __forceinline__ int sw_census(unsigned *census, unsigned seq, int G, int *group, int *member)
{
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;
    if (xcc >= 8u) { atomicExch(census + 9, 1u); return 0; }
    const unsigned idx = atomicAdd(census + xcc, 1u) - seq * 32u;
    const unsigned want = (seq + 1u) * 32u;
    int spins = 0;
    for (;;) {
        // the eight counts in one round trip; done when they add up to everybody
        unsigned c[8], sum = 0u;
#pragma unroll
        for (int x = 0; x < 8; ++x) c[x] = __hip_atomic_load(census + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int x = 0; x < 8; ++x) { sum += c[x]; ok = ok && c[x] == want; }
        if (sum == 8u * want) {
            if (!ok || idx >= 32u) { atomicExch(census + 9, 2u); return 0; }
            break;
        }
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SW_SPIN_MAX) { atomicExch(census + 9, 1u); return 0; }
    }
    *group = (int)xcc * (32 / G) + (int)idx / G;
    *member = (int)idx % G;
    return 1;
}

__global__ __launch_bounds__(SW_THREADS) void sweep_census_k(unsigned *census, unsigned seq)
{
    if (threadIdx.x == 0) { int g, mbr; (void)sw_census(census, seq, 32, &g, &mbr); }
}

// W columns per panel; loads run LAGL panels ahead of the axpy and DLAG ahead of the dots; NS = LAGL + 1 register stages.
// Timeline of a workgroup, interval `it` (one barrier per interval), written for (W, LAGL, DLAG) = (2, 8, 3):
//   streaming waves: issue the loads of panel it (stage it % 9) ; dots of panel it - 3 -> dotbuf ; BARRIER ;
//                    axpy of panel it - 8 with the scalars the service wave left in `scal`
//   service wave:    [per-column data: store what the last interval fetched, fetch panel it - 3] ;
//                    publish the workgroup's dots of panel it - 4 ; gather panel it - 8 (its granules were requested in
//                    the previous interval; polling only if they are late), scalar updates -> scal ; BARRIER
// A panel is published LAGL - DLAG - 1 intervals before it is gathered.  A member publishing panel q has gathered panel
// q + DLAG - LAGL, so every member has published that one and is gathering q + 2 DLAG - 2 LAGL + 1 or later: the slot of
// panel q - 16 is free.  Per-column data of panel q is fetched in interval q + DLAG and has arrived before the member
// publishes q in interval q + DLAG + 1; the writer of panel q's columns stores u / gP (in place) only after it has gathered
// q, i.e. after every member holds its copy.
// THIP_SWEEP_DBG (experiments of DESIGN.md 4.7: 1 no polling, 2 no wave reduction of the dots, 4 no barrier, 8 service wave
// idle, 16 no arithmetic) exists only in a -DSW_DEBUG build; otherwise the switches fold away
#ifdef SW_DEBUG
#define SW_DBG(a) ((a).dbg)
#else
#define SW_DBG(a) 0
#endif
template <int NSLOT, int W, int LAGL, int DLAG, int LS>
__global__ __launch_bounds__(SW_THREADS) void sweep_k(const SweepArgs a)
{
    constexpr int NS = LAGL + 1;
    constexpr int LAGT = LAGL + LS;                   // loads run this far ahead of the axpy: LAGL in registers, LS more in LDS
    constexpr int PF = LAGL - DLAG;
    constexpr int NL = (2 * W * 32 + 63) / 64;        // granule loads per service lane (G = 32)
    static_assert(SW_RING > 2 * (LAGT - DLAG) - 1 && SW_CR > LAGT - DLAG, "ring depths");
    __shared__ int s_role[4];
    __shared__ float dotbuf[2][SW_CW][2 * W];
    __shared__ float scal[2][2 * W];
    __shared__ float cold[SW_CR][8][W];
    extern __shared__ f32x4 sw_lds[];                 // LS panels the streaming threads park between registers and axpy
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SW_PROFILE
    // phase stamps of streaming wave 0 of (group 0, member 0): census[150 ..] = 10 ns ticks since kernel entry at: census done,
    // v / x_y in registers, fill block, steady loop, drain, stores
    const unsigned long long tp0 = __builtin_amdgcn_s_memrealtime();
#define SW_PHASE(i) do { if (wave == 0 && group == 0 && member == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        if (lane == 0) a.census[150 + (i)] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tp0); } } while (0)
#else
#define SW_PHASE(i) do { } while (0)
#endif

    if (tid == 0) {
        int g = 0, mbr = 0;
        s_role[2] = sw_ticket(a.census, a.seq, a.G, &g, &mbr);
        s_role[0] = g; s_role[1] = mbr;
    }
    __syncthreads();
    // the ticket comes first even when the loop has stopped: the host numbers the launches, and one that left before
    // counting itself would leave every later launch's tickets off by one
    // (a raised error word -- this launch's census, or an earlier sweep of the batch that gave up -- ends every later sweep
    // at entry: the host restores its snapshot of the iterate, thip_solver.hip sweep_recover)
    if (s_role[2] == 0 || *a.stop != 0 || __hip_atomic_load(a.census + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const int group = s_role[0], member = s_role[1];
    SW_PHASE(0);

    const int c0 = group * a.cols_per_group;
    const int c1 = min(a.n, c0 + a.cols_per_group);
    const int npan = c1 > c0 ? (c1 - c0 + W - 1) / W : 0;
    const int total = npan + LAGT;
    const int row0 = member * a.rows_per_member;
    const int row1 = min(a.m, row0 + a.rows_per_member);
    unsigned *const errflag = a.census + 9;

    if (wave < SW_CW) {
        // ---------------- streaming waves ----------------
        float vv[NSLOT][4], yv[NSLOT][4], acc1[NSLOT][4], acc2[NSLOT][4];
        int roff[NSLOT];
        bool valid[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int r = row0 + 4 * (tid + SW_CT * sl);
            valid[sl] = r + 4 <= row1;
            roff[sl] = valid[sl] ? r : (row0 + 4 <= a.m ? row0 : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vv[sl][k] = valid[sl] ? a.v[r + k] : 0.0f;
                yv[sl][k] = valid[sl] ? a.xy[r + k] : 0.0f;
                acc1[sl][k] = 0.0f; acc2[sl][k] = 0.0f;
            }
        }
        SW_PHASE(1);
        f32x4 stg[NS][W][NSLOT];
        const int jmax = a.n - 1;

#define SW_LOADS(S, P)                                                                                         \
        do {                                                                                                   \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                const int j = min(c0 + (P) * W + q, jmax);                                                     \
                const float *colp = a.A + (size_t)j * a.lda;                                                   \
                _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl)                                           \
                    stg[S][q][sl] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(colp + roff[sl])); \
            }                                                                                                  \
        } while (0)
#define SW_DOTS(S, P)                                                                                          \
        do {                                                                                                   \
            if (SW_DBG(a) & 16) { asm volatile("" :: "v"(stg[S][0][0][0]), "v"(stg[S][W - 1][NSLOT - 1][3])); break; } \
            float p_[2 * W];                                                                                   \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                float d1 = 0.0f, d2 = 0.0f;                                                                    \
                _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl)                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
                        d1 = fmaf(stg[S][q][sl][k], vv[sl][k], d1);                                            \
                        d2 = fmaf(stg[S][q][sl][k], yv[sl][k], d2);                                            \
                    }                                                                                          \
                p_[q] = d1; p_[W + q] = d2;                                                                    \
            }                                                                                                  \
            if (SW_DBG(a) & 2) { if (lane < 2 * W) dotbuf[(P) & 1][wave][lane] = p_[0]; }                          \
            else {                                                                                             \
            const float r_ = sw_reduce<2 * W>(p_, lane);                                                       \
            if ((lane & (64 / (2 * W) - 1)) == 0) dotbuf[(P) & 1][wave][lane / (64 / (2 * W))] = r_;          \
            }                                                                                                  \
        } while (0)
#define SW_AXPY(S, P)                                                                                          \
        do {                                                                                                   \
            if (SW_DBG(a) & 16) { asm volatile("" :: "v"(stg[S][0][0][1]), "v"(stg[S][W - 1][NSLOT - 1][2])); break; } \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                const float s1 = scal[(P) & 1][q], s2 = scal[(P) & 1][W + q];                                  \
                _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl)                                           \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
                        acc1[sl][k] = fmaf(stg[S][q][sl][k], s1, acc1[sl][k]);                                 \
                        acc2[sl][k] = fmaf(stg[S][q][sl][k], s2, acc2[sl][k]);                                 \
                    }                                                                                          \
            }                                                                                                  \
            /* pin the sums here: the optimiser otherwise sinks the axpys of all NS intervals to the end of the */ \
            /* unrolled block and keeps every stage and every interval's scalars alive until then */           \
            _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl)                                               \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                \
                    asm volatile("" : "+v"(acc1[sl][k]));                                                      \
                    asm volatile("" : "+v"(acc2[sl][k]));                                                      \
                }                                                                                              \
        } while (0)

#define SW_SPILL(S, P)                                                                                         \
        do {                                                                                                   \
            f32x4 *slot_ = sw_lds + (size_t)((P) % LS) * (W * NSLOT * SW_CT) + tid;                            \
            _Pragma("unroll") for (int q = 0; q < W; ++q)                                                      \
                _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl) slot_[(q * NSLOT + sl) * SW_CT] = stg[S][q][sl]; \
        } while (0)
#define SW_AXPY_LDS(P)                                                                                         \
        do {                                                                                                   \
            const f32x4 *slot_ = sw_lds + (size_t)((P) % LS) * (W * NSLOT * SW_CT) + tid;                      \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                const float s1 = scal[(P) & 1][q], s2 = scal[(P) & 1][W + q];                                  \
                _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl) {                                         \
                    const f32x4 d_ = slot_[(q * NSLOT + sl) * SW_CT];                                          \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
                        acc1[sl][k] = fmaf(d_[k], s1, acc1[sl][k]);                                            \
                        acc2[sl][k] = fmaf(d_[k], s2, acc2[sl][k]);                                            \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
            _Pragma("unroll") for (int sl = 0; sl < NSLOT; ++sl)                                               \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                \
                    asm volatile("" : "+v"(acc1[sl][k]));                                                      \
                    asm volatile("" : "+v"(acc2[sl][k]));                                                      \
                }                                                                                              \
        } while (0)
#define SW_GUARDED_BLOCK(IT0)                                                                                  \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                       \
            const int it = (IT0) + s;                                                                          \
            if (it < npan) SW_LOADS(s, it);                                                                    \
            if (it - DLAG >= 0 && it - DLAG < npan) SW_DOTS((s + NS - DLAG) % NS, it - DLAG);                  \
            if (it < total) sw_barrier_dbg(SW_DBG(a));                                                                      \
            if constexpr (LS == 0) {                                                                           \
                if (it - LAGL >= 0 && it - LAGL < npan) SW_AXPY((s + 1) % NS, it - LAGL);                      \
            } else {                                                                                           \
                if (it - LAGT >= 0 && it - LAGT < npan) SW_AXPY_LDS(it - LAGT);                                \
                if (it - LAGL >= 0 && it - LAGL < npan) SW_SPILL((s + 1) % NS, it - LAGL);                     \
            }                                                                                                  \
        }
        // fill (one block of NS intervals: LAGL < NS), steady state, drain -- three loops, so that the stage registers
        // have one assignment per loop (two forms of the body inside ONE loop doubled them)
        int it0 = 0;
        for (; it0 < LAGT; it0 += NS) { SW_GUARDED_BLOCK(it0); }
        SW_PHASE(2);
        for (; it0 + NS <= npan; it0 += NS) {
            // every phase active, no guards: the waits on the loads are counted, not drained
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int it = it0 + s;
                SW_LOADS(s, it);
                SW_DOTS((s + NS - DLAG) % NS, it - DLAG);
                sw_barrier_dbg(SW_DBG(a));
                if constexpr (LS == 0) { SW_AXPY((s + 1) % NS, it - LAGL); }
                else { SW_AXPY_LDS(it - LAGT); SW_SPILL((s + 1) % NS, it - LAGL); }
            }
        }
        SW_PHASE(3);
        for (; it0 < total; it0 += NS) { SW_GUARDED_BLOCK(it0); }
        SW_PHASE(4);
#undef SW_GUARDED_BLOCK
#undef SW_LOADS
#undef SW_DOTS
#undef SW_AXPY
#undef SW_AXPY_LDS
#undef SW_SPILL
        float *h1 = a.partH + ((size_t)group * 2 + 0) * a.mpad;
        float *h2 = a.partH + ((size_t)group * 2 + 1) * a.mpad;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl)
            if (valid[sl]) {
                const int r = row0 + 4 * (tid + SW_CT * sl);
                *reinterpret_cast<float4 *>(h1 + r) = make_float4(acc1[sl][0], acc1[sl][1], acc1[sl][2], acc1[sl][3]);
                *reinterpret_cast<float4 *>(h2 + r) = make_float4(acc2[sl][0], acc2[sl][1], acc2[sl][2], acc2[sl][3]);
            }
        SW_PHASE(5);
    } else {
        // ---------------- service wave ----------------
        // Its global loads (the granules of a panel, the per-column data of a panel) are issued TWO intervals before they
        // are used, as the last memory operations of an interval and always the same number of instructions (NL + 1,
        // addresses clamped, no branches around them): the interval then opens with s_waitcnt vmcnt(NL + 1) -- "everything
        // but the loads of the previous interval has arrived" -- and never waits for a round trip.  (Issued one interval
        // ahead and waited with vmcnt(0), every interval paid an L2 round trip: 1.59 us per panel instead of ~1.)
        float kappa = *a.kappa_p;
        const float rtau = *a.rtau_p;
        if (a.kappa_out != nullptr && !a.first) {
            // c.rx_x of the previous sweep (its workgroups' partials) + b.rx_y of the m-tail, in f64 like the other single-block sums
            double dc = 0.0, db = 0.0;
            for (int k = lane; k < a.pn_count; k += 64) dc += (double)a.pn_in[3 * a.pn_in_stride + k];
            for (int k = lane; k < a.np_m; k += 64) db += (double)a.pm_brx[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { dc += __shfl_xor(dc, o, 64); db += __shfl_xor(db, o, 64); }
            kappa = fminf(kappa + *a.skappa_p * ((float)dc + (float)db), 0.0f);
            if (group == 0 && member == 0 && lane == 0) *a.kappa_out = kappa;
        }
        const float tau = *a.tau_p;
        const bool conv = tau > a.eps_zero;
        const float rt = conv ? 1.0f / tau : 1.0f;
        float sdd = 0.0f, scx = 0.0f, scu = 0.0f, scrx = 0.0f;      // lanes < W: sums over the columns this workgroup writes
        const bool comp_u = a.ku != nullptr, comp_x = a.kx_in != nullptr;
        // per-column data: lane l < 8 W fetches field l / W of column l % W
        const int cf = lane / W, cq = lane % W;
        const bool clane = lane < 8 * W;
        const float *fp = a.c;                       // lanes without a field read c (a valid address) and drop the value
        bool fvalid = clane;
        if (clane) {
            switch (cf) {
            case 0: fp = a.c; break;      case 1: fp = a.Su; break;    case 2: fp = a.Tx; break;   case 3: fp = a.u; break;
            case 4: fp = a.ku; break;     case 5: fp = a.xx_in; break; case 6: fp = a.kx_in; break; default: fp = a.gP; break;
            }
            if (fp == nullptr) { fp = a.c; fvalid = false; }
        }
        unsigned long long *const gbase = a.gran + (size_t)group * SW_RING * a.G * (2 * W);
        const int nq = a.G * 2 * W;                   // granules per slot (<= 64 NL)
        const int jlast = max(c1 - 1, 0);
        bool dead = false;                            // a gather timed out (here or elsewhere): no more polling
        unsigned polls_total = 0u, polls_max = 0u;    // gathers that had to poll: how often, and the longest (census[18], [19])
        // register sets A / B alternate by interval parity
        unsigned long long xgA[NL], xgB[NL];
        float cvA = 0.0f, cvB = 0.0f;
#pragma unroll
        for (int i = 0; i < NL; ++i) { xgA[i] = 0ull; xgB[i] = 0ull; }

#ifdef SW_PROFILE
        unsigned long long tacc[6] = { 0, 0, 0, 0, 0, 0 }, tlast = __builtin_amdgcn_s_memrealtime();
        unsigned nmiss = 0, npoll = 0;
#define SW_STAMP(i) do { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); tacc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define SW_STAMP(i) do { } while (0)
#endif
        auto interval = [&](const int it, unsigned long long (&xg)[NL], float &cv) {
            if (SW_DBG(a) & 8) { sw_barrier_dbg(SW_DBG(a)); return; }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL + 1) : "memory");
            SW_STAMP(0);
            // per-column data fetched two intervals ago: panel it - 2 - (LAGL - PF - 1)
            {
                const int pc = it - 2 - LAGL + PF + 1;
                if (pc >= 0 && pc < npan && clane) cold[pc % SW_CR][cf][cq] = (fvalid && c0 + pc * W + cq < c1) ? cv : 0.0f;
            }
            // publish the workgroup's partial dots
            {
                const int pp = it - DLAG - 1;
                if (pp >= 0 && pp < npan && lane < 2 * W) {
                    float sum = dotbuf[pp & 1][0][lane];
#pragma unroll
                    for (int w = 1; w < SW_CW; ++w) sum += dotbuf[pp & 1][w][lane];
                    unsigned long long *g = gbase + ((size_t)(pp % SW_RING) * a.G + member) * (2 * W) + lane;
                    // TEST HOOK (thip_test_sweep_fault): one workgroup stops publishing half-way -- its group runs out of spins
                    const bool withheld = a.fault != 0 && group == 0 && member == a.G - 1 && pp >= npan / 2;
                    // pub_agent: the documented form (sc1 store, MI355X_MICROARCH.md inter-workgroup visibility); else a plain
                    // store that stays in the L2 the group shares (DESIGN.md 4.7 has the measured difference)
                    if (withheld) { }
                    else if (a.pub_agent) __hip_atomic_store(g, sw_pack(sum, a.tagbase + (unsigned)pp + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else __hip_atomic_store(g, sw_pack(sum, a.tagbase + (unsigned)pp + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            // gather, scalar updates
            SW_STAMP(1);
            const int pa = it - LAGT;
            if (pa >= 0 && pa < npan) {
                const unsigned tag = a.tagbase + (unsigned)pa + 1u;
                const unsigned long long *g = gbase + (size_t)(pa % SW_RING) * nq;
                float vsum = 0.0f;
                unsigned pend = 0;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    if (lane + 64 * i < nq) {
                        if ((unsigned)(xg[i] >> 32) == tag) vsum += __uint_as_float((unsigned)xg[i]);
                        else pend |= 1u << i;
                    }
                }
                int spins = 0;
                if (SW_DBG(a) & 1) pend = 0;
#ifdef SW_PROFILE
                if (!__all(pend == 0u)) {
                    if (group == 0 && member == 0 && nmiss < 40) {
                        const unsigned long long bal = __ballot(pend != 0u);
                        if (lane == 0) { a.census[24 + 3 * nmiss] = (unsigned)pa; a.census[25 + 3 * nmiss] = (unsigned)bal; a.census[26 + 3 * nmiss] = (unsigned)(bal >> 32); }
                    }
                    ++nmiss;
                }
#endif
                while (!dead && !__all(pend == 0u)) {
#pragma unroll
                    for (int i = 0; i < NL; ++i)
                        if ((pend >> i) & 1u) xg[i] = __hip_atomic_load(g + lane + 64 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int i = 0; i < NL; ++i)
                        if (((pend >> i) & 1u) && (unsigned)(xg[i] >> 32) == tag) {
                            vsum += __uint_as_float((unsigned)xg[i]);
                            pend &= ~(1u << i);
                        }
                    ++spins;
#ifdef SW_PROFILE
                    ++npoll;
#endif
                    if (spins > a.spin_max || ((spins & 255) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (lane == 0) atomicExch(errflag, 3u);
                        dead = true;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (spins > 0) { polls_total += (unsigned)spins; polls_max = max(polls_max, (unsigned)spins); }
                SW_STAMP(2);
                // lane l holds members l / 2W + 32 i / W .. of quantity l % 2W: sum over the lanes of equal l % 2W
                float sum = vsum;
#pragma unroll
                for (int o = 2 * W; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
                const float gT = sum;                                   // lanes 0 .. W-1: column's dot with v
                const float g3 = __shfl(sum, (lane + W) & 63, 64);      // ... and with x_y
                if (lane < W) {
                    const int j = c0 + pa * W + lane;
                    const bool real = j < c1 && !dead;
                    const int cs = pa % SW_CR;
                    const float cj = cold[cs][0][lane], Suj = cold[cs][1][lane], Txj = cold[cs][2][lane];
                    const float uj = cold[cs][3][lane], xxj = cold[cs][5][lane], gPj = cold[cs][7][lane];
                    float kuj = cold[cs][4][lane], kxj = cold[cs][6][lane];
                    float u_new = uj;
                    if (!a.first) {
                        const float g2 = gPj - 2.0f * g3;
                        u_new = sw_comp_add(uj, Suj * (-g2 - cj * rtau), comp_u, kuj);
                    }
                    const float x_new = sw_comp_add(xxj, Txj * (gT + cj * kappa), comp_x, kxj);
                    scal[pa & 1][lane] = real ? u_new : 0.0f;
                    scal[pa & 1][W + lane] = real ? x_new : 0.0f;
                    if (real && member == pa % a.G) {
                        if (!a.first) { a.u[j] = u_new; if (comp_u) a.ku[j] = kuj; }
                        a.xx_out[j] = x_new;
                        if (comp_x) a.kx_out[j] = kxj;
                        a.gP[j] = g3;
                        const float dj = conv ? fmaf(rt, g3, cj) : g3;      // solver.rs:596-597 / 634
                        sdd = fmaf(dj, dj, sdd);
                        scx = fmaf(cj, xxj, scx);
                        scu = fmaf(cj, u_new, scu);
                        scrx = fmaf(cj, xxj - 2.0f * x_new, scrx);
                    }
                }
            }
            // the loads of two intervals ahead, last and unconditional: the granules of panel pa + 2 (published
            // LAGL - DLAG - 3 intervals ago) and the per-column data of panel it - LAGL + PF + 1 (read before this
            // workgroup publishes that panel, two intervals from now)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SW_STAMP(3);
            {
                int pn = pa + 2;
                pn = pn < 0 ? 0 : pn;
                const unsigned long long *gn = gbase + (size_t)(pn % SW_RING) * nq;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int gi = min(lane + 64 * i, nq - 1);
                    xg[i] = __hip_atomic_load(gn + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int pc = it - LAGL + PF + 1;
                pc = pc < 0 ? 0 : pc;
                const int j = min(c0 + pc * W + cq, jlast);
                cv = __hip_atomic_load(fp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            SW_STAMP(4);
            sw_barrier_dbg(SW_DBG(a));
            SW_STAMP(5);
        };
        int it = 0;
        for (; it + 1 < total; it += 2) { interval(it, xgA, cvA); interval(it + 1, xgB, cvB); }
        if (it < total) interval(it, xgA, cvA);
        // what the hand-off through the group's L2 costs in polls (a visibility stall would show here long before a time-out)
        if (polls_total != 0u && lane == 0) { atomicAdd(a.census + 18, polls_total); atomicMax(a.census + 19, polls_max); }
        if (a.pn != nullptr) {
            if constexpr (W == 2) {
                sdd += __shfl_xor(sdd, 1, 64); scx += __shfl_xor(scx, 1, 64);
                scu += __shfl_xor(scu, 1, 64); scrx += __shfl_xor(scrx, 1, 64);
            }
            if (lane == 0) {
                float *o = a.pn + blockIdx.x;
                o[0] = sdd; o[a.pn_stride] = scx; o[2 * a.pn_stride] = scu; o[3 * a.pn_stride] = scrx;
            }
        }
#ifdef SW_PROFILE
        if (group == 0 && member == 0 && lane == 0)
        {
            for (int i = 0; i < 6; ++i) a.census[10 + i] = (unsigned)tacc[i];
            a.census[16] = nmiss; a.census[17] = npoll;
        }
#endif
#undef SW_STAMP
    }
}

}  // namespace thip

namespace thip {

namespace {
// THIP_SWEEP_CLASS (experiments): 0 = two columns per panel, 1 = one column per panel with twice the default group size,
// 2 / unset = the default geometry of sweep_plan_one(W = 1)
int sweep_class()
{
    static const int v = getenv("THIP_SWEEP_CLASS") ? atoi(getenv("THIP_SWEEP_CLASS")) : 2;
    return v >= 0 && v < 3 ? v : 0;
}
int sweep_variant()
{
    static const int v = getenv("THIP_SWEEP_VARIANT") ? atoi(getenv("THIP_SWEEP_VARIANT")) : 0;
    return v >= 0 && v < 4 ? v : 0;
}
}  // namespace

// One candidate geometry: W columns per panel (1 or 2: the two families of kernel instances), the smallest group size
// the family's row capacity allows times gmul.  0 when the kernel can take the matrix that way.
static int sweep_plan_one(size_t m, size_t n, size_t lda, const void *mat, int W, int gmul, SweepGeom *g, int force_G = 0)
{
    if (m == 0 || n == 0 || m % 4 != 0 || lda % 4 != 0 || ((uintptr_t)mat & 15u) != 0) return 1;
    if (n > ((size_t)1 << 30) || n < (size_t)40 * W) return 1;
    if (ctx().num_cu != 256) return 1;
    const size_t slot_rows = (size_t)SW_CT * 4;
    const size_t cap = slot_rows * (W == 1 ? 7 : 2);
    int G = 1;
    while (G < 32 && ((m + G - 1) / G + 3) / 4 * 4 > cap) G *= 2;
    if (((m + G - 1) / G + 3) / 4 * 4 > cap) return 1;
    for (int k = 1; k < gmul; k *= 2) { if (G >= 32) return 1; G *= 2; }
    // few columns: fewer, larger groups, so that every group has its 40 panels and no workgroup idles
    while (G < 32 && (size_t)(256 / G) * 40 * W > n) { if (gmul > 1) return 1; G *= 2; }
    if (force_G) {          // thip_test_sweep: a given group size (a power of two the rows fit)
        if (force_G < G || force_G > 32 || (force_G & (force_G - 1)) != 0) return 1;
        G = force_G;
    }
    const size_t rpm = ((m + G - 1) / G + 3) / 4 * 4;
    const int ngroups = 256 / G;
    size_t cpg = (n + ngroups - 1) / ngroups;
    if (cpg < (size_t)40 * W) cpg = (size_t)40 * W;      // with few columns some groups idle
    cpg = (cpg + W - 1) / W * W;
    const int need = (int)((rpm + slot_rows - 1) / slot_rows);
    g->G = G; g->ngroups = ngroups; g->rows_per_member = (int)rpm; g->cols_per_group = (int)cpg;
    // 16-byte slots per streaming thread: what the member's rows need (round 3 offered 4 or 7 only: the 10 000 rows per member
    // of the n = 10 000 LP ran 7 slots at 80 % of their lanes, the 7 829 of the k = 500 SDP at 62 %)
    g->nslot = W == 2 ? (need <= 1 ? 1 : 2) : need;
    g->mpad = (m + 63) / 64 * 64;
    g->w = W; g->variant = sweep_variant();
    g->npan = (int)(cpg / W);
    g->m_eff = (int)m;
    return 0;
}

// geometry for an m x n matrix; 0 when the sweep kernel can take it.  The default: one column per panel, the fewest
// workgroups per column the rows allow (8 at BASELINE configs[2]); THIP_SWEEP_CLASS = 0 / 1 force the others
int sweep_plan(size_t m, size_t n, size_t lda, const void *mat, SweepGeom *g)
{
    const int cls = sweep_class();
    if (cls == 0 && sweep_plan_one(m, n, lda, mat, 2, 1, g) == 0) return 0;
    if (cls == 1 && sweep_plan_one(m, n, lda, mat, 1, 2, g) == 0) return 0;
    return sweep_plan_one(m, n, lda, mat, 1, 1, g);
}

// the geometries worth timing on a given matrix (thip_solver.hip times them on the actual matrix, like the GEMV plans)
int sweep_candidates(size_t m, size_t n, size_t lda, const void *mat, SweepGeom *out, int max_out)
{
    static const int cand[6][2] = { { 1, 1 }, { 1, 2 }, { 1, 4 }, { 2, 1 }, { 2, 2 }, { 1, 8 } };
    int k = 0;
    for (int c = 0; c < 6 && k < max_out; ++c) {
        SweepGeom g;
        if (sweep_plan_one(m, n, lda, mat, cand[c][0], cand[c][1], &g) != 0) continue;
        bool dup = false;
        for (int j = 0; j < k; ++j) dup = dup || (out[j].G == g.G && out[j].w == g.w && out[j].nslot == g.nslot);
        if (!dup) out[k++] = g;
    }
    return k;
}

size_t sweep_gran_words(const SweepGeom &g) { return (size_t)g.ngroups * SW_RING * g.G * (2 * g.w); }

int sweep_census_dry_run(hipStream_t st, unsigned *census, unsigned seq)
{
    hipLaunchKernelGGL(sweep_census_k, dim3(256), dim3(SW_THREADS), 0, st, census, seq);
    THIP_LAUNCH_CHECK();
    return 0;
}

template <int NSLOT, int W, int L, int D, int LS>
static int sweep_go(hipStream_t st, const SweepArgs &a)
{
    const size_t lds = (size_t)LS * W * NSLOT * SW_CT * sizeof(f32x4);
    static bool attr_set = false;
    if (lds > 0 && !attr_set) {
        THIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_k<NSLOT, W, L, D, LS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((sweep_k<NSLOT, W, L, D, LS>), dim3(256), dim3(SW_THREADS), lds, st, a);
    THIP_LAUNCH_CHECK();
    return 0;
}

int sweep_launch(hipStream_t st, const SweepGeom &g, const SweepArgs &a)
{
    if (g.w == 2) {
        if (g.nslot == 2) return g.variant == 1 ? sweep_go<2, 2, 8, 3, 0>(st, a) : sweep_go<2, 2, 8, 3, 5>(st, a);
        return g.variant == 1 ? sweep_go<1, 2, 8, 3, 0>(st, a) : sweep_go<1, 2, 8, 3, 5>(st, a);
    }
    // <slots, columns per panel, LAGL, DLAG, LS>: register stages + LDS panels sized to the 512 VGPRs / 160 KB of a CU.
    // variant 2 / 3 (experiments, thip_sweep_test.variant): shallower rings -- a launch runs npan + LAGL + LS intervals, and the
    // last LAGL + LS of them load nothing
#define SW_CASE(N, L0, D0, S0)                                                          \
    case N:                                                                             \
        if (g.variant == 2) return sweep_go<N, 1, 2, 1, 3>(st, a);                      \
        if (g.variant == 3) return sweep_go<N, 1, 2, 1, 2>(st, a);                      \
        return sweep_go<N, 1, L0, D0, S0>(st, a);
    switch (g.nslot) {
    SW_CASE(1, 8, 3, 8)
    SW_CASE(2, 8, 3, 8)
    SW_CASE(3, 8, 3, 6)
    case 4:
        if (g.variant == 2) return sweep_go<4, 1, 2, 1, 3>(st, a);
        if (g.variant == 3) return sweep_go<4, 1, 2, 1, 2>(st, a);
        return g.variant == 1 ? sweep_go<4, 1, 8, 3, 3>(st, a) : sweep_go<4, 1, 8, 3, 5>(st, a);
    SW_CASE(5, 2, 1, 3)      // (3 register stages + 3 LDS panels measured 2 - 5 % ahead of 4 + 3 / 4 + 4 on short sweeps: fewer
    SW_CASE(6, 2, 1, 3)      //  intervals that load nothing at the end of a launch)
    default:
        if (g.variant == 3) return sweep_go<7, 1, 2, 1, 2>(st, a);
        return g.variant == 1 ? sweep_go<7, 1, 3, 1, 2>(st, a) : sweep_go<7, 1, 2, 1, 3>(st, a);
    }
#undef SW_CASE
}

}  // namespace thip

// ---------------------------------------------------------------------------------------------------
// thip_test_sweep: the kernel alone, for tests/test_gpu_sweep.py and tools (include/totsu_f32hip.h)
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ void sw_test_reduce_k(int m, int ngroups, size_t mpad, const float *__restrict__ partH, float *__restrict__ hN,
                                 float *__restrict__ h3)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)m; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.0f, b = 0.0f;
        for (int g = 0; g < ngroups; ++g) { a += partH[((size_t)g * 2 + 0) * mpad + i]; b += partH[((size_t)g * 2 + 1) * mpad + i]; }
        hN[i] = a; h3[i] = b;
    }
}
}  // namespace

extern "C" int thip_test_sweep(const thip_sweep_test *t, float *host_ms, int *host_info)
{
    THIP_NEED_INIT();
    if (!t) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    SweepGeom g;
    if ((t->force_members > 0 ? sweep_plan_one(t->m, t->n, t->lda, t->mat_a, 1, 1, &g, t->force_members)
                              : sweep_plan(t->m, t->n, t->lda, t->mat_a, &g)) != 0)
        return fail(THIP_E_INVALID, "the one-pass kernel cannot take this shape", __FILE__, __LINE__);
    if (t->variant > 0) g.variant = t->variant;
    hipStream_t st = ctx().stream;
    unsigned long long *gran = nullptr;
    unsigned *census = nullptr;
    float *partH = nullptr, *scal = nullptr;
    THIP_TRY(hipMalloc((void **)&gran, sweep_gran_words(g) * sizeof(unsigned long long)));
    THIP_TRY(hipMalloc((void **)&census, 160 * sizeof(unsigned)));
    THIP_TRY(hipMalloc((void **)&partH, (size_t)g.ngroups * 2 * g.mpad * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&scal, 4 * sizeof(float)));
    THIP_TRY(hipMemsetAsync(gran, 0, sweep_gran_words(g) * sizeof(unsigned long long), st));
    THIP_TRY(hipMemsetAsync(census, 0, 160 * sizeof(unsigned), st));
    THIP_TRY(hipMemsetAsync(partH, 0, (size_t)g.ngroups * 2 * g.mpad * sizeof(float), st));
    const float hs[4] = { 0.0f, t->kappa, t->rtau, 1.0f };       // [0] doubles as the stop flag (int 0)
    THIP_TRY(hipMemcpyAsync(scal, hs, sizeof(hs), hipMemcpyHostToDevice, st));
    SweepArgs a;
    a.A = t->mat_a; a.lda = t->lda; a.m = (int)t->m; a.n = (int)t->n;
    a.G = g.G; a.rows_per_member = g.rows_per_member; a.cols_per_group = g.cols_per_group;
    a.v = t->v; a.xy = t->xy; a.c = t->c; a.Su = t->su; a.Tx = t->tx; a.u = t->u; a.ku = t->ku;
    a.xx_in = t->xx_in; a.kx_in = t->kx_in; a.xx_out = t->xx_out; a.kx_out = t->kx_out; a.gP = t->gp;
    a.partH = partH; a.mpad = g.mpad; a.gran = gran; a.census = census;
    a.first = t->first;
    a.pn = nullptr; a.pn_stride = 0; a.tau_p = scal + 3; a.eps_zero = 1e-12f;
    a.kappa_out = nullptr; a.skappa_p = nullptr; a.pm_brx = nullptr; a.np_m = 0; a.pn_count = 0;
    a.pn_in = nullptr; a.pn_in_stride = 0; a.spin_max = SW_SPIN_MAX; a.fault = 0; a.pub_agent = t->pub_agent != 0;
    a.dbg = getenv("THIP_SWEEP_DBG") ? atoi(getenv("THIP_SWEEP_DBG")) : 0;
    a.stop = reinterpret_cast<const int *>(scal); a.kappa_p = scal + 1; a.rtau_p = scal + 2;
    unsigned seq = 0, tagbase = 0;
    THIP_RC(sweep_census_dry_run(st, census, seq++));
    hipEvent_t e0, e1;
    THIP_TRY(hipEventCreate(&e0)); THIP_TRY(hipEventCreate(&e1));
    const int reps = t->reps > 0 ? t->reps : 1;
    float best = 1e30f, tot = 0.0f;
    for (int r = 0; r < reps; ++r) {
        a.seq = seq++; a.tagbase = tagbase; tagbase += (unsigned)g.npan + 1u;
        THIP_TRY(hipEventRecord(e0, st));
        THIP_RC(sweep_launch(st, g, a));
        THIP_TRY(hipEventRecord(e1, st));
        THIP_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        THIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        tot += ms; if (ms < best) best = ms;
    }
    hipLaunchKernelGGL(sw_test_reduce_k, dim3(256), dim3(256), 0, st, (int)t->m, g.ngroups, g.mpad, partH, t->hn, t->h3);
    THIP_LAUNCH_CHECK();
    unsigned hc[160];
    THIP_TRY(hipMemcpyAsync(hc, census, sizeof(hc), hipMemcpyDeviceToHost, st));
    THIP_TRY(hipStreamSynchronize(st));
    if (host_ms) { host_ms[0] = best; host_ms[1] = tot / reps; }
    if (host_info) {
        host_info[0] = (int)hc[9]; host_info[1] = g.G; host_info[2] = g.ngroups; host_info[3] = g.npan; host_info[4] = g.nslot;
        host_info[5] = (int)hc[18]; host_info[6] = (int)hc[19];
    }
#ifdef SW_PROFILE
    fprintf(stderr, "service wave, 10 ns ticks: wait %u, cold+publish %u, tags+poll %u, reduce+math+stores %u, loads %u, barrier %u; intervals that polled %u, polls %u\n",
            hc[10], hc[11], hc[12], hc[13], hc[14], hc[15], hc[16], hc[17]);
    if (getenv("SW_PROFILE_MISSES"))
        for (int i = 0; i < 5; ++i) fprintf(stderr, "miss %d: panel %u lanes %08x%08x\n", i, hc[24 + 3 * i], hc[26 + 3 * i], hc[25 + 3 * i]);
    fprintf(stderr, "streaming wave 0 of workgroup (0, 0), us since kernel entry: census done %.2f, v / x_y loaded %.2f, fill done %.2f, steady loop done %.2f, drain done %.2f, stores done %.2f\n",
            hc[150] * 0.01, hc[151] * 0.01, hc[152] * 0.01, hc[153] * 0.01, hc[154] * 0.01, hc[155] * 0.01);
#endif
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(gran); hipFree(census); hipFree(partH); hipFree(scal);
    return 0;
}

// Can the one-pass kernel run on THIS device for an m x n_local block (leading dimension lda)?  Geometry + one placement
// census, no collective: a multi-rank host asks every rank and takes the minimum BEFORE it builds column-sharded solvers
// (a rank that found out inside thip_solver_init would leave the others waiting in their first all-reduce).
extern "C" int thip_sweep_probe(size_t m, size_t n_local, size_t lda, int *host_ok)
{
    THIP_NEED_INIT();
    if (!host_ok) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    *host_ok = 0;
    SweepGeom g;
    // (the matrix itself is not needed: any 16-byte aligned address stands for it)
    if (sweep_plan((m + 3) / 4 * 4, n_local, (lda + 3) / 4 * 4, reinterpret_cast<const void *>((uintptr_t)4096), &g) != 0) return 0;
    unsigned *census = nullptr;
    THIP_TRY(hipMalloc((void **)&census, 16 * sizeof(unsigned)));
    hipStream_t st = ctx().stream;
    THIP_TRY(hipMemsetAsync(census, 0, 16 * sizeof(unsigned), st));
    THIP_RC(sweep_census_dry_run(st, census, 0));
    unsigned hc[10];
    THIP_TRY(hipMemcpyAsync(hc, census, sizeof(hc), hipMemcpyDeviceToHost, st));
    THIP_TRY(hipStreamSynchronize(st));
    hipFree(census);
    *host_ok = hc[9] == 0u ? 1 : 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// thip_stream_probe: what THIS device streams -- a bare non-temporal read of `bytes` at dev_ptr (16-byte aligned; e.g. the
// solver's own A), eight 16-byte loads in flight per lane, nothing else.  bench.py prints it beside the sweep's rate, so
// that "fraction of what the box can read" is known for the box a line was measured on (boxes of one pool differ by 7 %).
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void stream_read_k(const f32x4 *__restrict__ p, size_t n4, float *out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (; i + 7 * stride < n4; i += 8 * stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += __builtin_nontemporal_load(p + i);
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;      // keeps the loads alive
}

// the same bytes as 32 KB contiguous pieces per workgroup and step (what a persistent kernel's workgroup reads per column):
// DRAM pages are walked in longer runs than by the grid-stride form; the probe reports the better of the two
__global__ __launch_bounds__(256) void stream_read_chunks_k(const f32x4 *__restrict__ p, size_t n4, float *out)
{
    constexpr size_t CH = 256 * 8;                  // 16-byte vectors per chunk
    const size_t nch = n4 / CH;
    f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (size_t c = blockIdx.x; c < nch; c += gridDim.x) {
        const f32x4 *q = p + c * CH + threadIdx.x;
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(q + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (size_t i = nch * CH + (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(p + i);
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}
}  // namespace

extern "C" int thip_stream_probe(const void *dev_ptr, size_t bytes, int reps, float *host_best_ms, float *host_avg_ms)
{
    THIP_NEED_INIT();
    if (!dev_ptr || bytes < 16 || ((uintptr_t)dev_ptr & 15u) != 0) return fail(THIP_E_INVALID, "bad buffer", __FILE__, __LINE__);
    hipStream_t st = ctx().stream;
    const size_t n4 = bytes / 16;
    hipEvent_t e0, e1;
    THIP_TRY(hipEventCreate(&e0)); THIP_TRY(hipEventCreate(&e1));
    if (reps < 1) reps = 1;
    float best = 1e30f, best_avg = 1e30f;
    // grid-stride form at the grids tools/stream_probe.hip found within 2 % of each other at 0.4 - 20 GB, and the chunked form
    // at 8 and 16 resident workgroups per CU; one warm-up each; the best form's best and average launch
    const unsigned grids[5] = { 4096u, 8192u, 16384u, 2048u, 4096u };
    for (int f = 0; f < 5; ++f) {
        const unsigned blocks = grids[f];
        const bool chunks = f >= 3;
        if ((size_t)blocks * 256 * 8 > n4 && f != 0) continue;
        float tot = 0.0f, mn = 1e30f;
        for (int r = 0; r <= reps; ++r) {
            THIP_TRY(hipEventRecord(e0, st));
            if (chunks) hipLaunchKernelGGL(stream_read_chunks_k, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(dev_ptr), n4, ctx().dev_scalar);
            else hipLaunchKernelGGL(stream_read_k, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(dev_ptr), n4, ctx().dev_scalar);
            THIP_TRY(hipEventRecord(e1, st));
            THIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            THIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r == 0) continue;
            tot += ms;
            if (ms < mn) mn = ms;
        }
        if (mn < best) { best = mn; best_avg = tot / reps; }
    }
    THIP_LAUNCH_CHECK();
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (host_best_ms) *host_best_ms = best;
    if (host_avg_ms) *host_avg_ms = best_avg;
    return 0;
}
