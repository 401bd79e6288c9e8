// thip_sweep_kernel.h -- the one-pass kernel itself (sweep_k) and its launcher template, shared by thip_sweep.hip (f32 A) and
// thip_sweep16.hip (A stored as bf16 / column-scaled f16): two translation units so that the instances compile in parallel.
// The description of the kernel is at the top of thip_sweep.hip.
#pragma once
#include "thip_common.h"

#include <type_traits>

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

// One 16-byte slot of a column as f32 values.  ELEM 0: four f32.  ELEM 1 (THIP_A_BF16): eight bf16 -- a bf16 is the high half
// of an f32, so widening is a shift or a mask.  ELEM 2 (THIP_A_F16): eight f16 (v_cvt_f32_f16); the column's power-of-two scale
// is applied to the dots and to the axpy scalars by the service wave, once per column.
template <int ELEM> struct SwElem { static constexpr int EPV = ELEM == 0 ? 4 : 8; static constexpr int ESIZE = ELEM == 0 ? 4 : 2; };
template <int ELEM>
__device__ __forceinline__ void sw_unpack(const f32x4 &raw, float (&e)[SwElem<ELEM>::EPV])
{
    if constexpr (ELEM == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = raw[k];
    } else if constexpr (ELEM == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned u = __float_as_uint(raw[k]);
            e[2 * k] = __uint_as_float(u << 16);
            e[2 * k + 1] = __uint_as_float(u & 0xffff0000u);
        }
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float f = raw[k];                     // (by value: __builtin_bit_cast of the element lvalue reads element 0)
            const h2 h = __builtin_bit_cast(h2, f);
            e[2 * k] = (float)h[0];
            e[2 * k + 1] = (float)h[1];
        }
    }
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

// x + (x of the lane a DPP control pairs this lane with): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141
// (i <-> 7 - i), row_mirror = 0x140 (i <-> 15 - i).  A VALU operand modifier -- no trip through the LDS crossbar, which is what a
// __shfl_xor (ds_bpermute) costs: ~120 cycles each, and the sums of an interval are a CHAIN of them.
template <int CTRL>
__device__ __forceinline__ float sw_dpp_add(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true);
    return x + __builtin_bit_cast(float, y);
}
// sum over the aligned block of N = 8 or 16 lanes this lane belongs to, left in every lane of the block (symmetric pairings:
// the same bits in every lane)
template <int N>
__device__ __forceinline__ float sw_block_sum(float x)
{
    static_assert(N == 8 || N == 16, "half a DPP row or a whole one");
    x = sw_dpp_add<0xB1>(x);
    x = sw_dpp_add<0x4E>(x);
    x = sw_dpp_add<0x141>(x);
    if constexpr (N == 16) x = sw_dpp_add<0x140>(x);
    return x;
}
__device__ __forceinline__ float sw_readlane(float x, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
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
// THIP_SWEEP_DBG (experiments of DESIGN.md 4.7 / 4.8: 1 no polling, 2 no row sums of the dots, 4 no barrier, 8 service wave
// idle, 16 no arithmetic, 32 column stores to the spare line, 64 nothing published (with 1), 256 service wave at priority 0)
// exists only in a -DSW_DEBUG build; otherwise the switches fold away
#ifdef SW_DEBUG
#define SW_DBG(a) ((a).dbg)
#else
#define SW_DBG(a) 0
#endif
template <int NSLOT, int W, int LAGL, int DLAG, int LS, int ELEM = 0>
__global__ __launch_bounds__(SW_THREADS) void sweep_k(const SweepArgs a)
{
    constexpr int EPV = SwElem<ELEM>::EPV;            // rows per 16-byte slot
    constexpr int ESIZE = SwElem<ELEM>::ESIZE;
    constexpr int NS = LAGL + 1;
    constexpr int LAGT = LAGL + LS;                   // loads run this far ahead of the axpy: LAGL in registers, LS more in LDS
    constexpr int PF = LAGL - DLAG;
    constexpr int NL = (2 * W * 32 + 63) / 64;        // granule loads per service lane (G = 32)
    static_assert(SW_RING > 2 * (LAGT - DLAG) - 1 && SW_CR > LAGT - DLAG, "ring depths");
    __shared__ int s_role[4];
    // partial dots of a panel: [parity][quantity][4 wave + DPP row] -- 28 row sums per quantity (entries 28 .. 31 are never read)
    __shared__ __attribute__((aligned(16))) float dotbuf[2][2 * W][32];
    __shared__ float scal[2][2 * W];
    __shared__ float cold[SW_CR][9][W];
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
        // ONE thread decides for the workgroup whether this launch runs: were every wave to read the stop flag and the error
        // word for itself, a peer raising the error word between two waves' loads would send some waves home and leave the
        // others streaming (axpys with a stale `scal`, or granules published from an unwritten dotbuf)
        int g = 0, mbr = 0;
        int go = sw_ticket(a.census, a.seq, a.G, &g, &mbr);
        s_role[0] = g; s_role[1] = mbr;
        if (*a.stop != 0 || __hip_atomic_load(a.census + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) go = 0;
        s_role[2] = go;
    }
    __syncthreads();
    // the ticket comes first even when the loop has stopped: the host numbers the launches, and one that left before
    // counting itself would leave every later launch's tickets off by one
    // (a raised error word -- this launch's census, or an earlier sweep of the batch that gave up -- ends every later sweep
    // at entry: the host restores its snapshot of the iterate, thip_solver.hip sweep_recover)
    if (s_role[2] == 0) return;
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
        float vv[NSLOT][EPV], yv[NSLOT][EPV], acc1[NSLOT][EPV], acc2[NSLOT][EPV];
        int roff[NSLOT];
        bool valid[NSLOT];
        // 16-bit storage (round 6): streaming wave 3 shares its SIMD with the service wave, whose chain takes most of a ~0.9 us
        // interval at priority 3 -- wave 3 is the one the barrier waits for.  A member's rows rarely fill its 7 x NSLOT wave-slots
        // (12 500 rows of bf16: 1563 of 1792 lane-slots), so the rows are dealt out WAVE BY WAVE in the order 0 1 2 4 5 6 3 -- a wave
        // takes all its slots before the next one starts -- and a wave runs the body with as many slots as it has rows for (no loads,
        // no dots, no axpy for the others).  Wave 3 then carries what is left: 27 lanes of one slot at BASELINE configs[2].
        // The f32 instances are untouched (ELEM == 0: slot-major rows, one body).
        const int pos = ELEM != 0 ? (wave == 3 ? SW_CW - 1 : (wave > 3 ? wave - 1 : wave)) : wave;
        // f16 storage: a BALANCED deal -- waves 4 5 6 (the partners of 0 1 2 on their SIMDs) take one slot less when wave 3 can hold what
        // that leaves, so that every SIMD carries about the same: two streaming waves of 4 + 3 slots, or wave 3's 3.4 slots + the
        // service wave (configs[2]: 448 / 448 / 448 / 219 lane-slots + the service chain, instead of 512 / 512 / 512 / 27).  f16 widens
        // every element with a v_cvt and gains 2.2 % from it on one box (666 -> 681 iter/s, sweep 1.487 -> 1.449 ms = 0.86 of 8 TB/s);
        // bf16 (a shift or a mask per element) loses 0.6 % and keeps the plain deal.
        constexpr bool BAL = ELEM == 2 && NSLOT > 1;
        int base_ls = pos * 64 * NSLOT;                   // first lane-slot of this wave
        bool short_cap = false;                           // this wave's capacity is NSLOT - 1 slots
        if constexpr (BAL) {
            const int tls = (row1 - row0 + EPV - 1) / EPV;                       // lane-slots the member needs
            const int rem = tls - 64 * (3 * NSLOT + 3 * (NSLOT - 1));            // what wave 3 would be left with
            if (rem >= 0 && rem <= 64 * NSLOT) {
                base_ls = pos <= 3 ? pos * 64 * NSLOT : 64 * (3 * NSLOT + (pos - 3) * (NSLOT - 1));
                short_cap = pos >= 3 && pos < 6;
            }
        }
        auto row_of = [&](const int sl) { return ELEM != 0 ? row0 + EPV * (base_ls + sl * 64 + lane) : row0 + EPV * (tid + SW_CT * sl); };
        // slots of this wave that hold a row (wave-uniform: lane 0's row of the slot)
        int my_slots = NSLOT;
        if constexpr (ELEM != 0) {
            my_slots = 0;
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) my_slots += (!(short_cap && sl == NSLOT - 1) && row0 + EPV * (base_ls + sl * 64) + EPV <= row1) ? 1 : 0;
            my_slots = __builtin_amdgcn_readfirstlane(my_slots);
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int r = row_of(sl);
            valid[sl] = r + EPV <= row1 && !(short_cap && sl == NSLOT - 1);
            roff[sl] = valid[sl] ? r : (row0 + EPV <= a.m ? row0 : 0);
#pragma unroll
            for (int k = 0; k < EPV; ++k) {
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
                const char *colp = reinterpret_cast<const char *>(a.A) + (size_t)j * a.lda * ESIZE;            \
                _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl)                                           \
                    stg[S][q][sl] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(colp + (size_t)roff[sl] * ESIZE)); \
            }                                                                                                  \
        } while (0)
#define SW_DOTS(S, P)                                                                                          \
        do {                                                                                                   \
            if (SW_DBG(a) & 16) { asm volatile("" :: "v"(stg[S][0][0][0]), "v"(stg[S][W - 1][NSLOT - 1][3])); break; } \
            float p_[2 * W];                                                                                   \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                float d1 = 0.0f, d2 = 0.0f;                                                                    \
                _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                         \
                    float e_[EPV];                                                                             \
                    sw_unpack<ELEM>(stg[S][q][sl], e_);                                                        \
                    _Pragma("unroll") for (int k = 0; k < EPV; ++k) {                                          \
                        d1 = fmaf(e_[k], vv[sl][k], d1);                                                       \
                        d2 = fmaf(e_[k], yv[sl][k], d2);                                                       \
                    }                                                                                          \
                }                                                                                              \
                p_[q] = d1; p_[W + q] = d2;                                                                    \
            }                                                                                                  \
            /* sums over each row of 16 lanes by DPP; lane q of a row writes the row's sum of quantity q -- the service */ \
            /* wave adds the 7 x 4 row sums (a full wave reduction by __shfl_xor was a chain of 6 .. 10 ds_bpermutes */   \
            /* in front of every barrier: with the service wave idle it alone cost 16 % of a 16-bit sweep) */            \
            float r_ = p_[0];                                                                                  \
            if (!(SW_DBG(a) & 2)) {                                                                            \
                _Pragma("unroll") for (int q = 0; q < 2 * W; ++q) {                                            \
                    const float s_ = sw_block_sum<16>(p_[q]);                                                  \
                    r_ = (lane & 15) == q ? s_ : r_;                                                           \
                }                                                                                              \
            }                                                                                                  \
            if ((lane & 15) < 2 * W) dotbuf[(P) & 1][lane & 15][4 * wave + (lane >> 4)] = r_;                  \
        } while (0)
#define SW_AXPY(S, P)                                                                                          \
        do {                                                                                                   \
            if (SW_DBG(a) & 16) { asm volatile("" :: "v"(stg[S][0][0][1]), "v"(stg[S][W - 1][NSLOT - 1][2])); break; } \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                const float s1 = scal[(P) & 1][q], s2 = scal[(P) & 1][W + q];                                  \
                _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                         \
                    float e_[EPV];                                                                             \
                    sw_unpack<ELEM>(stg[S][q][sl], e_);                                                        \
                    _Pragma("unroll") for (int k = 0; k < EPV; ++k) {                                          \
                        acc1[sl][k] = fmaf(e_[k], s1, acc1[sl][k]);                                            \
                        acc2[sl][k] = fmaf(e_[k], s2, acc2[sl][k]);                                            \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
            /* pin the sums here: the optimiser otherwise sinks the axpys of all NS intervals to the end of the */ \
            /* unrolled block and keeps every stage and every interval's scalars alive until then */           \
            _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl)                                               \
                _Pragma("unroll") for (int k = 0; k < EPV; ++k) {                                              \
                    asm volatile("" : "+v"(acc1[sl][k]));                                                      \
                    asm volatile("" : "+v"(acc2[sl][k]));                                                      \
                }                                                                                              \
        } while (0)

#define SW_SPILL(S, P)                                                                                         \
        do {                                                                                                   \
            f32x4 *slot_ = sw_lds + (size_t)((P) % LS) * (W * NSLOT * SW_CT) + tid;                            \
            _Pragma("unroll") for (int q = 0; q < W; ++q)                                                      \
                _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) slot_[(q * NSLOT + sl) * SW_CT] = stg[S][q][sl]; \
        } while (0)
#define SW_AXPY_LDS(P)                                                                                         \
        do {                                                                                                   \
            const f32x4 *slot_ = sw_lds + (size_t)((P) % LS) * (W * NSLOT * SW_CT) + tid;                      \
            _Pragma("unroll") for (int q = 0; q < W; ++q) {                                                    \
                const float s1 = scal[(P) & 1][q], s2 = scal[(P) & 1][W + q];                                  \
                _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                         \
                    const f32x4 d_ = slot_[(q * NSLOT + sl) * SW_CT];                                          \
                    float e_[EPV];                                                                             \
                    sw_unpack<ELEM>(d_, e_);                                                                   \
                    _Pragma("unroll") for (int k = 0; k < EPV; ++k) {                                          \
                        acc1[sl][k] = fmaf(e_[k], s1, acc1[sl][k]);                                            \
                        acc2[sl][k] = fmaf(e_[k], s2, acc2[sl][k]);                                            \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
            _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl)                                               \
                _Pragma("unroll") for (int k = 0; k < EPV; ++k) {                                              \
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
        auto stream = [&](auto nsl_c) {
            constexpr int NSL = decltype(nsl_c)::value;           // slots this wave streams: NSLOT, or NSLOT - 1 (see last_live)
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
        };
        if constexpr (ELEM != 0) {
            // (NSLOT <= 4 for 16-bit storage: at most five bodies)
            if (my_slots == NSLOT) stream(std::integral_constant<int, NSLOT>{});
            else if (NSLOT > 1 && my_slots == NSLOT - 1) stream(std::integral_constant<int, (NSLOT > 1 ? NSLOT - 1 : 0)>{});
            else if (NSLOT > 2 && my_slots == NSLOT - 2) stream(std::integral_constant<int, (NSLOT > 2 ? NSLOT - 2 : 0)>{});
            else if (NSLOT > 3 && my_slots == NSLOT - 3) stream(std::integral_constant<int, (NSLOT > 3 ? NSLOT - 3 : 0)>{});
            else stream(std::integral_constant<int, 0>{});
        } else {
            stream(std::integral_constant<int, NSLOT>{});
        }
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
                const int r = row_of(sl);
#pragma unroll
                for (int k = 0; k < EPV; k += 4) {
                    *reinterpret_cast<float4 *>(h1 + r + k) = make_float4(acc1[sl][k], acc1[sl][k + 1], acc1[sl][k + 2], acc1[sl][k + 3]);
                    *reinterpret_cast<float4 *>(h2 + r + k) = make_float4(acc2[sl][k], acc2[sl][k + 1], acc2[sl][k + 2], acc2[sl][k + 3]);
                }
            }
        SW_PHASE(5);
    } else {
        // ---------------- service wave ----------------
        // Its global loads (the granules of a panel, the per-column data of a panel) are issued TWO intervals before they
        // are used, as the last memory operations of an interval and always the same number of instructions (NL + 1,
        // addresses clamped, no branches around them): the interval then opens with s_waitcnt vmcnt(1 + NL + 1) -- "everything
        // but the loads of the previous interval and the granule store in front of them has arrived" -- and never waits for a
        // round trip.  (Issued one interval ahead and waited with vmcnt(0), every interval paid an L2 round trip: 1.59 us per
        // panel instead of ~1.)  The explicit wait only states the intent: the compiler inserts its own, and what keeps THOSE
        // from degenerating to vmcnt(0) is described at `spare` below.
        // The interval's chain runs on a SIMD it shares with a streaming wave: it goes first.
        if (!(SW_DBG(a) & 256)) __builtin_amdgcn_s_setprio(3);
        float kappa = *a.kappa_p;
        const float rtau = *a.rtau_p;
        // the kappa update (solver.rs:566-567) needs c.rx_x of the previous sweep (its workgroups' partials) and b.rx_y of the
        // m-tail: up to 2 x 512 partials.  Their loads are issued HERE and summed only after the ring's first intervals
        // (finish_kappa below): the first gather that uses kappa is LAGT intervals away, and the streaming waves' first barrier
        // would otherwise wait for this round trip and the f64 sums (~2.5 us per launch)
        const bool kupd = a.kappa_out != nullptr && !a.first;
        float kc[8], kb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            kc[j] = (kupd && k < a.pn_count) ? a.pn_in[3 * a.pn_in_stride + k] : 0.0f;
            kb[j] = (kupd && k < a.np_m) ? a.pm_brx[k] : 0.0f;
        }
        const float tau = *a.tau_p;
        const bool conv = tau > a.eps_zero;
        const float rt = conv ? 1.0f / tau : 1.0f;
        float sdd = 0.0f, scx = 0.0f, scu = 0.0f, scrx = 0.0f;      // lanes < W: sums over the columns this workgroup writes
        const bool comp_u = a.ku != nullptr, comp_x = a.kx_in != nullptr;
        // per-column data: lane l < 9 W fetches field l / W of column l % W (field 8: 1 / scale of an f16-stored column)
        const int cf = lane / W, cq = lane % W;
        const bool clane = lane < 9 * W;
        const float *fp = a.c;                       // lanes without a field read c (a valid address) and drop the value
        bool fvalid = clane;
        if (clane) {
            switch (cf) {
            case 0: fp = a.c; break;      case 1: fp = a.Su; break;    case 2: fp = a.Tx; break;   case 3: fp = a.u; break;
            case 4: fp = a.ku; break;     case 5: fp = a.xx_in; break; case 6: fp = a.kx_in; break; case 7: fp = a.gP; break;
            default: fp = a.inv_s; break;
            }
            if (fp == nullptr) { fp = a.c; fvalid = false; }
        }
        unsigned long long *const gbase = a.gran + (size_t)group * SW_RING * a.G * (2 * W);
        const int nq = a.G * 2 * W;                   // granules per slot (<= 64 NL)
        const int jlast = max(c1 - 1, 0);
        bool dead = false;                            // a gather timed out (here or elsewhere): no more polling
        unsigned polls_total = 0u, polls_max = 0u;    // gathers that had to poll: how often, and the longest (census[18], [19])
        // gather: the LPQ = 64 / 2W lanes [q LPQ, (q + 1) LPQ) hold quantity q (column q's dot with v for q < W, column q - W's
        // with x_y otherwise), lane position p the members p, p + LPQ, ...: the sum over a quantity's members is then a sum over
        // an aligned block of lanes -- DPP adds, no ds_bpermute
        constexpr int LPQ = 64 / (2 * W);
        static_assert(NL * LPQ == 32, "one granule load per LPQ members");
        const int gq = lane / LPQ, gpos = lane % LPQ;
        // publish: lane 8 q + j sums wave j's four row sums of quantity q (one ds_read_b128), the eight lanes are added by DPP
        const int dq = lane >> 3, dj = lane & 7;
        const bool dlane = dq < 2 * W && dj < SW_CW;
        const int cl = lane < W ? lane : 0;
        // Vector-memory instructions retire IN ORDER (vmcnt): the loads an interval issues for the interval after next sit behind
        // the stores issued before them, and a wait for those loads is a wait for every older store's write acknowledgement
        // (~1 us with the memory pipeline full).  Round 4 found every interval waiting for the acknowledgement of the granule
        // it had just published -- vmcnt(0)s the compiler put into the arithmetic because branches around the poll loop's loads
        // and around the granule store cost its wait-count analysis the thread: the floor under every 16-bit sweep (an interval
        // there is ~1 us).  Now: ONE granule store per interval, unconditional (a lane or an interval with nothing to publish
        // writes to the workgroup's spare line behind the granule ring), a poll loop without lane conditions around its loads,
        // and the column stores (one interval in G) behind a branch of their own: the compiler's counts come out as "the loads
        // of the interval before last have landed, the granule just published may still be on its way".
        unsigned long long *const spare = a.gran + (size_t)256 * SW_RING * (2 * W) + (size_t)blockIdx.x * 16;
        float *const sparef = reinterpret_cast<float *>(spare + 1);
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
        // An interval of the service wave is a CHAIN (LDS reads -> sums -> per-column arithmetic -> LDS writes -> barrier) that the
        // streaming waves wait for at the barrier: at two or four columns of 16-bit elements per panel an interval is ~1 us and the
        // chain was the longest thing in it (round 4: with this wave idle a bf16 sweep ran 23 % faster).  Hence: every LDS read of
        // the interval is issued first, the cross-lane sums are DPP adds, and nothing waits for a global round trip.
        // the columns' new values: by the lanes of real columns of the member whose turn it is (a lane without one: spare line)
        auto col_stores = [&](const bool wr, const int j, const float u_new, const float ku_new, const float x_new, const float kx_new,
                              const float g3) {
            *((wr && !a.first) ? a.u + j : sparef) = u_new;
            *((wr && !a.first && comp_u) ? a.ku + j : sparef + 1) = ku_new;
            *(wr ? a.xx_out + j : sparef + 2) = x_new;
            *((wr && comp_x) ? a.kx_out + j : sparef + 3) = kx_new;
            *(wr ? a.gP + j : sparef + 4) = g3;
        };
        auto interval = [&](const int it, unsigned long long (&xg)[NL], float &cv, const bool nowait) {
            if (SW_DBG(a) & 8) { sw_barrier_dbg(SW_DBG(a)); return; }
            // (intervals 0 and 1 use nothing that was fetched: no wait -- the kappa loads above are still in flight then)
            if (!nowait) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 + NL + 1) : "memory");
            SW_STAMP(0);
            const int pp = it - DLAG - 1;                 // panel whose partial dots this workgroup publishes
            const int pa = it - LAGT;                     // panel gathered
            const bool pub = pp >= 0 && pp < npan, gath = pa >= 0 && pa < npan;
            const int pac = gath ? pa : 0;
            const int cs = pac % SW_CR;
            f32x4 dsum = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (pub && dlane) dsum = *reinterpret_cast<const f32x4 *>(&dotbuf[pp & 1][dq][4 * dj]);
            // (no branch of an interval holds a vector-memory instruction -- the poll loop apart, which leaves nothing in flight:
            // the compiler's own wait counts then come out exact; an interval that gathers nothing runs the same instructions on
            // slot 0 and stores to the spare line)
            const float cj = cold[cs][0][cl], Suj = cold[cs][1][cl], Txj = cold[cs][2][cl], uj = cold[cs][3][cl];
            float kuj = cold[cs][4][cl], kxj = cold[cs][6][cl];
            const float xxj = cold[cs][5][cl], gPj = cold[cs][7][cl];
            float isc = 1.0f;
            if constexpr (ELEM == 2) isc = cold[cs][8][cl];                     // stored column = true column x scale
            // per-column data fetched two intervals ago: panel it - 2 - (LAGL - PF - 1)
            {
                const int pc = it - 2 - LAGL + PF + 1;
                if (pc >= 0 && pc < npan && clane) cold[pc % SW_CR][cf][cq] = (fvalid && c0 + pc * W + cq < c1) ? cv : (cf == 8 ? 1.0f : 0.0f);
            }
            // publish the workgroup's partial dots
            {
                float sum = ((dsum[0] + dsum[1]) + dsum[2]) + dsum[3];
                sum = sw_block_sum<8>(sum);
                // TEST HOOK (thip_test_sweep_fault): one workgroup stops publishing half-way -- its group runs out of spins
                const bool withheld = a.fault != 0 && group == 0 && member == a.G - 1 && pp >= npan / 2;
                const bool pw = pub && dj == 0 && dq < 2 * W && !withheld && !(SW_DBG(a) & 64);
                unsigned long long *g = gbase + ((size_t)((pub ? pp : 0) % SW_RING) * a.G + member) * (2 * W) + dq;
                // pub_agent: the documented form (sc1 store, MI355X_MICROARCH.md inter-workgroup visibility); else a plain
                // store that stays in the L2 the group shares (DESIGN.md 4.7 has the measured difference).  The plain store is
                // always issued (to the spare line when the other form is wanted): the default's instruction stream has no branch
                const unsigned long long gv = sw_pack(sum, a.tagbase + (unsigned)pp + 1u);
                __hip_atomic_store((pw && !a.pub_agent) ? g : spare, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (a.pub_agent) __hip_atomic_store(pw ? g : spare, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // gather, scalar updates
            SW_STAMP(1);
            {
                const unsigned tag = a.tagbase + (unsigned)pa + 1u;
                const unsigned long long *g = gbase + (size_t)(pac % SW_RING) * nq + gq;
                float vsum = 0.0f;
                unsigned pend = 0;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    if (gath && gpos + LPQ * i < a.G) {
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
                    // (no lane conditions around the loads: every lane re-reads its NL granules, clamped -- with branches around
                    // them the compiler's wait-count analysis lost track and put vmcnt(0) into the arithmetic below, i.e. every
                    // interval waited for the write acknowledgement of the granule it had just published)
                    unsigned long long yg[NL];
#pragma unroll
                    for (int i = 0; i < NL; ++i) {
                        const int mi = min(gpos + LPQ * i, a.G - 1);
                        yg[i] = __hip_atomic_load(g + (size_t)mi * (2 * W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int i = 0; i < NL; ++i) {
                        const bool hit = ((pend >> i) & 1u) != 0u && (unsigned)(yg[i] >> 32) == tag;
                        vsum += hit ? __uint_as_float((unsigned)yg[i]) : 0.0f;
                        pend = hit ? (pend & ~(1u << i)) : pend;
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
                // every lane of a block: the block's sum; lane w < W takes its column's two
                const float tot = sw_block_sum<(LPQ >= 16 ? 16 : 8)>(vsum);
                float gT = 0.0f, g3 = 0.0f;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    float s1 = sw_readlane(tot, w * LPQ), s2 = sw_readlane(tot, (W + w) * LPQ);
                    if constexpr (LPQ == 32) { s1 += sw_readlane(tot, w * LPQ + 16); s2 += sw_readlane(tot, (W + w) * LPQ + 16); }
                    if (lane == w) { gT = s1; g3 = s2; }
                }
                // (every lane runs the arithmetic -- lanes >= W on column 0's data with zero dots -- and every lane stores: only
                // the lanes of real columns of the member whose turn it is store to the vectors)
                const int j = c0 + pac * W + cl;
                const bool real = gath && lane < W && j < c1 && !dead;
                if constexpr (ELEM == 2) { gT *= isc; g3 *= isc; }
                float u_new = uj;
                if (!a.first) {
                    const float g2 = gPj - 2.0f * g3;
                    u_new = sw_comp_add(uj, Suj * (-g2 - cj * rtau), comp_u, kuj);
                }
                const float x_new = sw_comp_add(xxj, Txj * (gT + cj * kappa), comp_x, kxj);
                if (gath && lane < W) {
                    scal[pa & 1][lane] = real ? u_new * isc : 0.0f;
                    scal[pa & 1][W + lane] = real ? x_new * isc : 0.0f;
                }
                const bool wr = real && member == pa % a.G && !(SW_DBG(a) & 32);
                if (gath && member == pa % a.G) col_stores(wr, j, u_new, kuj, x_new, kxj, g3);
                if (wr) {
                    const float dj_ = conv ? fmaf(rt, g3, cj) : g3;      // solver.rs:596-597 / 634
                    sdd = fmaf(dj_, dj_, sdd);
                    scx = fmaf(cj, xxj, scx);
                    scu = fmaf(cj, u_new, scu);
                    scrx = fmaf(cj, xxj - 2.0f * x_new, scrx);
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
                const unsigned long long *gn = gbase + (size_t)(pn % SW_RING) * nq + gq;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int mi = min(gpos + LPQ * i, a.G - 1);
                    xg[i] = __hip_atomic_load(gn + (size_t)mi * (2 * W), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const int pre = min(LAGT & ~1, total & ~1);
        for (; it < pre; it += 2) { interval(it, xgA, cvA, it < 2); interval(it + 1, xgB, cvB, it < 2); }
        if (kupd) {
            // in f64, in the order of the single-block sums elsewhere (per lane ascending, then the butterfly)
            double dc = 0.0, db = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dc += (double)kc[j]; db += (double)kb[j]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { dc += __shfl_xor(dc, o, 64); db += __shfl_xor(db, o, 64); }
            kappa = fminf(kappa + *a.skappa_p * ((float)dc + (float)db), 0.0f);
            if (group == 0 && member == 0 && lane == 0) *a.kappa_out = kappa;
        }
        for (; it + 1 < total; it += 2) { interval(it, xgA, cvA, false); interval(it + 1, xgB, cvB, false); }
        if (it < total) interval(it, xgA, cvA, false);
        // what the hand-off through the group's L2 costs in polls (a visibility stall would show here long before a time-out)
        if (polls_total != 0u && lane == 0) { atomicAdd(a.census + 18, polls_total); atomicMax(a.census + 19, polls_max); }
        if (a.pn != nullptr) {
            // lanes 0 .. W - 1 hold the sums of their columns
#pragma unroll
            for (int o = 1; o < W; o <<= 1) {
                sdd += __shfl_xor(sdd, o, 64); scx += __shfl_xor(scx, o, 64);
                scu += __shfl_xor(scu, o, 64); scrx += __shfl_xor(scrx, o, 64);
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

template <int NSLOT, int W, int L, int D, int LS, int ELEM = 0>
static inline int sweep_go(hipStream_t st, const SweepArgs &a)
{
    const size_t lds = (size_t)LS * W * NSLOT * SW_CT * sizeof(f32x4);
    static bool attr_set = false;
    if (lds > 0 && !attr_set) {
        THIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_k<NSLOT, W, L, D, LS, ELEM>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((sweep_k<NSLOT, W, L, D, LS, ELEM>), dim3(256), dim3(SW_THREADS), lds, st, a);
    THIP_LAUNCH_CHECK();
    return 0;
}


}  // namespace thip
