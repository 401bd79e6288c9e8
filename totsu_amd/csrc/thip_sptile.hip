// thip_sptile.hip -- sparse operators held ONCE on the device, serving both A x and A^T y (the reference's
// example matrices -- l1reg_lp, partitioning_sdp, toruscompl_socp -- are mostly zeros; user operators follow
// examples/imgnr_udef/src/prob_op_a.rs:33-120).
//
// Why this format.  tools/scatter_probe.hip (profiles/r06_sparse_scatter_probe.txt) priced the alternatives on this part: per stored
// entry (4 B value + 4 B index, 16-byte loads) two GATHERS run at 4.0 TB/s of entries for contiguous indices and 0.3 - 1.0 TB/s for
// random ones, two global float ATOMICS at 0.30 / 0.085 TB/s whatever their scope (agent, or workgroup scope into one accumulator
// per XCD): a column-major one-pass sweep whose axpys are global scatters is 13x slower than two gather passes.  An index-only
// transpose (values once, a permutation for the other direction) stores 4 bytes of permutation where the second copy stores 4
// bytes of value -- it saves nothing and adds a dependent random load.  So: ONE copy of the values in 2-D TILES of 4096 x 4096
// (row block, column block), entry = {f32 value, u32 (local row | local column << 16)}, tiles ordered by (row block, column
// block), entries inside a tile in the caller's CSC order (column, then row), every tile padded to whole 16-byte quads.  Both
// products stream the same 8 bytes per entry with 16-byte loads; the scatter side of either product goes to an accumulator of
// one block (2 x 4096 fixed-point words) in LDS (integer LDS adds: no global atomic anywhere), the gather side reads the in-vector's block from
// LDS (staged once per tile visit) or, for a visit of fewer than 24 576 entries, straight from L2.
//   N product (A [x0 x1]): a workgroup owns (row block, slice): walks that row block's tiles -- contiguous in memory --, in = the
//       column block's slice of x, out = the row block's accumulators; entries of one column hit distinct rows: conflict-free.
//   T product (A^T [y0 y1]): a workgroup owns (column block, slice): walks the tiles of its column block (a list: one per row
//       block), in = the row block's slice of y, out = the column block's accumulators.  Entries are sorted by column, so a
//       wave's 256 entries mostly share ONE column: the lane adds its four products first, a wave whose entries all belong to one
//       column adds them up over the DPP network and issues one ds_add -- a 64-way same-address LDS atomic would serialise.
// Every (block, slice) writes its 2 x 4096 partial sums to part[slice][2][pad] when it is done; the consumer (the m-tail
// kernels of the one-pass schedule, sp_col_k below, finalize_partials) adds the slices in a fixed order.  Slices exist so
// that a matrix with few blocks still fills 256 CUs: the items are chosen by list-scheduling (build()), at most nnz / (16 dim) slices (the partials'
// traffic stays under 1/16 of the entries').  The order in which the waves of ONE workgroup reach an LDS accumulator is not fixed,
// but the accumulators are fixed-point words and the adds integer adds (see sp_tile_k): the sums are bitwise reproducible.
//
// The conic loop on this format (THIP_SCHED_SWEEP with thip_solver_set_sptile) is the dense one-pass schedule's recurrence in
// three launches: T product with [v, x_y] -> sp_col_k (per column: the two scalar updates of thip_sweep_kernel.h's service wave,
// kappa, the sums over n) -> N product with [u, x_x'] into the groups' shares the m-tail reads.  16 bytes per stored entry and
// iteration; the two-copy CSR gathers of round 5 (thip_sparse.hip, carried schedule) read 32.
//
// DENSE tiles.  A tile of full height whose every column holds all 4096 rows (the X / -X blocks of l1reg_lp: 99.98 % of its
// entries) needs no indices: entry e of the tile is row e mod 4096 of column e / 4096.  Such a tile stores its values only
// -- 4 bytes per entry, both on the device and per product -- and is streamed by two lean routines (sp_tile_k: dense_n /
// dense_t) whose sums stay in registers for a whole column (T) or a whole visit (N).
#include "thip_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

using namespace thip;

namespace thip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int SPT_TB = 4096;            // rows / columns per block
constexpr int SPT_THREADS = 1024;          // one workgroup per CU: 96 KB of LDS (two 64-bit accumulator blocks + the staged in-vector block)
constexpr int SPT_THREADS_LITE = 512;      // an operator without a visit worth staging: no in-vector block, 64 KB, two workgroups per CU
// entries of a tile visit from which the in-vector's block (32 KB) is staged in LDS.  8192 until the LITE instance walked its items
// flat: a 5-point Laplacian's diagonal tiles hold ~20 K entries, and with the threshold above them the operator runs LITE
// (T / N 0.230 / 0.203 -> 0.169 / 0.148 ms = 2.1 / 2.4 TB/s of entries; the random 1 % matrix, tiles of 162 K entries, is unmoved;
// at 65 536 its T product loses 8 %)
constexpr int SPT_STAGE_MIN = 24576;

// entries [e0, e0 + cnt) of `vals`, cnt % 4 == 0; their indices at [i0, i0 + cnt) of `idx` -- or nowhere (dense != 0): the
// tile is full, entry e0 + k is row k % 4096 of column k / 4096
struct SptTile { long long e0, i0; int cnt, rb, cw, dense; };
struct SptItem { int out_block, slice, ref0, ref1; long long e_first, e_last; };           // tile refs [ref0, ref1)

// where element i of a block lives in its LDS array: a lane's four entries are four CONSECUTIVE rows of a dense column (a
// dwordx4 of the stream), so instruction e of a wave touches rows 4 t + e -- a stride of four words, eight lanes per bank.
// De-interleaved by i mod 4 the same instruction touches 64 consecutive words (the first N-product kernel ran at 0.8 TB/s of
// entries with the plain layout: profiles/r06_sparse_lp_kernel_stats_first.txt)
__device__ __forceinline__ int spt_slot(int i) { return ((i & 3) << 10) | (i >> 2); }
static_assert(SPT_TB == 4096, "spt_slot de-interleaves a block of 4096");

struct SptArgs {
    const f32x4 *vals; const i32x4 *idx; const SptTile *tiles; const int *order; const SptItem *items;
    const float *in0, *in1;             // in1 == NULL: one right-hand side
    int in_len;
    float *part; size_t opad;           // [slice][2][opad]
    int abs_mode; const int *stop;
    int stage_min;                      // entries of a tile visit from which the in-vector's block is staged in LDS
    // fixed-point accumulation (see sp_tile_k): block maxima of |in0| / |in1| left by sp_absmax_k (nmax each, in1's behind in0's),
    // the exponent bound of the stored values and the bits of headroom for the longest row (N) / column (T)
    const float *xmax; int nmax; int a_exp; int head_bits;
};

constexpr int SPT_FIX_BITS = 50;        // bits of a fixed-point partial sum below the sign (see spt_add)
constexpr int SPT_NMAX = 256;           // block maxima per in-vector

// block maxima of |x0|, |x1| (x1 may be NULL): part[b], part[SPT_NMAX + b]; every block of the grid writes its slot
// (256 or 1024 threads: at most SPT_NMAX blocks, so a long vector -- a stencil's 9 M -- needs the wider block to keep the loads in flight)
__global__ __launch_bounds__(1024) void sp_absmax_k(const float *__restrict__ x0, const float *__restrict__ x1, int len, float *__restrict__ part)
{
    __shared__ float sh[16];
    float m0 = 0.0f, m1 = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
        // (a NaN counts as an infinite entry: the product then answers NaN instead of dropping it -- fmaxf would)
        const float v0 = fabsf(x0[i]);
        m0 = fmaxf(m0, v0 == v0 ? v0 : __builtin_inff());
        if (x1) { const float v1 = fabsf(x1[i]); m1 = fmaxf(m1, v1 == v1 ? v1 : __builtin_inff()); }
    }
    // (max of non-negative floats = max of their bit patterns: block_min on the negated values)
    m0 = -block_min(-m0, sh);
    m1 = -block_min(-m1, sh);
    if (threadIdx.x == 0) { part[blockIdx.x] = m0; part[SPT_NMAX + blockIdx.x] = m1; }
}

// The accumulators are 64-bit FIXED-POINT words and the LDS adds INTEGER adds (round 6, second half).  tools/lds_atomic_probe.hip:
// ds_add_f32 runs at 0.33 lane-adds per clock per CU on this part, ds_add_u32 at 7.4, ds_add_u64 at 5.4-6.7 (profiles/
// r06_lds_atomic_rates.txt) -- the float form was what bounded every scattered pattern (0.64 TB/s of entries).  A product p = a x is
// added as (int64)(p * 2^k): with |a| < 2^(a_exp + 1) (the matrix's, known at build time), |x| <= xmax (sp_absmax_k, one small launch
// in front of the product) and at most 2^head_bits terms per out element (the longest row / column), k = 50 - head_bits - (bound of
// the product's exponent) keeps every partial sum below 2^50 (so that a term converts with one f64 fma, spt_add); the resolution is 2^-(48 - head_bits) of the largest possible
// product -- finer than an f32 accumulator's.  And integer adds are associative: whatever order the waves reach an accumulator
// in, the sum is the same -- the products are BITWISE REPRODUCIBLE again (the register sums of the dense path and the wave sums
// of the T product have a fixed order by construction).
__device__ __forceinline__ double spt_scale(const float *xm, int nmax, int a_exp, int head_bits, float *sh, double *inv)
{
    // every workgroup forms the same maximum from the same block maxima
    float m = 0.0f;
    for (int i = threadIdx.x; i < nmax; i += blockDim.x) m = fmaxf(m, xm[i]);
    m = -block_min(-m, sh);
    int ex = 0;
    if (!(m < __builtin_inff())) { *inv = __builtin_nan(""); return 1.0; }    // an infinite / NaN entry in the in-vector: the product is NaN
    if (m > 0.0f) (void)frexpf(m, &ex);                                   // m < 2^ex
    const int k = SPT_FIX_BITS - head_bits - (a_exp + 1 + ex);
    *inv = ldexp(1.0, -k);
    return ldexp(1.0, k);
}
// The conversion is ONE f64 fma: |p S| < 2^51, so p S + 1.5 2^52 lies in [2^52, 2^53) and the low bits of its mantissa are
// round(p S) in two's complement -- the word to add is the difference of the bit patterns.  (long long)((double)p * S) -- f64 -> i64
// has no instruction: a dozen f64 operations -- made the scattered patterns VALU-bound: PMC on the random 1 % matrix, VALUBusy 73 %
// (T product) / 65 % with 33 % LDS bank conflicts (N), profiles/r06_sparse_patterns_pmc_counters.txt; an integer-only form (frexp,
// 24-bit mantissa, 64-bit shift) measured slower still.  The price: 50 instead of 62 bits below the sign.
__device__ __forceinline__ void spt_add(unsigned long long *acc, float p, double S)
{
    constexpr double M = 6755399441055744.0;        // 1.5 * 2^52
    const double d = fma((double)p, S, M);
    atomicAdd(acc, (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(M)));
}

// Inclusive SEGMENTED sums over the lanes of a wave on the DPP network: keys ascend with the lane (equal keys are neighbours), every
// lane ends with the sum of its key's lanes up to itself.  Row shifts 1, 2, 4, 8 scan inside the rows of 16; row_bcast 15 hands the
// last lane of rows 0 / 2 to rows 1 / 3, row_bcast 31 lane 31 to rows 2 and 3 -- a lane takes the carried sum when the carried key
// is its own (then the whole stretch in between has that key).  18 VALU moves; the __shfl_up form was 18 LDS-crossbar permutes
// on the unit that also serves the gathers and the accumulators.
__device__ __forceinline__ void seg_scan_dpp(const int key, float &a0, float &a1)
{
#define THIP_SEG_STEP(ctrl, rows) { \
        const int kk = __builtin_amdgcn_update_dpp(-1, key, ctrl, rows, 0xf, false); \
        const float t0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), ctrl, rows, 0xf, false)); \
        const float t1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), ctrl, rows, 0xf, false)); \
        if (kk == key) { a0 += t0; a1 += t1; } }
    THIP_SEG_STEP(0x111, 0xf)
    THIP_SEG_STEP(0x112, 0xf)
    THIP_SEG_STEP(0x114, 0xf)
    THIP_SEG_STEP(0x118, 0xf)
    THIP_SEG_STEP(0x142, 0xa)
    THIP_SEG_STEP(0x143, 0xc)
#undef THIP_SEG_STEP
}

// LITE: every tile visit of the operator is below the staging threshold (a very sparse operator: a stencil, the partitioning SDP) --
// no staged block, 64 KB of LDS, 512 threads: two workgroups per CU, whose fills and drains overlap
template <bool TPH, bool LITE = false>
__global__ __launch_bounds__(LITE ? SPT_THREADS_LITE : SPT_THREADS) void sp_tile_k(const SptArgs a)
{
    constexpr int SPT_THREADS = LITE ? thip::SPT_THREADS_LITE : thip::SPT_THREADS;
    extern __shared__ unsigned long long spt_lds[];
    unsigned long long *const lo0 = spt_lds, *const lo1 = spt_lds + SPT_TB;
    float2 *const lin = reinterpret_cast<float2 *>(spt_lds + 2 * SPT_TB);      // (LITE: never touched)
    __shared__ float shm[16];
    if (*a.stop != 0) return;
    const SptItem it = a.items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool two = a.in1 != nullptr;
    for (int i = tid; i < SPT_TB; i += SPT_THREADS) { lo0[i] = 0ull; lo1[i] = 0ull; }
    double inv0 = 1.0, inv1 = 1.0;
    const double S0 = a.abs_mode ? ldexp(1.0, SPT_FIX_BITS - a.head_bits - (a.a_exp + 2)) : spt_scale(a.xmax, a.nmax, a.a_exp, a.head_bits, shm, &inv0);
    const double S1 = (two && !a.abs_mode) ? spt_scale(a.xmax + SPT_NMAX, a.nmax, a.a_exp, a.head_bits, shm, &inv1) : 1.0;
    if (a.abs_mode) inv0 = ldexp(1.0, -(SPT_FIX_BITS - a.head_bits - (a.a_exp + 2)));
    // N product: in a dense column block no LDS add is needed at all: a column of a full tile is 1024 quads -- one step of this
    // loop, or half of one -- so a lane meets the SAME four rows in every step.  The lane keeps the sums of "its" rows in registers for as long
    // as the rows of its next quad are the ones it holds, and pays the LDS adds only when they change (every step, for a
    // scattered pattern: the old cost plus a compare).
    int hold[2][4] = { { -1, -1, -1, -1 }, { -1, -1, -1, -1 } };
    float h0[2][4] = { { 0.0f, 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f, 0.0f } }, h1[2][4] = { { 0.0f, 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f, 0.0f } };
    __syncthreads();                    // the accumulators are zero before any wave adds to them
    // one quad of the stream (slot u of the lane's two in flight): four entries `av` with their index words `iv`, the in-vector's
    // block from LDS (staged) or straight from L2 (in0b / in1b); `tag` tells the visits of one wave apart (the flat walk of LITE)
    auto process = [&](const int u, const f32x4 av, const i32x4 iv, const bool ok, const bool staged, const float *in0b, const float *in1b,
                       const int tag) {
                float p0[4], p1[4]; int oi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned id = (unsigned)iv[e];
                    const int lr = (int)(id & 0xffffu), lc = (int)(id >> 16);
                    const int ii = TPH ? lr : lc;
                    oi[e] = spt_slot(TPH ? lc : lr);
                    float v = ok ? av[e] : 0.0f;
                    float2 x;
                    if (a.abs_mode) { v = fabsf(v); x = make_float2(1.0f, 1.0f); }
                    else if (staged) x = lin[spt_slot(ii)];
                    else x = make_float2(in0b[ii], in1b[ii]);
                    p0[e] = v * x.x; p1[e] = v * x.y;
                }
                if (TPH) {
                    // entries are sorted by column: a lane's four mostly share it, and so does the wave
                    const bool same = oi[0] == oi[3];
                    const float s0 = (p0[0] + p0[1]) + (p0[2] + p0[3]), s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
                    const int tg = tag << 12;           // (a slot is 12 bits)
                    const int first = __builtin_amdgcn_readfirstlane(oi[0] | tg);
                    if (__all(ok && same && (oi[0] | tg) == first)) {
                        const float w0 = wave_sum_dpp(s0), w1 = two ? wave_sum_dpp(s1) : 0.0f;
                        if (lane == 0) { spt_add(&lo0[first & 0xfff], w0, S0); if (two) spt_add(&lo1[first & 0xfff], w1, S1); }
                    } else {
                        // several columns in the wave: a segmented sum over the lanes (keys ascend with the lane: the stream is
                        // column-sorted), one LDS add per column and wave instead of one per entry.  A lane whose quad straddles
                        // columns joins the segment of its LAST column and adds its earlier entries itself; a lane past the end
                        // carries the largest key and nothing.
                        // (equal keys are neighbours also where two visits meet inside a wave: the key carries the visit's tag)
                        const int key = ok ? (oi[3] | tg) : 0x7fffffff;
                        float a0 = 0.0f, a1 = 0.0f;
                        if (ok) {
                            if (same) { a0 = s0; a1 = s1; }
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (oi[e] == oi[3]) { a0 += p0[e]; a1 += p1[e]; }
                                    else { spt_add(&lo0[oi[e]], p0[e], S0); if (two) spt_add(&lo1[oi[e]], p1[e], S1); }
                                }
                            }
                        }
                        seg_scan_dpp(key, a0, a1);
                        const int kn = __builtin_amdgcn_update_dpp(-2, key, 0x130, 0xf, 0xf, false);     // the next lane's key (wave shift left)
                        if (kn != key && key != 0x7fffffff) { spt_add(&lo0[key & 0xfff], a0, S0); if (two) spt_add(&lo1[key & 0xfff], a1, S1); }
                    }
                } else if (ok) {
                    const bool keep = oi[0] == hold[u][0] && oi[1] == hold[u][1] && oi[2] == hold[u][2] && oi[3] == hold[u][3];
                    if (!keep) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (hold[u][e] >= 0) { spt_add(&lo0[hold[u][e]], h0[u][e], S0); if (two) spt_add(&lo1[hold[u][e]], h1[u][e], S1); }
                            hold[u][e] = oi[e]; h0[u][e] = 0.0f; h1[u][e] = 0.0f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h0[u][e] += p0[e]; h1[u][e] += p1[e]; }
                }
    };
    // (a dense tile away from its fast routines -- a clipped end, abs mode, LITE -- computes the index words it does not store)
    auto dense_idx = [&](const long long qq, const long long tile_e0) {
        const int k = (int)((qq << 2) - tile_e0);
        const int w = (k & (SPT_TB - 1)) | ((k >> 12) << 16);
        i32x4 iv;
        iv[0] = w; iv[1] = w + 1; iv[2] = w + 2; iv[3] = w + 3;
        return iv;
    };
    // one visit of a tile: entries [e0, e1) streamed by the threads t0, t0 + stride, ... (the whole workgroup: tid / THREADS; one wave
    // alone: lane / 64)
    auto visit = [&](const SptTile &tl, const long long e0, const long long e1, const bool staged, const float *in0b, const float *in1b,
                     const int t0, const int stride) {
        const long long q1 = e1 >> 2;
        const long long dq = (tl.i0 - tl.e0) >> 2;          // quad of `idx` that goes with a quad of `vals`
        const bool dense = tl.dense != 0;
        for (long long qb = e0 >> 2; qb < q1; qb += 2 * stride) {
            f32x4 av[2]; i32x4 iv[2]; bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long long q = qb + t0 + u * stride;
                ok[u] = q < q1;
                const long long qq = ok[u] ? q : q1 - 1;
                av[u] = __builtin_nontemporal_load(a.vals + qq);
                if (dense) iv[u] = dense_idx(qq, tl.e0);
                else iv[u] = __builtin_nontemporal_load(a.idx + qq + dq);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) process(u, av[u], iv[u], ok[u], staged, in0b, in1b, 0);
        }
    };
    // The fast routines of a DENSE tile (in-vector block staged; [e0, e1) cut on multiples of 256 entries from the tile's start, so a
    // wave's 64 quads lie in one column).  Four 16-byte loads of VALUES in flight per lane: the bytes in flight of the indexed form.
    // T product: a wave takes a contiguous run of 64-quad chunks, lanes keep their partial sums for as long as the column lasts:
    // one wave sum and one LDS add per column and wave.
    auto dense_t = [&](const SptTile &tl, const long long e0, const long long e1) {
        constexpr int NW = SPT_THREADS / 64;
        const long long c0 = (e0 - tl.e0) >> 8, nc = (e1 - e0) >> 8;     // chunks of 64 quads, counted from the tile's start
        const long long w0 = c0 + nc * wave / NW, w1 = c0 + nc * (wave + 1) / NW;
        const f32x4 *const tv = a.vals + (tl.e0 >> 2);
        int cur = -1;
        float s0 = 0.0f, s1 = 0.0f;
        auto flush = [&]() {
            if (cur < 0) return;
            const float t0 = wave_sum_dpp(s0), t1 = two ? wave_sum_dpp(s1) : 0.0f;
            if (lane == 0) { spt_add(&lo0[spt_slot(cur)], t0, S0); if (two) spt_add(&lo1[spt_slot(cur)], t1, S1); }
            s0 = 0.0f; s1 = 0.0f;
        };
        for (long long cb = w0; cb < w1; cb += 4) {
            f32x4 av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long c = cb + u < w1 ? cb + u : w1 - 1;
                av[u] = __builtin_nontemporal_load(tv + (c << 6) + lane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cb + u >= w1) break;                                // (wave-uniform)
                const int k = (int)((cb + u) << 6);                     // the chunk's first quad within the tile
                const int col = k >> 10;
                if (col != cur) { flush(); cur = col; }
                const int r = (((k & 1023) + lane) << 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float2 y = lin[spt_slot(r + e)];
                    s0 = fmaf(av[u][e], y.x, s0); s1 = fmaf(av[u][e], y.y, s1);
                }
            }
        }
        flush();
    };
    // N product: thread t takes the quads t, t + THREADS, ... of the visit -- always the same four rows (a column is 1024 quads) --
    // and keeps their sums in registers for the whole visit; x of the column is one LDS word for the whole wave.
    auto dense_n = [&](const SptTile &tl, const long long e0, const long long e1) {
        static_assert(1024 % SPT_THREADS == 0 || SPT_THREADS % 1024 == 0, "a thread must meet the same rows in every step");
        constexpr int STEP = SPT_THREADS < 1024 ? 1024 : SPT_THREADS;      // quads between a thread's loads
        const long long q0 = e0 >> 2, q1 = e1 >> 2, tq0 = tl.e0 >> 2;
        for (int sub = 0; sub < STEP / SPT_THREADS; ++sub) {
            const int t = tid + sub * SPT_THREADS;
            float r0[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, r1[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (long long qb = q0 + t; qb < q1; qb += 4 * STEP) {
                f32x4 av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long q = qb + u * STEP;
                    av[u] = __builtin_nontemporal_load(a.vals + (q < q1 ? q : qb));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long q = qb + u * STEP;
                    if (q >= q1) break;
                    const float2 x = lin[spt_slot((int)((q - tq0) >> 10))];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r0[e] = fmaf(av[u][e], x.x, r0[e]); r1[e] = fmaf(av[u][e], x.y, r1[e]); }
                }
            }
            if (q0 + t < q1) {
                const int row = (int)(((q0 + t - tq0) & 1023) << 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { spt_add(&lo0[spt_slot(row + e)], r0[e], S0); if (two) spt_add(&lo1[spt_slot(row + e)], r1[e], S1); }
            }
        }
    };
    auto clip = [&](const int ref, const SptTile &tl, long long &e0, long long &e1) {
        e0 = tl.e0; e1 = tl.e0 + tl.cnt;
        if (ref == it.ref0) e0 = it.e_first;
        if (ref == it.ref1 - 1) e1 = it.e_last;
    };
    if constexpr (LITE) {
        // LITE walks its item FLAT: the visits' records are fetched together (one lane each) into a table, their quad counts scanned,
        // and the workgroup streams the concatenation -- every load of the item is in flight at once instead of one chain of dependent
        // round trips (tile record, entries, in-vector) per visit.  A very sparse operator's item is a handful of small visits
        // (a stencil: 3 of ~7 000 entries; the partitioning SDP's equality rows: 31 of 16).
        constexpr int MAXV = 64;
        __shared__ long long v_pre[MAXV + 1], v_q0[MAXV], v_dq[MAXV], v_te0[MAXV];
        __shared__ int v_inb[MAXV];
        for (int base = it.ref0; base < it.ref1; base += MAXV) {
            const int nv = min(MAXV, it.ref1 - base);
            if (base != it.ref0) __syncthreads();           // the previous table is no longer read
            if (wave == 0) {
                long long c = 0;
                if (lane < nv) {
                    const int ref = base + lane;
                    const SptTile tl = a.tiles[a.order ? a.order[ref] : ref];
                    long long e0, e1;
                    clip(ref, tl, e0, e1);
                    v_q0[lane] = e0 >> 2; v_dq[lane] = (tl.i0 - tl.e0) >> 2; v_te0[lane] = tl.e0;
                    v_inb[lane] = (TPH ? tl.rb : tl.cw) | (tl.dense ? 1 << 30 : 0);
                    c = (e1 - e0) >> 2;
                }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(c, d, 64); if (lane >= d) c += t; }
                v_pre[lane + 1] = c;
                if (lane == 0) v_pre[0] = 0;
            }
            __syncthreads();
            const long long total = v_pre[nv];
            int v = 0;
            for (long long gb = 0; gb < total; gb += 2 * SPT_THREADS) {
                f32x4 av[2]; i32x4 iv[2]; bool ok[2]; int vv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const long long g = gb + tid + u * SPT_THREADS;
                    ok[u] = g < total;
                    const long long gc = ok[u] ? g : total - 1;
                    while (gc >= v_pre[v + 1]) ++v;             // (a lane's quads ascend: v only moves forward)
                    vv[u] = v;
                    const long long qq = v_q0[v] + (gc - v_pre[v]);
                    av[u] = __builtin_nontemporal_load(a.vals + qq);
                    if (v_inb[v] >> 30) iv[u] = dense_idx(qq, v_te0[v]);
                    else iv[u] = __builtin_nontemporal_load(a.idx + qq + v_dq[v]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const size_t inb = (size_t)(v_inb[vv[u]] & 0x3fffffff) * SPT_TB;
                    process(u, av[u], iv[u], ok[u], false, a.in0 + inb, (two ? a.in1 : a.in0) + inb, vv[u] + 1);
                }
            }
        }
    } else {
    // (1) the TINY visits (at most one step of a wave: 512 entries), one WAVE each, eight at a time: a visit is a chain of dependent
    // round trips -- tile record, entries, in-vector -- and an item of a very sparse operator walks dozens of them (the equality rows
    // of the partitioning SDP: 31 tiles of 16 entries); no barrier, the accumulators take LDS adds from any wave
    constexpr long long TINY = 2 * 64 * 4;
    for (int ref = it.ref0 + wave; ref < it.ref1; ref += SPT_THREADS / 64) {
        const SptTile tl = a.tiles[a.order ? a.order[ref] : ref];
        long long e0, e1;
        clip(ref, tl, e0, e1);
        if (e1 - e0 > TINY) continue;
        const int inb = TPH ? tl.rb : tl.cw;
        visit(tl, e0, e1, false, a.in0 + (size_t)inb * SPT_TB, (two ? a.in1 : a.in0) + (size_t)inb * SPT_TB, lane, 64);
    }
    // (2) the others, the whole workgroup on each; from stage_min entries on with the in-vector's block staged in LDS
    bool prev_staged = false;
    for (int ref = it.ref0; ref < it.ref1; ++ref) {
        const SptTile tl = a.tiles[a.order ? a.order[ref] : ref];
        long long e0, e1;
        clip(ref, tl, e0, e1);
        if (e1 - e0 <= TINY) continue;
        const int inb = TPH ? tl.rb : tl.cw;
        const float *in0b = a.in0 + (size_t)inb * SPT_TB;
        const float *in1b = (two ? a.in1 : a.in0) + (size_t)inb * SPT_TB;
        const bool fast = !LITE && tl.dense != 0 && !a.abs_mode && ((e0 - tl.e0) & 255) == 0 && ((e1 - tl.e0) & 255) == 0;
        const bool staged = !LITE && ((e1 - e0) >= a.stage_min || fast) && !a.abs_mode;
        if (staged) {
            if (prev_staged) __syncthreads();           // the previous visit's reads of `lin` are done
            const int lim = a.in_len - inb * SPT_TB;
            for (int i = tid; i < SPT_TB; i += SPT_THREADS) lin[spt_slot(i)] = i < lim ? make_float2(in0b[i], in1b[i]) : make_float2(0.0f, 0.0f);
            __syncthreads();
            prev_staged = true;
        }
        if (!LITE && fast) {
            if (TPH) dense_t(tl, e0, e1); else dense_n(tl, e0, e1);
        } else visit(tl, e0, e1, staged, in0b, in1b, tid, SPT_THREADS);
    }
    }
    if (!TPH) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (hold[u][e] >= 0) { spt_add(&lo0[hold[u][e]], h0[u][e], S0); if (two) spt_add(&lo1[hold[u][e]], h1[u][e], S1); }
    }
    __syncthreads();
    const size_t r0 = (size_t)it.out_block * SPT_TB;
    float *o0 = a.part + (size_t)it.slice * 2 * a.opad + r0, *o1 = o0 + a.opad;
    for (int i = tid; i < SPT_TB; i += SPT_THREADS)
        if (r0 + i < a.opad) {
            o0[i] = (float)((double)(long long)lo0[spt_slot(i)] * inv0);
            if (two) o1[i] = (float)((double)(long long)lo1[spt_slot(i)] * inv1);
        }
}

// The per-column step of the one-pass recurrence (the arithmetic of sweep_k's service wave, thip_sweep_kernel.h): gT / g3 = the
// slices' shares of A^T v and A^T x_y added up, then u_k[j], x_x_{k+1}[j], gP[j], and this workgroup's share of the sums over n.
struct SpColArgs { SweepArgs a; const float *partT; int nsl; size_t npad; float *xmax; };   // xmax: block maxima of |u_k|, |x_x_{k+1}| (the N product's in-vectors)

// (256 blocks of 256 threads, or of 1024 when a block has more than 1024 columns)
__global__ __launch_bounds__(1024) void sp_col_k(const SpColArgs ca)
{
    const SweepArgs &a = ca.a;
    __shared__ double shd[16];
    __shared__ float shf[16];
    if (*a.stop != 0) return;
    const int tid = threadIdx.x;
    float kappa = *a.kappa_p;
    const bool kupd = a.kappa_out != nullptr && !a.first;
    if (kupd) {
        // kappa_k (solver.rs:566-567): every workgroup forms it from the same partials in the same order
        double dc = 0.0, db = 0.0;
        for (int k = tid; k < a.pn_count; k += (int)blockDim.x) dc += (double)a.pn_in[3 * a.pn_in_stride + k];
        for (int k = tid; k < a.np_m; k += (int)blockDim.x) db += (double)a.pm_brx[k];
        dc = block_sum_d(dc, shd);
        db = block_sum_d(db, shd);
        kappa = fminf(kappa + *a.skappa_p * ((float)dc + (float)db), 0.0f);
        if (blockIdx.x == 0 && tid == 0) *a.kappa_out = kappa;
    }
    const float rtau = *a.rtau_p, tau = *a.tau_p;
    const bool conv = tau > a.eps_zero;
    const float rt = conv ? 1.0f / tau : 1.0f;
    const bool comp_u = a.ku != nullptr, comp_x = a.kx_in != nullptr;
    const int cpb = (a.n + (int)gridDim.x - 1) / (int)gridDim.x;
    const int j0 = blockIdx.x * cpb, j1 = min(a.n, j0 + cpb);
    float sdd = 0.0f, scx = 0.0f, scu = 0.0f, scrx = 0.0f;
    float mu = 0.0f, mx = 0.0f;         // sp_absmax_k's maxima of the two vectors this kernel writes, over this block's columns
    for (int j = j0 + tid; j < j1; j += (int)blockDim.x) {
        // (eight slices in flight per lane: the sum over up to 256 slices is a latency chain, not a bandwidth problem)
        float gT = 0.0f, g3 = 0.0f;
        {
            float t8[8], g8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { t8[u] = 0.0f; g8[u] = 0.0f; }
            int sl = 0;
            for (; sl + 8 <= ca.nsl; sl += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    t8[u] += ca.partT[((size_t)(sl + u) * 2 + 0) * ca.npad + j];
                    g8[u] += ca.partT[((size_t)(sl + u) * 2 + 1) * ca.npad + j];
                }
            }
            for (; sl < ca.nsl; ++sl) {
                t8[0] += ca.partT[((size_t)sl * 2 + 0) * ca.npad + j];
                g8[0] += ca.partT[((size_t)sl * 2 + 1) * ca.npad + j];
            }
            gT = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
            g3 = ((g8[0] + g8[1]) + (g8[2] + g8[3])) + ((g8[4] + g8[5]) + (g8[6] + g8[7]));
        }
        const float cj = a.c[j], uj = a.u[j], xxj = a.xx_in[j];
        float kuj = comp_u ? a.ku[j] : 0.0f, kxj = comp_x ? a.kx_in[j] : 0.0f;
        float u_new = uj;
        if (!a.first) {
            const float g2 = a.gP[j] - 2.0f * g3;
            const float inc = a.Su[j] * (-g2 - cj * rtau);
            if (comp_u) { const float y = inc - kuj; const float t = uj + y; kuj = (t - uj) - y; u_new = t; }
            else u_new = uj + inc;
            a.u[j] = u_new;
            if (comp_u) a.ku[j] = kuj;
        }
        float x_new;
        {
            const float inc = a.Tx[j] * (gT + cj * kappa);
            if (comp_x) { const float y = inc - kxj; const float t = xxj + y; kxj = (t - xxj) - y; x_new = t; }
            else x_new = xxj + inc;
        }
        a.xx_out[j] = x_new;
        if (comp_x) a.kx_out[j] = kxj;
        {
            const float au = fabsf(u_new), ax = fabsf(x_new);
            mu = fmaxf(mu, au == au ? au : __builtin_inff());
            mx = fmaxf(mx, ax == ax ? ax : __builtin_inff());
        }
        a.gP[j] = g3;
        const float dj = conv ? fmaf(rt, g3, cj) : g3;          // solver.rs:596-597 / 634
        sdd = fmaf(dj, dj, sdd);
        scx = fmaf(cj, xxj, scx);
        scu = fmaf(cj, u_new, scu);
        scrx = fmaf(cj, xxj - 2.0f * x_new, scrx);
    }
    if (ca.xmax != nullptr) {           // (gridDim.x == SPT_NMAX: every slot is written)
        mu = -block_min(-mu, shf);
        mx = -block_min(-mx, shf);
        if (tid == 0) { ca.xmax[blockIdx.x] = mu; ca.xmax[SPT_NMAX + blockIdx.x] = mx; }
    }
    if (a.pn != nullptr) {
        sdd = block_sum(sdd, shf); scx = block_sum(scx, shf); scu = block_sum(scu, shf); scrx = block_sum(scrx, shf);
        if (tid == 0) {
            float *o = a.pn + blockIdx.x;
            o[0] = sdd; o[a.pn_stride] = scx; o[2 * a.pn_stride] = scu; o[3 * a.pn_stride] = scrx;
        }
    }
}

}  // namespace thip

// ---------------------------------------------------------------------------------------------------
// the matrix object
// ---------------------------------------------------------------------------------------------------
struct thip_sptile {
    size_t m = 0, n = 0, nnz = 0, nnz_pad = 0;
    int nrb = 0, ncw = 0, ntiles = 0, nN = 0, nT = 0, slN = 1, slT = 1;
    size_t mpad = 0, npad = 0;
    f32x4 *vals = nullptr; i32x4 *idx = nullptr; SptTile *tiles = nullptr; int *order = nullptr;
    SptItem *itemsN = nullptr, *itemsT = nullptr;
    // partial sums of the trait-level products (thip_sptile_mv), made on first use
    float *partN = nullptr, *partT = nullptr;
    // fixed-point accumulation: |value| < 2^(a_exp + 1); at most 2^headN / 2^headT entries in a row / column; block maxima of the
    // in-vectors of the launch in flight (2 x SPT_NMAX floats)
    int a_exp = 0, headN = 1, headT = 1;
    float *xmax = nullptr;
    int64_t max_visit = 0;      // entries of the largest tile
    int ndense = 0;             // tiles stored without indices
    size_t nidx = 0;            // entries that carry an index
};

namespace thip {

size_t sptile_part_floats(const thip_sptile *M, bool tphase) { return tphase ? (size_t)M->slT * 2 * M->npad : (size_t)M->slN * 2 * M->mpad; }
int sptile_slices(const thip_sptile *M, bool tphase) { return tphase ? M->slT : M->slN; }
size_t sptile_pad(const thip_sptile *M, bool tphase) { return tphase ? M->npad : M->mpad; }
size_t sptile_bytes_per_pass(const thip_sptile *M) { return M->nnz_pad * 4 + M->nidx * 4; }
void sptile_dims(const thip_sptile *M, size_t *m, size_t *n, size_t *nnz) { *m = M->m; *n = M->n; *nnz = M->nnz; }

// part[slice][2][pad] <- the slices' shares of A [in0 in1] (tphase: of A^T [in0 in1]); in1 == NULL: one right-hand side (the
// second half of every slice is then left alone); abs_mode: |A| times ones.  `part` must have been zeroed once after its
// allocation: a (block, slice) without entries is never written.
// xmax_ready: the kernel that produced in0 / in1 left their block maxima in M->xmax (all SPT_NMAX slots of each: sp_col_k does, for
// the N product of the one-pass loop) -- no sp_absmax_k launch.
int sptile_product(hipStream_t st, const thip_sptile *M, bool tphase, const float *in0, const float *in1, float *part,
                   int abs_mode, const int *stop, bool xmax_ready)
{
    const int items = tphase ? M->nT : M->nN;
    if (items == 0) return 0;
    SptArgs a;
    a.vals = M->vals; a.idx = M->idx; a.tiles = M->tiles;
    a.order = tphase ? M->order : nullptr; a.items = tphase ? M->itemsT : M->itemsN;
    a.in0 = in0; a.in1 = in1; a.in_len = (int)(tphase ? M->m : M->n);
    a.part = part; a.opad = tphase ? M->npad : M->mpad;
    a.abs_mode = abs_mode; a.stop = stop ? stop : ctx().never_stop;
    a.xmax = M->xmax; a.a_exp = M->a_exp; a.head_bits = tphase ? M->headT : M->headN; a.nmax = 1;
    if (!abs_mode && xmax_ready) a.nmax = SPT_NMAX;
    else if (!abs_mode) {
        const int len = a.in_len;
        a.nmax = (int)std::min<size_t>(SPT_NMAX, std::max<size_t>(1, ((size_t)len + 1023) / 1024));
        hipLaunchKernelGGL(sp_absmax_k, dim3(a.nmax), dim3(len > SPT_NMAX * 1024 ? 1024 : 256), 0, st, in0, in1, len, M->xmax);
    }
    constexpr size_t lds_full = (size_t)SPT_TB * (2 * sizeof(unsigned long long) + sizeof(float2));
    constexpr size_t lds_lite = (size_t)SPT_TB * (2 * sizeof(unsigned long long));
    // (once per process, whichever thread comes first; a failure is reported to every caller)
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&]() {
        auto set = [&](const void *f, size_t bytes) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
        };
        set(reinterpret_cast<const void *>(&sp_tile_k<true, false>), lds_full);
        set(reinterpret_cast<const void *>(&sp_tile_k<false, false>), lds_full);
        set(reinterpret_cast<const void *>(&sp_tile_k<true, true>), lds_lite);
        set(reinterpret_cast<const void *>(&sp_tile_k<false, true>), lds_lite);
    });
    THIP_TRY(attr_err);
    static const int stage_min = getenv("THIP_SPT_STAGE_MIN") ? atoi(getenv("THIP_SPT_STAGE_MIN")) : SPT_STAGE_MIN;
    a.stage_min = stage_min;
    // (abs mode never stages either)
    const bool lite = M->max_visit < (int64_t)a.stage_min || abs_mode != 0;
    if (lite) {
        if (tphase) hipLaunchKernelGGL((sp_tile_k<true, true>), dim3(items), dim3(SPT_THREADS_LITE), lds_lite, st, a);
        else hipLaunchKernelGGL((sp_tile_k<false, true>), dim3(items), dim3(SPT_THREADS_LITE), lds_lite, st, a);
    } else {
        if (tphase) hipLaunchKernelGGL((sp_tile_k<true, false>), dim3(items), dim3(SPT_THREADS), lds_full, st, a);
        else hipLaunchKernelGGL((sp_tile_k<false, false>), dim3(items), dim3(SPT_THREADS), lds_full, st, a);
    }
    THIP_LAUNCH_CHECK();
    return 0;
}

int sptile_colupdate(hipStream_t st, const thip_sptile *M, const SweepArgs &a, const float *partT)
{
    SpColArgs ca;
    ca.a = a; ca.partT = partT; ca.nsl = M->slT; ca.npad = M->npad; ca.xmax = M->xmax;
    static_assert(SPT_NMAX == 256, "sp_col_k's grid fills every slot of the block maxima");
    hipLaunchKernelGGL(sp_col_k, dim3(256), dim3(M->n > (size_t)256 * 1024 ? 1024 : 256), 0, st, ca);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace thip

namespace {

template <class T>
int upload(const std::vector<T> &h, T **d)
{
    *d = nullptr;
    const size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(T);
    THIP_TRY(hipMalloc((void **)d, bytes));
    if (!h.empty()) THIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int build(thip_sptile *M, size_t m, size_t n, size_t nnz, const int64_t *colptr, const int32_t *rowidx, const float *vals)
{
    M->m = m; M->n = n; M->nnz = nnz;
    M->mpad = (m + 63) / 64 * 64; M->npad = (n + 63) / 64 * 64;
    const size_t nrb = (m + SPT_TB - 1) / SPT_TB, ncw = (n + SPT_TB - 1) / SPT_TB;
    if (m >= ((size_t)1 << 31) || n >= ((size_t)1 << 31) || nrb * ncw > ((size_t)1 << 27))
        return fail(THIP_E_INVALID, "sparse operator too large for the tile directory", __FILE__, __LINE__);
    M->nrb = (int)nrb; M->ncw = (int)ncw;
    if (nnz && (!colptr || !rowidx || !vals)) return fail(THIP_E_INVALID, "null CSC arrays", __FILE__, __LINE__);
    if (n && colptr && (colptr[0] != 0 || (size_t)colptr[n] != nnz)) return fail(THIP_E_INVALID, "column pointers do not span nnz", __FILE__, __LINE__);
    // entries per (row block, column block)
    std::vector<int64_t> cnt(nrb * ncw ? nrb * ncw : 1, 0);
    std::vector<int32_t> rowlen(m ? m : 1, 0);
    std::vector<char> unsorted(ncw ? ncw : 1, 0);   // a column block with a column whose rows do not ascend: no dense tiles there
    static const bool allow_dense = !(getenv("THIP_SPT_DENSE") && atoi(getenv("THIP_SPT_DENSE")) == 0);
    int64_t max_col = 0;
    float amax = 0.0f;
    for (size_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) return fail(THIP_E_INVALID, "column pointers decrease", __FILE__, __LINE__);
        int64_t *crow = cnt.data() + j / SPT_TB;
        max_col = std::max<int64_t>(max_col, colptr[j + 1] - colptr[j]);
        int32_t prev = -1;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            const int32_t r = rowidx[k];
            if (r < 0 || (size_t)r >= m) return fail(THIP_E_INVALID, "row index out of range", __FILE__, __LINE__);
            if (r <= prev) unsorted[j / SPT_TB] = 1;
            prev = r;
            ++crow[(size_t)(r / SPT_TB) * ncw];
            ++rowlen[r];
            const float av = std::fabs(vals[k]);
            if (av > amax && av < std::numeric_limits<float>::infinity()) amax = av;
        }
    }
    {
        // the fixed-point accumulators' bounds: |value| < 2^(a_exp + 1), at most 2^head terms per out element (+ 1 bit of slack)
        int ex = 0;
        if (amax > 0.0f) (void)std::frexp(amax, &ex);           // amax < 2^ex
        M->a_exp = ex - 1;
        const int64_t max_row = m ? *std::max_element(rowlen.begin(), rowlen.end()) : 0;
        auto bits = [](int64_t c) { int b = 0; while (((int64_t)1 << b) < c) ++b; return b + 1; };
        M->headN = bits(std::max<int64_t>(max_row, 1)); M->headT = bits(std::max<int64_t>(max_col, 1));
    }
    std::vector<SptTile> tiles;
    std::vector<int64_t> cur(cnt.size(), -1);       // write cursor of a tile; -1: empty
    std::vector<int> tile_of(cnt.size(), -1);
    int64_t e = 0, ie = 0;
    for (size_t rb = 0; rb < nrb; ++rb)
        for (size_t cw = 0; cw < ncw; ++cw) {
            const int64_t c = cnt[rb * ncw + cw];
            if (c == 0) continue;
            SptTile t; t.e0 = e; t.cnt = (int)((c + 3) / 4 * 4); t.rb = (int)rb; t.cw = (int)cw;
            // dense: full height, ascending rows in every column (then 4096 entries per column are all of its rows), at least four columns
            const int64_t wcols = (int64_t)std::min<size_t>(SPT_TB, n - cw * SPT_TB);
            t.dense = (allow_dense && !unsorted[cw] && (rb + 1) * SPT_TB <= m && wcols >= 4 && c == wcols * SPT_TB) ? 1 : 0;
            t.i0 = t.dense ? 0 : ie;
            if (t.dense) ++M->ndense; else ie += t.cnt;
            if ((c + 3) / 4 * 4 > 0x7fffffff) return fail(THIP_E_INVALID, "a tile holds more than 2^31 entries", __FILE__, __LINE__);
            tile_of[rb * ncw + cw] = (int)tiles.size();
            cur[rb * ncw + cw] = e;
            tiles.push_back(t);
            e += t.cnt;
        }
    M->ntiles = (int)tiles.size();
    for (const SptTile &t_ : tiles) M->max_visit = std::max<int64_t>(M->max_visit, t_.cnt);
    M->nnz_pad = (size_t)e;
    M->nidx = (size_t)ie;
    std::vector<float> hv(M->nnz_pad ? M->nnz_pad : 4, 0.0f);
    std::vector<int32_t> hi(M->nidx ? M->nidx : 4, 0);
    for (size_t j = 0; j < n; ++j) {
        const size_t cw = j / SPT_TB;
        const uint32_t lc = (uint32_t)(j % SPT_TB) << 16;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            const size_t r = (size_t)rowidx[k];
            const size_t ti = (r / SPT_TB) * ncw + cw;
            int64_t &c = cur[ti];
            const SptTile &t = tiles[tile_of[ti]];
            hv[c] = vals[k];
            if (!t.dense) hi[t.i0 + (c - t.e0)] = (int32_t)((uint32_t)(r % SPT_TB) | lc);
            ++c;
        }
    }
    for (const SptTile &t : tiles) {
        const int64_t real_end = cur[(size_t)t.rb * ncw + t.cw];
        for (int64_t k = real_end; k < t.e0 + t.cnt; ++k) { hv[k] = 0.0f; hi[t.i0 + (k - t.e0)] = hi[t.i0 + (real_end - 1 - t.e0)]; }
    }
    // the tiles of a column block, for the T product
    std::vector<int> order;
    order.reserve(tiles.size());
    for (size_t cw = 0; cw < ncw; ++cw)
        for (size_t rb = 0; rb < nrb; ++rb)
            if (tile_of[rb * ncw + cw] >= 0) order.push_back(tile_of[rb * ncw + cw]);
    // items: a block's entries are cut into S equal slices of about `per_item` entries, at least 32 768, at most nnz / (16 dim)
    // (<= 256) slices per block.  One workgroup is resident per CU and the workgroups are handed out in launch order, so a product
    // takes as long as the busiest CU: per_item is the candidate whose items, list-scheduled on 256 CUs at (entries + a fill and a
    // drain worth 32 768 entries) each, finish first.  (Measured on the l1reg_lp matrix with full tiles: 256 / 384 / 512 / 768 / 1024
    // items = 0.367 / 0.421 / 0.385 / 0.384 / 0.392 ms per product -- whole rounds win, a round with a few workgroups loses.)
    // THIP_SPT_ITEMS=k: per_item = stored entries / k, as before.
    static const int items_target = getenv("THIP_SPT_ITEMS") ? std::max(0, atoi(getenv("THIP_SPT_ITEMS"))) : 0;    // (0: choose)
    auto cap_of = [&](size_t dim) { return (int)std::min<size_t>(256, std::max<size_t>(1, M->nnz_pad / (16 * std::max<size_t>(dim, 1)))); };
    const int capN = cap_of(m), capT = cap_of(n);
    auto pick_per_item = [&](const std::vector<int64_t> &lens, const int cap) -> int64_t {
        if (items_target > 0) return std::max<int64_t>((int64_t)(M->nnz_pad / items_target), 32768);
        constexpr int CUS = 256;
        constexpr int64_t FIXED = 32768;
        int64_t best_p = std::max<int64_t>((int64_t)(M->nnz_pad / 512), 32768), best_t = std::numeric_limits<int64_t>::max();
        std::vector<int64_t> heap;
        for (int t = 128; t <= 1024; t += 8) {
            const int64_t P = std::max<int64_t>((int64_t)(M->nnz_pad / t), 32768);
            heap.assign(CUS, 0);            // (a min-heap of the CUs' finish times)
            int64_t span = 0;
            for (const int64_t len : lens) {
                const int S = (int)std::min<int64_t>(cap, std::max<int64_t>(1, (len + P - 1) / P));
                const int64_t cost = (len + S - 1) / S + FIXED;
                for (int k = 0; k < S; ++k) {
                    std::pop_heap(heap.begin(), heap.end(), std::greater<int64_t>());
                    heap.back() += cost;
                    span = std::max(span, heap.back());
                    std::push_heap(heap.begin(), heap.end(), std::greater<int64_t>());
                }
            }
            if (span < best_t) { best_t = span; best_p = P; }
            if (P == 32768) break;
        }
        return best_p;
    };
    std::vector<int64_t> lensN, lensT;
    for (size_t t0 = 0; t0 < tiles.size();) {
        size_t t1 = t0;
        int64_t len = 0;
        while (t1 < tiles.size() && tiles[t1].rb == tiles[t0].rb) len += tiles[t1++].cnt;
        lensN.push_back(len);
        t0 = t1;
    }
    for (size_t p0 = 0; p0 < order.size();) {
        size_t p1 = p0;
        int64_t len = 0;
        while (p1 < order.size() && tiles[order[p1]].cw == tiles[order[p0]].cw) len += tiles[order[p1++]].cnt;
        lensT.push_back(len);
        p0 = p1;
    }
    const int64_t per_itemN = pick_per_item(lensN, capN), per_itemT = pick_per_item(lensT, capT);
    std::vector<SptItem> itN, itT;
    M->slN = 1; M->slT = 1;
    {
        size_t t0 = 0;
        while (t0 < tiles.size()) {
            size_t t1 = t0;
            while (t1 < tiles.size() && tiles[t1].rb == tiles[t0].rb) ++t1;
            const int64_t E0 = tiles[t0].e0, E1 = tiles[t1 - 1].e0 + tiles[t1 - 1].cnt, len = E1 - E0;
            const int S = (int)std::min<int64_t>(capN, std::max<int64_t>(1, (len + per_itemN - 1) / per_itemN));
            M->slN = std::max(M->slN, S);
            size_t tr = t0;
            // a cut falls on a multiple of 256 entries FROM ITS TILE'S START: a wave streams 64 quads, and in a dense tile (columns of
            // 4096 entries) no wave then straddles two columns -- a straddling wave of the T product runs the segmented sum
            auto cutN = [&](int64_t c) -> int64_t {
                if (c <= E0) return E0;
                if (c >= E1) return E1;
                size_t t_ = t0;
                while (tiles[t_].e0 + tiles[t_].cnt <= c) ++t_;
                const int64_t r = std::min<int64_t>(((c - tiles[t_].e0 + 128) / 256) * 256, tiles[t_].cnt);
                return tiles[t_].e0 + r;
            };
            for (int s = 0; s < S; ++s) {
                const int64_t a0 = s == 0 ? E0 : cutN(E0 + (len / 4 * s / S) * 4), a1 = s + 1 == S ? E1 : cutN(E0 + (len / 4 * (s + 1) / S) * 4);
                if (a1 <= a0) {     // (cannot happen: len >= 4 S is not guaranteed for tiny blocks -- keep the slice, empty)
                    SptItem it_{ tiles[t0].rb, s, (int)t0, (int)t0, a0, a0 };
                    itN.push_back(it_);
                    continue;
                }
                while (tiles[tr].e0 + tiles[tr].cnt <= a0) ++tr;
                size_t tl = tr;
                while (tiles[tl].e0 + tiles[tl].cnt < a1) ++tl;
                SptItem it_{ tiles[t0].rb, s, (int)tr, (int)tl + 1, a0, a1 };
                itN.push_back(it_);
            }
            t0 = t1;
        }
    }
    {
        size_t p0 = 0;
        while (p0 < order.size()) {
            size_t p1 = p0;
            while (p1 < order.size() && tiles[order[p1]].cw == tiles[order[p0]].cw) ++p1;
            std::vector<int64_t> cum(p1 - p0 + 1, 0);
            for (size_t p = p0; p < p1; ++p) cum[p - p0 + 1] = cum[p - p0] + tiles[order[p]].cnt;
            const int64_t len = cum.back();
            const int S = (int)std::min<int64_t>(capT, std::max<int64_t>(1, (len + per_itemT - 1) / per_itemT));
            M->slT = std::max(M->slT, S);
            size_t pr = 0;
            auto cutT = [&](int64_t c) -> int64_t {         // (in the column block's cumulative entry space; see cutN)
                if (c <= 0) return (int64_t)0;
                if (c >= len) return len;
                size_t q_ = 0;
                while (cum[q_ + 1] <= c) ++q_;
                const int64_t r = std::min<int64_t>(((c - cum[q_] + 128) / 256) * 256, cum[q_ + 1] - cum[q_]);
                return cum[q_] + r;
            };
            for (int s = 0; s < S; ++s) {
                const int64_t a0 = s == 0 ? 0 : cutT((len / 4 * s / S) * 4), a1 = s + 1 == S ? len : cutT((len / 4 * (s + 1) / S) * 4);
                if (a1 <= a0) {
                    SptItem it_{ tiles[order[p0]].cw, s, (int)p0, (int)p0, 0, 0 };
                    itT.push_back(it_);
                    continue;
                }
                while (cum[pr + 1] <= a0) ++pr;
                size_t pl = pr;
                while (cum[pl + 1] < a1) ++pl;
                SptItem it_{ tiles[order[p0]].cw, s, (int)(p0 + pr), (int)(p0 + pl) + 1,
                             tiles[order[p0 + pr]].e0 + (a0 - cum[pr]), tiles[order[p0 + pl]].e0 + (a1 - cum[pl]) };
                itT.push_back(it_);
            }
            p0 = p1;
        }
    }
    M->nN = (int)itN.size(); M->nT = (int)itT.size();
    THIP_TRY(hipMalloc((void **)&M->vals, hv.size() * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&M->idx, hi.size() * sizeof(int32_t)));
    THIP_TRY(hipMemcpy(M->vals, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
    THIP_TRY(hipMemcpy(M->idx, hi.data(), hi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    THIP_RC(upload(tiles, &M->tiles));
    THIP_RC(upload(order, &M->order));
    THIP_RC(upload(itN, &M->itemsN));
    THIP_RC(upload(itT, &M->itemsT));
    THIP_TRY(hipMalloc((void **)&M->xmax, 2 * SPT_NMAX * sizeof(float)));
    THIP_TRY(hipMemset(M->xmax, 0, 2 * SPT_NMAX * sizeof(float)));
    return 0;
}

}  // namespace

extern "C" {

int thip_sptile_create(size_t n_row, size_t n_col, size_t nnz, const int64_t *host_colptr, const int32_t *host_rowidx,
                       const float *host_vals, thip_sptile **out)
{
    THIP_NEED_INIT();
    if (!out) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    *out = nullptr;
    thip_sptile *M = new thip_sptile();
    const int rc = build(M, n_row, n_col, nnz, host_colptr, host_rowidx, host_vals);
    if (rc != 0) { thip_sptile_destroy(M); return rc; }
    *out = M;
    return 0;
}

int thip_sptile_destroy(thip_sptile *M)
{
    if (!M) return 0;
    if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
    for (void *p : { (void *)M->vals, (void *)M->idx, (void *)M->tiles, (void *)M->order, (void *)M->itemsN, (void *)M->itemsT,
                     (void *)M->partN, (void *)M->partT, (void *)M->xmax })
        if (p) (void)hipFree(p);
    delete M;
    return 0;
}

int thip_sptile_info(const thip_sptile *M, size_t *host_nnz_stored, int *host_tiles, int *host_items_n, int *host_items_t,
                     int *host_slices_n, int *host_slices_t, size_t *host_bytes)
{
    if (!M) return fail(THIP_E_INVALID, "null matrix", __FILE__, __LINE__);
    if (host_nnz_stored) *host_nnz_stored = M->nnz_pad;
    if (host_tiles) *host_tiles = M->ntiles;
    if (host_items_n) *host_items_n = M->nN;
    if (host_items_t) *host_items_t = M->nT;
    if (host_slices_n) *host_slices_n = M->slN;
    if (host_slices_t) *host_slices_t = M->slT;
    if (host_bytes) *host_bytes = M->nnz_pad * 4 + M->nidx * 4 + (size_t)M->ntiles * (sizeof(SptTile) + sizeof(int))
                                  + (size_t)(M->nN + M->nT) * sizeof(SptItem);
    return 0;
}

int thip_sptile_layout(const thip_sptile *M, int *host_dense_tiles, size_t *host_indexed_entries, size_t *host_bytes_per_product)
{
    if (!M) return fail(THIP_E_INVALID, "null matrix", __FILE__, __LINE__);
    if (host_dense_tiles) *host_dense_tiles = M->ndense;
    if (host_indexed_entries) *host_indexed_entries = M->nidx;
    if (host_bytes_per_product) *host_bytes_per_product = sptile_bytes_per_pass(M);
    return 0;
}

// y = alpha * A x + beta * y (transpose != 0: A^T x); abs_mode != 0: |A| and x = 1 (MatOp::absadd_*, matop.rs:98-117).
// Replaces the reference's dense transform_ge call of MatOp::op_impl (matop.rs:76-86) for an operator held sparse.
int thip_sptile_mv(thip_sptile *M, int transpose, float alpha, const float *x, float beta, float *y, int abs_mode)
{
    THIP_NEED_INIT();
    if (!M) return fail(THIP_E_INVALID, "null matrix", __FILE__, __LINE__);
    hipStream_t st = ctx().stream;
    const bool t = transpose != 0;
    const size_t len = t ? M->n : M->m;
    if (len == 0) return 0;
    float *&part = t ? M->partT : M->partN;
    if (!part) {
        const size_t fl = sptile_part_floats(M, t);
        THIP_TRY(hipMalloc((void **)&part, fl * sizeof(float)));
        THIP_TRY(hipMemsetAsync(part, 0, fl * sizeof(float), st));
    }
    THIP_RC(sptile_product(st, M, t, abs_mode ? (const float *)M->vals : x, nullptr, part, abs_mode, nullptr));
    return finalize_partials(st, len, part, sptile_slices(M, t), 2 * sptile_pad(M, t), alpha, beta, y, nullptr);
}


int thip_test_sptile_time(thip_sptile *M, int reps, float *host_ms)
{
    THIP_NEED_INIT();
    if (!M || !host_ms) return fail(THIP_E_INVALID, "null argument", __FILE__, __LINE__);
    hipStream_t st = ctx().stream;
    const size_t nv = std::max(M->mpad, M->npad) + 64;
    float *ones = nullptr, *pt = nullptr, *pn = nullptr;
    THIP_TRY(hipMalloc((void **)&ones, 2 * nv * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&pt, std::max<size_t>(sptile_part_floats(M, true), 64) * sizeof(float)));
    THIP_TRY(hipMalloc((void **)&pn, std::max<size_t>(sptile_part_floats(M, false), 64) * sizeof(float)));
    std::vector<float> h(2 * nv, 1.0f);
    THIP_TRY(hipMemcpy(ones, h.data(), 2 * nv * sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    THIP_TRY(hipEventCreate(&e0)); THIP_TRY(hipEventCreate(&e1));
    if (reps < 1) reps = 1;
    for (int ph = 0; ph < 2; ++ph) {
        float best = 1e30f, tot = 0.0f;
        for (int r = 0; r <= reps; ++r) {
            THIP_TRY(hipEventRecord(e0, st));
            THIP_RC(sptile_product(st, M, ph == 0, ones, ones + nv, ph == 0 ? pt : pn, 0, nullptr));
            THIP_TRY(hipEventRecord(e1, st));
            THIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            THIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r == 0) continue;
            tot += ms; if (ms < best) best = ms;
        }
        host_ms[ph] = best; host_ms[2 + ph] = tot / reps;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(ones); hipFree(pt); hipFree(pn);
    return 0;
}

}  // extern "C"
