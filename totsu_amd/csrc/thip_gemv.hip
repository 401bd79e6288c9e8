// thip_gemv.hip -- dense GEMV on a column-major matrix for gfx950: the >99 %-of-bytes kernel of the
// conic iteration.  Replaces cublasSgemv at totsu_f32cuda/src/f32cuda.rs:144-171
// (semantic spec totsu_core/src/floatgeneric.rs:331-353) and MatOp::absadd_impl's per-column / per-row
// asum loops (totsu_core/src/matop.rs:98-117).
//
// One kernel template does y_N = A x_N and y_T = A^T x_T from ONE read of A ("dual GEMV"): the two
// products of SelfDualEmbed::op / ::trans_op / criteria_conv take independent inputs
// (solver.rs:122 vs 125, 146 vs 149, 594 vs 597), so a tile fetched for one is reused for the other.
//
// Tiling (HBM-bound: 0.5 flop/byte, so no LDS staging of A and no MFMA -- A streams straight to VGPRs):
//   block = 256 threads (4 waves); a block owns TILE = 256*VW*NJ consecutive rows and a chunk of columns;
//   lane t holds rows r0 + j*256*VW + t*VW + k  (k < VW, j < NJ): a wave-load is 64 x 16 B = 1 KiB
//   contiguous, the 4 waves cover 4 KiB contiguous of one column, KU columns are in flight (>= 8 loads
//   of 16 B per lane before the first use);
//   N: per-lane accumulators over the chunk's columns -> partial vector per chunk (float4 stores);
//   T: per-column partial dot over the lane's rows, KU columns reduced across the wave with a
//      transpose-reduce butterfly (K values on 64 lanes -> K sums in K+3 shuffles instead of 6K),
//      4 waves combined through LDS -> partial vector per row tile.
// The second reduction stage (over chunks / tiles) is deterministic: a finalize kernel, or folded into
// the consumer kernel of the fused iteration (thip_solver.hip).
#include "thip_common.h"

#include <cstdlib>

using namespace thip;

namespace {

constexpr int BLK = 256;
constexpr int MAXCW = 1024;   // max columns per chunk (LDS: 4 waves x MAXCW floats = 16 KiB)

template <int K, int O>
__device__ __forceinline__ void halve(float *v, int lane)
{
    const bool hi = (lane & O) != 0;
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {
        const float send = hi ? v[i] : v[i + K / 2];
        const float keep = hi ? v[i + K / 2] : v[i];
        v[i] = keep + __shfl_xor(send, O, 64);
    }
}

// K per-lane values on 64 lanes -> the K wave-wide sums; the lanes with (lane & (64/K - 1)) == 0 and
// (lane >> (6 - log2 K)) == c hold sum c (in fact every lane of that group does)
template <int K>
__device__ __forceinline__ float multi_reduce(float *v, int lane)
{
    if constexpr (K == 8) {
        halve<8, 32>(v, lane); halve<4, 16>(v, lane); halve<2, 8>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    } else if constexpr (K == 4) {
        halve<4, 32>(v, lane); halve<2, 16>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    } else if constexpr (K == 2) {
        halve<2, 32>(v, lane);
        float r = v[0];
        r += __shfl_xor(r, 16, 64); r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
        return r;
    } else {
        return wave_sum(v[0]);
    }
}

template <int K> struct Log2;
template <> struct Log2<1> { static constexpr int v = 0; };
template <> struct Log2<2> { static constexpr int v = 1; };
template <> struct Log2<4> { static constexpr int v = 2; };
template <> struct Log2<8> { static constexpr int v = 3; };

// FULL: every row of the block's tile is < m (uniform per block): loads are unconditional, so the KU * NJ loads of a
// step are issued back to back (with the per-lane row guard each load sits in its own branch followed by a
// vmcnt(0) wait, i.e. one load in flight per wave -- seen in the ISA, worth +1..+9 % depending on the shape)
// element types of the stored matrix: float, or bf16 as raw 16-bit patterns (the upper half of the f32 encoding)
// or IEEE f16 with one power-of-two scale per column (stored = f16(a * s_c); the kernel multiplies x_N[c] and the
// column's A^T product by 1 / s_c, so the result is that of the matrix stored / s_c)
typedef unsigned short bf16raw;
typedef _Float16 f16elt;
template <typename E> struct IsF16 { static constexpr bool v = false; };
template <> struct IsF16<f16elt> { static constexpr bool v = true; };
__device__ __forceinline__ float elt_f32(float v) { return v; }
__device__ __forceinline__ float elt_f32(bf16raw v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float elt_f32(f16elt v) { return (float)v; }
__device__ __forceinline__ unsigned raw16(bf16raw v) { return v; }
__device__ __forceinline__ unsigned raw16(f16elt v) { return (unsigned)__builtin_bit_cast(unsigned short, v); }

// FULL: every load of the tile is in bounds, no guards.  CLAMP (with FULL): the same unguarded code for a PARTIAL last row
// tile -- row group j of this lane loads from rofs[j], which is its own row where that exists and the tile's first row
// where it does not (one cache line for all such lanes: no extra HBM traffic); what the stray values would contribute is
// nullified by x_T = 0 on those rows and by the guarded store of the N sums.  The guarded form of a partial tile runs
// at a fraction of the rate (m = 125 250: 184 -> 163 us per launch).
template <typename E, int VW, int NJ, int K, bool DO_N, bool DO_T, bool ABS, bool NT, bool FULL, bool CLAMP = false>
__device__ __forceinline__ void step(const E *__restrict__ A, size_t lda, int m, int r_first, int c, int cc,
                                     const float *__restrict__ xn, const float (&xtv)[NJ][VW],
                                     float (&accN)[NJ][VW], float *ldsT_wave, int lane, const float *__restrict__ inv_s,
                                     const int (&rofs)[NJ])
{
    static_assert(FULL || !CLAMP, "CLAMP is a form of FULL");
    constexpr bool F16 = IsF16<E>::v;
    if constexpr (VW == 8) {
        // bf16 storage: 8 rows per 16-byte load.  The raw dwords stay in registers and are widened (one shift or
        // mask per element) right where they are consumed, so the live set is the loads themselves
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t raw[K][NJ];
        float xsv[K], isv[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            isv[u] = F16 ? inv_s[c + u] : 1.0f;
            xsv[u] = ((ABS || !DO_N) ? 1.0f : xn[c + u]) * isv[u];
        }
        if constexpr (FULL) {
            // All K * NJ loads are issued back to back and each column is consumed as soon as ITS loads have landed.
            // Left to itself the compiler either keeps every widened value alive (115+ VGPRs) or, once the updates are
            // pinned per column, splits the loads into two groups of four: the scheduling barrier below keeps the load
            // block together, and the compiler's own waitcnt insertion then counts the loads down column by column
            // (round 1 issued them from inline asm with hand-written vmcnt waits, which the compiler could not see).
#pragma unroll
            for (int u = 0; u < K; ++u) {
                const E *col = A + (size_t)(c + u) * lda + (CLAMP ? 0 : r_first);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const u32x4_t *src = reinterpret_cast<const u32x4_t *>(col + (CLAMP ? rofs[j] : j * (BLK * VW)));
                    raw[u][j] = NT ? __builtin_nontemporal_load(src) : *src;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int u = 0; u < K; ++u) {
                const E *col = A + (size_t)(c + u) * lda;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r = r_first + j * (BLK * VW);
                    if (r + 8 <= m) {
                        raw[u][j] = *reinterpret_cast<const u32x4_t *>(col + r);
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const unsigned lo = (r + 2 * d < m) ? raw16(col[r + 2 * d]) : 0u;
                            const unsigned hi = (r + 2 * d + 1 < m) ? raw16(col[r + 2 * d + 1]) : 0u;
                            raw[u][j][d] = lo | (hi << 16);
                        }
                    }
                }
            }
        }
        float p[K];
        constexpr unsigned HI = ABS ? 0x7fff0000u : 0xffff0000u;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            // packed f32 math (v_pk_fma_f32): the (low, high) halves of a dword are one 2-vector for both products
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t xs2 = { xsv[u], xsv[u] };
            f32x2_t s2 = { 0.0f, 0.0f };
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned q = raw[u][j][d];
                    f32x2_t a2;
                    if constexpr (F16) {
                        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                        const f16x2_t hv = __builtin_bit_cast(f16x2_t, q);
                        a2 = f32x2_t{ (float)hv[0], (float)hv[1] };
                        if constexpr (ABS) a2 = f32x2_t{ fabsf(a2[0]), fabsf(a2[1]) };
                    } else {
                        a2 = f32x2_t{ __uint_as_float((q << 16) & HI), __uint_as_float(q & HI) };
                    }
                    if constexpr (DO_N) {
                        f32x2_t acc2 = { accN[j][2 * d], accN[j][2 * d + 1] };
                        acc2 = __builtin_elementwise_fma(a2, xs2, acc2);
                        accN[j][2 * d] = acc2[0]; accN[j][2 * d + 1] = acc2[1];
                    }
                    if constexpr (DO_T) {
                        const f32x2_t t2 = { xtv[j][2 * d], xtv[j][2 * d + 1] };
                        s2 = __builtin_elementwise_fma(a2, t2, s2);
                    }
                }
            p[u] = (s2[0] + s2[1]) * isv[u];
            // pin column u's updates here: the scheduler otherwise defers every N update behind the T phase and
            // keeps 8 * K * NJ widened values alive (115+ VGPRs, half the waves per SIMD)
            if constexpr (DO_N) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(accN[j][k]));
            }
            asm volatile("" : "+v"(p[u]));
        }
        if constexpr (DO_T) {
            const float r = multi_reduce<K>(p, lane);
            constexpr int SH = 6 - Log2<K>::v;
            if ((lane & ((64 >> Log2<K>::v) - 1)) == 0) ldsT_wave[cc + (lane >> SH)] = r;
        }
        return;
    }
    float av[K][NJ][VW];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        const E *col = A + (size_t)(c + u) * lda;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = CLAMP ? rofs[j] : r_first + j * (BLK * VW);
            if constexpr (VW == 4) {
                if (FULL || r + 4 <= m) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    const f32x4_t *src = reinterpret_cast<const f32x4_t *>(col + r);
                    const f32x4_t q = NT ? __builtin_nontemporal_load(src) : *src;
                    av[u][j][0] = q[0]; av[u][j][1] = q[1]; av[u][j][2] = q[2]; av[u][j][3] = q[3];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) av[u][j][k] = (r + k < m) ? elt_f32(col[r + k]) : 0.0f;
                }
            } else {
                av[u][j][0] = (FULL || r < m) ? elt_f32(col[r]) : 0.0f;
            }
        }
    }
    if constexpr (ABS) {
#pragma unroll
        for (int u = 0; u < K; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < VW; ++k) av[u][j][k] = fabsf(av[u][j][k]);
    }
    if constexpr (DO_N) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const float xs = (ABS ? 1.0f : xn[c + u]) * (F16 ? inv_s[c + u] : 1.0f);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < VW; ++k) accN[j][k] = fmaf(av[u][j][k], xs, accN[j][k]);
        }
    }
    if constexpr (DO_T) {
        float p[K];
#pragma unroll
        for (int u = 0; u < K; ++u) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int k = 0; k < VW; ++k) s = fmaf(av[u][j][k], xtv[j][k], s);
            p[u] = F16 ? s * inv_s[c + u] : s;
        }
        const float r = multi_reduce<K>(p, lane);
        constexpr int SH = 6 - Log2<K>::v;
        if ((lane & ((64 >> Log2<K>::v) - 1)) == 0) ldsT_wave[cc + (lane >> SH)] = r;
    }
}

template <typename E, int VW, int NJ, int KU, bool DO_N, bool DO_T, bool ABS, bool NT>
__global__ __launch_bounds__(BLK) void dual_gemv_k(const E *__restrict__ A, size_t lda, int m, int n,
                                                   const float *__restrict__ xn, const float *__restrict__ xt,
                                                   float *__restrict__ partN, size_t strideN,
                                                   float *__restrict__ partT, size_t strideT,
                                                   int cols_per_chunk, const int *__restrict__ stop,
                                                   const float *__restrict__ inv_s, int m_load)
{
    if (stop != nullptr && *stop != 0) return;
    __shared__ float ldsT[DO_T ? 4 * MAXCW : 4];

    constexpr int TILE = BLK * VW * NJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int r_first = tile * TILE + tid * VW;
    const int c0 = chunk * cols_per_chunk;
    const int c1 = min(n, c0 + cols_per_chunk);

    float xtv[NJ][VW];
    float accN[NJ][VW];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            const int r = r_first + j * (BLK * VW) + k;
            accN[j][k] = 0.0f;
            xtv[j][k] = (DO_T && r < m) ? (ABS ? 1.0f : xt[r]) : 0.0f;
        }

    if constexpr (VW == 8 && DO_T) {
        // the bf16 step issues its loads from inline asm with its own vmcnt waits: make the compiler wait for the
        // x_T loads HERE (a use), or it places a vmcnt(0) for them at their first use inside the column loop
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int k = 0; k < VW; ++k) asm volatile("" : "+v"(xtv[j][k]));
    }

    float *ldsT_wave = ldsT + wave * (DO_T ? MAXCW : 1);
    int c = c0;
    int rofs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rofs[j] = 0;
    if ((tile + 1) * TILE <= m) {
        for (; c + KU <= c1; c += KU)
            step<E, VW, NJ, KU, DO_N, DO_T, ABS, NT, true>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
        for (; c < c1; ++c)
            step<E, VW, NJ, 1, DO_N, DO_T, ABS, NT, true>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
    } else if (VW > 1 && m_load > 0) {
        // the partial last tile, unguarded: m_load = the rows that may be loaded with whole vectors (m if it is a multiple
        // of VW, else roundup(m, VW) when the rows m .. lda - 1 are the library's own zeros)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = r_first + j * (BLK * VW);
            rofs[j] = (r + VW <= m_load) ? r : tile * TILE;
        }
        for (; c + KU <= c1; c += KU)
            step<E, VW, NJ, KU, DO_N, DO_T, ABS, NT, true, true>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
        for (; c < c1; ++c)
            step<E, VW, NJ, 1, DO_N, DO_T, ABS, NT, true, true>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
    } else {
        for (; c + KU <= c1; c += KU)
            step<E, VW, NJ, KU, DO_N, DO_T, ABS, NT, false>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
        for (; c < c1; ++c)
            step<E, VW, NJ, 1, DO_N, DO_T, ABS, NT, false>(A, lda, m, r_first, c, c - c0, xn, xtv, accN, ldsT_wave, lane, inv_s, rofs);
    }

    if constexpr (DO_N) {
        float *dst = partN + (size_t)chunk * strideN;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = r_first + j * (BLK * VW);
            if constexpr (VW >= 4) {
                if (r + VW <= m) {
#pragma unroll
                    for (int q = 0; q < VW; q += 4)
                        *reinterpret_cast<float4 *>(dst + r + q) =
                            make_float4(accN[j][q], accN[j][q + 1], accN[j][q + 2], accN[j][q + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < VW; ++k) if (r + k < m) dst[r + k] = accN[j][k];
                }
            } else {
                if (r < m) dst[r] = accN[j][0];
            }
        }
    }
    if constexpr (DO_T) {
        __syncthreads();
        float *dst = partT + (size_t)tile * strideT + c0;
        for (int t = tid; t < c1 - c0; t += BLK)
            dst[t] = (ldsT[t] + ldsT[MAXCW + t]) + (ldsT[2 * MAXCW + t] + ldsT[3 * MAXCW + t]);
    }
}

// Many small GEMVs in ONE launch (thip_lazy.hip: the deferred transform_ge calls of a composite operator, e.g. the 2000
// block products of ProbSOCPOpA::op / ::trans_op, socp.rs:77-130): blockIdx.z picks a descriptor.  The blocks are
// short (99 rows at BASELINE configs[2]) and start anywhere, so the unit of work is ONE WAVE: 128 rows (two per lane:
// l and l + 64) x a chunk of d.cpc columns, 8 columns = 16 guarded dword loads per lane in flight; the four waves of a
// workgroup take four consecutive column chunks of the same row tile and never synchronise.  A 99-row block keeps
// 99 / 128 of the lanes of EVERY wave busy (a 256-row tile would idle two waves of four and leave 3 KB per workgroup in
// flight instead of 16).  Partial sums: N -> part[chunk * nr + r]; T -> part[tile * nc + c] (8 columns reduced across
// the wave by the transpose-reduce butterfly, staged in LDS, stored coalesced).
constexpr int GROWS = 128;      // rows per wave tile
// DO_N && DO_T: both products of the SAME block from one read of it (the N and T calls of one stage of the loop land in
// the same record: SelfDualEmbed::op / trans_op issue a.trans_op and a.op back to back, solver.rs:122-125,146-149)
template <bool DO_N, bool DO_T>
__global__ __launch_bounds__(BLK) void grouped_gemv_k(const GroupDesc *__restrict__ tab)
{
    const GroupDesc d = tab[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x, chunk = blockIdx.y * 4 + wave;
    __shared__ float ldsT[DO_T ? 4 * MAXCW : 4];
    if (tile * GROWS >= d.nr || chunk * d.cpc >= d.nc) return;
    const int r0 = tile * GROWS + lane, r1 = r0 + 64;
    const bool ok0 = r0 < d.nr, ok1 = r1 < d.nr;
    const int c0 = chunk * d.cpc, c1 = min(d.nc, c0 + d.cpc);
    const float *__restrict__ A = d.A;
    const size_t lda = (size_t)d.nr;
    float xt0 = 0.0f, xt1 = 0.0f, acc0 = 0.0f, acc1 = 0.0f;
    if constexpr (DO_T) { xt0 = ok0 ? d.xt[r0] : 0.0f; xt1 = ok1 ? d.xt[r1] : 0.0f; }
    float *lw = ldsT + wave * (DO_T ? MAXCW : 1);
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        float a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float *col = A + (size_t)(c + u) * lda;
            a0[u] = ok0 ? __builtin_nontemporal_load(col + r0) : 0.0f;
            a1[u] = ok1 ? __builtin_nontemporal_load(col + r1) : 0.0f;
        }
        if constexpr (DO_N) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float xs = d.xn[c + u]; acc0 = fmaf(a0[u], xs, acc0); acc1 = fmaf(a1[u], xs, acc1); }
        }
        if constexpr (DO_T) {
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = fmaf(a0[u], xt0, a1[u] * xt1);
            const float r = multi_reduce<8>(p, lane);
            if ((lane & 7) == 0) lw[c - c0 + (lane >> 3)] = r;
        }
    }
    for (; c < c1; ++c) {
        const float *col = A + (size_t)c * lda;
        const float a0 = ok0 ? col[r0] : 0.0f, a1 = ok1 ? col[r1] : 0.0f;
        if constexpr (DO_N) { const float xs = d.xn[c]; acc0 = fmaf(a0, xs, acc0); acc1 = fmaf(a1, xs, acc1); }
        if constexpr (DO_T) { const float r = wave_sum(fmaf(a0, xt0, a1 * xt1)); if (lane == 0) lw[c - c0] = r; }
    }
    if constexpr (DO_N) {
        float *dst = d.partN + (size_t)chunk * d.nr;
        if (ok0) dst[r0] = acc0;
        if (ok1) dst[r1] = acc1;
    }
    if constexpr (DO_T) {
        // this wave's LDS writes are visible to itself after the wait (no cross-wave sharing)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        float *dst = d.partT + (size_t)tile * d.nc + c0;
        for (int t = lane; t < c1 - c0; t += 64) dst[t] = lw[t];
    }
}

// second stage: y[i] = alpha * sum_k part[k*stride + i] + beta * y[i]
__global__ void finalize_k(size_t n, const float *__restrict__ part, int np, size_t stride, float alpha, float beta,
                           float *__restrict__ y, const int *__restrict__ stop)
{
    if (stop != nullptr && *stop != 0) return;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLK) {
        double sd = 0.0;                       // the second stage accumulates in f64 (free; keeps the f32 floor down)
        for (int k = 0; k < np; ++k) sd += (double)part[(size_t)k * stride + i];
        const float s = (float)sd;
        y[i] = (beta == 0.0f) ? alpha * s : alpha * s + beta * y[i];
    }
}

// y = alpha * S x + beta * y, S packed upper by columns (floatgeneric.rs:356-376).  One wave per row.
__global__ void spmv_k(int n, float alpha, const float *__restrict__ sp, const float *__restrict__ x, float beta,
                       float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.0f;
    // c <= r: element (c, r) of column r: idx = r(r+1)/2 + c  (contiguous)
    const size_t base = (size_t)r * (r + 1) / 2;
    for (int c = lane; c <= r; c += 64) s = fmaf(sp[base + c], x[c], s);
    // c > r: element (r, c) of column c: idx = c(c+1)/2 + r
    for (int c = r + 1 + lane; c < n; c += 64) s = fmaf(sp[(size_t)c * (c + 1) / 2 + r], x[c], s);
    s = wave_sum(s);
    if (lane == 0) y[r] = (beta == 0.0f) ? alpha * s : alpha * s + beta * y[r];
}

// SymPack absadd (matop.rs:119-136): y[i] += sum_j |S(i,j)| over the full symmetric matrix
__global__ void sp_absadd_k(int n, const float *__restrict__ sp, float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.0f;
    const size_t base = (size_t)r * (r + 1) / 2;
    for (int c = lane; c <= r; c += 64) s += fabsf(sp[base + c]);
    for (int c = r + 1 + lane; c < n; c += 64) s += fabsf(sp[(size_t)c * (c + 1) / 2 + r]);
    s = wave_sum(s);
    if (lane == 0) y[r] = y[r] + s;
}

struct Plan {
    int vw, nj, ku, nt;
    int tiles, chunks, cols_per_chunk;
    int m_load = 0;             // > 0: rows of the partial last tile that whole vectors may load (dual_gemv_k)
    size_t strideN, strideT;
};

static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// tuning knobs (experiments only): THIP_GEMV_NJ = 1|2|4, THIP_GEMV_BLOCKS = target grid size, THIP_GEMV_NT = 0|1
static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// vw: rows per lane and load -- 4 (f32, 16-byte loads), 8 (bf16, 16-byte loads) or 1 (unaligned fallback)
Plan make_plan(size_t n_row, size_t n_col, int vw, const GemvHint *hint = nullptr)
{
    const bool vec_ok = vw > 1;
    static const int env_nj = env_int("THIP_GEMV_NJ", 0);
    static const int env_blocks = env_int("THIP_GEMV_BLOCKS", 0);
    static const int env_nt = env_int("THIP_GEMV_NT", -1);
    Plan p;
    p.vw = vw;
    if (vec_ok) {
        // measured on MI355X (gpurun_out/sweep_*.txt): 1 float4 row group per lane, 8 columns in
        // flight, is as fast as taller tiles once the grid is fine enough, and keeps partial sums small
        p.nj = 1; p.ku = 8;
        int nj = env_nj;
        if (hint && hint->nj > 0) nj = hint->nj;
        if (nj == 1) { p.nj = 1; p.ku = 8; }
        if (nj == 2) { p.nj = 2; p.ku = 4; }
        if (nj == 4) { p.nj = 4; p.ku = 2; }
    } else {
        if (n_row >= (size_t)BLK * 4 * 4) { p.nj = 4; p.ku = 4; }
        else { p.nj = 1; p.ku = 8; }
    }
    // A is read exactly once per launch: non-temporal loads (+6-8 % on MI355X: 6.0 -> 6.5 TB/s)
    p.nt = env_nt >= 0 ? env_nt : 1;
    const size_t tile = (size_t)BLK * p.vw * p.nj;
    p.tiles = (int)((n_row + tile - 1) / tile);
    // 8 to 32 workgroups per CU (0.2 MB of A each at the low end): a fine grid shortens the ramp and the tail of
    // the last round of workgroups (MI355X sweep: 8192 beat 2048 by 2-4 % at 20 GB, 4096 beat 1024 by 30 % at
    // 0.8 GB); the partial-sum traffic (chunks x m + tiles x n floats) stays at 1-2 % of A
    int target_blocks = (int)((double)n_row * (double)n_col * (vw == 8 ? 2.0 : 4.0) / 2.0e5);
    if (target_blocks < 2048) target_blocks = 2048;
    if (target_blocks > 8192) target_blocks = 8192;
    if (env_blocks > 0) target_blocks = env_blocks;
    if (hint && hint->target_blocks > 0) target_blocks = hint->target_blocks;
    int chunks = target_blocks / p.tiles;
    if (chunks < 1) chunks = 1;
    int cpc = (int)((n_col + chunks - 1) / chunks);
    // at least 32 columns per chunk (16 below 64 MB, where more workgroups matter more than fewer partials:
    // 4000 x 2000 measured 83 -> 75 us per iteration)
    static const int env_mincpc = env_int("THIP_GEMV_MINCPC", 0);
    const int mincpc = env_mincpc > 0 ? env_mincpc : (((double)n_row * (double)n_col * 4.0 < 64.0e6) ? 16 : 32);
    if (cpc < mincpc) cpc = mincpc;
    if (cpc > MAXCW) cpc = MAXCW;
    cpc = (int)round_up(cpc, 8);
    if (cpc > MAXCW) cpc = MAXCW;
    p.cols_per_chunk = cpc;
    p.chunks = (int)((n_col + cpc - 1) / cpc);
    p.strideN = round_up(n_row, 4);
    p.strideT = round_up(n_col, 4);
    return p;
}

template <typename E, bool DO_N, bool DO_T, bool ABS>
void launch_cfg(const Plan &p, hipStream_t st, const E *A, size_t lda, int m, int n, const float *xn,
                const float *xt, float *partN, float *partT, const int *stop, const float *inv_s)
{
    dim3 g(p.tiles, p.chunks), b(BLK);
    constexpr int VV = sizeof(E) == 2 ? 8 : 4;      // rows per 16-byte load
#define THIP_GEMV_LAUNCH(VW, NJ, KU)                                                                          \
    do {                                                                                                      \
        if (p.nt)                                                                                             \
            hipLaunchKernelGGL((dual_gemv_k<E, VW, NJ, KU, DO_N, DO_T, ABS, true>), g, b, 0, st, A, lda, m, n, xn, xt, \
                               partN, p.strideN, partT, p.strideT, p.cols_per_chunk, stop, inv_s, p.m_load);  \
        else                                                                                                  \
            hipLaunchKernelGGL((dual_gemv_k<E, VW, NJ, KU, DO_N, DO_T, ABS, false>), g, b, 0, st, A, lda, m, n, xn, xt, \
                               partN, p.strideN, partT, p.strideT, p.cols_per_chunk, stop, inv_s, p.m_load);  \
    } while (0)
    if (p.vw == VV) {
        if (p.nj == 4) THIP_GEMV_LAUNCH(VV, 4, 2);
        else if (p.nj == 2) THIP_GEMV_LAUNCH(VV, 2, 4);
        else THIP_GEMV_LAUNCH(VV, 1, 8);
    } else {
        if (p.nj == 4) THIP_GEMV_LAUNCH(1, 4, 4);
        else THIP_GEMV_LAUNCH(1, 1, 8);
    }
#undef THIP_GEMV_LAUNCH
}

template <typename E>
void launch_any(const Plan &p, hipStream_t st, const E *mat, size_t lda, int m, int n, const float *xn, const float *xt,
                bool do_n, bool do_t, bool abs_mode, float *partN, float *partT, const int *stop_flag,
                const float *inv_s = nullptr)
{
    if (abs_mode) {
        if (do_n && do_t) launch_cfg<E, true, true, true>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
        else if (do_n)    launch_cfg<E, true, false, true>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
        else              launch_cfg<E, false, true, true>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
    } else {
        if (do_n && do_t) launch_cfg<E, true, true, false>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
        else if (do_n)    launch_cfg<E, true, false, false>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
        else              launch_cfg<E, false, true, false>(p, st, mat, lda, m, n, xn, xt, partN, partT, stop_flag, inv_s);
    }
}

// f32 (ld = n_row) -> bf16 round-to-nearest-even (ld16 >= n_row, the padding rows are written as zeros)
__global__ void to_bf16_k(size_t n_row, size_t n_col, const float *__restrict__ src, bf16raw *__restrict__ dst, size_t ld16)
{
    const size_t total = ld16 * n_col;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < total; i += (size_t)gridDim.x * BLK) {
        const size_t c = i / ld16, r = i - c * ld16;
        unsigned out = 0;
        if (r < n_row) {
            const unsigned b = __float_as_uint(src[c * n_row + r]);
            if ((b & 0x7f800000u) == 0x7f800000u) out = (b >> 16) | ((b & 0xffffu) ? 0x40u : 0u);   // inf / nan
            else out = (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
        }
        dst[i] = (bf16raw)out;
    }
}

// per column: inv_s[c] = 1 / s_c with s_c the power of two that brings max_r |a(r, c)| into [2^13, 2^14) (f16 overflows
// at 65504 = 2^16; an all-zero column keeps s_c = 1).  One block per column.
__global__ __launch_bounds__(BLK) void f16_scale_k(size_t n_row, const float *__restrict__ src, float *__restrict__ inv_s)
{
    __shared__ float sh[16];
    const float *col = src + (size_t)blockIdx.x * n_row;
    float mx = 0.0f;
    for (size_t r = threadIdx.x; r < n_row; r += BLK) {
        const float v = fabsf(col[r]);
        if (v > mx && v <= 3.4028235e38f) mx = v;          // inf / nan do not set the scale
    }
    mx = -block_min(-mx, sh);
    if (threadIdx.x == 0) {
        float inv = 1.0f;
        if (mx > 0.0f) {
            int e;
            (void)frexpf(mx, &e);                          // mx = f * 2^e, f in [0.5, 1)  =>  mx * 2^(14 - e) in [2^13, 2^14)
            inv = ldexpf(1.0f, e - 14);
        }
        inv_s[blockIdx.x] = inv;
    }
}

// stored(r, c) = f16(a(r, c) / inv_s[c]) (round to nearest even), ld16 >= n_row, padding rows zero
__global__ void to_f16_k(size_t n_row, size_t n_col, const float *__restrict__ src, const float *__restrict__ inv_s,
                         f16elt *__restrict__ dst, size_t ld16)
{
    const size_t total = ld16 * n_col;
    for (size_t i = blockIdx.x * (size_t)BLK + threadIdx.x; i < total; i += (size_t)gridDim.x * BLK) {
        const size_t c = i / ld16, r = i - c * ld16;
        dst[i] = (r < n_row) ? (f16elt)(src[c * n_row + r] / inv_s[c]) : (f16elt)0.0f;
    }
}

}  // namespace

namespace thip {

const GemvHint *gemv_candidates(int *count)
{
    // measured on MI355X: fine grids of 1-row-group tiles win at 100k x 50k and at the 0.8 GB LP,
    // tall tiles with ~1k workgroups win for short-and-wide row shards
    static const GemvHint c[] = { {1, 8192}, {1, 4096}, {2, 8192}, {2, 2048}, {4, 8192}, {4, 4096}, {4, 2048}, {4, 1024}, {4, 768} };
    *count = (int)(sizeof(c) / sizeof(c[0]));
    return c;
}

size_t dual_gemv_scratch_floats(size_t n_row, size_t n_col)
{
    size_t best = 0;
    int nc = 0;
    const GemvHint *c = gemv_candidates(&nc);
    static const int vws[3] = { 4, 1, 8 };
    for (int i = -1; i < nc; ++i)
        for (int v = 0; v < 3; ++v) {
            const Plan p = make_plan(n_row, n_col, vws[v], i < 0 ? nullptr : &c[i]);
            const size_t f = (size_t)p.chunks * p.strideN + (size_t)p.tiles * p.strideT;
            if (f > best) best = f;
        }
    return best + 64;
}

int dual_gemv_partials(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
                       const float *xn, const float *xt, bool do_n, bool do_t, bool abs_mode,
                       float *scratch_base, size_t scratch_floats, GemvPartials *out, const int *stop_flag,
                       const GemvHint *hint, int a_kind, const float *inv_s, bool pad_zero)
{
    if (n_row == 0 || n_col == 0 || (!do_n && !do_t)) {
        out->partN = out->partT = nullptr; out->nN = out->nT = 0; out->strideN = out->strideT = 0;
        return 0;
    }
    if (n_row > 0x7fffffffull || n_col > 0x7fffffffull) return fail(THIP_E_INVALID, "matrix dimension > 2^31", __FILE__, __LINE__);
    const bool bf16 = a_kind == THIP_A_BF16, f16 = a_kind == THIP_A_F16;
    if (f16 && inv_s == nullptr) return fail(THIP_E_INVALID, "f16 storage needs the per-column scales", __FILE__, __LINE__);
    const bool vec_ok = (((uintptr_t)mat & 15u) == 0) && (lda % ((bf16 || f16) ? 8 : 4) == 0);
    Plan p = make_plan(n_row, n_col, vec_ok ? ((bf16 || f16) ? 8 : 4) : 1, hint);
    {
        // the partial last row tile runs unguarded when every vector it needs exists: m a multiple of the vector width,
        // or the rows m .. lda - 1 are the library's own zeros.  THIP_GEMV_CLAMP=0: the guarded form
        static const int clamp_on = getenv("THIP_GEMV_CLAMP") ? atoi(getenv("THIP_GEMV_CLAMP")) : 1;
        const size_t m_up = round_up(n_row, (size_t)p.vw);
        if (clamp_on && p.vw > 1 && (n_row % p.vw == 0 || (pad_zero && lda >= m_up))) p.m_load = (int)m_up;
    }
    const size_t needN = do_n ? (size_t)p.chunks * p.strideN : 0;
    const size_t needT = do_t ? (size_t)p.tiles * p.strideT : 0;
    if (needN + needT > scratch_floats) return fail(THIP_E_WORK, "gemv scratch too small", __FILE__, __LINE__);
    float *partN = scratch_base;
    float *partT = scratch_base + needN;
    const int m = (int)n_row, n = (int)n_col;
    if (bf16) launch_any(p, st, (const bf16raw *)mat, lda, m, n, xn, xt, do_n, do_t, abs_mode, partN, partT, stop_flag);
    else if (f16) launch_any(p, st, (const f16elt *)mat, lda, m, n, xn, xt, do_n, do_t, abs_mode, partN, partT, stop_flag, inv_s);
    else      launch_any(p, st, (const float *)mat, lda, m, n, xn, xt, do_n, do_t, abs_mode, partN, partT, stop_flag);
    THIP_LAUNCH_CHECK();
    out->partN = do_n ? partN : nullptr; out->nN = do_n ? p.chunks : 0; out->strideN = p.strideN;
    out->partT = do_t ? partT : nullptr; out->nT = do_t ? p.tiles : 0;  out->strideT = p.strideT;
    return 0;
}

// The product over columns [col0, col1) only, as one launch of its own: the tile height of the plan for (n_row x n_col)
// under `hint`, its own column chunks starting at col0.  Partial sums go where a consumer of the WHOLE product expects
// them: N partials into chunk rows chunk_row0 .. of partN (so that the launches over [0, c) and [c, n_col) fill
// consecutive chunk rows and the consumer sums them all), T partials into columns col0 .. col1 of every tile row.
// *chunks_used = chunk rows this launch filled.  `out` describes the layout (nN = chunk_row0 + *chunks_used).
// col0 only needs the alignment of the vector loads (a multiple of 4 floats; 8 for 16-bit storage).
int dual_gemv_partials_cols(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
                            const float *xn, const float *xt, bool do_n, bool do_t,
                            float *scratch_base, size_t scratch_floats, GemvPartials *out, const int *stop_flag,
                            const GemvHint *hint, int a_kind, const float *inv_s, bool pad_zero,
                            size_t col0, size_t col1, int chunk_row0, int max_chunk_rows, int *chunks_used)
{
    if (chunks_used) *chunks_used = 0;
    if (n_row == 0 || n_col == 0 || (!do_n && !do_t) || col1 <= col0) {
        out->partN = out->partT = nullptr; out->nN = out->nT = 0; out->strideN = out->strideT = 0;
        return 0;
    }
    if (n_row > 0x7fffffffull || n_col > 0x7fffffffull || col1 > n_col) return fail(THIP_E_INVALID, "bad column range", __FILE__, __LINE__);
    const bool bf16 = a_kind == THIP_A_BF16, f16 = a_kind == THIP_A_F16;
    if (f16 && inv_s == nullptr) return fail(THIP_E_INVALID, "f16 storage needs the per-column scales", __FILE__, __LINE__);
    const size_t esz = (bf16 || f16) ? 2 : 4;
    const char *base = (const char *)mat + col0 * lda * esz;
    const bool vec_ok = (((uintptr_t)mat & 15u) == 0) && (((uintptr_t)base & 15u) == 0) && (lda % ((bf16 || f16) ? 8 : 4) == 0);
    // tiling of the range's own shape (its columns decide the chunking); strides are those of the whole matrix
    Plan p = make_plan(n_row, col1 - col0, vec_ok ? ((bf16 || f16) ? 8 : 4) : 1, hint);
    p.strideN = round_up(n_row, 4);
    p.strideT = round_up(n_col, 4);
    {
        static const int clamp_on = getenv("THIP_GEMV_CLAMP") ? atoi(getenv("THIP_GEMV_CLAMP")) : 1;
        const size_t m_up = round_up(n_row, (size_t)p.vw);
        if (clamp_on && p.vw > 1 && (n_row % p.vw == 0 || (pad_zero && lda >= m_up))) p.m_load = (int)m_up;
    }
    if (chunk_row0 + p.chunks > max_chunk_rows) return fail(THIP_E_WORK, "gemv scratch: too many chunk rows", __FILE__, __LINE__);
    const size_t needN = do_n ? (size_t)max_chunk_rows * p.strideN : 0;
    const size_t needT = do_t ? (size_t)p.tiles * p.strideT : 0;
    if (needN + needT > scratch_floats) return fail(THIP_E_WORK, "gemv scratch too small", __FILE__, __LINE__);
    float *partN = scratch_base;
    float *partT = scratch_base + needN;
    out->partN = do_n ? partN : nullptr; out->nN = do_n ? chunk_row0 + p.chunks : 0; out->strideN = p.strideN;
    out->partT = do_t ? partT : nullptr; out->nT = do_t ? p.tiles : 0;  out->strideT = p.strideT;
    if (chunks_used) *chunks_used = p.chunks;
    const int m = (int)n_row, n = (int)(col1 - col0);
    float *pn = partN + (size_t)chunk_row0 * p.strideN;
    float *pt = partT + col0;
    const float *xnr = xn ? xn + col0 : nullptr;
    if (bf16) launch_any(p, st, (const bf16raw *)base, lda, m, n, xnr, xt, do_n, do_t, false, pn, pt, stop_flag);
    else if (f16) launch_any(p, st, (const f16elt *)base, lda, m, n, xnr, xt, do_n, do_t, false, pn, pt, stop_flag, inv_s + col0);
    else      launch_any(p, st, (const float *)base, lda, m, n, xnr, xt, do_n, do_t, false, pn, pt, stop_flag);
    THIP_LAUNCH_CHECK();
    return 0;
}

// chunk rows a launch over `cols` columns of an n_row-row matrix will fill under `hint` (and its tile count)
int dual_gemv_chunk_rows(size_t n_row, size_t cols, bool vec_ok, int a_kind, const GemvHint *hint, int *tiles)
{
    const bool h16 = a_kind == THIP_A_BF16 || a_kind == THIP_A_F16;
    const Plan p = make_plan(n_row, cols, vec_ok ? (h16 ? 8 : 4) : 1, hint);
    if (tiles) *tiles = p.tiles;
    return p.chunks;
}

// where dual_gemv_partials (f32, default plan) will leave its partial sums for this shape, without launching anything
int dual_gemv_partials_geometry(size_t n_row, size_t n_col, const void *mat, size_t lda, bool do_n, bool do_t,
                                float *scratch_base, GemvPartials *out)
{
    const bool vec_ok = (((uintptr_t)mat & 15u) == 0) && (lda % 4 == 0);
    const Plan p = make_plan(n_row, n_col, vec_ok ? 4 : 1, nullptr);
    const size_t needN = do_n ? (size_t)p.chunks * p.strideN : 0;
    float *partN = scratch_base, *partT = scratch_base + needN;
    out->partN = do_n ? partN : nullptr; out->nN = do_n ? p.chunks : 0; out->strideN = p.strideN;
    out->partT = do_t ? partT : nullptr; out->nT = do_t ? p.tiles : 0;  out->strideT = p.strideT;
    return 0;
}

int dual_gemv_cols_per_chunk(size_t n_row, size_t n_col, const void *mat, size_t lda, const GemvHint *hint, int a_kind,
                             int *chunks)
{
    const bool h16 = a_kind == THIP_A_BF16 || a_kind == THIP_A_F16;
    const bool vec_ok = (((uintptr_t)mat & 15u) == 0) && (lda % (h16 ? 8 : 4) == 0);
    const Plan p = make_plan(n_row, n_col, vec_ok ? (h16 ? 8 : 4) : 1, hint);
    if (chunks) *chunks = p.chunks;
    return p.cols_per_chunk;
}

int grouped_gemv(hipStream_t st, const GroupDesc *dev_tab, int n_desc, int max_tiles, int max_chunks, int mode)
{
    if (n_desc <= 0) return 0;
    dim3 g(max_tiles, max_chunks, n_desc);
    if (mode == 2)      hipLaunchKernelGGL((grouped_gemv_k<true, true>), g, dim3(BLK), 0, st, dev_tab);
    else if (mode == 1) hipLaunchKernelGGL((grouped_gemv_k<false, true>), g, dim3(BLK), 0, st, dev_tab);
    else                hipLaunchKernelGGL((grouped_gemv_k<true, false>), g, dim3(BLK), 0, st, dev_tab);
    THIP_LAUNCH_CHECK();
    return 0;
}

int finalize_partials(hipStream_t st, size_t n, const float *part, int np, size_t stride, float alpha, float beta,
                      float *y, const int *stop)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(finalize_k, dim3(grid_for(n, BLK, 2048)), dim3(BLK), 0, st, n, part, np, stride, alpha, beta, y, stop);
    THIP_LAUNCH_CHECK();
    return 0;
}

int to_bf16(hipStream_t st, size_t n_row, size_t n_col, const float *src, uint16_t *dst, size_t ld16)
{
    if (ld16 < n_row) return fail(THIP_E_INVALID, "ld16 < n_row", __FILE__, __LINE__);
    if (ld16 * n_col == 0) return 0;
    hipLaunchKernelGGL(to_bf16_k, dim3(grid_for(ld16 * n_col, BLK, 16384)), dim3(BLK), 0, st, n_row, n_col, src, dst, ld16);
    THIP_LAUNCH_CHECK();
    return 0;
}

int to_f16(hipStream_t st, size_t n_row, size_t n_col, const float *src, uint16_t *dst, size_t ld16, float *inv_s)
{
    if (ld16 < n_row) return fail(THIP_E_INVALID, "ld16 < n_row", __FILE__, __LINE__);
    if (n_col == 0) return 0;
    hipLaunchKernelGGL(f16_scale_k, dim3((unsigned)n_col), dim3(BLK), 0, st, n_row, src, inv_s);
    hipLaunchKernelGGL(to_f16_k, dim3(grid_for(ld16 * n_col, BLK, 16384)), dim3(BLK), 0, st, n_row, n_col, src, inv_s,
                       reinterpret_cast<f16elt *>(dst), ld16);
    THIP_LAUNCH_CHECK();
    return 0;
}

int dual_gemv(hipStream_t st, size_t n_row, size_t n_col, const void *mat, size_t lda,
              const float *xn, float alphaN, float betaN, float *outN,
              const float *xt, float alphaT, float betaT, float *outT,
              bool abs_mode, const int *stop_flag, int a_kind, const float *inv_s)
{
    const bool do_n = outN != nullptr, do_t = outT != nullptr;
    float *scr = nullptr;
    const size_t need = dual_gemv_scratch_floats(n_row, n_col);
    THIP_RC(scratch(need, &scr));
    GemvPartials gp;
    THIP_RC(dual_gemv_partials(st, n_row, n_col, mat, lda, xn, xt, do_n, do_t, abs_mode, scr, need, &gp, stop_flag, nullptr, a_kind, inv_s));
    if (do_n && n_row)
        hipLaunchKernelGGL(finalize_k, dim3(grid_for(n_row, BLK, 2048)), dim3(BLK), 0, st, n_row, gp.partN, gp.nN,
                           gp.strideN, alphaN, betaN, outN, stop_flag);
    if (do_t && n_col)
        hipLaunchKernelGGL(finalize_k, dim3(grid_for(n_col, BLK, 2048)), dim3(BLK), 0, st, n_col, gp.partT, gp.nT,
                           gp.strideT, alphaT, betaT, outT, stop_flag);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace thip

extern "C" {

int thip_transform_ge(int transpose, size_t n_row, size_t n_col, float alpha, const float *mat, const float *x,
                      float beta, float *y)
{
    THIP_NEED_INIT_NOFLUSH();
    // zero-sized operands: MatOp::op_impl never calls transform_ge then (matop.rs:79-85); be lenient anyway
    const size_t ylen = transpose ? n_col : n_row;
    if (ylen == 0) return 0;
    if (n_row == 0 || n_col == 0) return thip_scale(ylen, beta, y);
    // small products are deferred and batched (thip_lazy.hip); anything else runs now, after what is pending
    int deferred = 0;
    THIP_RC(lazy_push(transpose, n_row, n_col, alpha, mat, x, beta, y, &deferred));
    if (deferred) return 0;
    if (transpose)
        return dual_gemv(ctx().stream, n_row, n_col, mat, n_row, nullptr, 0.f, 0.f, nullptr, x, alpha, beta, y, false, nullptr);
    return dual_gemv(ctx().stream, n_row, n_col, mat, n_row, x, alpha, beta, y, nullptr, 0.f, 0.f, nullptr, false, nullptr);
}

int thip_to_bf16(size_t n_row, size_t n_col, const float *mat, uint16_t *mat16, size_t ld16)
{
    THIP_NEED_INIT();
    return to_bf16(ctx().stream, n_row, n_col, mat, mat16, ld16);
}

int thip_transform_ge_bf16(int transpose, size_t n_row, size_t n_col, float alpha, const uint16_t *mat16, size_t ld16,
                           const float *x, float beta, float *y)
{
    THIP_NEED_INIT();
    const size_t ylen = transpose ? n_col : n_row;
    if (ylen == 0) return 0;
    if (n_row == 0 || n_col == 0) return thip_scale(ylen, beta, y);
    if (ld16 < n_row) return fail(THIP_E_INVALID, "ld16 < n_row", __FILE__, __LINE__);
    if (transpose)
        return dual_gemv(ctx().stream, n_row, n_col, mat16, ld16, nullptr, 0.f, 0.f, nullptr, x, alpha, beta, y, false, nullptr, THIP_A_BF16);
    return dual_gemv(ctx().stream, n_row, n_col, mat16, ld16, x, alpha, beta, y, nullptr, 0.f, 0.f, nullptr, false, nullptr, THIP_A_BF16);
}

int thip_to_f16(size_t n_row, size_t n_col, const float *mat, uint16_t *mat16, size_t ld16, float *inv_scale)
{
    THIP_NEED_INIT();
    if (!inv_scale) return fail(THIP_E_INVALID, "inv_scale == NULL", __FILE__, __LINE__);
    return to_f16(ctx().stream, n_row, n_col, mat, mat16, ld16, inv_scale);
}

int thip_transform_ge_f16(int transpose, size_t n_row, size_t n_col, float alpha, const uint16_t *mat16, size_t ld16,
                          const float *inv_scale, const float *x, float beta, float *y)
{
    THIP_NEED_INIT();
    const size_t ylen = transpose ? n_col : n_row;
    if (ylen == 0) return 0;
    if (n_row == 0 || n_col == 0) return thip_scale(ylen, beta, y);
    if (ld16 < n_row) return fail(THIP_E_INVALID, "ld16 < n_row", __FILE__, __LINE__);
    if (transpose)
        return dual_gemv(ctx().stream, n_row, n_col, mat16, ld16, nullptr, 0.f, 0.f, nullptr, x, alpha, beta, y, false, nullptr, THIP_A_F16, inv_scale);
    return dual_gemv(ctx().stream, n_row, n_col, mat16, ld16, x, alpha, beta, y, nullptr, 0.f, 0.f, nullptr, false, nullptr, THIP_A_F16, inv_scale);
}

int thip_absadd_cols(size_t n_row, size_t n_col, const float *mat, float *tau)
{
    THIP_NEED_INIT();
    if (n_row == 0 || n_col == 0) return 0;
    return dual_gemv(ctx().stream, n_row, n_col, mat, n_row, nullptr, 0.f, 0.f, nullptr, nullptr, 1.0f, 1.0f, tau, true, nullptr);
}

int thip_absadd_rows(size_t n_row, size_t n_col, const float *mat, float *sigma)
{
    THIP_NEED_INIT();
    if (n_row == 0 || n_col == 0) return 0;
    return dual_gemv(ctx().stream, n_row, n_col, mat, n_row, nullptr, 1.0f, 1.0f, sigma, nullptr, 0.f, 0.f, nullptr, true, nullptr);
}

int thip_transform_sp(size_t n, float alpha, const float *mat, const float *x, float beta, float *y)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(spmv_k, dim3((unsigned)((n + 3) / 4)), dim3(BLK), 0, ctx().stream, (int)n, alpha, mat, x, beta, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

int thip_absadd_sympack(size_t n, const float *mat, float *y)
{
    THIP_NEED_INIT();
    if (n == 0) return 0;
    hipLaunchKernelGGL(sp_absadd_k, dim3((unsigned)((n + 3) / 4)), dim3(BLK), 0, ctx().stream, (int)n, mat, y);
    THIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
