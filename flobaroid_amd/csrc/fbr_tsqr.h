// fbr_tsqr.h -- blocked Householder TSQR on gfx950 (fp64, MFMA trailing update).
//
// R^T R = A^T A for a tall row stream A (rows = samples x N_OUT, columns = [Y | rhs], padded to a
// multiple of 16), WITHOUT forming A^T A (no squaring of the condition number):
//
//   level 0  every workgroup w owns a private upper-triangular R_w (global memory) and folds its share of the
//            MB-row blocks (MB = 16 SUB <= 64) into it with a triangular-pentagonal Householder QR ("TPQRT": QR of
//            [R_w ; B], only R_w's panel rows and the dense block take part).  The block lives in the VGPRs of the
//            workgroup's 8 waves for the whole fold (see fbr_tsqr_stream) -- only R_w is streamed:
//              panel (16 columns): owner wave, all in registers (DPP / permlane cross-lane ops), T built in-chain
//              trailing update:    W = T^T (R_rows + V^T C),  R_rows -= W,  C -= V W on v_mfma_f64_16x16x4_f64
//            Panels are software-pipelined ACROSS the waves: the owner of panel q+1 updates its tile first and
//            factorises it while the other waves are still applying panel q (LDS flags, no workgroup barrier).
//   level 1+ binary tree over the R_w (one launch per level; the partner's R is folded in MB-row chunks, panels
//            left of a chunk's first non-zero column are skipped).  Across ranks the same merge runs on R factors
//            exchanged over xGMI (flobaroid_amd/dist.py).
//
// Everything is deterministic (fixed block -> workgroup assignment, fixed tree, fixed evaluation order per wave).
#pragma once
#include <hip/hip_runtime.h>
#include "fbr_tsqr_work.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#ifndef FBR_TSQR_SUB2
#define FBR_TSQR_SUB2 6           // 16-row sub-blocks per fold of the two-tiles-per-wave kernels (n <= 256): 96-row blocks
#endif
#ifndef FBR_TSQR_SUB3
#define FBR_TSQR_SUB3 4           // three tiles per wave (n <= 384)
#endif
#ifndef FBR_TSQR_SUB4
#define FBR_TSQR_SUB4 4           // four tiles per wave (n <= 512: WALK-MAN's 481 columns)
#endif
#ifndef FBR_TSQR_INPLACE_MAX_TPW
#define FBR_TSQR_INPLACE_MAX_TPW 2  // panels are factorised in place in the wave's own tile registers up to this many tiles per wave
#endif
#define FBR_TSQR_RING_MAX 6        // published panels (V, T) kept in the LDS: how far the waves may drift apart (fewer for tall blocks)
// (128-row blocks: 5 slots fit the 160 KiB; four-wave workgroups, two per CU, have 80 KiB each: 3 slots -- enough for 4 waves)
// R rows through REGISTERS instead of LDS tiles (fbr_tsqr_stream, RREG): the four-wave workgroups that cover more than 20 column tiles
// (two 32-row folds per CU for WALK-MAN's 31 tiles) have no room for one LDS tile per column tile in their 80 KiB
template <int TPW, int W> __host__ __device__ constexpr bool fbr_tsqr_rreg() { return W < FBR_TSQR_WAVES && TPW > 5; }
template <int SUB, int W = FBR_TSQR_WAVES, bool RREG = false> __host__ __device__ constexpr int fbr_tsqr_ring()
{
    return RREG ? FBR_TSQR_RING_MAX : (W < FBR_TSQR_WAVES ? 3 : (SUB > 6 ? 5 : FBR_TSQR_RING_MAX));
}
#define FBR_TSQR_MAXN 768          // widest supported factor (columns incl. rhs, padded to 16)
#define FBR_TSQR_SPIN_LIMIT (1 << 20)  // ~50 ms of polling: far beyond any legitimate wait (a fold tail is ~0.1 ms)

typedef double fbr_td4 __attribute__((ext_vector_type(4)));

static thread_local std::string g_tsqr_err;
static inline const char *fbr_tsqr_error() { return g_tsqr_err.c_str(); }

// Row order of a chunk handed to the factorisation.  rows == 0: as given (sample-major stack).  rows > 0: the chunk holds `group`
// samples of `rows` regressor rows each and is stacked by regressor row: chunk row r * group + s is row r of sample s, and a
// block of rows is zero left of first_col[r] (device table) of the regressor rows it spans.  rows < 0: as given, and the input is
// upper triangular (a factor handed to fbr_tsqr_merge): the block that starts at row r0 is zero left of column r0.
struct FbrTsqrRowOrder {
    const int *first_col = nullptr;
    int rows = 0;
    long group = 0;
    // > 0: the chunk is stored COLUMN-major, element (row o, column c) at A[c * colmajor_ld + o] -- the layout the one-lane-per-sample
    // writers produce (fbr_kinid.h fbr_kinwrite_kernel: the 64 samples of a wave are 64 consecutive rows of one column = 512 contiguous
    // bytes per store instruction).  0: row-major, leading dimension n.
    long colmajor_ld = 0;
};
// input row (sample-major) of chunk row o
__device__ __forceinline__ long fbr_tsqr_in_row(long o, int rows, long group)
{
    if (rows <= 0) return o;
    const long r = o / group;
    return (o - r * group) * rows + r;
}

// A[r][c] (ld) = w[r] * [Y | rhs][r][c], zero in the padding columns / rows
// cols (optional, device): gather columns cols[0..P) of a Y with leading dimension ldy (cols[c] < 0: zero column)
__global__ __launch_bounds__(256) void fbr_tsqr_pack_kernel(long M, long Mpad, int P, int k, int ld,
                                                             const double *__restrict__ Y, int ldy, const int *__restrict__ cols,
                                                             const double *__restrict__ rhs, const double *__restrict__ w,
                                                             double *__restrict__ A, int orows, long ogroup)
{
    const long total = Mpad * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long o = i / ld;
        const int c = (int)(i - o * ld);
        double v = 0.0;
        if (o < M) {
            const long r = fbr_tsqr_in_row(o, orows, ogroup);
            if (c < P) {
                const int sc = cols ? cols[c] : c;  // (negative: a column the source does not have -- embedded group factors)
                v = sc >= 0 ? Y[r * ldy + sc] : 0.0;
            }
            else if (c < P + k)
                v = rhs[r * k + (c - P)];
            if (w) v *= w[r];
        }
        A[i] = v;
    }
}

// columns [P, ld) of rows < M: rhs then zeros; rows M..Mpad: all zeros  (completes a chunk whose first P columns were
// written in place by the regressor kernel)
__global__ __launch_bounds__(256) void fbr_tsqr_tail_kernel(long M, long Mpad, int P, int k, int ld, const double *__restrict__ rhs, double *__restrict__ A,
                                                             int orows, long ogroup)
{
    const int tw = ld - P;
    const long t1 = M * tw, total = t1 + (Mpad - M) * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i < t1) {
            const long r = i / tw;
            const int c = (int)(i - r * tw);
            A[r * ld + P + c] = (c < k) ? rhs[fbr_tsqr_in_row(r, orows, ogroup) * k + c] : 0.0;
        } else {
            A[M * ld + (i - t1)] = 0.0;
        }
    }
}

// Register-resident, wave-pipelined TPQRT.  A workgroup of 8 waves folds a block of MB = 16*SUB rows into its R:
//   * wave w owns the column tiles ct = w, w+8, ... (TPW per wave); the whole block lives in VGPRs in the MFMA
//     C/D layout (lane (kk, j) of tile/sub-tile holds row 16*sub + 4*reg + kk, column j), so
//       - V^T C consumes the block straight from the registers as the B operand (k-step = (sub, reg)),
//       - C -= V W accumulates straight into them;
//   * panel q (tile q) is factorised by its owner wave q % 8 in registers, the scaled Householder vectors V (MB x 16)
//     and T (16 x 16) are published in slot q % RING of an LDS ring, then `pub` (LDS) is advanced;
//   * every wave applies the panels in order to its own tiles as soon as they are published; the owner of panel q+1
//     updates tile q+1 first, factorises it, and only then finishes panel q on its other tiles -- the serial
//     dependency chain is  chain(q) -> one tile update -> chain(q+1), everything else runs beside it;
//   * only R (the panel's 16 rows) is streamed from global memory: each wave copies the R rows under its own tiles
//     to the LDS with LDS-DMA one panel ahead, and writes them back after the update.
#define FBR_TSQR_LDV 17  // LDS row stride of the published V panel (conflict-free for both operand walks)
#define FBR_TSQR_LDT 17  // LDS row stride of the published T (16 lanes write one row each: stride 16 would put them on two banks)
#define FBR_TSQR_TSZ (16 * FBR_TSQR_LDT)

// LDS carve (doubles): Rl[WAVES*TPW tiles][256] | Vr[RING][MB*17] | Tr[RING][256] | Rp[WAVES][256] | flags[16]
template <int TPW, int SUB, int W = FBR_TSQR_WAVES, bool XWG = false> __host__ __device__ static inline constexpr size_t fbr_tsqr_lds_doubles()
{
    constexpr bool RREG = fbr_tsqr_rreg<TPW, W>() || XWG;
    return (RREG ? 0 : (size_t)W * TPW * 256) + (size_t)fbr_tsqr_ring<SUB, W, RREG>() * (16 * SUB * FBR_TSQR_LDV + FBR_TSQR_TSZ) + (size_t)W * 256 + 16 +
           FBR_TSQR_RING_MAX + (XWG ? 32 : 0);  // (XWG: 64 ints at the very end for the blocks the workgroup has claimed)
}
typedef __attribute__((address_space(3))) void *fbr_tsqr_lds_ptr;
typedef const __attribute__((address_space(1))) void *fbr_tsqr_glb_ptr;

// Cross-lane primitives of the panel chain: all VALU (no LDS crossbar round trips).
//   fbr_dpp_bcast<J>  every lane gets the value of lane J of its own row of 16 lanes (DPP row_newbcast)
//   fbr_xor16_sum     v + (value of lane ^ 16);  fbr_xor32_sum  v + (value of lane ^ 32)   (gfx950 permlane swaps;
//                     both orders add the same two numbers, so every lane ends with bit-identical sums)
template <int J> __device__ __forceinline__ double fbr_dpp_bcast(double v)
{
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], 0x150 + J, 0xf, 0xf, true);
    b.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], 0x150 + J, 0xf, 0xf, true);
    return b.d;
}
__device__ __forceinline__ double fbr_xor16_sum(double v)
{
    union { double d; unsigned i[2]; } a, lo, hi;
    a.d = v;
    const auto r0 = __builtin_amdgcn_permlane16_swap(a.i[0], a.i[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(a.i[1], a.i[1], false, false);
    lo.i[0] = r0[0]; lo.i[1] = r1[0];
    hi.i[0] = r0[1]; hi.i[1] = r1[1];
    return lo.d + hi.d;
}
__device__ __forceinline__ double fbr_xor32_sum(double v)
{
    union { double d; unsigned i[2]; } a, lo, hi;
    a.d = v;
    const auto r0 = __builtin_amdgcn_permlane32_swap(a.i[0], a.i[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a.i[1], a.i[1], false, false);
    lo.i[0] = r0[0]; lo.i[1] = r1[0];
    hi.i[0] = r0[1]; hi.i[1] = r1[1];
    return lo.d + hi.d;
}

// acc += (value of lane J of the row of `bsrc`) * mul  in ONE instruction: gfx90a+ lets the DP-ALU VOP2 ops take a DPP
// row_newbcast source operand, which the compiler does not select from the intrinsics (it emits 2 v_mov_b32_dpp + FMA).
// The leading s_nop covers the VALU-write -> DPP-read hazard, which the hazard recogniser cannot see inside inline asm.
// NOP = false only where the DPP source register was written at least two VALU instructions earlier.
template <int J, bool NOP = true> __device__ __forceinline__ void fbr_fmac_bcast(double &acc, double bsrc, double mul)
{
#ifdef FBR_TSQR_SAFE_DPP
    constexpr bool nop = true;
#else
    constexpr bool nop = NOP;
#endif
    if constexpr (nop)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(J));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(J));
}

// acc += trow[l] * (z of lane l), l = 0..J-1   (column J of the T recurrence; lane i holds row i of T)
template <int J, int L = 0> struct FbrTAcc {
    static __device__ __forceinline__ void run(const double (&trow)[16], double z, double &a0, double &a1)
    {
        if constexpr (L < J) {
            if constexpr (L & 1)
                fbr_fmac_bcast<L, false>(a1, z, trow[L]);
            else
                fbr_fmac_bcast<L, (L == 0)>(a0, z, trow[L]);
            FbrTAcc<J, L + 1>::run(trow, z, a0, a1);
        }
    }
};

// One Householder step (column J) of the panel [R_pp ; B_p] held by one wave:
//   v[sb][reg]  lane (kk, c): B_p[16 sb + 4 reg + kk][c]   (MFMA C/D layout of the block tile)
//   Rp          LDS copy of the original R_pp (row-major 16 x 16, 0 below the diagonal)
//   rq[reg]     lane (kk, c): new R_pp[4 reg + kk][c];   trow[j]: lane (., i): T[i][j];   myscale: 1/(alpha - beta) of column c
// The vectors stay unscaled in v (scaled by myscale when published), so one fused multiply-add per element per step.
template <int SUB, int J>
__device__ __forceinline__ void fbr_tsqr_panel_step(fbr_td4 (&v)[SUB], const double *Rp, fbr_td4 &rq, double (&trow)[16],
                                                    double &myscale, int li, int kk)  // (li, kk by value: private opaque copies)
{
    // the lane masks of this step (li == J, li > J, ...) are rebuilt from an opaque copy of li: hoisted out of the panel / fold loops
    // (they are loop invariant) the 16 x 4 masks do not fit the SGPR file and come back as v_readlane pairs from spill lanes
    asm volatile("" : "+v"(li));
    const double rjc = Rp[J * 16 + li];
    const double alpha = fbr_dpp_bcast<J>(rjc);
    // s_c = x . B[:, c] with x = B[:, J] (lane J's registers, read through the DPP operand); lane J's own s is |x|^2
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg += 2) {
            if (sb == 0 && reg == 0)
                fbr_fmac_bcast<J, true>(s0, v[sb][reg], v[sb][reg]);
            else
                fbr_fmac_bcast<J, false>(s0, v[sb][reg], v[sb][reg]);  // written by the previous step's update, >= 3 instructions ago
            fbr_fmac_bcast<J, false>(s1, v[sb][reg + 1], v[sb][reg + 1]);
        }
    double s = fbr_xor32_sum(fbr_xor16_sum(s0 + s1));
    const double normsq = fbr_dpp_bcast<J>(s);
    // branch-free (a zero column gives tau = scale = 0, beta = alpha).  beta = -sign(alpha) g with g = sqrt(alpha^2 + |x|^2)
    // from the hardware rsq seed, one coupled Newton step on (g, h = 1/(2g)) and one residual correction of g;
    // tau = (beta - alpha) / beta = 1 + |alpha| / g = 1 + 2 |alpha| h;  scale = 1 / (alpha - beta) = sign(alpha) / (|alpha| + g)
    // by two Newton steps from a reciprocal seed taken on the unrefined g (off the dependency chain).
    const bool nz = normsq > 1e-290;  // (denormal-range column norms count as zero: the unscaled rsq below would overflow)
    const double aa = fabs(alpha);
    const double tt = fma(alpha, alpha, normsq);
    const double y0 = __builtin_amdgcn_rsq(tt);
    double g = tt * y0, h = 0.5 * y0;
    double r1 = __builtin_amdgcn_rcp(aa + g);
    const double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g);
    h = fma(h, rr, h);
    g = fma(fma(-g, g, tt), h, g);
    const double d1m = aa + g;
    r1 = fma(r1, fma(-d1m, r1, 1.0), r1);
    r1 = fma(r1, fma(-d1m, r1, 1.0), r1);
    const double scale = nz ? copysign(r1, alpha) : 0.0;
    const double tau = nz ? fma(aa, h + h, 1.0) : 0.0;
    const double beta = nz ? -copysign(g, alpha) : alpha;
    // column J of T: Z[l][J] = v_l . v_J = myscale_l * scale * s_l sits on lane l (l < J)
    {
        double a0 = 0.0, a1 = 0.0;
        FbrTAcc<J>::run(trow, scale * s * myscale, a0, a1);
        trow[J] = (li == J) ? tau : ((li < J) ? -tau * (a0 + a1) : 0.0);
    }
    const double wc = rjc + scale * s;
    const double ng = (li > J) ? -(tau * wc * scale) : 0.0;
    // B[:, c] -= g_c x: lane J itself has g = 0, so the broadcast source stays intact while the registers are rewritten
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            double e = v[sb][reg];
            if (sb == 0 && reg == 0)
                fbr_fmac_bcast<J, true>(e, e, ng);
            else
                fbr_fmac_bcast<J, false>(e, e, ng);
            v[sb][reg] = e;
        }
    if (li == J) myscale = scale;
    const double newr = (li > J) ? rjc - tau * wc : beta;
    asm volatile("" : "+v"(kk));
    if (kk == (J & 3)) rq[J >> 2] = newr;
    // pin the step's results here: otherwise their computation is sunk to the stores after the chain and every step's
    // inputs stay live (spills)
    asm volatile("" : "+v"(rq[J >> 2]));
    asm volatile("" : "+v"(trow[J]));
    asm volatile("" : "+v"(myscale));
    __builtin_amdgcn_sched_barrier(0);  // keep each step's side work (T column, R row) inside the step: bounded live ranges
}

template <int SUB, int... Js>
__device__ __forceinline__ void fbr_tsqr_panel_steps(fbr_td4 (&v)[SUB], const double *Rp, fbr_td4 &rq, double (&trow)[16],
                                                     double &myscale, int li, int kk, std::integer_sequence<int, Js...>)
{
    // The inline-asm DPP reads below are invisible to the hazard recogniser: make sure every earlier VALU write of the
    // tile registers (e.g. the 32-bit selects that copy the tile, which could sit right in front of the first step and
    // leave half of a double stale -- observed as 1e-8 relative errors) has retired before the first DPP operand read.
    asm volatile("s_nop 4" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    (fbr_tsqr_panel_step<SUB, Js>(v, Rp, rq, trow, myscale, li, kk), ...);
}

// Trailing update of column tile T of a wave with a published panel (V in Vl, T in Tm):
//   acc = R_rows + V^T C;  W = T^T acc;  R_rows -= W;  C -= V W        (Rl = LDS copy of the R_rows tiles, [tile][16][16])
// The MFMA C/D layout (reg r, lane (kk, j) = row 4r + kk, column j) is also the B-operand layout of k-step r, so acc
// and W feed the next product straight from the accumulator registers -- no LDS round trip, no barrier.
// Tiles are updated one at a time (a paired form ran dead / padding tiles through the MFMAs: +34 % MFMA work, slower;
// splitting V^T C over two accumulators to shorten the dependent chain was 3 % slower as well).
// Elements of a factor that ANOTHER workgroup reads or has written in the same launch (XWG: the cross-workgroup merge pipeline of the
// tree, fbr_tsqr_tree_x_kernel): 8-byte agent-scope relaxed atomics on both sides -- `sc1` stores write through, `sc1` loads bypass the
// reading CU's L1 (MI355X_MICROARCH.md, inter-workgroup visibility: "8-B agent atomics both sides"); ordering comes from the progress
// flags (fbr_xwg_publish / fbr_xwg_wait).
template <bool XWG> __device__ __forceinline__ double fbr_tsqr_ld(const double *p)
{
    if constexpr (XWG)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        return *p;
}
template <bool XWG> __device__ __forceinline__ void fbr_tsqr_st(double *p, double v)
{
    if constexpr (XWG)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// R rows of panel q under column tile ct, straight from global memory in the C/D layout (RREG: prefetched one tile ahead)
template <bool XWG = false>
__device__ __forceinline__ fbr_td4 fbr_tsqr_load_rrows(const double *__restrict__ R, unsigned ld, int q, int ct, int lane)
{
    const unsigned voff = (unsigned)(lane >> 4) * ld + (unsigned)(lane & 15);
    fbr_td4 r;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) r[reg] = fbr_tsqr_ld<XWG>(R + ((unsigned)(16 * q + 4 * reg) * ld + 16u * (unsigned)ct) + voff);
    return r;
}

template <int TPW, int SUB, int T, int W, bool RREG = false, bool XWG = false>
__device__ __forceinline__ void fbr_tsqr_update_tile(fbr_td4 (&C)[TPW][SUB], const double *Rl, double *__restrict__ R, unsigned ld, int q, int wave,
                                                     int lane, const double *Vl, const double *Tm, const fbr_td4 r0 = fbr_td4{0.0, 0.0, 0.0, 0.0})
{
    const int li = lane & 15, kk = lane >> 4;
    const unsigned j0 = 16u * (unsigned)q;
    const double *Rt = Rl + (wave + W * T) * 256;
    fbr_td4 acc, w2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int reg = 0; reg < 4; reg++) acc[reg] = RREG ? r0[reg] : Rt[(4 * reg + kk) * 16 + li];
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li], C[T][sb][reg], acc, 0, 0, 0);
        }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Tm[(4 * ks + kk) * FBR_TSQR_LDT + li], acc[ks], w2, 0, 0, 0);
    {
        // uniform (scalar) base + one per-lane offset shared by every store of the kernel
        const unsigned c0 = 16u * (unsigned)(wave + W * T);
        const unsigned voff = (unsigned)kk * ld + (unsigned)li;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) fbr_tsqr_st<XWG>(R + ((j0 + 4 * reg) * ld + c0) + voff, (RREG ? r0[reg] : Rt[(4 * reg + kk) * 16 + li]) - w2[reg]);
    }
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            C[T][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * sb + li) * FBR_TSQR_LDV + 4 * ks + kk], -w2[ks], C[T][sb], 0, 0, 0);
}

// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into the LDS, no VGPR data) written as inline assembly:
//   * scalar base + 32-bit per-lane byte offset (the builtin takes a 64-bit per-lane address: VGPR pairs and 64-bit adds);
//   * the compiler does not see an LDS write tracked by vmcnt, so it does not put s_waitcnt vmcnt(0) in front of every
//     later LDS read (the flag polls!) -- the consumer waits explicitly with fbr_dma_wait() before it reads the tiles.
__device__ __forceinline__ void fbr_dma16(const double *sbase, unsigned voff_bytes, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_bytes), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void fbr_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS-DMA of the R rows of panel q under the wave's tiles t >= t0 into their LDS slots (lane l -> row 8 h + l / 8,
// columns 2 (l % 8), +1: 16 bytes per lane).  Issued once per panel after the wave's whole update.
template <int TPW, int W>
__device__ __forceinline__ void fbr_tsqr_fetch_rows(int t0, int t1, double *Rl, const double *__restrict__ R, unsigned ld, int q, int wave, int lane)
{
    const unsigned voff = ((unsigned)(lane >> 3) * ld + 2u * (unsigned)(lane & 7)) * 8u;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(fbr_tsqr_lds_ptr)Rl;
#pragma unroll
    for (int t = 0; t < TPW; t++)
        if (t >= t0 && t < t1) {
            const int ct = wave + W * t;
#pragma unroll
            for (int h = 0; h < 2; h++)
                fbr_dma16(R + ((unsigned)(16 * q + 8 * h) * ld + 16u * (unsigned)ct), voff, lds0 + (unsigned)(ct * 256 + h * 128) * 8u);
        }
}

// the wave's tiles t0 <= t < t1, one at a time (static register indexing, a uniform branch per tile; dead and padding
// tiles cost nothing)
// (RREG: rn holds the R rows of tile t0 on entry; the rows of the next tile are requested before the current one is updated)
template <int TPW, int SUB, int W, int T = 0, bool RREG = false, bool XWG = false> struct FbrTsqrUpdateFrom {
    static __device__ __forceinline__ void run(int t0, int t1, fbr_td4 (&C)[TPW][SUB], double *Rl, double *__restrict__ R, unsigned ld, int q, int NP,
                                               int wave, int lane, const double *Vl, const double *Tm, fbr_td4 &rn)
    {
        if constexpr (T < TPW) {
            if (T >= t0 && T < t1) {
                if constexpr (RREG) {
                    const fbr_td4 r0 = rn;
                    if (T + 1 < t1) rn = fbr_tsqr_load_rrows<XWG>(R, ld, q, wave + W * (T + 1), lane);
                    fbr_tsqr_update_tile<TPW, SUB, T, W, true, XWG>(C, Rl, R, ld, q, wave, lane, Vl, Tm, r0);
                } else {
                    fbr_tsqr_update_tile<TPW, SUB, T, W>(C, Rl, R, ld, q, wave, lane, Vl, Tm);
                }
            }
            FbrTsqrUpdateFrom<TPW, SUB, W, T + 1, RREG, XWG>::run(t0, t1, C, Rl, R, ld, q, NP, wave, lane, Vl, Tm, rn);
        }
    }
};

// the single tile tp (uniform, selected by a static switch)
template <int TPW, int SUB, int W, int T = 0, bool RREG = false, bool XWG = false> struct FbrTsqrUpdateOne {
    static __device__ __forceinline__ void run(int tp, fbr_td4 (&C)[TPW][SUB], double *Rl, double *__restrict__ R, unsigned ld, int q, int NP,
                                               int wave, int lane, const double *Vl, const double *Tm, const fbr_td4 &r0)
    {
        if constexpr (T < TPW) {
            if (tp == T) fbr_tsqr_update_tile<TPW, SUB, T, W, RREG, XWG>(C, Rl, R, ld, q, wave, lane, Vl, Tm, r0);
            FbrTsqrUpdateOne<TPW, SUB, W, T + 1, RREG, XWG>::run(tp, C, Rl, R, ld, q, NP, wave, lane, Vl, Tm, r0);
        }
    }
};

// Ordering of the pipeline's LDS traffic.  All flags and payloads (V, T) live in the LDS, which executes a CU's
// operations in issue order, so publishing needs only "my LDS operations have been issued and returned" -- a workgroup
// fence would also drain the wave's global stores and LDS-DMA loads (vmcnt(0)), a full memory latency per panel.
__device__ __forceinline__ void fbr_lds_release() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void fbr_lds_acquire() { asm volatile("" ::: "memory"); }

// spin on an LDS word until it reaches `want` (bounded: a protocol error must not hang the device)
__device__ __forceinline__ bool fbr_tsqr_wait_ge(const int *flag, int want)
{
    for (int it = 0; it < FBR_TSQR_SPIN_LIMIT; it++) {
        if (__atomic_load_n(flag, __ATOMIC_RELAXED) >= want) {
            fbr_lds_acquire();
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// One fold = one MB-row block B folded into R.  The folds of a workgroup form ONE continuous pipeline: panels are
// numbered globally (G = panels of all earlier folds + q - q0), the flags only ever grow, and there is no barrier
// between folds -- a wave that has consumed all its tiles of fold f loads its tiles of fold f + 1 and goes on while
// the last, narrow panels of fold f are still being factorised by the other waves (up to RING panels of drift).
struct FbrTsqrFoldDesc {
    const double *B;  // block rows: element (r, c) at B[r * ldb + c * cs]  (row-major: ldb = leading dimension, cs = 1; column-major chunk:
                      // ldb = 1, cs = the chunk's column stride); rows >= mrows are zero
    int ldb, mrows, first_col;
    // XWG (cross-workgroup merge pipeline): per-wave progress counters (global memory, [waves]) of the block folded into the same factor
    // right before this one by ANOTHER workgroup (null: none), and of this block.  Counter of wave w: after iteration q it is q + 2 (0 =
    // not started; the block's opening iteration q0 - 1, which only factorises panel q0, counts -- for blocks of DENSE rows q0 = 0 and the
    // successor's first read, R_pp of panel 0, waits for it), and the R rows of panel q under the wave's tiles and the R_pp of panel
    // q + 1 (if the wave owns it) are final for this block and visible.
    const int *prev = nullptr;
    int *mine = nullptr;
    long cs = 1;
};

// progress flags of the cross-workgroup pipeline: relaxed agent-scope atomics (the data travels as agent-scope atomics as well; the
// publisher first waits for its own stores, the poller is bounded like the LDS waits)
__device__ __forceinline__ void fbr_xwg_publish(int *flag, int v, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool fbr_xwg_wait(const int *flag, int want)
{
    for (int it = 0; it < FBR_TSQR_SPIN_LIMIT; it++) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) {
            asm volatile("" ::: "memory");
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// TIMING: diagnostic instantiation (s_memtime cycles per phase into tacc[16]); the production kernels carry none of it
template <int TPW, int SUB, bool TIMING, int W, bool XWG = false, class FoldFn>
__device__ __forceinline__ void fbr_tsqr_stream(double *__restrict__ R, int n, int ldr, int nfolds, FoldFn fold_of, double *smem, unsigned *errflag,
                                                unsigned long long *tacc = nullptr)
{
    unsigned long long tk = TIMING ? __builtin_readcyclecounter() : 0;
#define FBR_TT(i)                                                         \
    if constexpr (TIMING) {                                               \
        const unsigned long long t1 = __builtin_readcyclecounter();       \
        tacc[i] += t1 - tk;                                               \
        tk = t1;                                                          \
    }
    constexpr int MB = 16 * SUB;
    constexpr bool RREG = fbr_tsqr_rreg<TPW, W>() || XWG;
    constexpr int FBR_TSQR_RING = fbr_tsqr_ring<SUB, W, RREG>();
    double *Rl = smem;  // 16-byte aligned tiles for the LDS-DMA
    double *Vr = Rl + (RREG ? 0 : W * TPW * 256);
    double *Tr = Vr + FBR_TSQR_RING * MB * FBR_TSQR_LDV;
    double *Rps = Tr + FBR_TSQR_RING * FBR_TSQR_TSZ;
    int *done = (int *)(Rps + W * 256);  // done[w]: wave w needs no panel < done[w] any more
    int *seq = done + 16;                // seq[slot]: 1 + global index of the panel published in ring slot `slot` (0 = none yet)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int NP = n / 16;
    const unsigned ld = (unsigned)ldr;
    double *Rp = Rps + wave * 256;  // R_pp staging of this wave's panel factorisations
    const int t1 = (NP - wave + W - 1) / W;  // this wave's tiles t < t1 are real column tiles (the rest is padding)

    if (tid < FBR_TSQR_RING) seq[tid] = 0;
    if (tid < W) done[tid] = 0;
    __syncthreads();  // the only workgroup barrier

    bool ok = true;
    unsigned fail_code = 0;
    int fail_q = 0;
    auto note = [&](bool good, unsigned code, int q) {
        if (ok && !good) {
            fail_code = code;
            fail_q = q;
        }
        ok = ok && good;
    };
    int gbase = 0;  // global index of panel q0 of the current fold
    for (int f = 0; f < nfolds; f++) {
        const FbrTsqrFoldDesc fd = fold_of(f);
        const int q0 = fd.first_col / 16;  // panels left of first_col: block columns are zero, identity reflectors
        if (q0 >= NP) continue;
        // ---- load this wave's tiles of the block
        fbr_td4 C[TPW][SUB];
#pragma unroll
        for (int t = 0; t < TPW; t++) {
            const int ct = wave + W * t;
#pragma unroll
            for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    // unconditional loads from a clamped address + select: a conditional load would have to complete
                    // before the other lanes may write the zero into the same register (one latency per element)
                    // (row group and tile clamped uniformly: scalar base + per-lane offset kk * ldb + li)
                    const int rg = 16 * sb + 4 * reg;
                    const bool gvalid = ct < NP && rg < fd.mrows;
                    const double *Bs = fd.B + ((long)((unsigned)(gvalid ? rg : 0) * (unsigned)fd.ldb) + (gvalid ? 16 * ct : 0) * fd.cs);
                    const bool valid = gvalid && rg + kk < fd.mrows;
                    const double x = Bs[(long)(valid ? (unsigned)kk * (unsigned)fd.ldb : 0u) + li * fd.cs];
                    C[t][sb][reg] = valid ? x : 0.0;
                }
        }
        // R_pp of the next panel this wave owns (lane (kk, c): rows kk, kk+4, kk+8, kk+12 of column c, 0 below the diagonal)
        fbr_td4 rpp = {0.0, 0.0, 0.0, 0.0};
        auto fetch_rpp = [&](int p) {
            if (p < NP) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const double *Rs = R + ((unsigned)(16 * p + 4 * reg) * ld + 16u * (unsigned)p);  // uniform
                    rpp[reg] = fbr_tsqr_ld<XWG>(Rs + ((unsigned)kk * ld + (unsigned)li));  // (entries below the diagonal are masked at use)
                }
            }
        };
        // XWG: what this block reads of the shared factor must have been finished by the block before it (another workgroup): the wave
        // waits on the progress counter of ITS OWN index there (same tile ownership in every block).  Only the owner of the first panel
        // needs something before the first iteration (R_pp of q0: final once the predecessor has chained it, in its iteration q0 - 1);
        // everything else is requested at the end of an iteration for the next one (below).
        const int *wprev = XWG && fd.prev ? fd.prev + wave : nullptr;
        int *wmine = XWG ? fd.mine + wave : nullptr;
        if constexpr (XWG) {
            if (wave == q0 % W) {
                if (wprev && ok) note(fbr_xwg_wait(wprev, q0 + 1), 3u, q0);
                fetch_rpp(q0);
            }
        } else {
            fetch_rpp(q0 + ((wave - q0) % W + W) % W);
        }
        // R rows of the first panel under this wave's tiles right of it (the wave is done with its tiles of the
        // previous fold, so their LDS slots are free)
        // RREG: the rows of the FIRST tile the wave will update with a panel travel in rn, requested an iteration ahead: the tile of the
        // next panel when the wave owns it (updated first, in front of the chain), else its first tile right of the panel
        fbr_td4 rn = {0.0, 0.0, 0.0, 0.0};
        auto first_tile = [&](int q) { return (q + 1 < NP && wave == (q + 1) % W) ? (q + 1) / W : ((q >= wave) ? (q - wave) / W + 1 : 0); };
        if constexpr (!RREG) fbr_tsqr_fetch_rows<TPW, W>((q0 >= wave) ? (q0 - wave) / W + 1 : 0, t1, Rl, R, ld, q0, wave, lane);
        FBR_TT(0)

        // factorise panel p = tile p of this wave (global index G): Householder QR of [R_pp ; tile] in registers,
        // publish V, T in ring slot G % RING
        // (v: the panel's tile -- for TPW <= 2 the wave's own registers C[t], which the factorisation consumes: the tile is dead
        // afterwards, and a copy of a tall tile would not fit the register file; else a copy selected from the wave's tiles, so that
        // the unrolled 16-step chain exists once in the code)
        auto chain_on = [&](fbr_td4 (&v)[SUB], int p, int G, unsigned long long tc0) __attribute__((always_inline)) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Rp[(4 * reg + kk) * 16 + li] = (li >= 4 * reg + kk) ? rpp[reg] : 0.0;
            fbr_lds_release();
            __builtin_amdgcn_wave_barrier();
            fbr_td4 rq = {0.0, 0.0, 0.0, 0.0};
            double trow[16];
            double myscale = 0.0;
            unsigned long long tq = TIMING ? __builtin_readcyclecounter() : 0;
            if constexpr (TIMING) tacc[7] += tq - tc0;  // tile copy, R_pp staging
            fbr_tsqr_panel_steps<SUB>(v, Rp, rq, trow, myscale, li, kk, std::make_integer_sequence<int, 16>{});
            if constexpr (TIMING) tacc[4] += __builtin_readcyclecounter() - tq;
            // the ring slot is free once no wave needs panel G - RING any more
            {
                const int need = G - FBR_TSQR_RING + 1;
                tq = TIMING ? __builtin_readcyclecounter() : 0;
                bool free_ = !ok;
                for (int it = 0; it < FBR_TSQR_SPIN_LIMIT && !free_; it++) {
                    const int d = __atomic_load_n(done + (lane & (W - 1)), __ATOMIC_RELAXED);
                    free_ = __builtin_amdgcn_ballot_w64(d < need) == 0;
                    if (!free_) __builtin_amdgcn_s_sleep(1);
                }
                fbr_lds_acquire();
                note(free_, 2u, G);
                if constexpr (TIMING) tacc[5] += __builtin_readcyclecounter() - tq;
            }
            double *Vl = Vr + (G % FBR_TSQR_RING) * (MB * FBR_TSQR_LDV);
            double *Tm = Tr + (G % FBR_TSQR_RING) * FBR_TSQR_TSZ;
#pragma unroll
            for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li] = v[sb][reg] * myscale;
            if (kk == 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) Tm[li * FBR_TSQR_LDT + j] = trow[j];
            }
            // published per ring slot: seq[slot] = G + 1 (the consumers take the panels in order anyway; the publisher does not have
            // to wait for its predecessor, which may belong to the previous fold and still be in flight)
            fbr_lds_release();
            if (lane == 0) __atomic_store_n(seq + G % FBR_TSQR_RING, G + 1, __ATOMIC_RELAXED);
            // R_pp back to global (upper triangle), then the R_pp of the next panel this wave owns
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 4 * reg + kk;
                double *Rs = R + ((unsigned)(16 * p + 4 * reg) * ld + 16u * (unsigned)p);  // uniform
                if (li >= i) fbr_tsqr_st<XWG>(Rs + ((unsigned)kk * ld + (unsigned)li), rq[reg]);
            }
            if constexpr (!XWG) fetch_rpp(p + W);  // (XWG: requested behind the predecessor's progress, at the end of iteration p + W - 2)
        };
        auto chain = [&](int p, int G) __attribute__((always_inline)) {
            const unsigned long long tc0 = TIMING ? __builtin_readcyclecounter() : 0;
            const int tp = p / W;
            if constexpr (TPW == 1) {
                chain_on(C[0], p, G, tc0);
            } else if constexpr (TPW == 2) {
                if (tp == 0)
                    chain_on(C[0], p, G, tc0);
                else
                    chain_on(C[1], p, G, tc0);
            } else if constexpr (TPW <= FBR_TSQR_INPLACE_MAX_TPW) {
                // one copy of the unrolled chain per tile of the wave: the price of factorising a tall tile where it lies
                if (tp == 0)
                    chain_on(C[0], p, G, tc0);
                else if (tp == 1)
                    chain_on(C[1], p, G, tc0);
                else if (tp == 2)
                    chain_on(C[2], p, G, tc0);
                else
                    chain_on(C[TPW - 1], p, G, tc0);
            } else {
                // copy of the panel's tile: a uniform BRANCH per tile (the empty asm keeps the compiler from turning the four cheap
                // blocks into 32 x 4 selects: 128 v_cndmask per panel on the fold's critical path)
                fbr_td4 v[SUB];
#pragma unroll
                for (int t = 0; t < TPW; t++)
                    if (t == tp) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int sb = 0; sb < SUB; sb++) v[sb] = C[t][sb];
                    }
                chain_on(v, p, G, tc0);
            }
        };

        // iteration q applies panel q; the owner of panel q + 1 factorises it inside iteration q (q = q0 - 1: only that)
        for (int q = q0 - 1; q < NP; q++) {
            const bool apply = q >= q0;
            const int G = gbase + (q - q0);
            const bool next_owner = q + 1 < NP && wave == (q + 1) % W;
            int t0 = (q >= wave) ? (q - wave) / W + 1 : 0;  // this wave's first tile right of panel q
            const bool need_panel = apply && (next_owner || t0 < t1);
            if (need_panel && ok) note(fbr_tsqr_wait_ge(seq + (G + FBR_TSQR_RING) % FBR_TSQR_RING, G + 1), 1u, q);
            if (need_panel || next_owner) fbr_dma_wait();  // vmcnt(0): this wave's R rows (LDS-DMA) / R_pp have landed
            FBR_TT(2)
            const int slot = (G + FBR_TSQR_RING) % FBR_TSQR_RING;
            const double *Vl = Vr + slot * (MB * FBR_TSQR_LDV);
            const double *Tm = Tr + slot * FBR_TSQR_TSZ;
            if (next_owner) {
                // next panel's owner: its tile first, then its factorisation, then the rest of panel q.  This is the
                // serial dependency chain of the fold: raise the wave's issue priority over the waves that only update
                __builtin_amdgcn_s_setprio(3);
                if (apply) FbrTsqrUpdateOne<TPW, SUB, W, 0, RREG, XWG>::run((q + 1) / W, C, Rl, R, ld, q, NP, wave, lane, Vl, Tm, rn);
                FBR_TT(3)
                t0 = (q + 1) / W + 1;
                // (RREG: the rows under the wave's next tile travel during the chain)
                if constexpr (RREG)
                    if (apply && t0 < t1) rn = fbr_tsqr_load_rrows<XWG>(R, ld, q, wave + W * t0, lane);
                chain(q + 1, G + 1);
                __builtin_amdgcn_s_setprio(0);
                FBR_TT(1)
            }
            if (apply) {
                FbrTsqrUpdateFrom<TPW, SUB, W, 0, RREG, XWG>::run(t0, t1, C, Rl, R, ld, q, NP, wave, lane, Vl, Tm, rn);
                fbr_lds_release();
                if (lane == 0) __atomic_store_n(done + wave, G + 1, __ATOMIC_RELAXED);
                // R rows of the next panel under the tiles right of it
                if constexpr (!RREG)
                    if (q + 1 < NP) fbr_tsqr_fetch_rows<TPW, W>((q + 1 >= wave) ? (q + 1 - wave) / W + 1 : 0, t1, Rl, R, ld, q + 1, wave, lane);
            }
            if constexpr (XWG) {
                // this wave's part of iteration q is final: publish, then -- behind the predecessor's iteration q + 1 -- request what
                // iteration q + 1 reads: the R rows under the first tile it will update and, for the owner of panel q + 2, its R_pp
                fbr_xwg_publish(wmine, q + 2, lane);
                if (q + 1 < NP) {
                    if (wprev && ok) note(fbr_xwg_wait(wprev, q + 3), 4u, q);
                    const int tf = first_tile(q + 1);
                    if (tf < t1) rn = fbr_tsqr_load_rrows<true>(R, ld, q + 1, wave + W * tf, lane);
                    if (q + 2 < NP && wave == (q + 2) % W) fetch_rpp(q + 2);
                }
            } else if constexpr (RREG) {
                if (q + 1 < NP) {
                    const int tf = first_tile(q + 1);
                    if (tf < t1) rn = fbr_tsqr_load_rrows(R, ld, q + 1, wave + W * tf, lane);
                }
            }
            FBR_TT(3)
        }
        gbase += NP - q0;
    }
    // first failing wave: which wait (1 = panel publish, 2 = ring slot, 3 = predecessor's R_pp, 4 = predecessor's iteration), wave, panel, block
    if (!ok && lane == 0) atomicCAS(errflag, 0u, fail_code | ((unsigned)wave << 4) | (((unsigned)fail_q & 0xffu) << 8) | ((unsigned)blockIdx.x << 16) | 0x80000000u);
#undef FBR_TT
}

// level 0: workgroup w folds blocks w, w+NW, ... of A into Rw[w]
// W = waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU, n <= 320: two independent folds in flight per CU)
template <int TPW, int SUB, bool TIMING, int W = FBR_TSQR_WAVES>
__global__ __launch_bounds__(64 * W, W == FBR_TSQR_WAVES ? 1 : 2) void fbr_tsqr_level0_kernel(const double *__restrict__ A, long Mpad, int n,
                                                                               double *__restrict__ Rw, long nblocks, unsigned *errflag,
                                                                               unsigned long long *dbg, const int *__restrict__ rowfc, int orows,
                                                                               long ogroup, long M, int slot_stride, long cm_ld)
{
    // slot_stride: workgroup w folds into working factor w * slot_stride (1 for data chunks; 2^l when rows are folded into the factors
    // that are still alive after l levels of the merge tree: the embedded group factors of the tree-structured TSQR)
    unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB;
    constexpr int LD = 16 * W * TPW;  // leading dimension of the working factors (>= n)
    double *R = Rw + (long)blockIdx.x * slot_stride * n * LD;
    const int nfolds = (int)((nblocks - blockIdx.x + gridDim.x - 1) / gridDim.x);
    auto fold_of = [&](int f) {
        const long r0 = ((long)blockIdx.x + (long)f * gridDim.x) * MB;
        int fc = 0;
        if (orows > 0) {  // chunk stacked by regressor row: the block is zero left of the first supported column of the rows it spans
            fc = n;
            if (r0 < M) {
                const int ra = (int)(r0 / ogroup), rb = (int)((std::min<long>(r0 + MB, M) - 1) / ogroup);
                for (int r = ra; r <= rb; r++) fc = min(fc, rowfc[r]);
            }
        } else if (orows < 0) {  // upper-triangular input
            fc = (int)std::min<long>(r0, n);
        }
        FbrTsqrFoldDesc fd{cm_ld > 0 ? A + r0 : A + r0 * n, cm_ld > 0 ? 1 : n, (int)std::min<long>(MB, Mpad - r0), fc};
        fd.cs = cm_ld > 0 ? cm_ld : 1;
        return fd;
    };
    fbr_tsqr_stream<TPW, SUB, TIMING, W, false>(R, n, LD, nfolds, fold_of, smem, errflag, tacc);
    if (TIMING && (threadIdx.x & 63) == 0) {
        unsigned long long *d = dbg + ((long)blockIdx.x * W + (threadIdx.x >> 6)) * 16;
        for (int i = 0; i < 16; i++) d[i] = tacc[i];
    }
}

// tree level: workgroup i folds Rw[(2i+1)*stride] (upper triangular, MB rows at a time) into Rw[2i*stride]
template <int TPW, int SUB, int W = FBR_TSQR_WAVES>
__global__ __launch_bounds__(64 * W, W == FBR_TSQR_WAVES ? 1 : 2) void fbr_tsqr_tree_kernel(double *__restrict__ Rw, int n, int stride, int count, unsigned *errflag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB;
    constexpr int LD = 16 * W * TPW;
    const long a = (long)2 * blockIdx.x * stride, b = a + stride;
    if (b >= count) return;
    const double *Rb = Rw + b * n * LD;
    auto fold_of = [&](int f) {
        const int i0 = f * MB;
        return FbrTsqrFoldDesc{Rb + (long)i0 * LD, LD, std::min(MB, n - i0), i0};
    };
    fbr_tsqr_stream<TPW, SUB, false, W, false>(Rw + a * n * LD, n, LD, (n + MB - 1) / MB, fold_of, smem, errflag);
}

// Tree level with the merges PIPELINED ACROSS WORKGROUPS.  A merge folds the partner's factor in MB-row blocks, and a block can follow
// the one before it through the panels as soon as that one is a panel ahead -- but one workgroup holds one block at a time, so in
// fbr_tsqr_tree_kernel the 8 blocks of a 496-column merge are 136 serial panel steps (0.93 ms per level, 8 levels: the fixed tail of
// every TSQR call).  Here G workgroups share a merge: workgroup j folds blocks j, j + G, ... and block b runs a panel or two behind
// block b - 1 of its neighbour, synchronised per wave and panel through progress counters in global memory (FbrTsqrFoldDesc::prev /
// mine); the factor's rows travel between the CUs as agent-scope atomics.  With G = 8 a merge takes ~40 panel steps instead of 136.
// The blocks are folded in the same order by the same arithmetic: the result is bit-identical to fbr_tsqr_tree_kernel's.
// prog: [pairs][blocks of a merge][8] ints, zero before the launch.  The workgroups of a merge are consecutive in the grid, so that
// in-order dispatch starts a block's predecessor first (a predecessor never waits for its successor: no deadlock even if the grid is
// not fully resident; the waits are bounded and report through errflag).
template <int TPW, int SUB, int W = FBR_TSQR_WAVES>
__global__ __launch_bounds__(64 * W, W == FBR_TSQR_WAVES ? 1 : 2) void fbr_tsqr_tree_x_kernel(double *__restrict__ Rw, int n, int stride, int count, int G,
                                                                                 int *__restrict__ prog, unsigned *errflag, int brows)
{
    // brows > 0: the partner is not a triangular factor but `brows` DENSE rows (the expanded reduced factor R_red E of a link-merged
    // model): every block is folded from column 0 and only the blocks that hold rows are visited
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB;
    constexpr int LD = 16 * W * TPW;
    const int p = blockIdx.x / G;
    const long a = (long)2 * p * stride, b = a + stride;
    if (b >= count) return;
    const double *Rb = Rw + b * n * LD;
    const int nblk = ((brows > 0 ? brows : n) + MB - 1) / MB;
    // Blocks are CLAIMED, not assigned: the workgroups of a merge take the next block of their pair from a counter as they come
    // (the first wave of a workgroup to reach its fold f claims for the workgroup, through the LDS).  A block's predecessor has then
    // always been claimed by a workgroup that is already running -- the waits below never depend on the order in which the hardware
    // dispatches the grid (HIP promises none; with blocks assigned by blockIdx a consumer could hold its CU while its producer
    // waited for one: seen as flag time-outs under mixed asynchronous submissions).  Which workgroup folds which block does not
    // change the arithmetic: the blocks still enter the factor in order.
    int *claimed = (int *)(smem + fbr_tsqr_lds_doubles<TPW, SUB, W, true>()) - 64;
    int *next = prog + (long)gridDim.x / G * nblk * W + p;  // (the pairs' claim counters sit behind the progress flags)
    if (threadIdx.x < 64) claimed[threadIdx.x] = -1;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    auto fold_of = [&](int f) {
        int v = 0;
        if (lane == 0) {
            v = atomicCAS(&claimed[f], -1, -2);
            if (v == -1) {
                v = __hip_atomic_fetch_add(next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_store_n(&claimed[f], v, __ATOMIC_RELAXED);
            } else {
                for (int it = 0; v < 0 && it < FBR_TSQR_SPIN_LIMIT; it++) {
                    __builtin_amdgcn_s_sleep(1);
                    v = __atomic_load_n(&claimed[f], __ATOMIC_RELAXED);
                }
                if (v < 0) v = 1 << 20;  // (gives up: the block is left out and the error word set by the waits that follow)
            }
        }
        const int bi = __builtin_amdgcn_readfirstlane(v);
        if (bi >= nblk) return FbrTsqrFoldDesc{Rb, LD, 0, n};  // nothing left to claim: an empty fold (first column = n)
        const int i0 = bi * MB;
        FbrTsqrFoldDesc fd{Rb + (long)i0 * LD, LD, std::min(MB, n - i0), brows > 0 ? 0 : i0};
        fd.prev = bi > 0 ? prog + ((long)p * nblk + bi - 1) * W : nullptr;
        fd.mine = prog + ((long)p * nblk + bi) * W;
        return fd;
    };
    fbr_tsqr_stream<TPW, SUB, false, W, true>(Rw + a * n * LD, n, LD, nblk, fold_of, smem, errflag);
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow factors (n <= 128 columns: left arm, KUKA): wave-private TSQR.  With 6-8 column tiles there is not enough
// trailing work to feed 8 waves and the fold is pure panel-chain latency, so here EVERY WAVE is an independent worker:
// it owns a private R (global memory, L2 / Infinity-Cache resident), holds a 32-row block of all NPT tiles in its
// VGPRs, factorises the panels itself and applies them to its own tiles.  No synchronisation at all between waves;
// 8 waves per CU (2 per SIMD) keep 8 dependency chains in flight per CU instead of one.
//   per-wave LDS (doubles): Vl[MB*17] | Tm[256] | Rp[256]   (A-operand staging of V and T, R_pp staging)
template <int SUB> __host__ __device__ constexpr size_t fbr_tsqr_narrow_lds_doubles() { return (size_t)16 * SUB * FBR_TSQR_LDV + 512; }
#define FBR_TSQR_NARROW_WAVES 4  // waves per workgroup of the narrow kernels (two workgroups per CU)

template <int NPT, int SUB>
__device__ __forceinline__ void fbr_tsqr_wave_fold(double *__restrict__ R, const double *__restrict__ B, unsigned ldb, int mrows, int first_col,
                                                   double *lds, int lane, long cs = 1)
{
    constexpr int MB = 16 * SUB;
    constexpr unsigned ld = 16 * NPT;
    double *Vl = lds;
    double *Tm = Vl + MB * FBR_TSQR_LDV;
    double *Rp = Tm + 256;
    const int li = lane & 15, kk = lane >> 4;
    const unsigned voff = (unsigned)kk * ld + (unsigned)li;
    const int q0 = first_col / 16;
    if (q0 >= NPT) return;

    fbr_td4 C[NPT][SUB];
#pragma unroll
    for (int t = 0; t < NPT; t++)
#pragma unroll
        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int rg = 16 * sb + 4 * reg;
                const bool gvalid = rg < mrows;
                const double *Bs = B + ((long)((unsigned)(gvalid ? rg : 0) * ldb) + 16 * t * cs);
                const bool valid = gvalid && rg + kk < mrows;
                const double x = Bs[(long)(valid ? (unsigned)kk * ldb : 0u) + li * cs];
                C[t][sb][reg] = valid ? x : 0.0;
            }
    // rows 4 reg + kk of the 16 x 16 tile (row panel p, column tile t) of R
    auto load_tile = [&](int p, int t) {
        fbr_td4 r;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) r[reg] = (R + ((unsigned)(16 * p + 4 * reg) * ld + 16u * (unsigned)t))[voff];
        return r;
    };
    fbr_td4 rpp = load_tile(q0, q0);
    for (int p = q0; p < NPT; p++) {
        fbr_td4 rn = {0.0, 0.0, 0.0, 0.0};
        if (p + 1 < NPT) rn = load_tile(p, p + 1);  // R rows of the first tile to update: in flight during the chain
        fbr_td4 v[SUB];
#pragma unroll
        for (int t = 0; t < NPT; t++)
            if (t == p) {
#pragma unroll
                for (int sb = 0; sb < SUB; sb++) v[sb] = C[t][sb];
            }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Rp[(4 * reg + kk) * 16 + li] = (li >= 4 * reg + kk) ? rpp[reg] : 0.0;
        fbr_lds_release();
        __builtin_amdgcn_wave_barrier();
        fbr_td4 rq = {0.0, 0.0, 0.0, 0.0};
        double trow[16];
        double myscale = 0.0;
        fbr_tsqr_panel_steps<SUB>(v, Rp, rq, trow, myscale, li, kk, std::make_integer_sequence<int, 16>{});
#pragma unroll
        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li] = v[sb][reg] * myscale;
        if (kk == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) Tm[li * 16 + j] = trow[j];
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            double *Rs = R + ((unsigned)(16 * p + 4 * reg) * ld + 16u * (unsigned)p);
            if (li >= 4 * reg + kk) Rs[voff] = rq[reg];
        }
        if (p + 1 < NPT) rpp = load_tile(p + 1, p + 1);  // row panel p + 1 is not touched by panel p's update
        fbr_lds_release();
        __builtin_amdgcn_wave_barrier();
        // ---- trailing update of the tiles right of the panel (static register indexing, uniform branch per tile).
        //      The A operands (V for V^T C, V for V W, T) are the same for every tile: read from the LDS once per panel.
        // (CACHE: the operands stay in registers for all tiles of the panel; tall blocks of many tiles have no room for them and read
        // them from the LDS in front of every MFMA instead)
        constexpr bool CACHE = NPT * SUB <= 12;
        constexpr int CS = CACHE ? SUB : 1;
        double va[CS][4], vb[CS][4], ta[4];
        if constexpr (CACHE) {
#pragma unroll
            for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    va[sb][r4] = Vl[(16 * sb + 4 * r4 + kk) * FBR_TSQR_LDV + li];
                    vb[sb][r4] = Vl[(16 * sb + li) * FBR_TSQR_LDV + 4 * r4 + kk];
                }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) ta[ks] = Tm[(4 * ks + kk) * 16 + li];
#pragma unroll
        for (int t = 0; t < NPT; t++)
            if (t > p) {
                const fbr_td4 r0 = rn;
                if (t + 1 < NPT) rn = load_tile(p, t + 1);
                fbr_td4 acc = r0, w2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(CACHE ? va[CACHE ? sb : 0][reg] : Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li], C[t][sb][reg], acc, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ks], acc[ks], w2, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) (R + ((unsigned)(16 * p + 4 * reg) * ld + 16u * (unsigned)t))[voff] = r0[reg] - w2[reg];
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++)
                        C[t][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(CACHE ? vb[CACHE ? sb : 0][ks] : Vl[(16 * sb + li) * FBR_TSQR_LDV + 4 * ks + kk], -w2[ks], C[t][sb], 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();  // (the wave's own LDS reads are in order with the next panel's writes)
    }
}

// level 0: wave g folds blocks g, g + NWV, ... of A into its private Rw[g]
// SUB = 6 (96-row blocks, round 6): ONE workgroup of four waves per CU, a wave may use all 512 registers of its SIMD lane (288 of them hold
// the block of a six-tile factor) -- the 36 fixed instructions of a Householder step are amortised over twice the rows of the 48-row shape
template <int NPT, int SUB>
__global__ __launch_bounds__(FBR_TSQR_NARROW_WAVES * 64, (SUB > 3 ? 1 : 2)) void fbr_tsqr_narrow_level0_kernel(const double *__restrict__ A, long Mpad,
                                                                                                 double *__restrict__ Rw, long nblocks, int nwaves,
                                                                                                 const int *__restrict__ rowfc, int orows, long ogroup, long M, long cm_ld)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB, n = 16 * NPT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x * FBR_TSQR_NARROW_WAVES + wave;
    if (g >= nwaves) return;
    double *lds = smem + wave * fbr_tsqr_narrow_lds_doubles<SUB>();
    double *R = Rw + (long)g * n * n;
    for (long b = g; b < nblocks; b += nwaves) {
        const long r0 = b * MB;
        int fc = 0;
        if (orows > 0) {  // chunk stacked by regressor row (FbrTsqrRowOrder)
            fc = n;
            if (r0 < M) {
                const int ra = (int)(r0 / ogroup), rb = (int)((std::min<long>(r0 + MB, M) - 1) / ogroup);
                for (int r = ra; r <= rb; r++) fc = min(fc, rowfc[r]);
            }
        } else if (orows < 0) {  // upper-triangular input
            fc = (int)std::min<long>(r0, n);
        }
        if (cm_ld > 0)
            fbr_tsqr_wave_fold<NPT, SUB>(R, A + r0, 1, (int)std::min<long>(MB, Mpad - r0), fc, lds, lane, cm_ld);
        else
            fbr_tsqr_wave_fold<NPT, SUB>(R, A + r0 * n, n, (int)std::min<long>(MB, Mpad - r0), fc, lds, lane);
    }
}

// tree level: wave i folds Rw[(2i+1)*stride] (upper triangular, MB rows at a time) into Rw[2i*stride]
template <int NPT, int SUB>
__global__ __launch_bounds__(FBR_TSQR_NARROW_WAVES * 64, 2) void fbr_tsqr_narrow_tree_kernel(double *__restrict__ Rw, int stride, int count)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB, n = 16 * NPT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long i = (long)blockIdx.x * FBR_TSQR_NARROW_WAVES + wave;
    const long a = 2 * i * stride, b = a + stride;
    if (b >= count) return;
    double *lds = smem + wave * fbr_tsqr_narrow_lds_doubles<SUB>();
    const double *Rb = Rw + b * n * n;
    for (int i0 = 0; i0 < n; i0 += MB) fbr_tsqr_wave_fold<NPT, SUB>(Rw + a * n * n, Rb + (long)i0 * n, n, std::min(MB, n - i0), i0, lds, lane);
}

// the same level for up to FBR_TSQR_NARROW_BATCH factorisations of one shape at once (blockIdx.y): the trees of narrow factors are
// launch bound (11 levels of 50 - 100 us over 2048 wave-private factors), and the row groups of a symmetric robot come in equal pairs
// (two arms, two legs) -- one launch sequence serves both
#define FBR_TSQR_NARROW_BATCH 4
struct FbrTsqrNarrowBatch {
    double *Rw[FBR_TSQR_NARROW_BATCH];
};
template <int NPT, int SUB>
__global__ __launch_bounds__(FBR_TSQR_NARROW_WAVES * 64, 2) void fbr_tsqr_narrow_tree_batch_kernel(FbrTsqrNarrowBatch bt, int stride, int count)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB, n = 16 * NPT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long i = (long)blockIdx.x * FBR_TSQR_NARROW_WAVES + wave;
    const long a = 2 * i * stride, b = a + stride;
    if (b >= count) return;
    double *Rw = bt.Rw[blockIdx.y];
    double *lds = smem + wave * fbr_tsqr_narrow_lds_doubles<SUB>();
    const double *Rb = Rw + b * n * n;
    for (int i0 = 0; i0 < n; i0 += MB) fbr_tsqr_wave_fold<NPT, SUB>(Rw + a * n * n, Rb + (long)i0 * n, n, std::min(MB, n - i0), i0, lds, lane);
}

// progress flags / claim counters of a merge tree := 0 (a kernel instead of hipMemsetAsync: the runtime's fill path costs 40 - 80 us per
// call on the stream, and this clear sits between the last level-0 fold and the first tree level of every factorisation)
__global__ __launch_bounds__(256) void fbr_tsqr_zero_kernel(int *__restrict__ p, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0;
}

// copy between the caller's Pa x Pa factor and the padded n x n working factor (upper triangle only)
__global__ void fbr_tsqr_copy_kernel(int Pa, const double *__restrict__ src, int lds, double *__restrict__ dst, int ldd,
                                     int rows_dst, int cols_dst)
{
    const long total = (long)rows_dst * cols_dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_dst), c = (int)(i % cols_dst);
        double v = 0.0;
        if (r < Pa && c < Pa && c >= r) v = src[(long)r * lds + c];
        dst[(long)r * ldd + c] = v;
    }
}


#define TSQR_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            g_tsqr_err = std::string(#call) + ": " + hipGetErrorString(e__);             \
            return -3;                                                                   \
        }                                                                                \
    } while (0)

// samples per chunk so that the packed chunk stays around 4 GiB
static inline long fbr_tsqr_chunk_samples(int rows, int Pa)
{
    const int n = (Pa + 15) & ~15;
    const double per = (double)rows * n * 8.0;
    return std::max(1L, (long)(4.0 * 1024 * 1024 * 1024 / per));
}

// kernel instantiations: 8 waves per workgroup, one workgroup per CU; tiles per wave 1..6 (n <= 768); block rows
// 128 (n <= 256: the base regressor [YBase | tau] of WALK-MAN) / 64 / 48 / 32 so that the block (TPW * SUB * 8 VGPRs) and the panel
// chain fit the 256 VGPRs of a wave.  The panel chain costs ~120 VALU instructions per Householder step of which only 2 * 4 SUB
// depend on the block height, so taller blocks amortise it (fbr_tsqr_panel_step)
#define FBR_TSQR_DISPATCH(TPWV, CALL)                  \
    switch (TPWV) {                                    \
    case 1: { constexpr int TPW = 1, SUB = 4; CALL; } break; \
    case 2: { constexpr int TPW = 2, SUB = FBR_TSQR_SUB2; CALL; } break; \
    case 3: { constexpr int TPW = 3, SUB = FBR_TSQR_SUB3; CALL; } break; \
    case 4: { constexpr int TPW = 4, SUB = FBR_TSQR_SUB4; CALL; } break; \
    case 5: { constexpr int TPW = 5, SUB = 3; CALL; } break; \
    default: { constexpr int TPW = 6, SUB = 2; CALL; } break; \
    }
// four-wave workgroups, two per CU (9..20 column tiles: the base regressor of WALK-MAN, the waist chain's group): the fold of a wide
// factor is the serial sum of its panel chains with the MFMA pipes two thirds idle, so for the widths whose
// block fits HALF the register file at the same height two independent folds per CU run side by side
#define FBR_TSQR_HALF_WAVES 4
#define FBR_TSQR_HALF_MAX_TILES 20
// Wider than 20 tiles the four-wave workgroups keep the R rows in registers instead of LDS tiles (RREG) and fold 48- / 32-row blocks:
// two independent folds per CU for WALK-MAN's 31 tiles (6 tiles per wave: 21..24 column tiles, 8 per wave: 25..32)
#ifndef FBR_TSQR_SUB6H
#define FBR_TSQR_SUB6H 3
#endif
#ifndef FBR_TSQR_SUB8H
#define FBR_TSQR_SUB8H 2
#endif
#define FBR_TSQR_DUAL_MAX_TILES 32
#define FBR_TSQR_DISPATCH_HALF(TPWV, CALL)             \
    switch (TPWV) {                                    \
    case 3: { constexpr int TPW = 3, SUB = FBR_TSQR_SUB3; CALL; } break; \
    case 4: { constexpr int TPW = 4, SUB = FBR_TSQR_SUB4; CALL; } break; \
    default: { constexpr int TPW = 5, SUB = 3; CALL; } break; \
    }
static inline int fbr_tsqr_sub_for(int tpw) { return tpw == 2 ? FBR_TSQR_SUB2 : (tpw == 3 ? FBR_TSQR_SUB3 : (tpw == 4 ? FBR_TSQR_SUB4 : (tpw <= 1 ? 4 : (tpw == 5 ? 3 : 2)))); }

// narrow (wave-private) kernels: column tiles 1..8, 32-row blocks
#define FBR_TSQR_NARROW_MAX_TILES 8
#ifndef FBR_TSQR_NARROW_SUB_LE6
#define FBR_TSQR_NARROW_SUB_LE6 3  // <= 6 column tiles (left arm, KUKA): 48-row blocks still fit the 256 VGPRs of a wave
#endif
static inline int fbr_tsqr_narrow_sub_for(int npt) { return npt <= 6 ? FBR_TSQR_NARROW_SUB_LE6 : 2; }
#define FBR_TSQR_NARROW_TALL_SUB 6  // level-0 folds of long calls over <= 6 column tiles: 96-row blocks, one wave per SIMD (the trees keep the 48-row kernels)
#define FBR_TSQR_NARROW_DISPATCH_TALL(NPTV, CALL)      \
    switch (NPTV) {                                    \
    case 1: { constexpr int NPT = 1, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    case 2: { constexpr int NPT = 2, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    case 3: { constexpr int NPT = 3, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    case 4: { constexpr int NPT = 4, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    case 5: { constexpr int NPT = 5, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    default: { constexpr int NPT = 6, SUB = FBR_TSQR_NARROW_TALL_SUB; CALL; } break; \
    }
#define FBR_TSQR_NARROW_DISPATCH(NPTV, CALL)           \
    switch (NPTV) {                                    \
    case 1: { constexpr int NPT = 1, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 2: { constexpr int NPT = 2, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 3: { constexpr int NPT = 3, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 4: { constexpr int NPT = 4, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 5: { constexpr int NPT = 5, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 6: { constexpr int NPT = 6, SUB = FBR_TSQR_NARROW_SUB_LE6; CALL; } break; \
    case 7: { constexpr int NPT = 7, SUB = 2; CALL; } break; \
    default: { constexpr int NPT = 8, SUB = 2; CALL; } break; \
    }

// Kernel shape of a factorisation of width Pa over about rows_hint rows (shared by fbr_tsqr_begin and the work counter)
struct FbrTsqrShape {
    int n, tpw, sub, mb, NW, ld;
    bool narrow;
    int waves;  // waves per workgroup of the wide kernels (8, or 4 with two workgroups per CU)
    int ttpw, tsub, tmb;  // tiles per wave / block rows of the kernel that runs the merge tree (the eight-wave one for the RREG shapes)
};
static inline int fbr_tsqr_shape(int Pa, int num_cus, long rows_hint, FbrTsqrShape *out, const FbrTsqrOpts &opts = FbrTsqrOpts())
{
    const int n = (Pa + 15) & ~15;
    if (n > FBR_TSQR_MAXN) {
        g_tsqr_err = "TSQR supports at most " + std::to_string(FBR_TSQR_MAXN) + " columns";
        return -4;
    }
    const bool narrow = n / 16 <= FBR_TSQR_NARROW_MAX_TILES && opts.narrow;
    // (two 32- / 48-row folds per CU with the R rows through registers, for more than 20 column tiles: measured 12 % slower in round 4
    // -- 1.5 x the panel-chain work per row, 11 % more MFMAs, twice the R traffic per row -- and removed in round 5: DESIGN.md 10)
    const bool dual = false;
    const bool half = !narrow && n / 16 > FBR_TSQR_NARROW_MAX_TILES && n / 16 <= FBR_TSQR_HALF_MAX_TILES && !opts.timing;
    const int waves = half ? FBR_TSQR_HALF_WAVES : FBR_TSQR_WAVES;
    const int tpw8 = (n / 16 + FBR_TSQR_WAVES - 1) / FBR_TSQR_WAVES;
    const int tpw = narrow ? n / 16 : (dual ? (n / 16 <= 24 ? 6 : 8) : (n / 16 + waves - 1) / waves);
    // 96-row blocks at one wave per SIMD for narrow factors of at most 6 tiles, when every wave still gets a few blocks to fold
    const bool tall = narrow && opts.narrow_tall && tpw <= 6 && rows_hint >= 16L * FBR_TSQR_NARROW_TALL_SUB * 4 * (FBR_TSQR_NARROW_WAVES * (long)num_cus);
    const int sub = narrow ? (tall ? FBR_TSQR_NARROW_TALL_SUB : fbr_tsqr_narrow_sub_for(tpw)) : (dual ? (tpw == 6 ? FBR_TSQR_SUB6H : FBR_TSQR_SUB8H) : fbr_tsqr_sub_for(tpw));
    const int mb = 16 * sub;
    const long want = (rows_hint + mb - 1) / mb;
    const long per_cu = narrow ? (tall ? 1 : 2) * FBR_TSQR_NARROW_WAVES : (half ? 2 : 1);  // narrow: private R per WAVE
    const int NW = (int)std::max(1L, std::min<long>(per_cu * num_cus, want));
    const int ld = narrow ? n : 16 * waves * tpw;
    // (the RREG shapes share their leading dimension with the eight-wave kernel of the same width: 16 x 4 x 6 = 16 x 8 x 3, 16 x 4 x 8 = 16 x 8 x 4)
    // (tall narrow shape: the merge trees run the 48-row wave-private kernels)
    const int tsub = dual ? fbr_tsqr_sub_for(tpw8) : (tall ? fbr_tsqr_narrow_sub_for(tpw) : sub);
    *out = FbrTsqrShape{n, tpw, sub, mb, NW, ld, narrow, waves, dual ? tpw8 : tpw, tsub, 16 * tsub};
    return 0;
}

// Start a factorisation of width Pa: working factors zeroed, R_in (device, Pa x Pa, may be null) seeded into slot 0.
static inline int fbr_tsqr_begin(FbrTsqrWork &wk, hipStream_t st, int Pa, const double *R_in, int num_cus, long rows_hint, unsigned *shared_err = nullptr)
{
    FbrTsqrShape sh;
    if (int rc = fbr_tsqr_shape(Pa, num_cus, rows_hint, &sh, wk.opts)) return rc;
    const int n = sh.n, tpw = sh.tpw, sub = sh.sub, mb = sh.mb, NW = sh.NW, ld = sh.ld;
    const bool narrow = sh.narrow;
    const size_t need = (size_t)NW * n * ld * sizeof(double);
    if (need > wk.rw_bytes) {
        if (wk.Rw) (void)hipFree(wk.Rw);
        wk.Rw = nullptr;
        wk.rw_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.Rw, need));
        wk.rw_bytes = need;
    }
    wk.n = n; wk.ld = ld; wk.NW = NW; wk.Pa = Pa; wk.mb = mb; wk.tpw = tpw; wk.sub = sub; wk.narrow = narrow; wk.waves = sh.waves; wk.ttpw = sh.ttpw;
    if (shared_err) {
        if (wk.err && wk.own_err) (void)hipFree(wk.err);
        wk.err = shared_err;  // (cleared by the caller once per call: several factorisations of one call report into it)
        wk.own_err = false;
    } else {
        if (!wk.err || !wk.own_err) {
            wk.err = nullptr;
            wk.own_err = true;
            TSQR_HIP(hipMalloc((void **)&wk.err, sizeof(unsigned)));
        }
        TSQR_HIP(hipMemsetAsync(wk.err, 0, sizeof(unsigned), st));
    }
    TSQR_HIP(hipMemsetAsync(wk.Rw, 0, need, st));
    if (R_in) {
        hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, Pa, R_in, Pa, wk.Rw, ld, n, ld);
        TSQR_HIP(hipGetLastError());
    }
    wk.active = true;
    return 0;
}

// Packed chunk of the factorisation for M rows ([Mpad][n] doubles, Mpad = M rounded up to 16): (re)allocated on demand.
static inline int fbr_tsqr_chunk_buffer(FbrTsqrWork &wk, long M, double **A)
{
    if (!wk.active) {
        g_tsqr_err = "tsqr chunk without begin";
        return -1;
    }
    const long Mpad = (M + 15) & ~15L;
    const size_t need = (size_t)std::max(Mpad, 16L) * wk.n * sizeof(double);
    if (need > wk.a_bytes) {
        if (wk.A) (void)hipFree(wk.A);
        wk.A = nullptr;
        wk.a_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.A, need));
        wk.a_bytes = need;
        wk.clean_key = -1;
    }
    *A = wk.A;
    return 0;
}

// Before a writer fills the chunk in place for fbr_tsqr_fold_chunk(k = 0): make the padding columns zero once per (buffer, width).
static inline int fbr_tsqr_chunk_clean(FbrTsqrWork &wk, hipStream_t st)
{
    const long key = (long)wk.Pa * 4096 + wk.n;
    if (wk.clean_key == key || !wk.A) return 0;
    TSQR_HIP(hipMemsetAsync(wk.A, 0, wk.a_bytes, st));
    wk.clean_key = key;
    return 0;
}

// level 0 over the packed chunk wk.A (M rows)
// Asrc (optional): another packed chunk than wk.A.  slot_stride / max_wgs: the blocks are dealt to at most max_wgs workgroups and
// workgroup w folds into working factor w * slot_stride (wide kernels only: rows folded into the factors still alive inside the tree).
static inline int fbr_tsqr_fold_packed(FbrTsqrWork &wk, hipStream_t st, long M, const FbrTsqrRowOrder &ro = FbrTsqrRowOrder(),
                                       const double *Asrc = nullptr, int slot_stride = 1, int max_wgs = 0)
{
    const int n = wk.n;
    const long Mpad = (M + 15) & ~15L;
    const long nblocks = (Mpad + wk.mb - 1) / wk.mb;
    const double *A = Asrc ? Asrc : wk.A;
    if (wk.narrow) {
        if (slot_stride != 1 || max_wgs) {
            g_tsqr_err = "slot stride is not supported by the wave-private kernels";
            return -1;
        }
        const int nwaves = (int)std::min<long>(wk.NW, nblocks);
        const int grid = (nwaves + FBR_TSQR_NARROW_WAVES - 1) / FBR_TSQR_NARROW_WAVES;
        if (wk.sub == FBR_TSQR_NARROW_TALL_SUB) {
            FBR_TSQR_NARROW_DISPATCH_TALL(wk.tpw, {
                constexpr size_t lb = FBR_TSQR_NARROW_WAVES * fbr_tsqr_narrow_lds_doubles<SUB>() * sizeof(double);
                (void)hipFuncSetAttribute((const void *)fbr_tsqr_narrow_level0_kernel<NPT, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((fbr_tsqr_narrow_level0_kernel<NPT, SUB>), dim3(grid), dim3(FBR_TSQR_NARROW_WAVES * 64), lb, st, A, Mpad, wk.Rw, nblocks, nwaves,
                                   ro.first_col, ro.rows, ro.group, M, ro.colmajor_ld);
            });
            TSQR_HIP(hipGetLastError());
            return 0;
        }
        FBR_TSQR_NARROW_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_narrow_level0_kernel<NPT, SUB>), dim3(grid), dim3(FBR_TSQR_NARROW_WAVES * 64),
                                                            (FBR_TSQR_NARROW_WAVES * fbr_tsqr_narrow_lds_doubles<SUB>() * sizeof(double)), st, A, Mpad,
                                                            wk.Rw, nblocks, nwaves, ro.first_col, ro.rows, ro.group, M, ro.colmajor_ld));
        TSQR_HIP(hipGetLastError());
        return 0;
    }
    const int grid = (int)std::min<long>(max_wgs > 0 ? std::min(max_wgs, wk.NW) : wk.NW, nblocks);
    unsigned long long *dbg = nullptr;
    if (wk.waves == FBR_TSQR_HALF_WAVES) {
        constexpr int HW = FBR_TSQR_HALF_WAVES;
        FBR_TSQR_DISPATCH_HALF(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_level0_kernel<TPW, SUB, false, HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 (int)(fbr_tsqr_lds_doubles<TPW, SUB, HW>() * sizeof(double))));
        FBR_TSQR_DISPATCH_HALF(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_level0_kernel<TPW, SUB, false, HW>), dim3(grid), dim3(64 * HW),
                                                          (fbr_tsqr_lds_doubles<TPW, SUB, HW>() * sizeof(double)), st, A, Mpad, n, wk.Rw, nblocks, wk.err, dbg, ro.first_col, ro.rows, ro.group, M, slot_stride, ro.colmajor_ld));
        TSQR_HIP(hipGetLastError());
        return 0;
    }
    if (wk.opts.timing) {
        TSQR_HIP(hipMalloc((void **)&dbg, (size_t)grid * FBR_TSQR_WAVES * 16 * 8));
        TSQR_HIP(hipMemsetAsync(dbg, 0, (size_t)grid * FBR_TSQR_WAVES * 16 * 8, st));
    }
    if (dbg) {
        FBR_TSQR_DISPATCH(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_level0_kernel<TPW, SUB, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            (int)(fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double))));
        FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_level0_kernel<TPW, SUB, true>), dim3(grid), dim3(FBR_TSQR_THREADS),
                                                     (fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double)), st, A, Mpad, n, wk.Rw, nblocks, wk.err, dbg, ro.first_col, ro.rows, ro.group, M, slot_stride, ro.colmajor_ld));
    } else {
        FBR_TSQR_DISPATCH(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_level0_kernel<TPW, SUB, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            (int)(fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double))));
        FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_level0_kernel<TPW, SUB, false>), dim3(grid), dim3(FBR_TSQR_THREADS),
                                                     (fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double)), st, A, Mpad, n, wk.Rw, nblocks, wk.err, dbg, ro.first_col, ro.rows, ro.group, M, slot_stride, ro.colmajor_ld));
    }
    TSQR_HIP(hipGetLastError());
    if (dbg) {
        std::vector<unsigned long long> hb((size_t)grid * FBR_TSQR_WAVES * 16);
        TSQR_HIP(hipMemcpyAsync(hb.data(), dbg, hb.size() * 8, hipMemcpyDeviceToHost, st));
        TSQR_HIP(hipStreamSynchronize(st));
        double sum[16] = {0};
        for (size_t i = 0; i < hb.size(); i++) sum[i & 15] += (double)hb[i];
        const double folds = (double)nblocks * FBR_TSQR_WAVES;
        fprintf(stderr, "[fbr tsqr timing] cycles per fold per wave: load+init=%.0f chain=%.0f (before steps %.0f, 16 steps %.0f, ring wait %.0f, up to V/T written %.0f, order wait %.0f) wait_panel=%.0f update=%.0f  (mb=%d n=%d, %ld folds)\n",
                sum[0] / folds, sum[1] / folds, sum[7] / folds, sum[4] / folds, sum[5] / folds, sum[8] / folds, sum[6] / folds, sum[2] / folds, sum[3] / folds, wk.mb, n, nblocks);
        (void)hipFree(dbg);
    }
    return 0;
}

// Fold M rows of [Y (M x P) | rhs (M x k)] (row weights w optional, column gather optional) into the working factors.
static inline int fbr_tsqr_fold_rows(FbrTsqrWork &wk, hipStream_t st, long M, int P, const double *Y, int k, const double *rhs,
                                     const double *w, int ldy = 0, const int *cols = nullptr, const FbrTsqrRowOrder &ro = FbrTsqrRowOrder())
{
    if (ldy <= 0) ldy = P;
    if (!wk.active || P + k != wk.Pa) {
        g_tsqr_err = "tsqr fold without matching begin";
        return -1;
    }
    if (M <= 0) return 0;
    double *A = nullptr;
    int rc = fbr_tsqr_chunk_buffer(wk, M, &A);
    if (rc) return rc;
    const long Mpad = (M + 15) & ~15L;
    hipLaunchKernelGGL(fbr_tsqr_pack_kernel, dim3(2048), dim3(256), 0, st, M, Mpad, P, k, wk.n, Y, ldy, cols, rhs, w, A, ro.rows, ro.group);
    TSQR_HIP(hipGetLastError());
    return fbr_tsqr_fold_packed(wk, st, M, ro);  // (the pack kernel writes zeros into the padding columns: a clean buffer of this width stays clean)
}

// Fold the chunk whose first P columns were already written into fbr_tsqr_chunk_buffer() (leading dimension wk.n):
// append the rhs columns and the zero padding, then level 0.
static inline int fbr_tsqr_fold_chunk(FbrTsqrWork &wk, hipStream_t st, long M, int P, int k, const double *rhs,
                                      const FbrTsqrRowOrder &ro = FbrTsqrRowOrder())
{
    if (!wk.active || P + k != wk.Pa) {
        g_tsqr_err = "tsqr fold without matching begin";
        return -1;
    }
    if (M <= 0) return 0;
    const long Mpad = (M + 15) & ~15L;
    if (ro.colmajor_ld > 0) {
        // column-major chunk of a lane writer: every column < Pa written, padding columns and padding rows cleared by the writer's own
        // clear kernel (fbr_groups_clear_cm_kernel); whole 16-row groups only
        if (k != 0 || Mpad != M) {
            g_tsqr_err = "column-major chunks carry their rhs columns and whole 16-row groups";
            return -1;
        }
        wk.clean_key = -1;
    } else if (k == 0) {
        // the writer has stored every column < Pa (rhs columns included): what is left are the zero padding columns -- written once per
        // buffer instead of once per chunk (WALK-MAN: 15 columns x 6 M base-wrench rows per 1 M samples) -- and the rows M..Mpad
        const long key = (long)wk.Pa * 4096 + wk.n;
        if (wk.clean_key != key) {
            TSQR_HIP(hipMemsetAsync(wk.A, 0, wk.a_bytes, st));
            g_tsqr_err = "chunk buffer cleared after it was filled";  // (callers clear BEFORE the writer runs: fbr_tsqr_chunk_clean)
            return -1;
        }
        if (Mpad > M) TSQR_HIP(hipMemsetAsync(wk.A + M * wk.n, 0, (size_t)(Mpad - M) * wk.n * sizeof(double), st));
    } else {
        hipLaunchKernelGGL(fbr_tsqr_tail_kernel, dim3(1024), dim3(256), 0, st, M, Mpad, P, k, wk.n, rhs, wk.A, ro.rows, ro.group);
        TSQR_HIP(hipGetLastError());
        wk.clean_key = -1;
    }
    return fbr_tsqr_fold_packed(wk, st, M, ro);
}

// Binary tree over the working factors, result (Pa x Pa, upper triangular) to R_out (device): enqueued on st, not waited for
// (several factorisations can run their latency-bound trees on different streams; fbr_tsqr_check() collects the error word).
// Levels of the binary tree with stride_from <= stride < stride_to (strides are powers of two; the factors alive after the level of
// stride s are the slots that are multiples of 2 s).
// brows > 0 (one level over two working factors only): the partner slot holds `brows` dense rows instead of a triangular factor
static inline int fbr_tsqr_tree_levels(FbrTsqrWork &wk, hipStream_t st, int stride_from, int stride_to, int brows = 0)
{
    if (!wk.active) {
        g_tsqr_err = "tsqr finish without begin";
        return -1;
    }
    const int n = wk.n;
    stride_to = std::min(stride_to, wk.NW);
    // dense partner rows are understood by the cross-workgroup merge kernel only (fbr_tsqr_tree_x_kernel); every other tree kernel would
    // fold the partner slot as a triangular factor and silently return a wrong R
    const bool x_kernel = !wk.narrow && !wk.opts.tree_one_wg && wk.n / 16 > FBR_TSQR_NARROW_MAX_TILES && (wk.waves != FBR_TSQR_HALF_WAVES || wk.ttpw == wk.tpw);
    if (brows > 0 && !x_kernel) {
        g_tsqr_err = "dense partner rows need the cross-workgroup merge kernel (wide factor, tsqr_tree_one_wg off)";
        return -1;
    }
    if (wk.narrow) {
        // 128-column factors (8 tiles: the leading dimension of the wave-private layout equals that of the eight-wave kernels with one
        // tile per wave): once a level has no more merges than CUs, a merge is folded by a whole workgroup -- panel chain on one wave, the
        // seven tile updates beside it -- instead of by one wave: the late levels of the tree are pure latency (one merge = 125 ... 170 us
        // by a single wave; eleven levels over 2048 private factors were 1.8 ms at the end of a short call)
        int cus = 0, dev = 0;
        TSQR_HIP(hipGetDevice(&dev));
        TSQR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const bool coop = wk.tpw == FBR_TSQR_NARROW_MAX_TILES && wk.opts.short_calls;
        if (coop)
            TSQR_HIP(hipFuncSetAttribute((const void *)fbr_tsqr_tree_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(fbr_tsqr_lds_doubles<1, 4>() * sizeof(double))));
        for (int stride = stride_from; stride < stride_to; stride *= 2) {
            const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
            if (coop && pairs <= std::max(cus, 1)) {
                hipLaunchKernelGGL((fbr_tsqr_tree_kernel<1, 4>), dim3(pairs), dim3(FBR_TSQR_THREADS), (fbr_tsqr_lds_doubles<1, 4>() * sizeof(double)), st, wk.Rw, n, stride,
                                   wk.NW, wk.err);
                TSQR_HIP(hipGetLastError());
                continue;
            }
            const int grid = (pairs + FBR_TSQR_NARROW_WAVES - 1) / FBR_TSQR_NARROW_WAVES;
            FBR_TSQR_NARROW_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_narrow_tree_kernel<NPT, SUB>), dim3(grid), dim3(FBR_TSQR_NARROW_WAVES * 64),
                                                                (FBR_TSQR_NARROW_WAVES * fbr_tsqr_narrow_lds_doubles<SUB>() * sizeof(double)), st, wk.Rw,
                                                                stride, wk.NW));
            TSQR_HIP(hipGetLastError());
        }
    } else if (wk.waves == FBR_TSQR_HALF_WAVES && wk.ttpw == wk.tpw && !wk.opts.tree_one_wg) {
        // four-wave shapes (two workgroups per CU): the same cross-workgroup merge pipeline (fbr_tsqr_tree_x_kernel)
        constexpr int HW = FBR_TSQR_HALF_WAVES;
        const int nblk = ((brows > 0 ? brows : n) + wk.mb - 1) / wk.mb;
        auto level_ints = [&](int stride) { const size_t pr = (wk.NW + 2 * stride - 1) / (2 * stride); return pr * nblk * HW + pr; };  // flags + claim counters
        size_t all_ints = 0;
        for (int stride = 1; stride < wk.NW; stride *= 2) all_ints += level_ints(stride);
        const size_t need = std::max<size_t>(all_ints, 1) * sizeof(int);
        if (need > wk.prog_bytes) {
            if (wk.prog) (void)hipFree(wk.prog);
            wk.prog = nullptr;
            wk.prog_bytes = 0;
            TSQR_HIP(hipMalloc((void **)&wk.prog, need));
            wk.prog_bytes = need;
        }
        int cus = 0, dev = 0;
        TSQR_HIP(hipGetDevice(&dev));
        TSQR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        FBR_TSQR_DISPATCH_HALF(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_tree_x_kernel<TPW, SUB, HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 (int)(fbr_tsqr_lds_doubles<TPW, SUB, HW, true>() * sizeof(double))));
        if (stride_from == 1) {
            hipLaunchKernelGGL(fbr_tsqr_zero_kernel, dim3((unsigned)std::min<size_t>((need / sizeof(int) + 255) / 256, 1024)), dim3(256), 0, st, wk.prog, (long)(need / sizeof(int)));
            TSQR_HIP(hipGetLastError());
        }
        size_t off = 0;
        for (int stride = 1; stride < stride_from; stride *= 2) off += level_ints(stride);
        for (int stride = stride_from; stride < stride_to; stride *= 2) {
            const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
            int G = 1;
            while (G < 8 && 2 * G <= nblk && pairs * 2 * G <= 2 * std::max(cus, 1)) G *= 2;
            FBR_TSQR_DISPATCH_HALF(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_tree_x_kernel<TPW, SUB, HW>), dim3(pairs * G), dim3(64 * HW),
                                                              (fbr_tsqr_lds_doubles<TPW, SUB, HW, true>() * sizeof(double)), st, wk.Rw, n, stride, wk.NW, G,
                                                              wk.prog + off, wk.err, brows));
            off += level_ints(stride);
            TSQR_HIP(hipGetLastError());
        }
    } else if (wk.waves == FBR_TSQR_HALF_WAVES && wk.ttpw == wk.tpw) {
        constexpr int HW = FBR_TSQR_HALF_WAVES;
        FBR_TSQR_DISPATCH_HALF(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_tree_kernel<TPW, SUB, HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 (int)(fbr_tsqr_lds_doubles<TPW, SUB, HW>() * sizeof(double))));
        for (int stride = stride_from; stride < stride_to; stride *= 2) {
            const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
            FBR_TSQR_DISPATCH_HALF(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_tree_kernel<TPW, SUB, HW>), dim3(pairs), dim3(64 * HW),
                                                              (fbr_tsqr_lds_doubles<TPW, SUB, HW>() * sizeof(double)), st, wk.Rw, n, stride, wk.NW, wk.err));
            TSQR_HIP(hipGetLastError());
        }
    } else {
    // (wk.ttpw: the RREG shapes run their merge tree on the eight-wave kernel of the same leading dimension: taller blocks, half the
    // serial panel steps per merge)
    if (!wk.opts.tree_one_wg && wk.n / 16 > FBR_TSQR_NARROW_MAX_TILES) {
        // merges pipelined across workgroups (fbr_tsqr_tree_x_kernel): as many workgroups per merge as keep the level's grid within one
        // round of CUs (2 at the 128 merges of level 1, 4 at 64, 8 from 32 merges on)
        const int sub_t = fbr_tsqr_sub_for(wk.ttpw), nblk = ((brows > 0 ? brows : n) + 16 * sub_t - 1) / (16 * sub_t);
        // progress flags + claim counters of EVERY level, one region per level, cleared once in front of the first level (a clear per
        // level was a 20 - 40 us launch on the tree's critical path each time)
        auto level_ints = [&](int stride) { const size_t pr = (wk.NW + 2 * stride - 1) / (2 * stride); return pr * nblk * FBR_TSQR_WAVES + pr; };
        size_t all_ints = 0;
        for (int stride = 1; stride < wk.NW; stride *= 2) all_ints += level_ints(stride);
        const size_t need = std::max<size_t>(all_ints, 1) * sizeof(int);
        if (need > wk.prog_bytes) {
            if (wk.prog) (void)hipFree(wk.prog);
            wk.prog = nullptr;
            wk.prog_bytes = 0;
            TSQR_HIP(hipMalloc((void **)&wk.prog, need));
            wk.prog_bytes = need;
        }
        int cus = 0, dev = 0;
        TSQR_HIP(hipGetDevice(&dev));
        TSQR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        FBR_TSQR_DISPATCH(wk.ttpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_tree_x_kernel<TPW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                            (int)(fbr_tsqr_lds_doubles<TPW, SUB, FBR_TSQR_WAVES, true>() * sizeof(double))));
        if (stride_from == 1) {
            hipLaunchKernelGGL(fbr_tsqr_zero_kernel, dim3((unsigned)std::min<size_t>((need / sizeof(int) + 255) / 256, 1024)), dim3(256), 0, st, wk.prog, (long)(need / sizeof(int)));
            TSQR_HIP(hipGetLastError());
        }
        size_t off = 0;
        for (int stride = 1; stride < stride_from; stride *= 2) off += level_ints(stride);
        for (int stride = stride_from; stride < stride_to; stride *= 2) {
            const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
            int G = 1;
            while (G < 8 && 2 * G <= nblk && pairs * 2 * G <= std::max(cus, 1)) G *= 2;
            FBR_TSQR_DISPATCH(wk.ttpw, hipLaunchKernelGGL((fbr_tsqr_tree_x_kernel<TPW, SUB>), dim3(pairs * G), dim3(FBR_TSQR_THREADS),
                                                          (fbr_tsqr_lds_doubles<TPW, SUB, FBR_TSQR_WAVES, true>() * sizeof(double)), st, wk.Rw, n, stride, wk.NW, G,
                                                          wk.prog + off, wk.err, brows));
            off += level_ints(stride);
            TSQR_HIP(hipGetLastError());
        }
        return 0;
    }
    FBR_TSQR_DISPATCH(wk.ttpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_tree_kernel<TPW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)(fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double))));
    for (int stride = stride_from; stride < stride_to; stride *= 2) {
        const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
        FBR_TSQR_DISPATCH(wk.ttpw, hipLaunchKernelGGL((fbr_tsqr_tree_kernel<TPW, SUB>), dim3(pairs), dim3(FBR_TSQR_THREADS),
                                                     (fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double)), st, wk.Rw, n, stride, wk.NW, wk.err));
        TSQR_HIP(hipGetLastError());
    }
    }
    return 0;
}

// Working factor 0 (the root of the tree) to R_out (device, Pa x Pa): ends the factorisation.
static inline int fbr_tsqr_copy_out(FbrTsqrWork &wk, hipStream_t st, double *R_out)
{
    hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, wk.Pa, wk.Rw, wk.ld, R_out, wk.Pa, wk.Pa, wk.Pa);
    TSQR_HIP(hipGetLastError());
    wk.active = false;
    return 0;
}

static inline int fbr_tsqr_finish_async(FbrTsqrWork &wk, hipStream_t st, double *R_out)
{
    if (int rc = fbr_tsqr_tree_levels(wk, st, 1, 1 << 30)) return rc;
    return fbr_tsqr_copy_out(wk, st, R_out);
}

// Trees of several narrow factorisations of one shape (same width, same number of private factors) level by level in shared launches.
static inline int fbr_tsqr_finish_narrow_batch(FbrTsqrWork **wks, int nw, hipStream_t st, double **R_out)
{
    FbrTsqrWork &w0 = *wks[0];
    FbrTsqrNarrowBatch bt;
    for (int i = 0; i < FBR_TSQR_NARROW_BATCH; i++) bt.Rw[i] = wks[std::min(i, nw - 1)]->Rw;
    for (int stride = 1; stride < w0.NW; stride *= 2) {
        const int pairs = (w0.NW + 2 * stride - 1) / (2 * stride);
        const int grid = (pairs + FBR_TSQR_NARROW_WAVES - 1) / FBR_TSQR_NARROW_WAVES;
        FBR_TSQR_NARROW_DISPATCH(w0.tpw, hipLaunchKernelGGL((fbr_tsqr_narrow_tree_batch_kernel<NPT, SUB>), dim3(grid, nw), dim3(FBR_TSQR_NARROW_WAVES * 64),
                                                            (FBR_TSQR_NARROW_WAVES * fbr_tsqr_narrow_lds_doubles<SUB>() * sizeof(double)), st, bt, stride, w0.NW));
        TSQR_HIP(hipGetLastError());
    }
    for (int i = 0; i < nw; i++)
        if (int rc = fbr_tsqr_copy_out(*wks[i], st, R_out[i])) return rc;
    return 0;
}

// Wait for the stream and report a pipeline time-out of the factorisation's kernels (the device error word).
static inline int fbr_tsqr_check(FbrTsqrWork &wk, hipStream_t st)
{
    unsigned herr = 0;
    TSQR_HIP(hipMemcpyAsync(&herr, wk.err, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    TSQR_HIP(hipStreamSynchronize(st));
    if (herr) {
        g_tsqr_err = "TSQR pipeline flag wait timed out (internal error)";
        return -5;
    }
    return 0;
}

// tree + wait + error check
static inline int fbr_tsqr_finish(FbrTsqrWork &wk, hipStream_t st, double *R_out)
{
    if (int rc = fbr_tsqr_finish_async(wk, st, R_out)) return rc;
    return fbr_tsqr_check(wk, st);
}
