// fbr_tsqr.h -- blocked Householder TSQR on gfx950 (fp64, MFMA trailing update).
//
// R^T R = A^T A for a tall row stream A (rows = samples x N_OUT, columns = [Y | rhs], padded to a
// multiple of 16), WITHOUT forming A^T A (no squaring of the condition number):
//
//   level 0  every workgroup w owns a private upper-triangular R_w (global memory, L2/MALL resident) and folds
//            its share of the 64-row blocks into it with a triangular-pentagonal Householder QR ("TPQRT": QR of
//            [R_w ; B], only R_w's panel rows and the dense block take part).  The block lives in the VGPRs of the
//            workgroup's 8 waves for the whole fold (see fbr_tsqr_fold_regs) -- only R_w is streamed:
//              panel (16 columns): owner wave, in registers; T from V^T V by MFMA + 16x16 triangular recurrence
//              trailing update:    W = T^T (R_rows + V^T C),  R_rows -= W,  C -= V W on v_mfma_f64_16x16x4_f64
//   level 1+ binary tree over the R_w (one launch per level; the partner's R is folded in 64-row chunks, panels
//            left of a chunk's first non-zero column are skipped).  Across ranks the same merge runs on R factors
//            exchanged over xGMI (flobaroid_amd/dist.py).
//
// Everything is deterministic (fixed block -> workgroup assignment, fixed tree).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#define FBR_TSQR_THREADS 256
#define FBR_TSQR_WAVES (FBR_TSQR_THREADS / 64)
#define FBR_TSQR_MAXN 768          // widest supported factor (columns incl. rhs, padded to 16)

typedef double fbr_td4 __attribute__((ext_vector_type(4)));

static thread_local std::string g_tsqr_err;
static inline const char *fbr_tsqr_error() { return g_tsqr_err.c_str(); }

// A[r][c] (ld) = w[r] * [Y | rhs][r][c], zero in the padding columns / rows
// cols (optional, device): gather columns cols[0..P) of a Y with leading dimension ldy
__global__ __launch_bounds__(256) void fbr_tsqr_pack_kernel(long M, long Mpad, int P, int k, int ld,
                                                             const double *__restrict__ Y, int ldy, const int *__restrict__ cols,
                                                             const double *__restrict__ rhs, const double *__restrict__ w,
                                                             double *__restrict__ A)
{
    const long total = Mpad * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ld;
        const int c = (int)(i - r * ld);
        double v = 0.0;
        if (r < M) {
            if (c < P)
                v = Y[r * ldy + (cols ? cols[c] : c)];
            else if (c < P + k)
                v = rhs[r * k + (c - P)];
            if (w) v *= w[r];
        }
        A[i] = v;
    }
}

// Register-resident TPQRT.  A workgroup of 8 waves folds a block of MB = 16*SUB rows into its private R:
//   * wave w owns the column tiles ct = w, w+8, ... (TPW per wave); the whole block lives in VGPRs in the MFMA
//     C/D layout (lane (kk, j) of tile/sub-tile holds row 16*sub + 4*reg + kk, column j), so
//       - V^T C consumes the block straight from the registers as the B operand (k-step = (sub, reg)),
//       - C -= V W accumulates straight into them;
//   * the panel (tile p) is factorised by its owner wave in registers (cross-lane broadcasts, no barrier per
//     column), the scaled Householder vectors V (MB x 16) and T (16 x 16) are published through LDS;
//   * only R (the panel's 16 rows) is streamed from global memory: read once, written once per fold.
#define FBR_TSQR_LDV 17  // LDS row stride of the published V panel (conflict-free for both operand walks)

// LDS carve (doubles): Rl[WAVES*TPW tiles][256] | Vl[MB*17] | Tm[256] | Rp[256]
template <int TPW, int SUB> static inline size_t fbr_tsqr_lds_doubles()
{
    return (size_t)FBR_TSQR_WAVES * TPW * 256 + (size_t)16 * SUB * FBR_TSQR_LDV + 512;
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

// acc += trow[l] * (z of lane l), l = 0..J-1   (column J of the T recurrence; lane i holds row i of T)
template <int J, int L = 0> struct FbrTAcc {
    static __device__ __forceinline__ void run(const double (&trow)[16], double z, double &a0, double &a1)
    {
        if constexpr (L < J) {
            if constexpr (L & 1)
                a1 += trow[L] * fbr_dpp_bcast<L>(z);
            else
                a0 += trow[L] * fbr_dpp_bcast<L>(z);
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
                                                    double &myscale, int li, int kk)
{
    const double rjc = Rp[J * 16 + li];
    const double alpha = fbr_dpp_bcast<J>(rjc);
    // s_c = x . B[:, c] with x = B[:, J] (lane J's registers); lane J's own s is |x|^2
    fbr_td4 x[SUB];
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg += 2) {
            x[sb][reg] = fbr_dpp_bcast<J>(v[sb][reg]);
            x[sb][reg + 1] = fbr_dpp_bcast<J>(v[sb][reg + 1]);
            s0 += x[sb][reg] * v[sb][reg];
            s1 += x[sb][reg + 1] * v[sb][reg + 1];
        }
    double s = fbr_xor32_sum(fbr_xor16_sum(s0 + s1));
    const double normsq = fbr_dpp_bcast<J>(s);
    // branch-free (a zero column gives tau = scale = 0, beta = alpha); two independent reciprocals (hardware seed +
    // 2 Newton steps) instead of two divisions
    const bool nz = normsq > 0.0;
    const double bet = -copysign(sqrt(alpha * alpha + normsq), alpha);
    const double d1 = alpha - bet;
    double r1 = __builtin_amdgcn_rcp(d1), r2 = __builtin_amdgcn_rcp(bet);
    r1 = r1 * (2.0 - d1 * r1);
    r2 = r2 * (2.0 - bet * r2);
    r1 = r1 * (2.0 - d1 * r1);
    r2 = r2 * (2.0 - bet * r2);
    const double scale = nz ? r1 : 0.0;
    const double tau = nz ? -d1 * r2 : 0.0;
    const double beta = nz ? bet : alpha;
    // column J of T: Z[l][J] = v_l . v_J = myscale_l * scale * s_l sits on lane l (l < J)
    {
        double a0 = 0.0, a1 = 0.0;
        FbrTAcc<J>::run(trow, scale * s * myscale, a0, a1);
        trow[J] = (li == J) ? tau : ((li < J) ? -tau * (a0 + a1) : 0.0);
    }
    const double wc = rjc + scale * s;
    const double g = (li > J) ? tau * wc * scale : 0.0;
#pragma unroll
    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) v[sb][reg] -= g * x[sb][reg];
    if (li == J) myscale = scale;
    const double newr = (li > J) ? rjc - tau * wc : beta;
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
    (fbr_tsqr_panel_step<SUB, Js>(v, Rp, rq, trow, myscale, li, kk), ...);
}

// Trailing update of a wave's column tiles t >= t0 with the published panel (V in Vl, T in Tm), two tiles at a time:
//   acc = R_rows + V^T C;  W = T^T acc;  R_rows -= W;  C -= V W        (Rl = LDS copy of the R_rows tiles, [tile][16][16])
// The MFMA C/D layout (reg r, lane (kk, j) = row 4r + kk, column j) is also the B-operand layout of k-step r, so acc
// and W feed the next product straight from the accumulator registers -- no LDS round trip, no barrier.
// Tiles are indexed statically (one code path, pairs skipped by a uniform branch): a pair that straddles t0 also runs
// its dead left tile (already consumed as a panel) through the MFMAs, only its R store is suppressed.
template <int TPW, int SUB, int T = 0> struct FbrTsqrUpdate {
    static __device__ __forceinline__ void run(int t0, fbr_td4 (&C)[TPW][SUB], const double *Rl, double *__restrict__ R, unsigned ld, unsigned j0,
                                               int wave, int li, int kk, const double *Vl, const double *Tm)
    {
        if constexpr (T < TPW) {
            constexpr int NB = (T + 1 < TPW) ? 2 : 1;
            if (T + NB - 1 >= t0) {
                fbr_td4 acc[NB], w2[NB];
#pragma unroll
                for (int b = 0; b < NB; b++) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) acc[b][reg] = Rl[(wave + FBR_TSQR_WAVES * (T + b)) * 256 + (4 * reg + kk) * 16 + li];
                    w2[b] = fbr_td4{0.0, 0.0, 0.0, 0.0};
                }
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const double a = Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li];
#pragma unroll
                        for (int b = 0; b < NB; b++) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, C[T + b][sb][reg], acc[b], 0, 0, 0);
                    }
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const double a = Tm[(4 * ks + kk) * 16 + li];
#pragma unroll
                    for (int b = 0; b < NB; b++) w2[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[b][ks], w2[b], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < NB; b++)
                    if (T + b >= t0) {
                        const unsigned c0 = 16u * (unsigned)(wave + FBR_TSQR_WAVES * (T + b));
#pragma unroll
                        for (int reg = 0; reg < 4; reg++)
                            R[(j0 + 4 * reg + kk) * ld + c0 + li] = Rl[(wave + FBR_TSQR_WAVES * (T + b)) * 256 + (4 * reg + kk) * 16 + li] - w2[b][reg];
                    }
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        const double a = Vl[(16 * sb + li) * FBR_TSQR_LDV + 4 * ks + kk];
#pragma unroll
                        for (int b = 0; b < NB; b++) C[T + b][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, -w2[b][ks], C[T + b][sb], 0, 0, 0);
                    }
            }
            FbrTsqrUpdate<TPW, SUB, T + NB>::run(t0, C, Rl, R, ld, j0, wave, li, kk, Vl, Tm);
        }
    }
};

template <int TPW, int SUB>
__device__ __forceinline__ void fbr_tsqr_fold_regs(double *__restrict__ R, int n, int ldr, const double *__restrict__ B, int ldb, int mrows, int first_col,
                                   double *smem, unsigned long long *tacc = nullptr)
{
    unsigned long long tk = tacc ? __builtin_readcyclecounter() : 0;
#define FBR_TT(i)                                                         \
    if (tacc) {                                                           \
        const unsigned long long t1 = __builtin_readcyclecounter();       \
        tacc[i] += t1 - tk;                                               \
        tk = t1;                                                          \
    }
    constexpr int MB = 16 * SUB;
    double *Rl = smem;  // 16-byte aligned tiles for the LDS-DMA
    double *Vl = Rl + FBR_TSQR_WAVES * TPW * 256;
    double *Tm = Vl + MB * FBR_TSQR_LDV;
    double *Rp = Tm + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int NP = n / 16;
    const unsigned ld = (unsigned)ldr;

    // ---- load this wave's tiles of the block (rows >= mrows are zero)
    fbr_td4 C[TPW][SUB];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int ct = wave + FBR_TSQR_WAVES * t;
#pragma unroll
        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = 16 * sb + 4 * reg + kk;
                C[t][sb][reg] = (ct < NP && r < mrows) ? B[(unsigned)r * (unsigned)ldb + 16 * ct + li] : 0.0;
            }
    }

    // R_pp of the first panel (its owner only); later panels are prefetched one panel ahead
    // (lane (kk, c) holds rows kk, kk+4, kk+8, kk+12 of column c, 0 below the diagonal)
    fbr_td4 rpp = {0.0, 0.0, 0.0, 0.0};
    {
        const int p0 = first_col / 16;
        if (p0 < NP && wave == p0 % FBR_TSQR_WAVES) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 4 * reg + kk;
                rpp[reg] = (li >= i) ? R[(unsigned)(16 * p0 + i) * ld + 16 * p0 + li] : 0.0;
            }
        }
    }
    for (int p = first_col / 16; p < NP; p++) {  // panels left of first_col: block columns are zero, identity reflectors
        const int ow = p % FBR_TSQR_WAVES, tp = p / FBR_TSQR_WAVES;
        const int j0 = 16 * p;
        const int t0 = (p >= wave) ? (p - wave) / FBR_TSQR_WAVES + 1 : 0;  // this wave's first tile right of the panel
        // the R rows of the panel under this wave's tiles (final since the previous fold) are copied to the LDS by
        // LDS-DMA (no VGPRs) while the panel is being factorised; every wave fetches and consumes its own tiles only
        auto fetch_rows = [&]() {
#pragma unroll
            for (int t = 0; t < TPW; t++)
                if (t >= t0) {
                    const int ct = wave + FBR_TSQR_WAVES * t;
#pragma unroll
                    for (int h = 0; h < 2; h++)  // lane l -> row 8 h + l / 8, columns 2 (l % 8), +1  (16 bytes)
                        __builtin_amdgcn_global_load_lds((fbr_tsqr_glb_ptr)(R + (unsigned)(j0 + 8 * h + (lane >> 3)) * ld + 16 * ct + 2 * (lane & 7)),
                                                         (fbr_tsqr_lds_ptr)(Rl + ct * 256 + h * 128), 16, 0, 0);
                }
        };
        FBR_TT(3)
        __syncthreads();  // every wave is done with the previous panel's V / T
        FBR_TT(0)
        if (wave == ow) {
            // ---- panel: Householder factorisation of [R_pp ; V] in this wave's registers; the tile is copied out of
            //      the block registers with a static switch (it is dead afterwards)
            fbr_td4 v[SUB];
#pragma unroll
            for (int t = 0; t < TPW; t++)
                if (t == tp) {
#pragma unroll
                    for (int sb = 0; sb < SUB; sb++) v[sb] = C[t][sb];
                }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Rp[(4 * reg + kk) * 16 + li] = rpp[reg];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            fetch_rows();
            unsigned long long tq = tacc ? __builtin_readcyclecounter() : 0;
            if (tacc) { tacc[4] += tq - tk; }
            fbr_td4 rq = {0.0, 0.0, 0.0, 0.0};
            double trow[16];
            double myscale = 0.0;
            fbr_tsqr_panel_steps<SUB>(v, Rp, rq, trow, myscale, li, kk, std::make_integer_sequence<int, 16>{});
            if (tacc) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[5] += t1 - tq; tq = t1; }
            // ---- publish the scaled vectors V and T
#pragma unroll
            for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li] = v[sb][reg] * myscale;
            if (kk == 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) Tm[li * 16 + j] = trow[j];
            }
            if (tacc) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[6] += t1 - tq; tq = t1; }
            // R_pp back to global (upper triangle)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 4 * reg + kk;
                if (li >= i) R[(unsigned)(j0 + i) * ld + j0 + li] = rq[reg];
            }
        } else {
            fetch_rows();
        }
        FBR_TT(1)
        __syncthreads();  // panel published, R rows landed (the barrier waits for vmcnt(0))
        FBR_TT(2)
        // the owner of the next panel fetches its R_pp now (panel p only touches its own 16 rows of R)
        if (p + 1 < NP && wave == (p + 1) % FBR_TSQR_WAVES) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 4 * reg + kk;
                rpp[reg] = (li >= i) ? R[(unsigned)(j0 + 16 + i) * ld + j0 + 16 + li] : 0.0;
            }
        }
        // ---- trailing update of this wave's tiles right of the panel (tiles past the last column tile are zero columns
        //      of the padded factor: ld = 16 * WAVES * TPW)
        FbrTsqrUpdate<TPW, SUB>::run(t0, C, Rl, R, ld, (unsigned)j0, wave, li, kk, Vl, Tm);
    }
    FBR_TT(3)
    __syncthreads();
#undef FBR_TT
}

// level 0: workgroup w folds blocks w, w+NW, ... of A into Rw[w]
template <int TPW, int SUB>
__global__ __launch_bounds__(FBR_TSQR_THREADS, 2) void fbr_tsqr_level0_kernel(const double *__restrict__ A, long Mpad, int n,
                                                                               double *__restrict__ Rw, long nblocks,
                                                                               unsigned long long *dbg)
{
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB;
    constexpr int LD = 16 * FBR_TSQR_WAVES * TPW;  // leading dimension of the working factors (>= n)
    double *R = Rw + (long)blockIdx.x * n * LD;
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long r0 = b * MB;
        const int m = (int)std::min<long>(MB, Mpad - r0);
        fbr_tsqr_fold_regs<TPW, SUB>(R, n, LD, A + r0 * n, n, m, 0, smem, dbg ? tacc : nullptr);
    }
    if (dbg && (threadIdx.x & 63) == 0) {
        unsigned long long *d = dbg + ((long)blockIdx.x * FBR_TSQR_WAVES + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 8; i++) d[i] = tacc[i];
    }
}

// tree level: workgroup i folds Rw[(2i+1)*stride] (upper triangular, MB rows at a time) into Rw[2i*stride]
template <int TPW, int SUB>
__global__ __launch_bounds__(FBR_TSQR_THREADS, 2) void fbr_tsqr_tree_kernel(double *__restrict__ Rw, int n, int stride, int count)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int MB = 16 * SUB;
    constexpr int LD = 16 * FBR_TSQR_WAVES * TPW;
    const long a = (long)2 * blockIdx.x * stride, b = a + stride;
    if (b >= count) return;
    for (int i0 = 0; i0 < n; i0 += MB) {
        const int m = std::min(MB, n - i0);
        fbr_tsqr_fold_regs<TPW, SUB>(Rw + a * n * LD, n, LD, Rw + b * n * LD + (long)i0 * LD, LD, m, i0, smem);
    }
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

struct FbrTsqrWork {
    double *Rw = nullptr;   // [NW][n][n]
    double *A = nullptr;    // packed chunk [Mpad][n]
    size_t rw_bytes = 0, a_bytes = 0;
    int n = 0, ld = 0, NW = 0, Pa = 0, mb = 0, tpw = 0, sub = 0;
    bool active = false;
    void release()
    {
        if (Rw) (void)hipFree(Rw);
        if (A) (void)hipFree(A);
        Rw = A = nullptr;
        rw_bytes = a_bytes = 0;
        active = false;
    }
};

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

// kernel instantiations: 4 waves per workgroup, two workgroups per CU (one's serial panel factorisation overlaps the
// other's trailing update); tiles per wave 1..12; block rows 64 / 32 / 16 so that the block fits the VGPRs
#define FBR_TSQR_DISPATCH(TPWV, CALL)                  \
    switch (TPWV) {                                    \
    case 1: { constexpr int TPW = 1, SUB = 4; CALL; } break; \
    case 2: { constexpr int TPW = 2, SUB = 4; CALL; } break; \
    case 3: { constexpr int TPW = 3, SUB = 4; CALL; } break; \
    case 4: { constexpr int TPW = 4, SUB = 3; CALL; } break; \
    case 5: { constexpr int TPW = 5, SUB = 2; CALL; } break; \
    case 6: { constexpr int TPW = 6, SUB = 2; CALL; } break; \
    case 7: { constexpr int TPW = 7, SUB = 2; CALL; } break; \
    case 8: { constexpr int TPW = 8, SUB = 2; CALL; } break; \
    case 9: { constexpr int TPW = 9, SUB = 1; CALL; } break; \
    case 10: { constexpr int TPW = 10, SUB = 1; CALL; } break; \
    case 11: { constexpr int TPW = 11, SUB = 1; CALL; } break; \
    default: { constexpr int TPW = 12, SUB = 1; CALL; } break; \
    }
static inline int fbr_tsqr_sub_for(int tpw) { return tpw <= 3 ? 4 : (tpw == 4 ? 3 : (tpw <= 8 ? 2 : 1)); }

// Start a factorisation of width Pa: working factors zeroed, R_in (device, Pa x Pa, may be null) seeded into slot 0.
static inline int fbr_tsqr_begin(FbrTsqrWork &wk, hipStream_t st, int Pa, const double *R_in, int num_cus, long rows_hint)
{
    const int n = (Pa + 15) & ~15;
    if (n > FBR_TSQR_MAXN) {
        g_tsqr_err = "TSQR supports at most " + std::to_string(FBR_TSQR_MAXN) + " columns";
        return -4;
    }
    const int tpw = (n / 16 + FBR_TSQR_WAVES - 1) / FBR_TSQR_WAVES;
    const int sub = fbr_tsqr_sub_for(tpw);
    const int mb = 16 * sub;
    const long want = (rows_hint + mb - 1) / mb;
    const char *envw = getenv("FBR_TSQR_WG_PER_CU");  // experiments only
    const long per_cu = envw ? std::max(1, atoi(envw)) : 2;
    const int NW = (int)std::max(1L, std::min<long>(per_cu * num_cus, want));
    const int ld = 16 * FBR_TSQR_WAVES * tpw;
    const size_t need = (size_t)NW * n * ld * sizeof(double);
    if (need > wk.rw_bytes) {
        if (wk.Rw) (void)hipFree(wk.Rw);
        wk.Rw = nullptr;
        wk.rw_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.Rw, need));
        wk.rw_bytes = need;
    }
    wk.n = n; wk.ld = ld; wk.NW = NW; wk.Pa = Pa; wk.mb = mb; wk.tpw = tpw; wk.sub = sub;
    TSQR_HIP(hipMemsetAsync(wk.Rw, 0, need, st));
    if (R_in) {
        hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, Pa, R_in, Pa, wk.Rw, ld, n, ld);
        TSQR_HIP(hipGetLastError());
    }
    wk.active = true;
    return 0;
}

// Fold M rows of [Y (M x P) | rhs (M x k)] (row weights w optional) into the working factors.
static inline int fbr_tsqr_fold_rows(FbrTsqrWork &wk, hipStream_t st, long M, int P, const double *Y, int k, const double *rhs,
                                     const double *w, int ldy = 0, const int *cols = nullptr)
{
    if (ldy <= 0) ldy = P;
    if (!wk.active || P + k != wk.Pa) {
        g_tsqr_err = "tsqr fold without matching begin";
        return -1;
    }
    if (M <= 0) return 0;
    const int n = wk.n;
    const long Mpad = (M + 15) & ~15L;
    const size_t need = (size_t)Mpad * n * sizeof(double);
    if (need > wk.a_bytes) {
        if (wk.A) (void)hipFree(wk.A);
        wk.A = nullptr;
        wk.a_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.A, need));
        wk.a_bytes = need;
    }
    hipLaunchKernelGGL(fbr_tsqr_pack_kernel, dim3(2048), dim3(256), 0, st, M, Mpad, P, k, n, Y, ldy, cols, rhs, w, wk.A);
    TSQR_HIP(hipGetLastError());
    const long nblocks = (Mpad + wk.mb - 1) / wk.mb;
    const int grid = (int)std::min<long>(wk.NW, nblocks);
    unsigned long long *dbg = nullptr;
    if (getenv("FBR_TSQR_TIMING")) {
        TSQR_HIP(hipMalloc((void **)&dbg, (size_t)grid * FBR_TSQR_WAVES * 8 * 8));
        TSQR_HIP(hipMemsetAsync(dbg, 0, (size_t)grid * FBR_TSQR_WAVES * 8 * 8, st));
    }
    FBR_TSQR_DISPATCH(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_level0_kernel<TPW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)(fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double))));
    FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_level0_kernel<TPW, SUB>), dim3(grid), dim3(FBR_TSQR_THREADS),
                                                 (fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double)), st, wk.A, Mpad, n, wk.Rw, nblocks, dbg));
    TSQR_HIP(hipGetLastError());
    if (dbg) {
        std::vector<unsigned long long> hb((size_t)grid * FBR_TSQR_WAVES * 8);
        TSQR_HIP(hipMemcpyAsync(hb.data(), dbg, hb.size() * 8, hipMemcpyDeviceToHost, st));
        TSQR_HIP(hipStreamSynchronize(st));
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < hb.size(); i++) sum[i & 7] += (double)hb[i];
        const double folds = (double)nblocks * FBR_TSQR_WAVES;
        fprintf(stderr, "[fbr tsqr timing] cycles per fold per wave: barrier_in=%.0f panel(or idle)=%.0f barrier_pub=%.0f update=%.0f  (mb=%d n=%d)\n",
                sum[0] / folds, sum[1] / folds, sum[2] / folds, sum[3] / folds, wk.mb, n);
        fprintf(stderr, "[fbr tsqr timing]   panel owner split per fold per wave: load_Rpp=%.0f steps=%.0f ZT=%.0f store=%.0f\n", sum[4] / folds,
                sum[5] / folds, sum[6] / folds, sum[7] / folds);
        (void)hipFree(dbg);
    }
    return 0;
}

// Binary tree over the working factors, result (Pa x Pa, upper triangular) to R_out (device).
static inline int fbr_tsqr_finish(FbrTsqrWork &wk, hipStream_t st, double *R_out)
{
    if (!wk.active) {
        g_tsqr_err = "tsqr finish without begin";
        return -1;
    }
    const int n = wk.n;
    FBR_TSQR_DISPATCH(wk.tpw, (void)hipFuncSetAttribute((const void *)fbr_tsqr_tree_kernel<TPW, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)(fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double))));
    for (int stride = 1; stride < wk.NW; stride *= 2) {
        const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
        FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_tree_kernel<TPW, SUB>), dim3(pairs), dim3(FBR_TSQR_THREADS),
                                                     (fbr_tsqr_lds_doubles<TPW, SUB>() * sizeof(double)), st, wk.Rw, n, stride, wk.NW));
        TSQR_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, wk.Pa, wk.Rw, wk.ld, R_out, wk.Pa, wk.Pa, wk.Pa);
    TSQR_HIP(hipGetLastError());
    wk.active = false;
    return 0;
}
