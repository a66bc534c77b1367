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

// LDS carve (doubles): Vl[MB*17] | Rp[256] | Rq[256] | Tm[256] | Tau[16] | Wt[WAVES*256]
template <int SUB> static inline size_t fbr_tsqr_lds_doubles() { return (size_t)16 * SUB * FBR_TSQR_LDV + 3 * 256 + 16 + (size_t)FBR_TSQR_WAVES * 256; }

// broadcast, inside every group of 16 lanes, the value of lane j of that group
__device__ __forceinline__ double fbr_row_bcast(double v, int j, int lane)
{
    return __shfl(v, (lane & 48) | j, 64);
}

template <int TPW, int SUB>
__device__ __forceinline__ void fbr_tsqr_fold_regs(double *__restrict__ R, int n, const double *__restrict__ B, long ldb, int mrows, int first_col,
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
    double *Vl = smem;
    double *Rp = Vl + MB * FBR_TSQR_LDV;
    double *Rq = Rp + 256;
    double *Tm = Rq + 256;
    double *Tau = Tm + 256;
    double *Wt = Tau + 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int NP = n / 16;
    double *Wm = Wt + wave * 256;

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
                C[t][sb][reg] = (ct < NP && r < mrows) ? B[(long)r * ldb + 16 * ct + li] : 0.0;
            }
    }

    // R_pp of the first panel (its owner only); later panels are prefetched one panel ahead
    fbr_td4 rpp = {0.0, 0.0, 0.0, 0.0};
    {
        const int p0 = first_col / 16;
        if (p0 < NP && wave == p0 % FBR_TSQR_WAVES) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 4 * reg + kk;
                rpp[reg] = (li >= i) ? R[(long)(16 * p0 + i) * n + 16 * p0 + li] : 0.0;
            }
        }
    }
    for (int p = first_col / 16; p < NP; p++) {  // panels left of first_col: block columns are zero, identity reflectors
        {
            const int ow = p % FBR_TSQR_WAVES, tp = p / FBR_TSQR_WAVES;
            const int j0 = 16 * p;
            FBR_TT(3)
            __syncthreads();  // every wave is done with the previous panel's V / T / R rows
            FBR_TT(0)
            if (wave == ow) {
                // ---- panel: Householder factorisation of [R_pp ; V]; the tile is copied out of the block registers
                //      with a static switch (it is dead afterwards), so the block array is only indexed statically
                fbr_td4 v[SUB];
#pragma unroll
                for (int t = 0; t < TPW; t++)
                    if (t == tp) {
#pragma unroll
                        for (int sb = 0; sb < SUB; sb++) v[sb] = C[t][sb];
                    }
                {
                    // R_pp (upper triangle): prefetched by this wave during the previous panel's update phase
                    // (lane (kk, c) holds rows kk, kk+4, kk+8, kk+12 of column c)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int i = 4 * reg + kk;
                        Rp[i * 16 + li] = rpp[reg];
                        Rq[i * 16 + li] = rpp[reg];
                        Tm[i * 16 + li] = 0.0;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                unsigned long long tq = tacc ? __builtin_readcyclecounter() : 0;
                if (tacc) { tacc[4] += tq - tk; }
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    // (loads first: they do not depend on the chain below)
                    const double alpha = Rp[j * 16 + j];
                    const double rjc = Rp[j * 16 + li];
                    // x = column j of the panel at this lane's rows; s = x . B[:, c], q = |x|^2 (every lane, no broadcast)
                    fbr_td4 x[SUB];
                    double sa = 0.0, sb2 = 0.0, qa = 0.0, qb = 0.0;
#pragma unroll
                    for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                        for (int reg = 0; reg < 4; reg += 2) {
                            x[sb][reg] = fbr_row_bcast(v[sb][reg], j, lane);
                            x[sb][reg + 1] = fbr_row_bcast(v[sb][reg + 1], j, lane);
                            sa += x[sb][reg] * v[sb][reg];
                            sb2 += x[sb][reg + 1] * v[sb][reg + 1];
                            qa += x[sb][reg] * x[sb][reg];
                            qb += x[sb][reg + 1] * x[sb][reg + 1];
                        }
                    double s = sa + sb2, normsq = qa + qb;
                    s += __shfl_xor(s, 16, 64);
                    normsq += __shfl_xor(normsq, 16, 64);
                    s += __shfl_xor(s, 32, 64);
                    normsq += __shfl_xor(normsq, 32, 64);
                    double tau = 0.0, scale = 0.0, beta = alpha;
                    if (normsq > 0.0) {
                        beta = -copysign(sqrt(alpha * alpha + normsq), alpha);
                        // two independent reciprocals (hardware seed + 2 Newton steps) instead of two divisions
                        const double d1 = alpha - beta;
                        double r1 = __builtin_amdgcn_rcp(d1), r2 = __builtin_amdgcn_rcp(beta);
                        r1 = r1 * (2.0 - d1 * r1);
                        r2 = r2 * (2.0 - beta * r2);
                        r1 = r1 * (2.0 - d1 * r1);
                        r2 = r2 * (2.0 - beta * r2);
                        scale = r1;
                        tau = -d1 * r2;
                    }
                    if (li > j) {
                        const double wc = rjc + scale * s;
                        const double f = tau * wc * scale;
#pragma unroll
                        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) v[sb][reg] -= f * x[sb][reg];
                        if (kk == 0) Rq[j * 16 + li] = rjc - tau * wc;
                    } else if (li == j) {
#pragma unroll
                        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) v[sb][reg] *= scale;
                        if (kk == 0) {
                            Rq[j * 16 + j] = beta;
                            Tau[j] = tau;
                        }
                    }
                }
                if (tacc) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[5] += t1 - tq; tq = t1; }
                // ---- publish V; Z = V^T V straight from the registers; T by the triangular recurrence
                fbr_td4 z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li] = v[sb][reg];
                        z = __builtin_amdgcn_mfma_f64_16x16x4f64(v[sb][reg], v[sb][reg], z, 0, 0, 0);
                    }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Wm[(4 * reg + kk) * 16 + li] = z[reg];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                {
                    // T[i][j] = tau_j (i == j), -tau_j * sum_{l<j} T[i][l] Z[l][j] (i < j); lane i keeps row i in registers
                    // (T[i][l] = 0 for l < i, so the sum may start at 0); Z is read with static LDS offsets
                    const int i = li;
                    double trow[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                        for (int l = 0; l + 1 < j; l += 2) {
                            acc0 += trow[l] * Wm[l * 16 + j];
                            acc1 += trow[l + 1] * Wm[(l + 1) * 16 + j];
                        }
                        if (j & 1) acc0 += trow[j - 1] * Wm[(j - 1) * 16 + j];
                        const double tj = Tau[j];
                        trow[j] = (i == j) ? tj : ((i < j) ? -tj * (acc0 + acc1) : 0.0);
                    }
                    if (kk == 0) {
#pragma unroll
                        for (int j = 0; j < 16; j++) Tm[i * 16 + j] = trow[j];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (tacc) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[6] += t1 - tq; tq = t1; }
                // R_pp back to global (upper triangle)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int i = 4 * reg + kk;
                    if (li >= i) R[(long)(j0 + i) * n + j0 + li] = Rq[i * 16 + li];
                }
            }
            FBR_TT(1)
            __syncthreads();  // panel published
            FBR_TT(2)
            // the owner of the next panel fetches its R_pp now (panel p only touches its own 16 rows of R)
            if (p + 1 < NP && wave == (p + 1) % FBR_TSQR_WAVES) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int i = 4 * reg + kk;
                    rpp[reg] = (li >= i) ? R[(long)(j0 + 16 + i) * n + j0 + 16 + li] : 0.0;
                }
            }
            // ---- trailing update of this wave's tiles right of the panel
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                const int ct = wave + FBR_TSQR_WAVES * t;
                if (ct <= p || ct >= NP) continue;
                const int c0 = 16 * ct;
                fbr_td4 r0;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) r0[reg] = R[(long)(j0 + 4 * reg + kk) * n + c0 + li];
                fbr_td4 acc = r0;
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * sb + 4 * reg + kk) * FBR_TSQR_LDV + li], C[t][sb][reg], acc, 0, 0, 0);
                // W2 = T^T acc (through LDS: C/D layout -> B operand)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Wm[(4 * reg + kk) * 16 + li] = acc[reg];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                fbr_td4 w2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Tm[(4 * ks + kk) * 16 + li], Wm[(4 * ks + kk) * 16 + li], w2, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) R[(long)(j0 + 4 * reg + kk) * n + c0 + li] = r0[reg] - w2[reg];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Wm[(4 * reg + kk) * 16 + li] = -w2[reg];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                double wb[4];
#pragma unroll
                for (int ks = 0; ks < 4; ks++) wb[ks] = Wm[(4 * ks + kk) * 16 + li];
                // C += V (-W2), accumulated straight into the register tile
#pragma unroll
                for (int sb = 0; sb < SUB; sb++)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++)
                        C[t][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vl[(16 * sb + li) * FBR_TSQR_LDV + 4 * ks + kk], wb[ks], C[t][sb], 0, 0, 0);
                __builtin_amdgcn_wave_barrier();
            }
        }
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
    double *R = Rw + (long)blockIdx.x * n * n;
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long r0 = b * MB;
        const int m = (int)std::min<long>(MB, Mpad - r0);
        fbr_tsqr_fold_regs<TPW, SUB>(R, n, A + r0 * n, n, m, 0, smem, dbg ? tacc : nullptr);
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
    const long a = (long)2 * blockIdx.x * stride, b = a + stride;
    if (b >= count) return;
    for (int i0 = 0; i0 < n; i0 += MB) {
        const int m = std::min(MB, n - i0);
        fbr_tsqr_fold_regs<TPW, SUB>(Rw + a * n * n, n, Rw + b * n * n + (long)i0 * n, n, m, i0, smem);
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
    int n = 0, NW = 0, Pa = 0, mb = 0, tpw = 0, sub = 0;
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
    const int NW = (int)std::max(1L, std::min<long>(2L * num_cus, want));
    const size_t need = (size_t)NW * n * n * sizeof(double);
    if (need > wk.rw_bytes) {
        if (wk.Rw) (void)hipFree(wk.Rw);
        wk.Rw = nullptr;
        wk.rw_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.Rw, need));
        wk.rw_bytes = need;
    }
    wk.n = n; wk.NW = NW; wk.Pa = Pa; wk.mb = mb; wk.tpw = tpw; wk.sub = sub;
    TSQR_HIP(hipMemsetAsync(wk.Rw, 0, need, st));
    if (R_in) {
        hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, Pa, R_in, Pa, wk.Rw, n, n, n);
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
    FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_level0_kernel<TPW, SUB>), dim3(grid), dim3(FBR_TSQR_THREADS),
                                                 fbr_tsqr_lds_doubles<SUB>() * sizeof(double), st, wk.A, Mpad, n, wk.Rw, nblocks, dbg));
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
    for (int stride = 1; stride < wk.NW; stride *= 2) {
        const int pairs = (wk.NW + 2 * stride - 1) / (2 * stride);
        FBR_TSQR_DISPATCH(wk.tpw, hipLaunchKernelGGL((fbr_tsqr_tree_kernel<TPW, SUB>), dim3(pairs), dim3(FBR_TSQR_THREADS),
                                                     fbr_tsqr_lds_doubles<SUB>() * sizeof(double), st, wk.Rw, n, stride, wk.NW));
        TSQR_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, wk.Pa, wk.Rw, n, R_out, wk.Pa, wk.Pa, wk.Pa);
    TSQR_HIP(hipGetLastError());
    wk.active = false;
    return 0;
}
