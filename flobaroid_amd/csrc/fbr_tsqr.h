// fbr_tsqr.h -- blocked Householder TSQR on gfx950 (fp64, MFMA trailing update).
//
// R^T R = A^T A for a tall row stream A (rows = samples x N_OUT, columns = [Y | rhs], padded to a
// multiple of 16), WITHOUT forming A^T A (no squaring of the condition number):
//
//   level 0  every workgroup w owns a private upper-triangular R_w and folds its share of the row
//            blocks (FBR_TSQR_MB rows each) into it with a triangular-pentagonal Householder QR
//            ("TPQRT": QR of [R_w ; B] where only R_w's panel rows and the dense block take part):
//              panel (16 columns): Householder vectors held in registers, one block reduction per column
//              T factor:           V^T V by MFMA, 16x16 triangular recurrence by one wave
//              trailing update:    W = T^T (R_rows + V^T C),  R_rows -= W,  C -= V W
//                                  with v_mfma_f64_16x16x4_f64, V (mb x 16) resident in LDS
//   level 1+ binary tree over the R_w (one kernel launch per level, same fold routine, block = the
//            partner's R), the last survivor is the result.  Across ranks the same merge runs on R factors
//            exchanged over xGMI (flobaroid_amd/dist.py).
//
// Everything is deterministic (fixed block -> workgroup assignment, fixed tree).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#define FBR_TSQR_MB 768       // rows per block (V panel = MB x 17 doubles of LDS)
#define FBR_TSQR_LDV 17       // LDS row stride of the V panel (conflict-free for both MFMA operand walks)
#define FBR_TSQR_RPT (FBR_TSQR_MB / 256)

typedef double fbr_td4 __attribute__((ext_vector_type(4)));

static thread_local std::string g_tsqr_err;
static inline const char *fbr_tsqr_error() { return g_tsqr_err.c_str(); }

// A[r][c] (ld) = w[r] * [Y | rhs][r][c], zero in the padding columns / rows
__global__ __launch_bounds__(256) void fbr_tsqr_pack_kernel(long M, long Mpad, int P, int k, int ld,
                                                             const double *__restrict__ Y, const double *__restrict__ rhs,
                                                             const double *__restrict__ w, double *__restrict__ A)
{
    const long total = Mpad * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ld;
        const int c = (int)(i - r * ld);
        double v = 0.0;
        if (r < M) {
            if (c < P)
                v = Y[r * P + c];
            else if (c < P + k)
                v = rhs[r * k + (c - P)];
            if (w) v *= w[r];
        }
        A[i] = v;
    }
}

// sum over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ double fbr_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Fold the dense block B (m rows, leading dimension ld, m <= FBR_TSQR_MB, rows m..m16-1 readable zeros) into the
// upper-triangular R (n x n, leading dimension ldr, n multiple of 16).  One workgroup of 256 threads.
// B is destroyed.  smem: Vb[MB*17] | Rp[256] | Rq[256] | Tm[256] | Z[4*256] | Wt[4*256] | red[2*4*16] | Tau[16]
__device__ void fbr_tsqr_fold_block(double *__restrict__ R, int ldr, int n, double *__restrict__ B, int ld, int m, double *smem)
{
    double *Vb = smem;
    double *Rp = Vb + FBR_TSQR_MB * FBR_TSQR_LDV;
    double *Rq = Rp + 256;   // updated R_pp (Rp stays read-only during the factorisation: no cross-wave races)
    double *Tm = Rq + 256;
    double *Z = Tm + 256;
    double *Wt = Z + 4 * 256;
    double *red = Wt + 4 * 256;
    double *Tau = red + 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int m16 = (m + 15) & ~15;
    const int NP = n / 16;

    for (int p = 0; p < NP; p++) {
        const int j0 = 16 * p;
        __syncthreads();  // previous panel's trailing update (global C, R rows) is complete and visible
        // ---- load the panel: this thread's rows into registers, R_pp into LDS
        double v[FBR_TSQR_RPT][16];
#pragma unroll
        for (int i = 0; i < FBR_TSQR_RPT; i++) {
            const int r = tid + 256 * i;
#pragma unroll
            for (int c = 0; c < 16; c++) v[i][c] = (r < m) ? B[(long)r * ld + j0 + c] : 0.0;
        }
        {
            const int i = tid >> 4, c = tid & 15;
            const double rv = (c >= i) ? R[(long)(j0 + i) * ldr + j0 + c] : 0.0;
            Rp[tid] = rv;
            Rq[tid] = rv;
            Tm[tid] = 0.0;
        }
        __syncthreads();
        // ---- Householder factorisation of [R_pp ; V] column by column
#pragma unroll
        for (int j = 0; j < 16; j++) {
            // s[c] = sum_r x_r * v_rc for c = j..15 (c == j: |x|^2), x = current column j
            double s[16];
#pragma unroll
            for (int c = j; c < 16; c++) {
                double a = 0.0;
#pragma unroll
                for (int i = 0; i < FBR_TSQR_RPT; i++) a += v[i][j] * v[i][c];
                s[c] = fbr_wave_sum(a);
            }
            double *rb = red + (j & 1) * 64;
            if (lane == 0) {
#pragma unroll
                for (int c = j; c < 16; c++) rb[wave * 16 + c] = s[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = j; c < 16; c++) s[c] = rb[c] + rb[16 + c] + rb[32 + c] + rb[48 + c];
            const double alpha = Rp[j * 16 + j];
            const double normsq = s[j];
            double tau = 0.0, scale = 0.0, beta = alpha;
            if (normsq > 0.0) {
                beta = -copysign(sqrt(alpha * alpha + normsq), alpha);
                tau = (beta - alpha) / beta;
                scale = 1.0 / (alpha - beta);
            }
            if (tid == 0) Tau[j] = tau;
            // v_j = x * scale ; remaining columns: w_c = R[j][c] + scale * s[c]
#pragma unroll
            for (int c = j + 1; c < 16; c++) {
                const double wc = Rp[j * 16 + c] + scale * s[c];
                const double f = tau * wc * scale;
#pragma unroll
                for (int i = 0; i < FBR_TSQR_RPT; i++) v[i][c] -= f * v[i][j];
                if (tid == c) Rq[j * 16 + c] = Rp[j * 16 + c] - tau * wc;  // only row j is touched at step j
            }
#pragma unroll
            for (int i = 0; i < FBR_TSQR_RPT; i++) v[i][j] *= scale;
            if (tid == j) Rq[j * 16 + j] = beta;
        }
        // ---- V to LDS (zero rows up to m16 + one spare k-step)
#pragma unroll
        for (int i = 0; i < FBR_TSQR_RPT; i++) {
            const int r = tid + 256 * i;
#pragma unroll
            for (int c = 0; c < 16; c++) Vb[r * FBR_TSQR_LDV + c] = v[i][c];
        }
        __syncthreads();
        // ---- R_pp back to global; Z = V^T V by MFMA (each wave a quarter of the k-steps)
        {
            const int i = tid >> 4, c = tid & 15;
            if (c >= i) R[(long)(j0 + i) * ldr + j0 + c] = Rq[tid];
        }
        {
            fbr_td4 z = {0.0, 0.0, 0.0, 0.0};
            const int nks = m16 / 4;
            for (int ks = wave; ks < nks; ks += 4) {
                const double a = Vb[(4 * ks + kk) * FBR_TSQR_LDV + li];
                z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, z, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Z[wave * 256 + (kk + 4 * reg) * 16 + li] = z[reg];
        }
        __syncthreads();
        // ---- T (16x16 upper triangular): T[j][j] = tau_j, T[0:j, j] = -tau_j T[0:j,0:j] z[0:j, j]; one wave, lanes 0..15
        if (wave == 0) {
            if (lane < 16) {
                const int i = lane;  // row of T
                for (int j = 0; j < 16; j++) {
                    double t = 0.0;
                    if (i == j) {
                        t = Tau[j];
                    } else if (i < j) {
                        double acc = 0.0;
                        for (int l = i; l < j; l++) {
                            const double zz = Z[l * 16 + j] + Z[256 + l * 16 + j] + Z[512 + l * 16 + j] + Z[768 + l * 16 + j];
                            acc += Tm[i * 16 + l] * zz;
                        }
                        t = -Tau[j] * acc;
                    }
                    Tm[i * 16 + j] = t;
                }
            }
        }
        __syncthreads();
        // ---- trailing update, one column tile per wave at a time
        double *Wm = Wt + wave * 256;
        for (int ct = p + 1 + wave; ct < NP; ct += 4) {
            const int c0 = 16 * ct;
            fbr_td4 r0, acc;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) r0[reg] = R[(long)(j0 + kk + 4 * reg) * ldr + c0 + li];
            acc = r0;
            const int nks = m16 / 4;
            const double *bp = B + (long)kk * ld + c0 + li;
            for (int ks = 0; ks < nks; ks++) {
                const double a = Vb[(4 * ks + kk) * FBR_TSQR_LDV + li];
                const double b = bp[(long)(4 * ks) * ld];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            // W2 = T^T acc  (through LDS to turn the C/D layout into a B operand)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Wm[(kk + 4 * reg) * 16 + li] = acc[reg];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            fbr_td4 w2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const double a = Tm[(4 * ks + kk) * 16 + li];  // A[i][k] = T[k][i]
                const double b = Wm[(4 * ks + kk) * 16 + li];
                w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, w2, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) R[(long)(j0 + kk + 4 * reg) * ldr + c0 + li] = r0[reg] - w2[reg];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Wm[(kk + 4 * reg) * 16 + li] = w2[reg];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            double wb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) wb[ks] = Wm[(4 * ks + kk) * 16 + li];
            // C -= V W2, 16 rows at a time
            const int nrt = m16 / 16;
            for (int rt = 0; rt < nrt; rt++) {
                fbr_td4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const double a = Vb[(16 * rt + li) * FBR_TSQR_LDV + 4 * ks + kk];  // A[i][k] = V[16rt+i][k]
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wb[ks], d, 0, 0, 0);
                }
                double *cp = B + (long)(16 * rt + kk) * ld + c0 + li;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const double c = cp[(long)(4 * reg) * ld];
                    cp[(long)(4 * reg) * ld] = c - d[reg];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

// level 0: workgroup w folds blocks w, w+NW, ... of A into Rw[w]
__global__ __launch_bounds__(256, 1) void fbr_tsqr_level0_kernel(double *__restrict__ A, long Mpad, int ld, int n,
                                                                  double *__restrict__ Rw, long nblocks)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *R = Rw + (long)blockIdx.x * n * n;
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const long r0 = b * FBR_TSQR_MB;
        const int m = (int)std::min<long>(FBR_TSQR_MB, Mpad - r0);
        fbr_tsqr_fold_block(R, n, n, A + r0 * ld, ld, m, smem);
    }
}

// tree level: workgroup i folds Rw[(2i+1)*stride] (as a dense n-row block) into Rw[2i*stride]
__global__ __launch_bounds__(256, 1) void fbr_tsqr_tree_kernel(double *__restrict__ Rw, int n, int stride, int count)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long a = (long)2 * blockIdx.x * stride, b = a + stride;
    if (b >= count) return;
    fbr_tsqr_fold_block(Rw + a * n * n, n, n, Rw + b * n * n, n, n, smem);
}

// copy between the caller's Pa x Pa factor and the padded n x n working factor (upper triangle only)
__global__ void fbr_tsqr_copy_kernel(int Pa, int n, const double *__restrict__ src, int lds, double *__restrict__ dst, int ldd,
                                     int rows_dst, int cols_dst)
{
    const long total = (long)rows_dst * cols_dst;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_dst), c = (int)(i % cols_dst);
        double v = 0.0;
        if (r < Pa && c < Pa && c >= r) v = src[(long)r * lds + c];
        dst[(long)r * ldd + c] = v;
    }
    (void)n;
}

struct FbrTsqrWork {
    double *Rw = nullptr;   // [NW][n][n]
    double *A = nullptr;    // packed chunk [Mpad][n]
    size_t rw_bytes = 0, a_bytes = 0;
    int n = 0, NW = 0, Pa = 0;
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

static inline size_t fbr_tsqr_lds_bytes()
{
    return (size_t)(FBR_TSQR_MB * FBR_TSQR_LDV + 256 + 256 + 256 + 4 * 256 + 4 * 256 + 2 * 64 + 16) * sizeof(double);
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

// Start a factorisation of width Pa: working factors zeroed, R_in (device, Pa x Pa, may be null) seeded into slot 0.
static inline int fbr_tsqr_begin(FbrTsqrWork &wk, hipStream_t st, int Pa, const double *R_in, int num_cus, long rows_hint)
{
    const int n = (Pa + 15) & ~15;
    if (n > FBR_TSQR_MB) {
        g_tsqr_err = "TSQR supports at most " + std::to_string(FBR_TSQR_MB) + " columns";
        return -4;
    }
    const long want = (rows_hint + FBR_TSQR_MB - 1) / FBR_TSQR_MB;
    const int NW = (int)std::max(1L, std::min<long>(num_cus, want));
    const size_t need = (size_t)NW * n * n * sizeof(double);
    if (need > wk.rw_bytes) {
        if (wk.Rw) (void)hipFree(wk.Rw);
        wk.Rw = nullptr;
        wk.rw_bytes = 0;
        TSQR_HIP(hipMalloc((void **)&wk.Rw, need));
        wk.rw_bytes = need;
    }
    wk.n = n; wk.NW = NW; wk.Pa = Pa;
    TSQR_HIP(hipMemsetAsync(wk.Rw, 0, need, st));
    if (R_in) {
        hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, Pa, n, R_in, Pa, wk.Rw, n, n, n);
        TSQR_HIP(hipGetLastError());
    }
    TSQR_HIP(hipFuncSetAttribute((const void *)fbr_tsqr_level0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)fbr_tsqr_lds_bytes()));
    TSQR_HIP(hipFuncSetAttribute((const void *)fbr_tsqr_tree_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)fbr_tsqr_lds_bytes()));
    wk.active = true;
    return 0;
}

// Fold M rows of [Y (M x P) | rhs (M x k)] (row weights w optional) into the working factors.
static inline int fbr_tsqr_fold_rows(FbrTsqrWork &wk, hipStream_t st, long M, int P, const double *Y, int k, const double *rhs,
                                     const double *w)
{
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
    hipLaunchKernelGGL(fbr_tsqr_pack_kernel, dim3(2048), dim3(256), 0, st, M, Mpad, P, k, n, Y, rhs, w, wk.A);
    TSQR_HIP(hipGetLastError());
    const long nblocks = (Mpad + FBR_TSQR_MB - 1) / FBR_TSQR_MB;
    const int grid = (int)std::min<long>(wk.NW, nblocks);
    hipLaunchKernelGGL(fbr_tsqr_level0_kernel, dim3(grid), dim3(256), fbr_tsqr_lds_bytes(), st, wk.A, Mpad, n, n, wk.Rw, nblocks);
    TSQR_HIP(hipGetLastError());
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
        hipLaunchKernelGGL(fbr_tsqr_tree_kernel, dim3(pairs), dim3(256), fbr_tsqr_lds_bytes(), st, wk.Rw, n, stride, wk.NW);
        TSQR_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, st, wk.Pa, n, wk.Rw, n, R_out, wk.Pa, wk.Pa, wk.Pa);
    TSQR_HIP(hipGetLastError());
    wk.active = false;
    return 0;
}
