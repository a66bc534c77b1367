// fbr_signal.h -- the step before the path on the device: the signal conditioning of Data.preprocess
// (identification/data.py:369-619) -- zero-phase Butterworth low-passes (scipy.signal.filtfilt), median filters (scipy.signal.medfilt)
// and the 4th-order central differences -- on S x ncols channel arrays that stay in HBM.
//
// filtfilt is an IIR recursion along time: sequential per channel, and a measurement file has ~100 channels for millions of samples.
// It is parallelised over TIME by a blocked scan of the filter state (transposed direct form II, the recurrence of scipy's lfilter):
//   z' = M z + g x,  y = z_0 + b_0 x        (M, g from the normalised coefficients)
//   pass A  every (channel, block of FBR_SIG_LB samples) runs the recurrence from a zero state and keeps its final state;
//   pass B  one thread per channel chains the blocks: z_start[blk + 1] = M^LB z_start[blk] + z_zero_state[blk];
//   pass C  every (channel, block) runs the recurrence again from its true start state and writes the outputs
// -- the outputs are those of the sequential recurrence up to the rounding of the start states.  Forward and backward sweep, scipy's
// default odd extension (padlen = 3 * ncoef) and its steady-state initial conditions (zi * first sample) included.
#pragma once
#include <hip/hip_runtime.h>

#define FBR_SIG_MAXC 12   // filter coefficients (order + 1) supported
#define FBR_SIG_LB 2048   // samples per block of the state scan (2 M x 29 channels: 256 -> 28.6 ms, 1024 -> 9.9, 2048 -> 9.3, 4096 -> 13.2)
#define FBR_SIG_MAXK 31   // median window

struct FbrIir {
    int nc;                     // coefficients (order + 1)
    double b[FBR_SIG_MAXC], a[FBR_SIG_MAXC];   // normalised by a[0]
    double zi[FBR_SIG_MAXC];    // lfilter_zi(b, a): steady-state of a unit step
    double Mp[(FBR_SIG_MAXC - 1) * (FBR_SIG_MAXC - 1)];  // M^LB, row-major p x p
};

// sample t of the odd-extended column (scipy.signal._arraytools.odd_ext along axis 0)
__device__ __forceinline__ double fbr_sig_ext(const double *__restrict__ x, long S, long ld, int pad, long t)
{
    if (t < pad) return 2.0 * x[0] - x[(long)(pad - t) * ld];
    if (t < pad + S) return x[(t - pad) * ld];
    return 2.0 * x[(S - 1) * ld] - x[(S - 2 - (t - pad - S)) * ld];
}

// one step of lfilter's transposed direct form II (same operation order as scipy's C loop)
__device__ __forceinline__ double fbr_sig_step(const FbrIir &f, double (&z)[FBR_SIG_MAXC - 1], double x)
{
    const int p = f.nc - 1;
    const double y = z[0] + f.b[0] * x;
#pragma unroll
    for (int i = 0; i < FBR_SIG_MAXC - 2; i++)
        if (i < p - 1) z[i] = z[i + 1] + x * f.b[i + 1] - y * f.a[i + 1];
    z[p - 1] = x * f.b[p] - y * f.a[p];
    return y;
}

// mode 0: pass A (zero start state, final state -> zs); mode 1: pass C (start state from zst, outputs written)
// dir 0: forward over the odd-extended input X -> Y1 [Le][ncols]; dir 1: backward over Y1 (reversed) -> central part into X
__global__ __launch_bounds__(256) void fbr_sig_iir_kernel(FbrIir f, int mode, int dir, double *__restrict__ X, long S, int ncols, long ld, int pad,
                                                           double *__restrict__ Y1, double *__restrict__ zs, const double *__restrict__ zst, long nblk)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(gid % ncols);
    const long blk = gid / ncols;
    if (blk >= nblk) return;
    const int p = f.nc - 1;
    const long Le = S + 2L * pad;
    const long t0 = blk * FBR_SIG_LB, t1 = min(Le, t0 + FBR_SIG_LB);
    double z[FBR_SIG_MAXC - 1];
#pragma unroll
    for (int i = 0; i < FBR_SIG_MAXC - 1; i++) z[i] = (mode == 1 && i < p) ? zst[(blk * ncols + c) * (FBR_SIG_MAXC - 1) + i] : 0.0;
    for (long t = t0; t < t1; t++) {
        const double x = dir == 0 ? fbr_sig_ext(X + c, S, ld, pad, t) : Y1[(Le - 1 - t) * ncols + c];
        const double y = fbr_sig_step(f, z, x);
        if (mode == 1) {
            if (dir == 0) {
                Y1[t * ncols + c] = y;
            } else {
                const long i = Le - 1 - t - pad;  // position in the un-extended signal
                if (i >= 0 && i < S) X[i * ld + c] = y;
            }
        }
    }
    if (mode == 0)
        for (int i = 0; i < p; i++) zs[(blk * ncols + c) * (FBR_SIG_MAXC - 1) + i] = z[i];
}

// pass B: start state of every block of one channel (thread per channel)
__global__ void fbr_sig_chain_kernel(FbrIir f, int dir, const double *__restrict__ X, long S, int ncols, long ld, int pad, const double *__restrict__ Y1,
                                     const double *__restrict__ zs, double *__restrict__ zst, long nblk)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const int p = f.nc - 1;
    const long Le = S + 2L * pad;
    const double u0 = dir == 0 ? fbr_sig_ext(X + c, S, ld, pad, 0) : Y1[(Le - 1) * ncols + c];
    double z[FBR_SIG_MAXC - 1], zn[FBR_SIG_MAXC - 1];
    for (int i = 0; i < p; i++) z[i] = f.zi[i] * u0;
    for (long blk = 0; blk < nblk; blk++) {
        for (int i = 0; i < p; i++) zst[(blk * ncols + c) * (FBR_SIG_MAXC - 1) + i] = z[i];
        for (int i = 0; i < p; i++) {
            double acc = zs[(blk * ncols + c) * (FBR_SIG_MAXC - 1) + i];
            for (int j = 0; j < p; j++) acc += f.Mp[i * p + j] * z[j];
            zn[i] = acc;
        }
        for (int i = 0; i < p; i++) z[i] = zn[i];
    }
}

// scipy.signal.medfilt(X, (k, 1)): median over the window of k samples of a column, zeros beyond the ends
__global__ __launch_bounds__(256) void fbr_sig_median_kernel(int k, const double *__restrict__ src, double *__restrict__ X, long S, int ncols, long ld)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * ncols) return;
    const long t = gid / ncols;
    const int c = (int)(gid - t * ncols);
    const int h = k / 2;
    double w[FBR_SIG_MAXK];
    for (int i = 0; i < k; i++) {
        const long u = t - h + i;
        const double v = (u >= 0 && u < S) ? src[u * ncols + c] : 0.0;
        int j = i;  // insertion sort
        while (j > 0 && w[j - 1] > v) {
            w[j] = w[j - 1];
            j--;
        }
        w[j] = v;
    }
    X[t * ld + c] = w[h];
}

__global__ __launch_bounds__(256) void fbr_sig_gather_kernel(const double *__restrict__ X, long S, int ncols, long ld, double *__restrict__ dst)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * ncols) return;
    const long t = gid / ncols;
    dst[gid] = X[t * ld + (gid - t * ncols)];
}

// 4th-order central difference of data.py:396-418 with its edge rules (see flobaroid_amd/data.py: Data._central_diff)
__global__ __launch_bounds__(256) void fbr_sig_cdiff_kernel(const double *__restrict__ A, const double *__restrict__ T, double *__restrict__ D, long S, int ncols)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * ncols) return;
    const long t = gid / ncols;
    const int c = (int)(gid - t * ncols);
    auto a = [&](long i) { return A[i * ncols + c]; };
    const double div0 = T[1] - T[0];
    const double div_last = S > 4 ? T[S - 3] - T[S - 4] : div0;
    double d;
    if (t == 0)
        d = (a(1) - a(0)) / div0;
    else if (t == 1)
        d = (a(2) - a(0)) / (2 * div0);
    else if (t == S - 2)
        d = (a(S - 1) - a(S - 3)) / (2 * div_last);
    else if (t == S - 1)
        d = (a(S - 1) - a(S - 2)) / div_last;
    else
        d = (-a(t + 2) + 8 * a(t + 1) - 8 * a(t - 1) + a(t - 2)) / (12 * (T[t] - T[t - 1]));
    D[gid] = d;
}
