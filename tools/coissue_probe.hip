// Microbenchmark: do fp64 VALU FMAs of one wave run beside the fp64 MFMAs of the other wave of the same SIMD?
// (decides whether a producer fused into the Gram workgroup can hide behind its MFMA phase)
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 fp64 FMAs (plain or with LDS reads).
// hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = waves 0-3 run MFMA, 2 = waves 4-7 run VALU fp64, 4 = waves 4-7 run MFMA too, 8 = VALU waves also read LDS
__global__ __launch_bounds__(512, 2) void k(double *out, unsigned long long *cyc, int iters, int mode, double a0)
{
    __shared__ double lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = a0 + i * 1e-6;
    __syncthreads();
    double a = a0 + threadIdx.x * 1e-9, b = 1.0 + a0;
    double s = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool mf = (wave < 4 && (mode & 1)) || (wave >= 4 && (mode & 4));
    const int lanes = (mode & 16) ? 16 : (mode & 32) ? 32 : 64;  // active lanes of the VALU waves (EXEC mask)
    const bool va = wave >= 4 && (mode & 2) && (int)(threadIdx.x & 63) < lanes;
    if (mf) {
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (va) {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = a + i;
        const double *lp = lds + (threadIdx.x & 63);
        for (int it = 0; it < iters; it++) {
            // 32 fp64 FMAs per iteration (8 independent chains x 4) = the issue time of ~4 MFMAs' worth of pipe (4 x 64 cycles)?
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double m = b;
                if (mode & 8) m = lp[((it + r) & 31) * 64];
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = __builtin_fma(x[i], m, a);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s += x[i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    double *out;
    unsigned long long *cyc, h[8 * 1024];
    hipMalloc(&out, (size_t)blocks * 512 * 8);
    hipMalloc(&cyc, (size_t)blocks * 8 * 8);
    const int iters = 20000;
    const char *names[] = {"", "MFMA on waves 0-3 only", "fp64 VALU on waves 4-7 only", "MFMA (0-3) beside fp64 VALU (4-7)", "", "MFMA on all 8 waves", "", "",
                           "", "", "VALU + LDS reads only", "MFMA beside VALU + LDS reads", "", "", "", "", "", "", "fp64 VALU, 16 active lanes", "MFMA beside fp64 VALU of 16 lanes",
                           "", "", "", "", "", "", "", "", "", "", "", "", "", "", "fp64 VALU, 32 active lanes", "MFMA beside fp64 VALU of 32 lanes"};
    for (int mode : {1, 2, 3, 5, 10, 11, 18, 19, 34, 35}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, 100, mode, 1.0);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mode, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, (size_t)blocks * 8 * 8, hipMemcpyDeviceToHost);
        double cm = 0, cv = 0;
        for (int b = 0; b < blocks; b++)
            for (int w = 0; w < 8; w++) (w < 4 ? cm : cv) += (double)h[b * 8 + w];
        cm /= blocks * 4.0;
        cv /= blocks * 4.0;
        const int mw = (mode & 1 ? 4 : 0) + (mode & 4 ? 4 : 0);
        printf("mode %2d %-36s %8.3f ms | waves 0-3: %9.0f cycles (%.1f per MFMA) | waves 4-7: %9.0f cycles (%.2f per fp64 FMA, %.1f per MFMA) | %.1f TFLOP/s MFMA\n", mode,
               names[mode], ms, cm, cm / (iters * 4.0), cv, cv / (iters * 32.0), cv / (iters * 4.0), mw ? (double)blocks * mw * iters * 4 * 2048.0 / ms / 1e9 : 0.0);
    }
    return 0;
}
