// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (confirms the fp64 matrix peak used
// as the roofline denominator).  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> void run(int blocks, int iters)
{
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 /*waves*/ * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.2f TFLOP/s\n", NACC, blocks, iters, ms, flop / ms / 1e9);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs=%d clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<1>(p.multiProcessorCount * 4, 20000);
    run<2>(p.multiProcessorCount * 4, 10000);
    run<4>(p.multiProcessorCount * 4, 10000);
    run<8>(p.multiProcessorCount * 2, 10000);
    run<4>(p.multiProcessorCount * 8, 10000);
    return 0;
}
