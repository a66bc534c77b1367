// Microbenchmark / self-check of the TSQR panel chain (one wave): cycles per Householder step and a comparison of the
// register-resident chain against a plain serial Householder QR of the same 16-column panel done by lane 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/tsqr_chain_probe tools/tsqr_chain_probe.hip
#include "../flobaroid_amd/csrc/fbr_tsqr.h"
#include <cmath>
#include <cstdio>
#include <vector>

template <int SUB> __global__ void chain_kernel(const double *Bp, const double *Rpp, double *Vout, double *Tout, double *Rout, long long *cyc, int reps)
{
    __shared__ double Rp[256];
    const int lane = threadIdx.x, li = lane & 15, kk = lane >> 4;
    fbr_td4 v[SUB], rq;
    double trow[16], myscale = 0.0;
    long long t0 = 0, t1 = 0;
    for (int r = 0; r < reps; r++) {
        for (int sb = 0; sb < SUB; sb++)
            for (int reg = 0; reg < 4; reg++) v[sb][reg] = Bp[(16 * sb + 4 * reg + kk) * 16 + li];
        for (int reg = 0; reg < 4; reg++) Rp[(4 * reg + kk) * 16 + li] = Rpp[(4 * reg + kk) * 16 + li];
        rq = fbr_td4{0, 0, 0, 0};
        myscale = 0.0;
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const long long ta = __builtin_readcyclecounter();
        fbr_tsqr_panel_steps<SUB>(v, Rp, rq, trow, myscale, li, kk, std::make_integer_sequence<int, 16>{});
        asm volatile("" : "+v"(v[0][0]));
        const long long tb = __builtin_readcyclecounter();
        if (r > 0) t1 += tb - ta;
        __syncthreads();
    }
    for (int sb = 0; sb < SUB; sb++)
        for (int reg = 0; reg < 4; reg++) Vout[(16 * sb + 4 * reg + kk) * 16 + li] = v[sb][reg] * myscale;
    if (kk == 0)
        for (int j = 0; j < 16; j++) Tout[li * 16 + j] = trow[j];
    for (int reg = 0; reg < 4; reg++) Rout[(4 * reg + kk) * 16 + li] = rq[reg];
    if (lane == 0) *cyc = (t1 - t0) / (reps - 1);
}

template <int SUB> static int run()
{
    const int MB = 16 * SUB;
    std::vector<double> B(MB * 16), R(256, 0.0), V(MB * 16), T(256), Rn(256);
    srand(7 + SUB);
    for (auto &x : B) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; i++)
        for (int j = i; j < 16; j++) R[i * 16 + j] = rand() / (double)RAND_MAX - 0.5;
    for (int r = 0; r < MB; r++) B[r * 16 + 5] = 0.0;  // a zero column below a zero diagonal entry: tau = 0 path
    for (int i = 0; i <= 5; i++) R[i * 16 + 5] = (i < 5) ? R[i * 16 + 5] : 0.0;
    double *dB, *dR, *dV, *dT, *dRn;
    long long *dc, cyc = 0;
    hipMalloc(&dB, B.size() * 8); hipMalloc(&dR, 2048); hipMalloc(&dV, V.size() * 8); hipMalloc(&dT, 2048); hipMalloc(&dRn, 2048); hipMalloc(&dc, 8);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, R.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain_kernel<SUB>, dim3(1), dim3(64), 0, 0, dB, dR, dV, dT, dRn, dc, 50);
    hipMemcpy(V.data(), dV, V.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(T.data(), dT, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(Rn.data(), dRn, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    // host check: A = [R; B] (16+MB x 16);  Q = I - Vf T Vf^T with Vf = [I_unit ; V]  must give  Q^T A = [Rn; 0]
    const int M = 16 + MB;
    std::vector<double> A(M * 16), Vf(M * 16, 0.0);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) { A[i * 16 + j] = R[i * 16 + j]; Vf[i * 16 + j] = (i == j) ? 1.0 : 0.0; }
    for (int r = 0; r < MB; r++)
        for (int j = 0; j < 16; j++) { A[(16 + r) * 16 + j] = B[r * 16 + j]; Vf[(16 + r) * 16 + j] = V[r * 16 + j]; }
    // W = Vf^T A (16x16), W2 = T^T W, QtA = A - Vf W2
    std::vector<double> W(256, 0.0), W2(256, 0.0);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) { double a = 0; for (int r = 0; r < M; r++) a += Vf[r * 16 + i] * A[r * 16 + j]; W[i * 16 + j] = a; }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) { double a = 0; for (int k = 0; k < 16; k++) a += T[k * 16 + i] * W[k * 16 + j]; W2[i * 16 + j] = a; }
    double err = 0, nrm = 0;
    for (int r = 0; r < M; r++)
        for (int j = 0; j < 16; j++) {
            double a = A[r * 16 + j];
            for (int k = 0; k < 16; k++) a -= Vf[r * 16 + k] * W2[k * 16 + j];
            const double want = (r < 16 && j >= r) ? Rn[r * 16 + j] : 0.0;
            err = fmax(err, fabs(a - want));
            nrm = fmax(nrm, fabs(A[r * 16 + j]));
        }
    // orthogonality of Q: |(Vf^T Vf) vs T^-1 + T^-T| via  T (Vf^T Vf) T^T = T + T^T
    printf("SUB=%d  cycles/panel=%lld  cycles/step=%.0f  max|Q^T A - [R;0]|=%.3e (|A|max %.2f)  tau5=%.3g\n", SUB, cyc, cyc / 16.0, err, nrm, T[5 * 16 + 5]);
    hipFree(dB); hipFree(dR); hipFree(dV); hipFree(dT); hipFree(dRn); hipFree(dc);
    return err < 1e-13 ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += run<1>();
    bad += run<2>();
    bad += run<3>();
    bad += run<4>();
    printf(bad ? "CHAIN PROBE FAILED\n" : "chain probe ok\n");
    return bad;
}
