// Measured bound for "take the Householder panel chain of the TSQR off the fp64 VALU" (DESIGN.md section 10).
//
// The MFMA-heavy alternative for the 16-column panel [R_pp ; B] (16 + MB rows) is CholeskyQR + Householder reconstruction.  With a
// triangular R_pp on top the reconstruction needs no LU (V_top = I):
//     G = R_pp^T R_pp + B^T B  (MFMA)      R_c = chol(G)      R_new = -D R_c,  D = sign(diag R_pp)      U = R_pp + D R_c
//     V_b = B U^-1             T = U R_c^-1 D                 (H = I - [I; V_b] T [I; V_b]^T,  H^T [R_pp; B] = [R_new; 0])
// so that the existing trailing update W = T^T (R_rows + V^T C), C -= V W stays as it is.  What is left on the serial path of the panel
// besides the 4 SUB + 4 Gram MFMAs: a 16 x 16 Cholesky, R_c^-T (free-ish as an augmented elimination beside the Cholesky) and U^-1 (a second
// 16-step triangular elimination of the same shape), plus V_b = B U^-1 (4 SUB MFMAs after an LDS round trip of the block, which holds B
// with lanes over columns -- the A-operand layout of a right-multiplication has lanes over rows).
//
// This probe times, in ONE wave exactly like the product's panel owner:
//   (a) fbr_tsqr_panel_steps (the product's chain, for reference -- same as tools/tsqr_chain_probe.hip);
//   (b) the Gram MFMAs + the register-resident Cholesky with the augmented inverse (16 steps: cross-row broadcast of row j, rsq + Newton,
//       4 + 4 DPP fused multiply-adds) -- i.e. the new formulation WITHOUT U^-1, V_b and T;
// and checks (b) against a host Cholesky.  If (b) alone is not far below (a), the complete formulation (roughly (b) + another 16-step
// elimination + 2 x 4 SUB MFMAs) cannot beat the chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -o tools/_build/panel_mfma_probe tools/panel_mfma_probe.hip
#include "../flobaroid_amd/csrc/fbr_tsqr.h"
#include <cmath>
#include <cstdio>
#include <vector>

// m[r], x1[r]: lane (kk, li) holds row 4 r + kk, column li.  One step of the symmetric right-looking Cholesky of M with the same
// elimination applied to X1 (starts as the identity, ends as L^-1 = R_c^-T); rc / xi receive row J of R_c and of L^-1.
template <int J> __device__ __forceinline__ void chol_step(fbr_td4 &m, fbr_td4 &x1, fbr_td4 &rc, fbr_td4 &xi, int li, int kk)
{
    const int src = (J & 3) * 16 + li;
    const double rowj = __shfl(m[J >> 2], src, 64);   // M[J][li] on every lane (cross-row broadcast through the LDS crossbar)
    const double x1j = __shfl(x1[J >> 2], src, 64);
    const double pv = fbr_dpp_bcast<J>(rowj);         // M[J][J]
    double y = __builtin_amdgcn_rsq(pv);
    y = y * fma(-0.5 * pv, y * y, 1.5);               // one Newton step: 1 / d
    y = y * fma(-0.5 * pv, y * y, 1.5);
    const double inv2 = y * y;
    const double nx = -x1j * inv2, nm = -rowj * inv2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double e = x1[r];
        fbr_fmac_bcast<J, true>(e, m[r], nx);         // X1[i] -= M[i][J] X1[J] / d^2   (rows i != J; row J is taken out below)
        x1[r] = (4 * r + kk == J) ? x1[r] : e;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double e = m[r];
        fbr_fmac_bcast<J, true>(e, e, nm);            // M[i][c] -= M[i][J] M[J][c] / d^2
        m[r] = e;
    }
    if (kk == (J & 3)) {
        rc[J >> 2] = rowj * y;
        xi[J >> 2] = x1j * y;
    }
    asm volatile("" : "+v"(rc[J >> 2]), "+v"(xi[J >> 2]));
    __builtin_amdgcn_sched_barrier(0);
}
template <int... Js> __device__ __forceinline__ void chol_steps(fbr_td4 &m, fbr_td4 &x1, fbr_td4 &rc, fbr_td4 &xi, int li, int kk, std::integer_sequence<int, Js...>)
{
    asm volatile("s_nop 4" ::: "memory");
    (chol_step<Js>(m, x1, rc, xi, li, kk), ...);
}

template <int SUB> __global__ void probe_kernel(const double *Bp, const double *Rpp, double *Rc_out, double *Xi_out, long long *cyc, int reps)
{
    __shared__ double Rp[256];
    const int lane = threadIdx.x, li = lane & 15, kk = lane >> 4;
    fbr_td4 v[SUB], rq, rc = {0, 0, 0, 0}, xi = {0, 0, 0, 0};
    double trow[16], myscale = 0.0;
    long long ta_sum = 0, tb_sum = 0;
    for (int r = 0; r < reps; r++) {
        for (int sb = 0; sb < SUB; sb++)
            for (int reg = 0; reg < 4; reg++) v[sb][reg] = Bp[(16 * sb + 4 * reg + kk) * 16 + li];
        for (int reg = 0; reg < 4; reg++) Rp[(4 * reg + kk) * 16 + li] = Rpp[(4 * reg + kk) * 16 + li];
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // ---- (b) Gram by MFMA + Cholesky with the augmented inverse
        const long long t0 = __builtin_readcyclecounter();
        fbr_td4 g = {0, 0, 0, 0}, rp;
        for (int reg = 0; reg < 4; reg++) rp[reg] = Rp[(4 * reg + kk) * 16 + li];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) g = __builtin_amdgcn_mfma_f64_16x16x4f64(rp[reg], rp[reg], g, 0, 0, 0);
#pragma unroll
        for (int sb = 0; sb < SUB; sb++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) g = __builtin_amdgcn_mfma_f64_16x16x4f64(v[sb][reg], v[sb][reg], g, 0, 0, 0);
        fbr_td4 x1;
        for (int reg = 0; reg < 4; reg++) x1[reg] = (4 * reg + kk == li) ? 1.0 : 0.0;
        chol_steps(g, x1, rc, xi, li, kk, std::make_integer_sequence<int, 16>{});
        asm volatile("" : "+v"(rc[0]), "+v"(xi[0]));
        const long long t1 = __builtin_readcyclecounter();
        // ---- (a) the product's Householder chain on the same panel
        rq = fbr_td4{0, 0, 0, 0};
        myscale = 0.0;
        fbr_tsqr_panel_steps<SUB>(v, Rp, rq, trow, myscale, li, kk, std::make_integer_sequence<int, 16>{});
        asm volatile("" : "+v"(v[0][0]), "+v"(rq[0]));
        const long long t2 = __builtin_readcyclecounter();
        if (r > 0) {
            tb_sum += t1 - t0;
            ta_sum += t2 - t1;
        }
        __syncthreads();
    }
    for (int reg = 0; reg < 4; reg++) {
        Rc_out[(4 * reg + kk) * 16 + li] = rc[reg];
        Xi_out[(4 * reg + kk) * 16 + li] = xi[reg];
    }
    if (lane == 0) {
        cyc[0] = ta_sum / (reps - 1);
        cyc[1] = tb_sum / (reps - 1);
    }
}

template <int SUB> static int run()
{
    const int MB = 16 * SUB;
    std::vector<double> B(MB * 16), R(256, 0.0), Rc(256), Xi(256);
    srand(11 + SUB);
    for (auto &x : B) x = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; i++)
        for (int j = i; j < 16; j++) R[i * 16 + j] = (rand() / (double)RAND_MAX - 0.5) + (i == j ? 2.0 : 0.0);
    double *dB, *dR, *dRc, *dXi;
    long long *dc, cyc[2] = {0, 0};
    hipMalloc(&dB, B.size() * 8); hipMalloc(&dR, 2048); hipMalloc(&dRc, 2048); hipMalloc(&dXi, 2048); hipMalloc(&dc, 16);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, R.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel<SUB>, dim3(1), dim3(64), 0, 0, dB, dR, dRc, dXi, dc, 50);
    hipMemcpy(Rc.data(), dRc, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(Xi.data(), dXi, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
    // host: G = R^T R + B^T B, check Rc^T Rc = G and Xi Rc^T = I
    std::vector<double> G(256, 0.0);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double a = 0;
            for (int k = 0; k < 16; k++) a += R[k * 16 + i] * R[k * 16 + j];
            for (int r = 0; r < MB; r++) a += B[r * 16 + i] * B[r * 16 + j];
            G[i * 16 + j] = a;
        }
    double e1 = 0, e2 = 0, gmax = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double a = 0, b = 0;
            for (int k = 0; k < 16; k++) {
                a += Rc[k * 16 + i] * Rc[k * 16 + j];
                b += Xi[i * 16 + k] * Rc[j * 16 + k];  // (L^-1 L)[i][j], L = Rc^T
            }
            e1 = fmax(e1, fabs(a - G[i * 16 + j]));
            e2 = fmax(e2, fabs(b - (i == j ? 1.0 : 0.0)));
            gmax = fmax(gmax, fabs(G[i * 16 + j]));
        }
    printf("SUB=%d (%d-row block): Householder chain %lld cycles/panel | Gram MFMAs + Cholesky + augmented inverse %lld cycles (= %.2f of the chain; "
           "without U^-1, V_b = B U^-1, T)   |Rc^T Rc - G| = %.2e (|G| %.1f)  |L^-1 L - I| = %.2e\n",
           SUB, MB, cyc[0], cyc[1], (double)cyc[1] / (double)cyc[0], e1, gmax, e2);
    hipFree(dB); hipFree(dR); hipFree(dRc); hipFree(dXi); hipFree(dc);
    return (e1 < 1e-11 * gmax && e2 < 1e-11) ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += run<2>();
    bad += run<3>();
    bad += run<4>();
    printf(bad ? "PANEL PROBE FAILED\n" : "panel probe ok\n");
    return bad;
}
