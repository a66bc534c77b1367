// fbr_math.h -- per-sample rigid-body arithmetic shared by the HIP kernels.
//
// Formulation: "base-frame composite regressor".  All link wrenches are expressed in ONE
// frame A (world-aligned axes, origin at the base-link origin == iDynTree's MIXED representation for
// the zero-position world_T_base the reference always passes, identification/model.py:431-432), so
//   base rows of link l        = W_l              (6 x 10, wrench of unit parameters moved to A)
//   joint row d (d ancestor l) = S_d^T W_l        (S_d = joint motion vector in A)
// which needs no per-hop 6x6 transforms.  Link kinematics use the classical (non-spatial) recursion:
// angular velocity/acceleration and the PROPER linear acceleration of the link origin in link axes.
//
// The functions are __host__ __device__ so tests can compile this header with g++ and check the
// device arithmetic on the CPU (tests/emul/).
#pragma once

#if defined(__HIPCC__)
#define FBR_HD __host__ __device__ __forceinline__
#else
#define FBR_HD inline
#endif

#include <math.h>

// per-link record: R(9) p(3) w(3) dw(3) a(3)
#define FBR_LINK_REC 21
#define FBR_OFF_R 0
#define FBR_OFF_P 9
#define FBR_OFF_W 12
#define FBR_OFF_DW 15
#define FBR_OFF_A 18
// per-dof record: S = [lin(3); ang(3)] in frame A
#define FBR_DOF_REC 6

FBR_HD void fbr_cross(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
FBR_HD void fbr_mv(const double *A, const double *x, double *y)  // y = A x (row-major 3x3)
{
    y[0] = A[0] * x[0] + A[1] * x[1] + A[2] * x[2];
    y[1] = A[3] * x[0] + A[4] * x[1] + A[5] * x[2];
    y[2] = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
}
FBR_HD void fbr_mtv(const double *A, const double *x, double *y)  // y = A^T x
{
    y[0] = A[0] * x[0] + A[3] * x[1] + A[6] * x[2];
    y[1] = A[1] * x[0] + A[4] * x[1] + A[7] * x[2];
    y[2] = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
}
FBR_HD void fbr_mm(const double *A, const double *B, double *C)  // C = A B
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Base link record.  Floating: world_R_base = RPY(rpy)^T; w, dw = R^T (mixed angular vel/acc);
// a = R^T (mixed linear acc - g).  The mixed LINEAR velocity drops out of the classical recursion.
FBR_HD void fbr_kin_base(int floating, const double *g, const double *base_vel, const double *base_acc,
                         const double *rpy, double *rec)
{
    double *R = rec + FBR_OFF_R;
    if (floating) {
        const double cr = cos(rpy[0]), sr = sin(rpy[0]);
        const double cp = cos(rpy[1]), sp = sin(rpy[1]);
        const double cy = cos(rpy[2]), sy = sin(rpy[2]);
        // transpose of Rz(y)Ry(p)Rx(r)
        R[0] = cy * cp;
        R[3] = cy * sp * sr - sy * cr;
        R[6] = cy * sp * cr + sy * sr;
        R[1] = sy * cp;
        R[4] = sy * sp * sr + cy * cr;
        R[7] = sy * sp * cr - cy * sr;
        R[2] = -sp;
        R[5] = cp * sr;
        R[8] = cp * cr;
        double al[3] = {base_acc[0] - g[0], base_acc[1] - g[1], base_acc[2] - g[2]};
        fbr_mtv(R, base_vel + 3, rec + FBR_OFF_W);
        fbr_mtv(R, base_acc + 3, rec + FBR_OFF_DW);
        fbr_mtv(R, al, rec + FBR_OFF_A);
    } else {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        for (int i = 0; i < 3; i++) {
            rec[FBR_OFF_W + i] = 0.0;
            rec[FBR_OFF_DW + i] = 0.0;
            rec[FBR_OFF_A + i] = -g[i];
        }
    }
    rec[FBR_OFF_P] = rec[FBR_OFF_P + 1] = rec[FBR_OFF_P + 2] = 0.0;
}

// Child link record from the parent's.  jt: 0 fixed, 1 revolute about `s` (unit, child frame) by q, 2 prismatic along `s` by q.
// Srec (6) receives the joint motion vector in A when jt != 0.
FBR_HD void fbr_kin_child(const double *par, const double *restR, const double *r0, const double *s, int jt,
                          double q, double dq, double ddq, double *rec, double *Srec)
{
    double Rj[9], r[3] = {r0[0], r0[1], r0[2]}, sp[3] = {0, 0, 0};
    if (jt == 1) {
        const double c = cos(q), sn = sin(q), v = 1.0 - c;
        double Rq[9];
        Rq[0] = c + s[0] * s[0] * v;
        Rq[1] = s[0] * s[1] * v - s[2] * sn;
        Rq[2] = s[0] * s[2] * v + s[1] * sn;
        Rq[3] = s[1] * s[0] * v + s[2] * sn;
        Rq[4] = c + s[1] * s[1] * v;
        Rq[5] = s[1] * s[2] * v - s[0] * sn;
        Rq[6] = s[2] * s[0] * v - s[1] * sn;
        Rq[7] = s[2] * s[1] * v + s[0] * sn;
        Rq[8] = c + s[2] * s[2] * v;
        fbr_mm(restR, Rq, Rj);
    } else {
        for (int i = 0; i < 9; i++) Rj[i] = restR[i];
        if (jt == 2) {  // the child origin slides along the axis (parent axes: restR s)
            fbr_mv(restR, s, sp);
            for (int i = 0; i < 3; i++) r[i] += sp[i] * q;
        }
    }
    const double *Rp = par + FBR_OFF_R, *wp = par + FBR_OFF_W, *dwp = par + FBR_OFF_DW, *ap = par + FBR_OFF_A;
    fbr_mm(Rp, Rj, rec + FBR_OFF_R);
    double t[3];
    fbr_mv(Rp, r, t);
    for (int i = 0; i < 3; i++) rec[FBR_OFF_P + i] = par[FBR_OFF_P + i] + t[i];
    // acceleration of the child origin in parent axes
    double dwxr[3], wxr[3], wxwxr[3], acc[3];
    fbr_cross(dwp, r, dwxr);
    fbr_cross(wp, r, wxr);
    fbr_cross(wp, wxr, wxwxr);
    for (int i = 0; i < 3; i++) acc[i] = ap[i] + dwxr[i] + wxwxr[i];
    if (jt == 2) {  // relative motion of the origin along the axis: Coriolis 2 w x (s dq) + s ddq
        double wxs[3];
        fbr_cross(wp, sp, wxs);
        for (int i = 0; i < 3; i++) acc[i] += 2.0 * wxs[i] * dq + sp[i] * ddq;
    }
    fbr_mtv(Rj, acc, rec + FBR_OFF_A);
    double wl[3], dwl[3];
    fbr_mtv(Rj, wp, wl);
    fbr_mtv(Rj, dwp, dwl);
    if (jt == 1) {
        double wxs[3];
        fbr_cross(wl, s, wxs);
        for (int i = 0; i < 3; i++) {
            rec[FBR_OFF_W + i] = wl[i] + s[i] * dq;
            rec[FBR_OFF_DW + i] = dwl[i] + s[i] * ddq + wxs[i] * dq;
        }
        // S = [p x sA ; sA]
        fbr_mv(rec + FBR_OFF_R, s, Srec + 3);
        fbr_cross(rec + FBR_OFF_P, Srec + 3, Srec);
    } else {
        for (int i = 0; i < 3; i++) {
            rec[FBR_OFF_W + i] = wl[i];
            rec[FBR_OFF_DW + i] = dwl[i];
        }
        if (jt == 2) {  // S = [sA ; 0]: a unit velocity along the axis, the same at every point
            fbr_mv(rec + FBR_OFF_R, s, Srec);
            Srec[3] = Srec[4] = Srec[5] = 0.0;
        }
    }
}

// Wrench (in frame A, about the base origin) produced by unit standard parameter `pidx` (0..9) of a link
// with record `rec`:  column pidx of W_l.
FBR_HD void fbr_unit_wrench(const double *rec, int pidx, double *w6)
{
    const double *w = rec + FBR_OFF_W, *dw = rec + FBR_OFF_DW, *a = rec + FBR_OFF_A;
    double f[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (pidx == 0) {
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2];
    } else if (pidx < 4) {
        // h = e_k:  f = dw x e + w x (w x e),  n = e x a
        double e[3] = {0, 0, 0};
        e[pidx - 1] = 1.0;
        double dwxe[3], wxe[3], wxwxe[3];
        fbr_cross(dw, e, dwxe);
        fbr_cross(w, e, wxe);
        fbr_cross(w, wxe, wxwxe);
        for (int i = 0; i < 3; i++) f[i] = dwxe[i] + wxwxe[i];
        fbr_cross(e, a, n);
    } else {
        // Ibar = E_ab (symmetric unit):  n = E dw + w x (E w)
        // vech order xx,xy,xz,yy,yz,zz -> (a,b)
        const int ia = (pidx == 4 || pidx == 5 || pidx == 6) ? 0 : ((pidx == 7 || pidx == 8) ? 1 : 2);
        const int ib = (pidx == 4) ? 0 : ((pidx == 5 || pidx == 7) ? 1 : 2);
        double Edw[3] = {0, 0, 0}, Ew[3] = {0, 0, 0};
        Edw[ia] = dw[ib];
        Ew[ia] = w[ib];
        if (ia != ib) {
            Edw[ib] = dw[ia];
            Ew[ib] = w[ia];
        }
        double wxEw[3];
        fbr_cross(w, Ew, wxEw);
        for (int i = 0; i < 3; i++) n[i] = Edw[i] + wxEw[i];
    }
    const double *R = rec + FBR_OFF_R, *p = rec + FBR_OFF_P;
    fbr_mv(R, f, w6);
    double nA[3], pxf[3];
    fbr_mv(R, n, nA);
    fbr_cross(p, w6, pxf);
    for (int i = 0; i < 3; i++) w6[3 + i] = nA[i] + pxf[i];
}

// Inertia parameters (pidx 4..9) produce a pure moment: the force half of fbr_unit_wrench's column is zero and its moment half is R n.
// n3 = that moment in frame A -- bit for bit w6[3..5] of fbr_unit_wrench up to the sign of a zero.
FBR_HD void fbr_unit_moment3(const double *rec, int pidx, double *n3)
{
    const double *w = rec + FBR_OFF_W, *dw = rec + FBR_OFF_DW;
    const int ia = (pidx == 4 || pidx == 5 || pidx == 6) ? 0 : ((pidx == 7 || pidx == 8) ? 1 : 2);
    const int ib = (pidx == 4) ? 0 : ((pidx == 5 || pidx == 7) ? 1 : 2);
    double Edw[3] = {0, 0, 0}, Ew[3] = {0, 0, 0};
    Edw[ia] = dw[ib];
    Ew[ia] = w[ib];
    if (ia != ib) {
        Edw[ib] = dw[ia];
        Ew[ib] = w[ia];
    }
    double wxEw[3], n[3];
    fbr_cross(w, Ew, wxEw);
    for (int i = 0; i < 3; i++) n[i] = Edw[i] + wxEw[i];
    fbr_mv(rec + FBR_OFF_R, n, n3);
}

// Net wrench (frame A) of a link with the 10 parameters pi:  W_l pi  (used by inverse dynamics / predict)
FBR_HD void fbr_link_wrench(const double *rec, const double *pi, double *w6)
{
    const double *w = rec + FBR_OFF_W, *dw = rec + FBR_OFF_DW, *a = rec + FBR_OFF_A;
    const double m = pi[0], *h = pi + 1;
    double dwxh[3], wxh[3], wxwxh[3], hxa[3];
    fbr_cross(dw, h, dwxh);
    fbr_cross(w, h, wxh);
    fbr_cross(w, wxh, wxwxh);
    fbr_cross(h, a, hxa);
    double Idw[3] = {pi[4] * dw[0] + pi[5] * dw[1] + pi[6] * dw[2], pi[5] * dw[0] + pi[7] * dw[1] + pi[8] * dw[2],
                     pi[6] * dw[0] + pi[8] * dw[1] + pi[9] * dw[2]};
    double Iw[3] = {pi[4] * w[0] + pi[5] * w[1] + pi[6] * w[2], pi[5] * w[0] + pi[7] * w[1] + pi[8] * w[2],
                    pi[6] * w[0] + pi[8] * w[1] + pi[9] * w[2]};
    double wxIw[3];
    fbr_cross(w, Iw, wxIw);
    double f[3], n[3];
    for (int i = 0; i < 3; i++) {
        f[i] = m * a[i] + dwxh[i] + wxwxh[i];
        n[i] = Idw[i] + wxIw[i] + hxa[i];
    }
    const double *R = rec + FBR_OFF_R, *p = rec + FBR_OFF_P;
    fbr_mv(R, f, w6);
    double nA[3], pxf[3];
    fbr_mv(R, n, nA);
    fbr_cross(p, w6, pxf);
    for (int i = 0; i < 3; i++) w6[3 + i] = nA[i] + pxf[i];
}

FBR_HD double fbr_dot6(const double *S, const double *w6)
{
    return S[0] * w6[0] + S[1] * w6[1] + S[2] * w6[2] + S[3] * w6[3] + S[4] * w6[4] + S[5] * w6[5];
}

// friction column value: kind 0 Coulomb(sign) 1 viscous(sym) 2 viscous+ 3 viscous- 4 offset 5 Stribeck
FBR_HD double fbr_friction_value(int kind, double dq, double sign, double vs)
{
    switch (kind) {
    case 0: return sign;
    case 1: return dq;
    case 2: return dq < 0 ? 0.0 : dq;
    case 3: return dq > 0 ? 0.0 : dq;
    case 4: return 1.0;
    default: {
        const double sg = (dq > 0) - (dq < 0);
        return exp(-fabs(dq) / vs) * sg;
    }
    }
}
