// fbr_reduce.h -- the column reductions of the Gram / TSQR (host side, no HIP): the reduced robot a model's reductions run on and the
// constant matrix E that expands their results.  Used by fbr_api.hip (build_reduction) and, unchanged, by the CPU emulation of the
// kernels (tests/emul), which checks E^T G_red E against the oracle's Gram of ALL columns without a GPU.
#ifndef FBR_REDUCE_H
#define FBR_REDUCE_H
#include <cmath>
#include <vector>

#include "fbr_math.h"
#include "fbr_program.h"

// ------------------------------------------------------------------------------------------------
// Link merging.  A link attached to its parent by a FIXED joint has no kinematics of its own: a rigid body with parameters pi_c given
// in the link's frame c is the same body with parameters T pi_c in the frame a of the moving body it rides on (x_a = R x_c + r):
//     m' = m,   h' = R h + m r,   I' = R I R^T + 2 (r . Rh) 1 - (r (Rh)^T + (Rh) r^T) + m (|r|^2 1 - r r^T)
// (first moment h = m c, inertia about the frame origin -- the reference's parameter convention, model.py:220-231,
// helpers.py:374-407).  Hence the link's regressor columns are Y_c = Y_a T, for every row and every state: exact column dependencies
// known from the URDF alone (they are why WALK-MAN's 480 columns have rank 213).  The reductions therefore run on the REDUCED robot --
// moving bodies only (WALK-MAN: 30 of 48 links, 300 of 480 columns; rest transforms composed across the fixed links) -- and are
// expanded with the constant E = [T blocks | identity]:  [Y|rhs] = [Y_red|rhs] E^  =>  G = E^T G_red E,  R = qr(R_red E).
// Same results (to rounding), 0.39 of the column pairs.
// ------------------------------------------------------------------------------------------------
static inline void fbr_param_transform(const double *R, const double *r, double T[10][10])
{
    static const int I0[6] = {0, 0, 0, 1, 1, 2}, I1[6] = {0, 1, 2, 1, 2, 2};
    for (int p = 0; p < 10; p++) {
        double pi[10] = {0};
        pi[p] = 1.0;
        const double mass = pi[0], *h = pi + 1;
        double I[3][3];
        for (int e = 0; e < 6; e++) I[I0[e]][I1[e]] = I[I1[e]][I0[e]] = pi[4 + e];
        double Rh[3], RI[3][3], Ia[3][3];
        for (int i = 0; i < 3; i++) Rh[i] = R[3 * i] * h[0] + R[3 * i + 1] * h[1] + R[3 * i + 2] * h[2];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) RI[i][j] = R[3 * i] * I[0][j] + R[3 * i + 1] * I[1][j] + R[3 * i + 2] * I[2][j];
        const double rRh = r[0] * Rh[0] + r[1] * Rh[1] + r[2] * Rh[2], rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                Ia[i][j] = RI[i][0] * R[3 * j] + RI[i][1] * R[3 * j + 1] + RI[i][2] * R[3 * j + 2];
                Ia[i][j] += (i == j ? 2.0 * rRh : 0.0) - (r[i] * Rh[j] + Rh[i] * r[j]) + mass * ((i == j ? rr : 0.0) - r[i] * r[j]);
            }
        T[0][p] = mass;
        for (int i = 0; i < 3; i++) T[1 + i][p] = Rh[i] + mass * r[i];
        for (int e = 0; e < 6; e++) T[4 + e][p] = Ia[I0[e]][I1[e]];
    }
}

// rotation Q (row-major) whose third column is the unit vector along a: the frame in which a joint axis is z
static inline void fbr_axis_frame(const double *a_in, double *Q)
{
    const double na = std::sqrt(a_in[0] * a_in[0] + a_in[1] * a_in[1] + a_in[2] * a_in[2]);
    double a[3] = {a_in[0] / na, a_in[1] / na, a_in[2] / na};
    int i = 0;
    for (int c = 1; c < 3; c++)
        if (std::fabs(a[c]) < std::fabs(a[i])) i = c;
    double u[3] = {0, 0, 0};
    u[i] = 1.0;
    const double ua = a[i];
    for (int c = 0; c < 3; c++) u[c] -= ua * a[c];
    const double nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int c = 0; c < 3; c++) u[c] /= nu;
    const double v[3] = {a[1] * u[2] - a[2] * u[1], a[2] * u[0] - a[0] * u[2], a[0] * u[1] - a[1] * u[0]};
    for (int r = 0; r < 3; r++) {
        Q[3 * r] = u[r];
        Q[3 * r + 1] = v[r];
        Q[3 * r + 2] = a[r];
    }
}

struct FbrReducedRobot {
    int Lr = 0;
    std::vector<int32_t> parent, dof, jtype;   // [Lr]
    std::vector<double> restR, restp, axis;    // [9 Lr], [3 Lr], [3 Lr]
    std::vector<unsigned short> masks;         // [Lr] identified parameters of every reduced link (0x3ff: all ten)
    bool masked = false;                       // some link has a mask (regrouping)
    std::vector<int> red_of, body;             // [L] reduced index of a moving link (-1: fixed) ; moving body a link rides on
    std::vector<double> bR, bp;                // [9 L], [3 L]  x_body = bR x_link + bp
};

// which = 0: fixed links merged into the moving bodies; which = 1: the same, and the regroupable parameters of every link behind a joint
// dropped (revolute: three -- m, h along the axis, the inertia 1 - a a^T; prismatic: the six inertia entries).  false: nothing to reduce (no such links, or a gravity-only model).
static inline bool fbr_reduce_robot(const FbrHostModel &hm, int which, FbrReducedRobot &rr)
{
    const int L = hm.L;
    const bool regroup = which == 1;
    int nfixed = 0, nrev = 0;
    for (int l = 0; l < L; l++) {
        nfixed += hm.parent[l] >= 0 && hm.dof[l] < 0;
        nrev += hm.parent[l] >= 0 && hm.dof[l] >= 0;
    }
    if (regroup ? nrev == 0 : nfixed == 0) return false;
    // gravity-only models keep 4 of a link's 10 columns (m, h): the frame change feeds m and h into the INERTIA of the body as well, so
    // the kept columns of a fixed link are not combinations of the kept columns of its body -- nothing is reduced there
    if (hm.grav_only) return false;
    // moving bodies in link order; for every link: its body and the constant transform body <- link
    std::vector<int> &red_of = rr.red_of, &body = rr.body, moving;
    red_of.assign(L, -1);
    body.assign(L, -1);
    for (int l = 0; l < L; l++)
        if (hm.parent[l] < 0 || hm.dof[l] >= 0) {
            red_of[l] = (int)moving.size();
            moving.push_back(l);
        }
    // (regrouping: the frame of a body behind a joint is turned so that the joint axis becomes z, x_link = Q x_body)
    std::vector<double> Q((size_t)9 * L, 0.0);
    for (int l = 0; l < L; l++) {
        for (int i = 0; i < 3; i++) Q[9 * l + 4 * i] = 1.0;
        if (regroup && hm.parent[l] >= 0 && hm.dof[l] >= 0 && hm.jtype[l] == 1) fbr_axis_frame(&hm.axis[3 * l], &Q[9 * l]);  // (revolute)
    }
    std::vector<double> &bR = rr.bR, &bp = rr.bp;  // x_body = bR x_link + bp
    bR.assign((size_t)9 * L, 0.0);
    bp.assign((size_t)3 * L, 0.0);
    for (int l : hm.order) {  // parents first
        double *R = &bR[9 * l], *p = &bp[3 * l];
        if (red_of[l] >= 0) {
            body[l] = l;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R[3 * i + j] = Q[9 * l + 3 * j + i];  // Q^T
            p[0] = p[1] = p[2] = 0.0;
        } else {
            const int q = hm.parent[l];
            body[l] = body[q];
            const double *Rq = &bR[9 * q], *pq = &bp[3 * q], *Rl = &hm.restR[9 * l], *pl = &hm.restp[3 * l];
            fbr_mm(Rq, Rl, R);
            double tmp[3];
            fbr_mv(Rq, pl, tmp);
            for (int i = 0; i < 3; i++) p[i] = pq[i] + tmp[i];
        }
    }
    // the reduced robot: a moving link hangs off the body of its parent, its rest transform composed across the fixed links between
    const int Lr = (int)moving.size();
    rr.Lr = Lr;
    std::vector<int32_t> &rparent = rr.parent, &rdof = rr.dof;
    rr.jtype.assign(moving.size(), 0);
    std::vector<double> &rR = rr.restR, &rp = rr.restp, &rax = rr.axis;
    std::vector<unsigned short> &masks = rr.masks;
    rparent.assign(Lr, -1);
    rdof.assign(Lr, -1);
    rR.assign((size_t)9 * Lr, 0.0);
    rp.assign((size_t)3 * Lr, 0.0);
    rax.assign((size_t)3 * Lr, 0.0);
    masks.assign(Lr, 0x3ff);
    for (int i = 0; i < Lr; i++) {
        const int l = moving[i], q = hm.parent[l];
        rdof[i] = hm.dof[l];
        rr.jtype[i] = hm.jtype[l];
        if (q < 0) {
            rparent[i] = -1;
            for (int c = 0; c < 9; c++) rR[9 * i + c] = hm.restR[9 * l + c];
            for (int c = 0; c < 3; c++) rp[3 * i + c] = hm.restp[3 * l + c];
            for (int c = 0; c < 3; c++) rax[3 * i + c] = hm.axis[3 * l + c];
        } else {
            rparent[i] = red_of[body[q]];
            double tmpR[9], tmp[3];
            fbr_mm(&bR[9 * q], &hm.restR[9 * l], tmpR);
            fbr_mm(tmpR, &Q[9 * l], &rR[9 * i]);
            fbr_mv(&bR[9 * q], &hm.restp[3 * l], tmp);
            for (int c = 0; c < 3; c++) rp[3 * i + c] = bp[3 * q + c] + tmp[c];
            for (int c = 0; c < 3; c++)  // Q^T axis (= |axis| z when regrouping)
                rax[3 * i + c] = Q[9 * l + c] * hm.axis[3 * l] + Q[9 * l + 3 + c] * hm.axis[3 * l + 1] + Q[9 * l + 6 + c] * hm.axis[3 * l + 2];
            // revolute: m, h_z, I_yy dropped.  prismatic: the link turns with its parent whatever the joint does, so all six inertia entries
            // act like inertia of the parent body (Y_i[I] = Y_par T e_I); m and h feel the displacement and stay
            if (regroup) masks[i] = hm.jtype[l] == 2 ? 0x00f : (0x3ff & ~((1u << 0) | (1u << 3) | (1u << 7)));
        }
    }
    rr.masked = regroup;
    return true;
}

// CSC of the augmented E [(cols_red + 16) x (cols + 16)] for the reduced model rh built from rr: column j of the full layout = sum of
// val[e] x (reduced column row[e]), e in [beg[j], beg[j + 1]); the 16 rhs columns are an identity
static inline void fbr_reduction_matrix(const FbrHostModel &hm, const FbrReducedRobot &rr, const FbrHostModel &rh, std::vector<int> &beg,
                                        std::vector<int> &row, std::vector<double> &val)
{
    const int L = hm.L, Lr = rr.Lr;
    const std::vector<int32_t> &rparent = rr.parent;
    const std::vector<double> &rR = rr.restR, &rp = rr.restp, &bR = rr.bR, &bp = rr.bp;
    const std::vector<unsigned short> &masks = rr.masks;
    const std::vector<int> &red_of = rr.red_of, &body = rr.body;
    // X_i [Pr x 10]: the 10 standard columns of reduced link i (its own frame) in terms of the columns the reduced model computes.
    // Kept parameters: unit vectors.  Dropped ones (regrouping), with T the frame change link i -> parent body at q = 0:
    //     Y_i[m] = Y_par T e_m,   Y_i[h_z] = Y_par T e_hz,   Y_i[I_yy] = Y_par T (e_Ixx + e_Iyy) - Y_i[I_xx]
    // (mass, first moment along the axis and the inertia 1 - z z^T of the link do not notice the joint's rotation)
    const int Pr = rh.cols, Pf = hm.cols, cpl = hm.cpl;
    std::vector<std::vector<double>> X((size_t)Lr * 10, std::vector<double>(Pr, 0.0));
    std::vector<int> colof((size_t)Lr * 10, -1);
    for (int c = 0; c < rh.ninert; c++) colof[(size_t)rh.coldesc[c].link * 10 + rh.coldesc[c].pidx] = c;
    for (int i : rh.order) {
        for (int p = 0; p < 10; p++)
            if (colof[(size_t)i * 10 + p] >= 0) X[(size_t)i * 10 + p][colof[(size_t)i * 10 + p]] = 1.0;
        if (masks[i] == 0x3ff) continue;
        double T[10][10];
        fbr_param_transform(&rR[9 * i], &rp[3 * i], T);
        const int par = rparent[i];
        auto add = [&](int pdst, int psrc, double sgn) {
            std::vector<double> &dst = X[(size_t)i * 10 + pdst];
            for (int pr = 0; pr < 10; pr++) {
                const double tv = sgn * T[pr][psrc];
                if (tv == 0.0) continue;
                const std::vector<double> &src = X[(size_t)par * 10 + pr];
                for (int c = 0; c < Pr; c++) dst[c] += tv * src[c];
            }
        };
        if (masks[i] == 0x00f) {  // prismatic: the whole inertia tensor rides on the parent
            for (int e = 4; e < 10; e++) add(e, e, 1.0);
            continue;
        }
        add(0, 0, 1.0);
        add(3, 3, 1.0);
        add(7, 4, 1.0);
        add(7, 7, 1.0);
        for (int c = 0; c < Pr; c++) X[(size_t)i * 10 + 7][c] -= X[(size_t)i * 10 + 4][c];
    }
    // E, augmented with 16 rhs columns, column by column of the FULL layout
    beg.assign(Pf + FBR_MAX_RHS + 1, 0);
    row.clear();
    val.clear();
    std::vector<double> col(Pr);
    for (int l = 0; l < L; l++) {
        double T[10][10];
        fbr_param_transform(&bR[9 * l], &bp[3 * l], T);
        const int b = red_of[body[l]];
        for (int p = 0; p < cpl; p++) {
            beg[cpl * l + p] = (int)row.size();
            std::fill(col.begin(), col.end(), 0.0);
            for (int pr = 0; pr < 10; pr++) {
                if (T[pr][p] == 0.0) continue;
                const std::vector<double> &src = X[(size_t)b * 10 + pr];
                for (int c = 0; c < Pr; c++) col[c] += T[pr][p] * src[c];
            }
            for (int c = 0; c < Pr; c++)
                if (col[c] != 0.0) {
                    row.push_back(c);
                    val.push_back(col[c]);
                }
        }
    }
    for (int j = hm.ninert; j < Pf; j++) {  // friction columns: the same joints in the same layout
        beg[j] = (int)row.size();
        row.push_back(rh.ninert + (j - hm.ninert));
        val.push_back(1.0);
    }
    for (int r = 0; r < FBR_MAX_RHS; r++) {
        beg[Pf + r] = (int)row.size();
        row.push_back(Pr + r);
        val.push_back(1.0);
    }
    beg[Pf + FBR_MAX_RHS] = (int)row.size();
}
#endif  // FBR_REDUCE_H
