// fbr_emul.cpp -- CPU emulation of the HIP kernels' data flow (TEST ONLY, built by tests/conftest.py with g++).
//
// It compiles the SAME headers the kernels use (csrc/fbr_math.h, csrc/fbr_program.h) and mirrors the
// kernels' indexing (records, packed tile image, pair slots, MFMA lane mapping), so table / index bugs
// show up in the CPU test-suite instead of costing a GPU round trip.  Nothing here is shipped or used
// by the product path.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../flobaroid_amd/csrc/fbr_math.h"
#include "../../flobaroid_amd/csrc/fbr_program.h"
#include "../../flobaroid_amd/csrc/fbr_reduce.h"
#include "../../flobaroid_amd/csrc/fbr_kinid.h"
#include "../../flobaroid_amd/csrc/fbr_gram64.h"

extern "C" {

struct EmulTopo {
    int L, n;
    const int32_t *parent, *dof;
    const double *restR, *restp, *axis;
    int floating;
    double gravity[3];
    int fric, fric_sym, grav_only;
    double stribeck;
    const unsigned short *masks;  // per link: identified parameters (column masks of the regrouped model), or null
    const int32_t *jtype;         // per link: 1 revolute, 2 prismatic (links with a DOF), or null: all revolute
};

static void make(const EmulTopo *t, FbrHostModel &hm)
{
    hm.build(t->L, t->n, t->parent, t->dof, t->restR, t->restp, t->axis, t->floating, t->gravity, t->fric,
             t->fric_sym, t->grav_only, t->stribeck, t->masks, t->jtype);
}

// The reduced robot of a model's column reductions and the expansion matrix (csrc/fbr_reduce.h, what fbr_api.hip's build_reduction
// runs): which = 0 fixed links merged, 1 merged + regrouped.  Returns the reduced link count (0: nothing to reduce); the arrays hold at
// least L links; E is dense [Pr][Pf] (reduced x full identified columns, without the rhs identity), Pr is returned through *Pr_out.
int emul_reduction(const EmulTopo *t, int which, int32_t *parent, int32_t *dof, double *restR, double *restp, double *axis,
                   unsigned short *masks, int *masked, int *Pr_out, double *E, long E_cap, int32_t *jtype)
{
    FbrHostModel hm;
    make(t, hm);
    FbrReducedRobot rr;
    if (!fbr_reduce_robot(hm, which, rr)) return 0;
    for (int i = 0; i < rr.Lr; i++) {
        parent[i] = rr.parent[i];
        dof[i] = rr.dof[i];
        masks[i] = rr.masks[i];
        jtype[i] = rr.jtype[i];
        for (int c = 0; c < 9; c++) restR[9 * i + c] = rr.restR[9 * i + c];
        for (int c = 0; c < 3; c++) {
            restp[3 * i + c] = rr.restp[3 * i + c];
            axis[3 * i + c] = rr.axis[3 * i + c];
        }
    }
    *masked = rr.masked ? 1 : 0;
    FbrHostModel rh;
    rh.build(rr.Lr, t->n, rr.parent.data(), rr.dof.data(), rr.restR.data(), rr.restp.data(), rr.axis.data(), t->floating, t->gravity,
             t->fric, t->fric_sym, t->grav_only, t->stribeck, rr.masked ? rr.masks.data() : nullptr, rr.jtype.data());
    std::vector<int> beg, row;
    std::vector<double> val;
    fbr_reduction_matrix(hm, rr, rh, beg, row, val);
    *Pr_out = rh.cols;
    if ((long)rh.cols * hm.cols > E_cap) return -1;
    for (long i = 0; i < (long)rh.cols * hm.cols; i++) E[i] = 0.0;
    for (int j = 0; j < hm.cols; j++)
        for (int e = beg[j]; e < beg[j + 1]; e++) E[(long)row[e] * hm.cols + j] = val[e];
    return rr.Lr;
}

// mirrors fbr_kin_kernel: one "lane" per sample, links in traversal order, AoS record
static void kin_sample(const FbrHostModel &hm, const double *q, const double *dq, const double *ddq, const double *bv,
                       const double *ba, const double *rpy, double *rec)
{
    for (int l : hm.order) {
        double *r = rec + FBR_LINK_REC * l;
        if (hm.parent[l] < 0) {
            fbr_kin_base(hm.floating, hm.gravity, bv, ba, rpy, r);
        } else {
            int d = hm.dof[l];
            fbr_kin_child(rec + FBR_LINK_REC * hm.parent[l], &hm.restR[9 * l], &hm.restp[3 * l], &hm.axis[3 * l], hm.jtype[l],
                          d >= 0 ? q[d] : 0.0, d >= 0 ? dq[d] : 0.0, d >= 0 ? ddq[d] : 0.0, r,
                          d >= 0 ? rec + FBR_LINK_REC * hm.L + 6 * d : nullptr);
        }
    }
}

int emul_dims(const EmulTopo *t, int *rows, int *cols, int *rec)
{
    FbrHostModel hm;
    make(t, hm);
    *rows = hm.rows; *cols = hm.cols; *rec = hm.rec_size();
    return 0;
}

int emul_kin(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv,
             const double *ba, const double *rpy, double *rec_out)
{
    FbrHostModel hm;
    make(t, hm);
    const int REC = hm.rec_size();
    for (long s = 0; s < S; s++)
        kin_sample(hm, q + s * hm.n, dq + s * hm.n, ddq + s * hm.n, hm.floating ? bv + 6 * s : nullptr,
                   hm.floating ? ba + 6 * s : nullptr, hm.floating ? rpy + 3 * s : nullptr, rec_out + s * REC);
    return 0;
}

// mirrors fbr_regressor_kernel: one thread per column, rows emitted in order
int emul_regressor(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv,
                   const double *ba, const double *rpy, const double *sign, double *Y)
{
    FbrHostModel hm;
    make(t, hm);
    const int REC = hm.rec_size();
    std::vector<double> rec(REC);
    for (long s = 0; s < S; s++) {
        kin_sample(hm, q + s * hm.n, dq + s * hm.n, ddq + s * hm.n, hm.floating ? bv + 6 * s : nullptr,
                   hm.floating ? ba + 6 * s : nullptr, hm.floating ? rpy + 3 * s : nullptr, rec.data());
        double *Ys = Y + (size_t)s * hm.rows * hm.cols;
        for (int c = 0; c < hm.cols; c++) {
            const FbrCol &cd = hm.coldesc[c];
            for (int r = 0; r < hm.rows; r++) Ys[(size_t)r * hm.cols + c] = 0.0;
            if (cd.kind == 0) {
                double w6[6];
                fbr_unit_wrench(&rec[FBR_LINK_REC * cd.link], cd.pidx, w6);
                for (int r = 0; r < hm.fb; r++) Ys[(size_t)r * hm.cols + c] = w6[r];
                for (int d : hm.path[cd.link])
                    Ys[(size_t)(hm.fb + d) * hm.cols + c] = fbr_dot6(&rec[FBR_LINK_REC * hm.L + 6 * d], w6);
            } else {
                int j = cd.joint;
                Ys[(size_t)(hm.fb + j) * hm.cols + c] =
                    fbr_friction_value(cd.pidx, dq[s * hm.n + j], sign ? sign[s * hm.n + j] : 0.0, hm.stribeck);
            }
        }
    }
    return 0;
}

// mirrors fbr_id_kernel: per link net wrench in frame A, subtree sums, S^T projection, friction model
// mode 0: x is the full standard vector (10 per link + friction slots, model.py:299-326 semantics)
// mode 1: x is an identified-parameter vector (cols), tau = Y x   (predict)
int emul_id(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv,
            const double *ba, const double *rpy, const double *sign, const double *vel_sign, const double *x, int mode,
            double *tau)
{
    FbrHostModel hm;
    make(t, hm);
    const int REC = hm.rec_size();
    std::vector<double> rec(REC), F(6 * hm.L);
    for (long s = 0; s < S; s++) {
        kin_sample(hm, q + s * hm.n, dq + s * hm.n, ddq + s * hm.n, hm.floating ? bv + 6 * s : nullptr,
                   hm.floating ? ba + 6 * s : nullptr, hm.floating ? rpy + 3 * s : nullptr, rec.data());
        double *ts = tau + (size_t)s * hm.rows;
        for (int r = 0; r < hm.rows; r++) ts[r] = 0.0;
        for (int l = 0; l < hm.L; l++) {
            double pi[10] = {0};
            if (mode == 0) {
                for (int p = 0; p < 10; p++) pi[p] = x[10 * l + p];
            } else {
                for (int p = 0; p < hm.cpl; p++) pi[p] = x[hm.cpl * l + p];
            }
            fbr_link_wrench(&rec[FBR_LINK_REC * l], pi, &F[6 * l]);
            for (int r = 0; r < hm.fb; r++) ts[r] += F[6 * l + r];
            for (int d : hm.path[l]) ts[hm.fb + d] += fbr_dot6(&rec[FBR_LINK_REC * hm.L + 6 * d], &F[6 * l]);
        }
        if (hm.fric) {
            const int n = hm.n;
            if (mode == 0) {
                const int f0 = hm.friction_start();
                for (int j = 0; j < n; j++) {
                    double sg = sign[s * n + j];
                    double v = sg * x[f0 + j];
                    if (!hm.grav_only) {
                        v += x[f0 + n + j] * dq[s * n + j];
                        const int poff = f0 + 2 * n;
                        v += x[poff + j];
                        if (hm.stribeck > 0) {
                            double sgn = (sg > 0) - (sg < 0);
                            v += x[poff + n + j] * exp(-fabs(vel_sign[s * n + j]) / hm.stribeck) * sgn;
                        }
                    }
                    ts[hm.fb + j] += v;
                }
            } else {
                for (int c = hm.cpl * hm.L; c < hm.cols; c++) {
                    const FbrCol &cd = hm.coldesc[c];
                    ts[hm.fb + cd.joint] +=
                        x[c] * fbr_friction_value(cd.pidx, dq[s * n + cd.joint], sign[s * n + cd.joint], hm.stribeck);
                }
            }
        }
    }
    return 0;
}

// mirrors fbr_kinid_kernel (csrc/fbr_kinid.h): the SAME host-built step program and the SAME lane body as the device kernel, one "lane"
// per sample, branch-point records in slots, torques written when their level of the joint stack is taken again.  info (optional, 3):
// nsteps, maxlvl, nslots of the program.  Returns -1 when the tree is deeper than the kernel instances cover.
int emul_kinid(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv, const double *ba,
               const double *rpy, const double *sign, const double *vel_sign, const double *x, int mode, double *tau, int *info, int flink,
               const double *fp)
{
    FbrHostModel hm;
    make(t, hm);
    if (hm.maxdepth > FBR_KINID_MAXD) return -1;
    FbrKinIdProgram p;
    fbr_kinid_build(hm, p);
    if (info) {
        info[0] = p.nsteps;
        info[1] = p.maxlvl;
        info[2] = p.nslots;
    }
    const int n = hm.n;
    std::vector<double> slots((size_t)std::max(p.nslots, 1) * FBR_LINK_REC);
    for (long s = 0; s < S; s++) {
        double *ts = tau + (size_t)s * hm.rows;
        for (int r = 0; r < hm.rows; r++) ts[r] = std::nan("");  // every row must be written exactly once
        std::vector<int> written(hm.rows, 0);
        for (auto &v : slots) v = std::nan("");
        auto state = [&](int d, double &a, double &b, double &c) {
            a = q[s * n + d];
            b = dq[s * n + d];
            c = ddq[s * n + d];
        };
        auto basest = [&](double *v6, double *a6, double *e3) {
            for (int i = 0; i < 6; i++) {
                v6[i] = bv[6 * s + i];
                a6[i] = ba[6 * s + i];
            }
            for (int i = 0; i < 3; i++) e3[i] = rpy[3 * s + i];
        };
        auto save = [&](int b, int i, double v) { slots[(size_t)b * FBR_LINK_REC + i] = v; };
        auto load = [&](int b, int i) { return slots[(size_t)b * FBR_LINK_REC + i]; };
        auto link = [&](int l, int, const double *rec, const double (*)[6], const int *, double *F) {
            if (mode == 2) {  // contact wrench at (flink, fp): x = [S][6]
                if (l != flink) return;
                const double *w = x + s * 6;
                double tt[3], pf[3], f[3] = {w[0], w[1], w[2]}, pxf[3];
                fbr_mv(rec + FBR_OFF_R, fp, tt);
                for (int i = 0; i < 3; i++) pf[i] = rec[FBR_OFF_P + i] + tt[i];
                fbr_cross(pf, f, pxf);
                for (int i = 0; i < 3; i++) {
                    F[i] = f[i];
                    F[3 + i] = w[3 + i] + pxf[i];
                }
                return;
            }
            double pi[10];
            for (int c = 0; c < 10; c++) pi[c] = mode == 0 ? x[10 * l + c] : (c < hm.cpl ? x[hm.cpl * l + c] : 0.0);
            fbr_link_wrench(rec, pi, F);
        };
        auto consts = [&](int l, double *rR, double *rp, double *ax) {
            for (int i = 0; i < 9; i++) rR[i] = hm.restR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = hm.restp[3 * l + i];
                ax[i] = hm.axis[3 * l + i];
            }
        };
        auto emit = [&](int r, double v) {
            if (r >= hm.fb && hm.fric && mode != 2) {
                const int d = r - hm.fb;
                const double dqv = dq[s * n + d], sg = sign[s * n + d];
                if (mode == 0) {
                    const int f0 = hm.friction_start();
                    double tt = sg * x[f0 + d];
                    if (!hm.grav_only) {
                        tt += x[f0 + n + d] * dqv;
                        const int poff = f0 + 2 * n;
                        tt += x[poff + d];
                        if (hm.stribeck > 0) {
                            const double sgn = (sg > 0) - (sg < 0);
                            tt += x[poff + n + d] * exp(-fabs(vel_sign[s * n + d]) / hm.stribeck) * sgn;
                        }
                    }
                    v += tt;
                } else {
                    for (int c = hm.cpl * hm.L; c < hm.cols; c++) {
                        const FbrCol &cd = hm.coldesc[c];
                        if (cd.joint == d) v += x[c] * fbr_friction_value(cd.pidx, dqv, sg, hm.stribeck);
                    }
                }
            }
            written[r]++;
            ts[r] = v;
        };
        fbr_kinid_lane<FBR_KINID_MAXD, true>(p.nsteps, p.maxlvl, p.steps.data(), p.endflush.data(), hm.floating, hm.gravity, hm.fb, state, basest,
                                             save, load, link, emit, consts);
        for (int r = 0; r < hm.rows; r++)
            if (written[r] != 1) return -2 - r;  // a row written twice or never: a bug of the flush lists
    }
    return 0;
}

// The tree cut into parts (fbr_kinid_build_parts: what the waves of the lane writer's workgroups walk): every part's program run on its own,
// a link contributing its wrench only in the part that owns it -- the parts' torques must add up to the whole robot's.  No friction.
int emul_kinid_parts(const EmulTopo *t, int nparts, long S, const double *q, const double *dq, const double *ddq, const double *bv, const double *ba,
                     const double *rpy, const double *x, double *tau, int *steps_out)
{
    FbrHostModel hm;
    make(t, hm);
    if (hm.maxdepth > FBR_KINID_MAXD) return -1;
    std::vector<FbrKinIdProgram> progs;
    std::vector<std::vector<char>> own;
    std::vector<double> cost(hm.L, 1.0);
    for (int l = 0; l < hm.L; l++) cost[l] += (double)hm.path[l].size();
    fbr_kinid_build_parts(hm, cost, nparts, progs, own);
    const int n = hm.n;
    for (int l = 0; l < hm.L; l++) {  // every link is owned exactly once
        int c = 0;
        for (size_t p = 0; p < own.size(); p++) c += own[p][l];
        if (c != 1) return -3;
    }
    for (long s = 0; s < S; s++) {
        double *ts = tau + (size_t)s * hm.rows;
        for (int r = 0; r < hm.rows; r++) ts[r] = 0.0;
        for (size_t pi = 0; pi < progs.size(); pi++) {
            const FbrKinIdProgram &p = progs[pi];
            if (steps_out) steps_out[pi] = p.nsteps;
            std::vector<double> slots((size_t)std::max(p.nslots, 1) * FBR_LINK_REC, std::nan(""));
            auto state = [&](int d, double &a, double &b, double &c) {
                a = q[s * n + d];
                b = dq[s * n + d];
                c = ddq[s * n + d];
            };
            auto basest = [&](double *v6, double *a6, double *e3) {
                for (int i = 0; i < 6; i++) {
                    v6[i] = bv[6 * s + i];
                    a6[i] = ba[6 * s + i];
                }
                for (int i = 0; i < 3; i++) e3[i] = rpy[3 * s + i];
            };
            auto save = [&](int b, int i, double v) { slots[(size_t)b * FBR_LINK_REC + i] = v; };
            auto load = [&](int b, int i) { return slots[(size_t)b * FBR_LINK_REC + i]; };
            auto link = [&](int l, int, const double *rec, const double (*)[6], const int *, double *F) {
                if (!own[pi][l]) return;
                double pi10[10];
                for (int c = 0; c < 10; c++) pi10[c] = x[10 * l + c];
                fbr_link_wrench(rec, pi10, F);
            };
            auto consts = [&](int l, double *rR, double *rp, double *ax) {
                for (int i = 0; i < 9; i++) rR[i] = hm.restR[9 * l + i];
                for (int i = 0; i < 3; i++) {
                    rp[i] = hm.restp[3 * l + i];
                    ax[i] = hm.axis[3 * l + i];
                }
            };
            auto emit = [&](int r, double v) { ts[r] += v; };
            fbr_kinid_lane<FBR_KINID_MAXD, true>(p.nsteps, p.maxlvl, p.steps.data(), p.endflush.data(), hm.floating, hm.gravity, hm.fb, state, basest, save,
                                                 load, link, emit, consts);
        }
    }
    return (int)progs.size();
}

// mirrors fbr_kinfd_kernel: one "lane" per evaluation e = s (1 + 3 n) + j of the finite-difference sweep, score[e] = sum W_s . Y_e
int emul_kinfd(const EmulTopo *t, long S, double eps, const double *q, const double *dq, const double *ddq, const double *bv, const double *ba,
               const double *rpy, const double *sign, const double *W, double *out)
{
    FbrHostModel hm;
    make(t, hm);
    if (hm.maxdepth > FBR_KINID_MAXD || hm.masked) return -1;
    FbrKinIdProgram p;
    fbr_kinid_build(hm, p);
    const int n = hm.n, nper = 1 + 3 * n;
    std::vector<double> slots((size_t)std::max(p.nslots, 1) * FBR_LINK_REC);
    for (long e = 0; e < S * nper; e++) {
        const long s = e / nper;
        const int j = (int)(e - s * nper);
        const int kind = (j == 0) ? -1 : (j - 1) / n, dj = (j == 0) ? -1 : (j - 1) % n;
        const double *Ws = W + (size_t)s * hm.rows * hm.cols;
        double score = 0.0;
        auto state = [&](int d, double &a, double &b, double &c) {
            a = q[s * n + d] + ((kind == 0 && d == dj) ? eps : 0.0);
            b = dq[s * n + d] + ((kind == 1 && d == dj) ? eps : 0.0);
            c = ddq[s * n + d] + ((kind == 2 && d == dj) ? eps : 0.0);
        };
        auto basest = [&](double *v6, double *a6, double *e3) {
            for (int i = 0; i < 6; i++) {
                v6[i] = bv[6 * s + i];
                a6[i] = ba[6 * s + i];
            }
            for (int i = 0; i < 3; i++) e3[i] = rpy[3 * s + i];
        };
        auto save = [&](int b, int i, double v) { slots[(size_t)b * FBR_LINK_REC + i] = v; };
        auto load = [&](int b, int i) { return slots[(size_t)b * FBR_LINK_REC + i]; };
        auto consts = [&](int l, double *rR, double *rp, double *ax) {
            for (int i = 0; i < 9; i++) rR[i] = hm.restR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = hm.restp[3 * l + i];
                ax[i] = hm.axis[3 * l + i];
            }
        };
        auto Wr = [&](int r, int c) { return Ws[(size_t)r * hm.cols + c]; };
        auto link = [&](int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, double *) {
            score += fbr_kinfd_link_score<FBR_KINID_MAXD>(l, depth, rec, Sst, lvd, hm.cpl, hm.fb, Wr);
        };
        auto emit = [&](int, double) {};
        fbr_kinid_lane<FBR_KINID_MAXD, false>(p.nsteps, p.maxlvl, p.steps.data(), p.endflush.data(), hm.floating, hm.gravity, hm.fb, state, basest,
                                              save, load, link, emit, consts);
        for (int c = hm.cpl * hm.L; c < hm.cols; c++) {
            const FbrCol &cd = hm.coldesc[c];
            const int jj = cd.joint;
            const double dqv = dq[s * n + jj] + ((kind == 1 && jj == dj) ? eps : 0.0);
            score += Ws[(size_t)(hm.fb + jj) * hm.cols + c] * fbr_friction_value(cd.pidx, dqv, sign ? sign[s * n + jj] : 0.0, hm.stribeck);
        }
        out[e] = score;
    }
    return 0;
}

// 0: the product's choice (fbr_gram_build_best), 1: force the one-workgroup-per-CU shape, 2: force the two-per-CU shape
static int g_shape = 0;
void emul_set_gram_shape(int shape) { g_shape = shape; }
// 1: few rhs columns get no tiles, their products are accumulated where the image is packed (fbr_gram_rhs_moments: what
// fbr_gram_accumulate runs for k <= 2); 0: dense rhs tiles whatever k
static int g_moments = 0;
void emul_set_rhs_moments(int on) { g_moments = on; }
static void build_program(FbrGramProgram &gp, const FbrHostModel &hm, int k)
{
    fbr_gram_build_best(gp, hm, k, g_shape, !(g_moments && fbr_gram_rhs_moments(hm, k)));
}

int emul_program_info(const EmulTopo *t, int k, int *NT, int *npairs, long *mfma, int *T, int *image_doubles,
                      int *items_total, long *mfma_uniform)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    build_program(gp, hm, k);
    *NT = gp.NT; *npairs = (int)gp.pairs.size(); *mfma = gp.mfma_per_sample; *T = gp.T;
    *image_doubles = gp.part_image_max;
    long dma = 0;
    for (int p = 0; p < gp.T; p++) dma += gp.part_image[p];
    *items_total = (int)dma;  // doubles copied per sample summed over the parts
    *mfma_uniform = gp.mfma_uniform;
    return 0;
}

// row statistics of the executed k-steps: out[0] = executed rows (4 per k-step, even-sample count), out[1] = rows that are real in
// both tiles, out[2] = base-section rows executed, out[3] = base-section rows real in both
int emul_row_stats(const EmulTopo *t, int k, long *out)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    build_program(gp, hm, k);
    out[0] = out[1] = out[2] = out[3] = 0;
    for (size_t s = 0; s < gp.slots.size(); s++) {
        const int pi = gp.slots[s].pair;
        if (pi < 0) continue;
        const FbrPair &p = gp.pairs[pi];
        const FbrTile &a = gp.tiles[p.I], &b = gp.tiles[p.J];
        for (int ks = gp.slots[s].kb; ks < p.nkend(); ks++)
            for (int r = 4 * ks; r < 4 * ks + 4; r++) {
                bool real = false;
                if (r < p.common) {
                    if (p.mode == 0) real = a.posnz[r] && b.posnz[r];
                    else if (p.mode == 1) real = a.posnz[r] && b.rownz[a.rowid[r]];
                    else real = a.rownz[r] && b.rownz[r];
                }
                out[0]++;
                out[1] += real;
                if (r < hm.fbp && p.mode != 2) { out[2]++; out[3] += real; }
            }
    }
    return 0;
}

// histogram of n = MFMAs per k-step instance (row segment, k-step) of one even sample: out[n] for n = 1..8, out[0] = row segments
int emul_nstats(const EmulTopo *t, int k, long *out)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    build_program(gp, hm, k);
    for (int i = 0; i < 9; i++) out[i] = 0;
    const int SEGW = gp.cfg.segw, NSEG = gp.cfg.nseg, NPW = gp.cfg.npw();
    for (int tw = 0; tw < gp.T * FBR_WPB; tw++)
        for (int sg = 0; sg < NSEG; sg++) {
            int kb = -1, kmax = 0, ends[16], cnt = 0;
            for (int j = 0; j < SEGW; j++) {
                const FbrSlot &sl = gp.slots[(size_t)tw * NPW + sg * SEGW + j];
                if (sl.pair < 0) continue;
                kb = sl.kb;
                ends[cnt++] = gp.pairs[sl.pair].nkend();
                kmax = std::max(kmax, ends[cnt - 1]);
            }
            if (!cnt) continue;
            out[0]++;
            for (int ks = kb; ks < kmax; ks++) {
                int n = 0;
                for (int j = 0; j < cnt; j++) n += ends[j] > ks;
                out[std::min(n, 8)]++;
            }
        }
    return 0;
}

// per part: [load of the most loaded wave, MFMAs, image doubles]; out holds 3 * T ints
int emul_part_stats(const EmulTopo *t, int k, int *out, int cap)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    build_program(gp, hm, k);
    for (int p = 0; p < gp.T && p < cap; p++) {
        out[3 * p] = gp.part_load[p];
        out[3 * p + 1] = gp.part_mfma[p];
        out[3 * p + 2] = gp.part_image[p];
    }
    return gp.T;
}

// mirrors fbr_pack_kernel + fbr_gram_kernel + fbr_gram_reduce_kernel: the packed image of each sample is
// produced once; every part copies its DMA pieces into a part-local image, each (wave, slot) runs its k-steps
// with the 16x16x4 MFMA lane mapping on part-local offsets, results are scattered into G.
// The Gram pass over sample-contiguous images (csrc/fbr_gram64.h: fbr_kinimg_kernel / fbr_gram64_kernel), from the SAME host tables:
// producer = per (part, link, parameter) one destination word, level stride 1024 doubles, the column swizzle inside a 32-sample run;
// consumer = stage (level, half) staged through the pieces table, operand reads with the kernel's indexing, the pairs of a wave's
// segments below their common depth; rhs moments as per-lane running sums.  Returns < 0 when the model is outside that pass.
// stats (10 longs, optional): tile rows, MFMAs per block, levels, widest stage, sum over levels of the busiest wave's active pairs,
// sum over levels of ceil(active pairs / waves), parts, pairs, force tiles, stages.
static int g_force_tiles = 1, g_wide16 = 0;
void emul_set_force_tiles(int on) { g_force_tiles = on; }
void emul_set_wide16(int on) { g_wide16 = on; }
int emul_gram64(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv, const double *ba,
                const double *rpy, const double *sign, const double *rhs, int k, const double *wts, double *G, long *stats)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    fbr_gram_build_best(gp, hm, k, g_shape, !fbr_gram_rhs_moments(hm, k));
    FbrGram64 g;
    FbrGram64Producer pr;
    if (k > 1 || !fbr_gram64_build(hm, gp, g, g_force_tiles != 0, g_wide16 != 0) || !fbr_gram64_build_producer(hm, gp, g, pr)) return -1;
    const int W = g.wpb, npw = g.npw, NTT = g.NT + g.NF;
    if (stats) {
        long npairs = 0;
        for (size_t i = 0; i < g.wmeta.size(); i += 3) npairs += g.wmeta[i] >= 0;
        stats[0] = g.ntr; stats[1] = g.mfma_per_block; stats[2] = g.nlev; stats[3] = g.maxact;
        stats[4] = g.busiest; stats[5] = g.balanced; stats[6] = pr.nparts; stats[7] = npairs; stats[8] = g.NF; stats[9] = g.nstage;
    }
    if (S <= 0) return 0;
    const int REC = hm.rec_size(), P = hm.cols, Pa = P + k;
    const long nblk = (S + 63) / 64;
    std::vector<double> rec(REC), img((size_t)g.blk_doubles), buf((size_t)g.maxact * 512);
    std::vector<double> acc((size_t)W * npw * 256, 0.0), mom((size_t)(P + 1) * 64, 0.0);
    std::vector<char> written((size_t)g.blk_doubles);
    for (long blk = 0; blk < nblk; blk++) {
        std::fill(img.begin(), img.end(), 0.0);  // (the buffer is zeroed once at allocation: what is never written stays zero)
        std::fill(written.begin(), written.end(), 0);
        const int valid = (int)std::min(64L, S - blk * 64);
        for (int lane = 0; lane < 64; lane++) {
            const bool live = lane < valid;
            const long s = blk * 64 + std::min(lane, valid - 1);
            kin_sample(hm, q + s * hm.n, dq + s * hm.n, ddq + s * hm.n, hm.floating ? bv + 6 * s : nullptr, hm.floating ? ba + 6 * s : nullptr,
                       hm.floating ? rpy + 3 * s : nullptr, rec.data());
            const double *ws = wts ? wts + (size_t)s * hm.rows : nullptr;
            auto tv = [&](int r) { const double wv = ws ? ws[r] : 1.0; return wv * wv * rhs[(size_t)s * hm.rows + r]; };
            for (int pq = 0; pq < pr.nparts; pq++)
                for (int l = 0; l < hm.L; l++)
                    for (int pp = 0; pp < 10; pp++) {
                        const long long *w14 = &pr.rel[((size_t)pq * hm.L + l) * FBR_G64_WORDS];
                        const long long d0 = w14[pp], dF = pp < 4 ? w14[10 + pp] : 0;
                        if (!d0) {
                            if (dF) return -2;
                            continue;
                        }
                        const int c = pr.lcol[((size_t)pq * hm.L + l) * 10 + pp];
                        if (c < 0 || hm.coldesc[c].link != l || hm.coldesc[c].pidx != pp) return -2;
                        double w6[6];
                        fbr_unit_wrench(&rec[FBR_LINK_REC * l], pp, w6);
                        double mc = 0.0;
                        auto put = [&](long long dw, int lv, double v, int r) {
                            if (!dw) return false;
                            const long base = (long)((dw & ~(1LL << 62) & ~0xffLL) / 8);
                            const int x = (int)(dw & 0xff);
                            const long at = base + (long)lv * 1024 + (lane >> 5) * 512 + ((lane & 31) ^ x);
                            if (at < 0 || at >= g.blk_doubles) return false;
                            img[at] = live ? v * (ws ? ws[r] : 1.0) : 0.0;
                            written[at]++;
                            if (k) mc += v * tv(r);
                            return true;
                        };
                        for (int i = (pp >= 4 ? 3 : 0); i < hm.fb; i++)
                            if (!put(i < g.flev ? dF : d0, i, w6[i], i)) return -3;
                        if (pp >= 4)
                            for (int i = 0; i < std::min(3, hm.fb); i++)
                                if (w6[i] != 0.0) return -4;  // (the force rows of an inertia column are structural zeros)
                        int j = 0;
                        for (int d : hm.path[l]) {
                            if (!put(d0, hm.fb + j, fbr_dot6(&rec[FBR_LINK_REC * hm.L + 6 * d], w6), hm.fb + d)) return -3;
                            j++;
                        }
                        if (k && live) mom[(size_t)c * 64 + lane] += mc;
                    }
            // friction columns: written by the part that owns the joint's link, on the row of that joint (the last level of the link's path)
            for (int pq = 0; pq < pr.nparts; pq++)
                for (int l = 0; l < hm.L; l++)
                    for (int pf = 0; pf < FBR_G64_FRIC; pf++) {
                        const long long dw = pr.rel[((size_t)pq * hm.L + l) * FBR_G64_WORDS + 14 + pf];
                        if (!dw) continue;
                        const int c = pr.lcol[(size_t)pr.nparts * 10 * hm.L + ((size_t)pq * hm.L + l) * FBR_G64_FRIC + pf];
                        if (c < hm.ninert || c >= hm.cols || hm.dof[l] != hm.coldesc[c].joint) return -12;
                        const int d = hm.dof[l], lv = hm.fb + (int)hm.path[l].size() - 1, r = hm.fb + d;
                        const double fv = fbr_friction_value(hm.coldesc[c].pidx, dq[s * hm.n + d], sign ? sign[s * hm.n + d] : 0.0, hm.stribeck);
                        const long base = (long)((dw & ~(1LL << 62) & ~0xffLL) / 8);
                        const long at = base + (long)lv * 1024 + (lane >> 5) * 512 + ((lane & 31) ^ (int)(dw & 0xff));
                        if (at < 0 || at >= g.blk_doubles) return -3;
                        img[at] = live ? fv * (ws ? ws[r] : 1.0) : 0.0;
                        written[at]++;
                        if (k && live) mom[(size_t)c * 64 + lane] += fv * tv(r);
                    }
            if (k && live) {
                double tt = 0.0;
                for (int r = 0; r < hm.rows; r++) {
                    const double v = rhs[(size_t)s * hm.rows + r] * (ws ? ws[r] : 1.0);
                    tt += v * v;
                }
                mom[(size_t)P * 64 + lane] += tt;
            }
        }
        for (char wv : written)
            if (wv > 1) return -5;  // two writers of one image position
        // consumer: steps (half, stage), a stage = consecutive levels staged together
        for (int half = 0; half < 2; half++)
            for (int lv = 0; lv < g.nlev; lv++) {
                int sgi = 0;
                while (g.stage_lev[sgi + 1] <= lv) sgi++;
                if (lv == g.stage_lev[sgi]) {
                    std::fill(buf.begin(), buf.end(), 1e300);  // (what the stage does not bring must not be read)
                    std::vector<char> hit(buf.size(), 0);
                    for (int i = g.lev_begin[g.stage_lev[sgi]]; i < g.lev_begin[g.stage_lev[sgi + 1]]; i++)
                        for (int e = 0; e < 128; e++) {
                            if (g.pieces[2 * i + 1] + e >= (int)buf.size() || hit[g.pieces[2 * i + 1] + e]++) return -11;  // outside the buffer / two pieces on one place
                            buf[g.pieces[2 * i + 1] + e] = img[g.pieces[2 * i] + half * 512 + e];
                        }
                }
                const int *sl = &g.slab[(size_t)lv * NTT];
                for (int w = 0; w < W; w++)
                    for (int qs = 0; qs < npw; qs++) {
                        const int *mm = &g.wmeta[((size_t)w * npw + qs) * 3];
                        const int tI = mm[0], tJ = mm[1], lo = mm[2] & 0xff, hi = mm[2] >> 8;
                        if (tI < 0 || lv < lo || lv >= hi) continue;
                        if (sl[tI] < 0 || sl[tJ] < 0) return -6;
                        double *a4 = &acc[((size_t)w * npw + qs) * 256];
                        for (int ks = 0; ks < 8; ks++) {
                            double A[16][4], B[4][16];
                            for (int lane = 0; lane < 64; lane++) {
                                const int li = lane & 15, kk = lane >> 4, sx = FBR_G64_SWZ(li);
                                A[li][kk] = buf[(size_t)sl[tI] * 512 + li * 32 + ((4 * ks + kk) ^ sx)];
                                B[kk][li] = buf[(size_t)sl[tJ] * 512 + li * 32 + ((4 * ks + kk) ^ sx)];
                            }
                            for (int lane = 0; lane < 64; lane++)
                                for (int reg = 0; reg < 4; reg++) {
                                    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
                                    double sum = 0;
                                    for (int kk = 0; kk < 4; kk++) sum += A[row][kk] * B[kk][col];
                                    a4[reg * 64 + lane] += sum;
                                }
                        }
                    }
            }
    }
    for (int kind = 0; kind < 2; kind++)  // the two reductions: main blocks, force blocks
    for (int w = 0; w < W; w++)
        for (int sl = 0; sl < npw; sl++) {
            const size_t sr = ((size_t)(w & 7) * (W / 8) + (size_t)(w >> 3)) * npw + sl;  // (the reduction's slot order)
            const int tI = g.slot_tiles[2 * ((size_t)kind * W * npw + sr)], tJ = g.slot_tiles[2 * ((size_t)kind * W * npw + sr) + 1];
            if (tI < 0) continue;
            const double *a4 = &acc[((size_t)w * npw + sl) * 256];
            for (int lane = 0; lane < 64; lane++)
                for (int reg = 0; reg < 4; reg++) {
                    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
                    const int ci = g.tilecol[(size_t)tI * FBR_TILE + row], cj = g.tilecol[(size_t)tJ * FBR_TILE + col];
                    if (ci < 0 || cj < 0 || ci >= P || cj >= P) continue;
                    G[(size_t)ci * Pa + cj] += a4[reg * 64 + lane];
                    if (tI != tJ) G[(size_t)cj * Pa + ci] += a4[reg * 64 + lane];
                }
        }
    if (k)
        for (int c = 0; c <= P; c++) {
            double sum = 0.0;
            for (int lane = 0; lane < 64; lane++) sum += mom[(size_t)c * 64 + lane];
            if (c == P) {
                G[(size_t)P * Pa + P] += sum;
            } else {
                G[(size_t)c * Pa + P] += sum;
                G[(size_t)P * Pa + c] += sum;
            }
        }
    return 0;
}

int emul_gram(const EmulTopo *t, long S, const double *q, const double *dq, const double *ddq, const double *bv,
              const double *ba, const double *rpy, const double *sign, const double *rhs, int k, const double *wts,
              double *G)
{
    FbrHostModel hm;
    make(t, hm);
    FbrGramProgram gp;
    build_program(gp, hm, k);
    const int REC = hm.rec_size();
    const int Pa = gp.Pa;
    const int FBR_NPW = gp.cfg.npw();
    std::vector<double> rec(REC), loc(gp.part_image_max, 0.0);
    // pack kernel: the images of all samples first (an odd sample writes its base rows into its partner's image); every image
    // starts as garbage where the kernel never writes and never reads ... except the structural zeros, which are zero
    std::vector<double> imgs((size_t)S * gp.image_doubles, 0.0);
    // rhs moments (mirror of the pack kernel's `mom`): per work item its column's products with the rhs columns, and rhs^T rhs
    const bool moments = !gp.rhs_tiles;
    std::vector<double> mom(gp.items.size() * 2, 0.0);
    double mtt[3] = {0, 0, 0};
    for (long s = 0; s < S; s++) {
        kin_sample(hm, q + s * hm.n, dq + s * hm.n, ddq + s * hm.n, hm.floating ? bv + 6 * s : nullptr,
                   hm.floating ? ba + 6 * s : nullptr, hm.floating ? rpy + 3 * s : nullptr, rec.data());
        const double *ws = wts ? wts + (size_t)s * hm.rows : nullptr;
        double *img = &imgs[(size_t)s * gp.image_doubles];
        const bool odd = (s & 1) != 0, partner = !odd && s + 1 < S;
        auto bpos = [&](int r) { return odd ? (r < 2 ? 6 + r : 2 + r) : r; };
        auto bimg = [&](int r) { return (odd && r < 2) ? img - gp.image_doubles : img; };
        // poison what a stale image could hold: the mirror must overwrite, clear or never read it, like the kernel
        for (const FbrItem &it : gp.items) {
            if (it.kind == 0 && hm.fb && !odd) img[it.off + 6 * FBR_TILE] = img[it.off + 7 * FBR_TILE] = 1e300;
            if (it.kind == 0 && hm.fb && odd)
                for (int r = 0; r < 4; r++) img[it.off + r * FBR_TILE] = 1e300;  // never read: k-step 0 skipped
            if (it.kind == 2 && hm.fb)
                for (int r = 0; r < 8; r++) img[it.off + r * FBR_TILE] = 1e300;
        }
        auto rw = [&](int r, int i) { return rhs[((size_t)s * hm.rows + r) * k + i] * (ws ? ws[r] : 1.0); };
        if (moments)
            for (int r = 0; r < hm.rows; r++) {
                const double a = rw(r, 0), b = k > 1 ? rw(r, 1) : 0.0;
                mtt[0] += a * a; mtt[1] += a * b; mtt[2] += b * b;
            }
        for (size_t ii = 0; ii < gp.items.size(); ii++) {
            const FbrItem &it = gp.items[ii];
            auto note = [&](double v, int r) {
                if (moments)
                    for (int i = 0; i < k; i++) mom[2 * ii + i] += v * rw(r, i);
            };
            if (it.kind == 0) {
                double w6[6];
                fbr_unit_wrench(&rec[FBR_LINK_REC * it.a], it.b, w6);
                for (int r = 0; r < hm.fb; r++) {
                    bimg(r)[it.off + bpos(r) * FBR_TILE] = w6[r] * (ws ? ws[r] : 1.0);
                    note(w6[r] * (ws ? ws[r] : 1.0), r);
                }
                if (hm.fb && !odd && !partner) img[it.off + 6 * FBR_TILE] = img[it.off + 7 * FBR_TILE] = 0.0;
                int j = 0;
                for (int d : hm.path[it.a]) {
                    img[it.off + hm.ppos[it.a][j] * FBR_TILE] = fbr_dot6(&rec[FBR_LINK_REC * hm.L + 6 * d], w6) * (ws ? ws[hm.fb + d] : 1.0);
                    note(img[it.off + hm.ppos[it.a][j] * FBR_TILE], hm.fb + d);
                    j++;
                }
            } else if (it.kind == 1) {
                int r = hm.fb + it.a;
                img[it.off] =
                    fbr_friction_value(it.b, dq[s * hm.n + it.a], sign ? sign[s * hm.n + it.a] : 0.0, hm.stribeck) *
                    (ws ? ws[r] : 1.0);
                note(img[it.off], r);
            } else {
                for (int r = 0; r < hm.rows; r++) {
                    const double v = rhs[((size_t)s * hm.rows + r) * k + it.a] * (ws ? ws[r] : 1.0);
                    if (r < hm.fb)
                        bimg(r)[it.off + bpos(r) * FBR_TILE] = v;
                    else
                        img[it.off + (hm.fbp + r - hm.fb) * FBR_TILE] = v;
                }
                if (hm.fb && odd)
                    for (int r = 0; r < 4; r++) img[it.off + r * FBR_TILE] = 0.0;
                if (hm.fb && !odd && !partner) img[it.off + 6 * FBR_TILE] = img[it.off + 7 * FBR_TILE] = 0.0;
            }
        }
    }
    std::vector<double> acc((size_t)gp.T * FBR_WPB * FBR_NPW * 256, 0.0);
    for (long s = 0; s < S; s++) {
        const double *img = &imgs[(size_t)s * gp.image_doubles];
        const int kskip = (s & 1) ? gp.base_ks : 0;
        for (int part = 0; part < gp.T; part++) {
            std::fill(loc.begin(), loc.end(), 0.0);
            for (const FbrPiece &pc : gp.pieces[part]) {
                const int nd = pc.half ? 64 : 128;
                for (int i = 0; i < nd; i++) loc[pc.loff + i] = img[pc.goff + i];
            }
            // part-image row -> regressor row
            std::vector<int> ridl(gp.part_image_max / FBR_TILE, 0);
            for (int ti : gp.part_tiles[part])
                for (size_t j = 0; j < gp.tiles[ti].rowid.size(); j++) ridl[gp.part_tile_off[part][ti] / FBR_TILE + j] = gp.tiles[ti].rowid[j];
            for (int w = 0; w < FBR_WPB; w++)
                for (int sl = 0; sl < FBR_NPW; sl++) {
                    int pi = gp.slots[((size_t)part * FBR_WPB + w) * FBR_NPW + sl].pair;
                    if (pi < 0) continue;
                    const FbrPair &p = gp.pairs[pi];
                    const int oA = gp.part_tile_off[part][p.I], oB = gp.part_tile_off[part][p.J];
                    if (oA < 0 || oB < 0 || (oA % 64) || (oB % 64)) return -7;
                    double *a4 = &acc[(((size_t)part * FBR_WPB + w) * FBR_NPW + sl) * 256];
                    const int kb = gp.slots[((size_t)part * FBR_WPB + w) * FBR_NPW + sl].kb;
                    if (kb > p.kbegin()) return -8;
                    for (int ks = (gp.tiles[p.I].type == 0) ? std::max(kb, kskip) : kb; ks < p.nkend(); ks++) {  // superset of the pair's k-step mask, like the kernel
                        double A[16][4], B[4][16];
                        for (int lane = 0; lane < 64; lane++) {
                            int i = lane & 15, kk = lane >> 4;
                            int pos = 4 * ks + kk;
                            double a = loc[oA + 64 * ks + lane];  // no masking: see FbrHostModel::ppos
                            double b = (p.mode == 1) ? loc[oB + ridl[(oA >> 4) + pos] * FBR_TILE + i] : loc[oB + 64 * ks + lane];
                            A[i][kk] = a;
                            B[kk][i] = b;
                        }
                        for (int lane = 0; lane < 64; lane++)
                            for (int reg = 0; reg < 4; reg++) {
                                int row = (lane >> 4) + 4 * reg, col = lane & 15;
                                double sum = 0;
                                for (int kk = 0; kk < 4; kk++) sum += A[row][kk] * B[kk][col];
                                a4[reg * 64 + lane] += sum;
                            }
                    }
                }
        }
    }
    // reduce / scatter
    for (int part = 0; part < gp.T; part++)
        for (int w = 0; w < FBR_WPB; w++)
            for (int sl = 0; sl < FBR_NPW; sl++) {
                int pi = gp.slots[((size_t)part * FBR_WPB + w) * FBR_NPW + sl].pair;
                if (pi < 0) continue;
                const FbrPair &p = gp.pairs[pi];
                const double *a4 = &acc[(((size_t)part * FBR_WPB + w) * FBR_NPW + sl) * 256];
                for (int lane = 0; lane < 64; lane++)
                    for (int reg = 0; reg < 4; reg++) {
                        int row = (lane >> 4) + 4 * reg, col = lane & 15;
                        int ci = gp.tiles[p.I].col[row], cj = gp.tiles[p.J].col[col];
                        if (ci < 0 || cj < 0) continue;
                        double v = a4[reg * 64 + lane];
                        G[(size_t)ci * Pa + cj] += v;
                        if (p.I != p.J) G[(size_t)cj * Pa + ci] += v;
                    }
            }
    if (moments) {  // mirror of fbr_gram_mom_reduce_kernel
        const int P = hm.cols;
        // on the device one workgroup per item adds into G[c][P] without atomics: a column owned by two items is a lost update there
        // (sequential code would not notice)
        std::vector<int> owners(P + k, 0);
        for (size_t ii = 0; ii < gp.items.size(); ii++)
            if (gp.items[ii].col >= 0 && gp.items[ii].col < P + k && ++owners[gp.items[ii].col] > 1) return -10;
        for (size_t ii = 0; ii < gp.items.size(); ii++) {
            const int c = gp.items[ii].col;
            if (c < 0) return -9;
            for (int i = 0; i < k; i++) {
                G[(size_t)c * Pa + P + i] += mom[2 * ii + i];
                G[(size_t)(P + i) * Pa + c] += mom[2 * ii + i];
            }
        }
        G[(size_t)P * Pa + P] += mtt[0];
        if (k > 1) {
            G[(size_t)P * Pa + P + 1] += mtt[1];
            G[(size_t)(P + 1) * Pa + P] += mtt[1];
            G[(size_t)(P + 1) * Pa + P + 1] += mtt[2];
        }
    }
    return 0;
}
}
