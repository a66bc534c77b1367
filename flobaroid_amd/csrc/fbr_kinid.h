// fbr_kinid.h -- kinematics + inverse dynamics / prediction FUSED, one lane per sample, the link records never leave the registers.
//
// Replaces the pair fbr_kin_kernel (one lane per sample, 9.5 KB of records written per WALK-MAN sample in partially filled lines) +
// fbr_id_kernel (one wave per sample, the records read back) on the calls that need torques only: fbr_predict (A9, identifier.py:135-141),
// fbr_inverse_dynamics_batch (A3, identification/model.py:239-331, and the simulated base wrench of every floating-base pass,
// model.py:398-413).  The round-5 review priced the pair at 18.9 KB per sample staged through HBM for 1.1 KB of algorithmic I/O.
//
// Formulation (base-frame composite): with F_l = W_l pi_l the wrench of link l in frame A and S_d the motion vector of
// joint d in A,   base rows = sum_l F_l,   tau_d = S_d . sum_{l below d} F_l.   The links are walked parents first (FbrHostModel::order,
// a depth-first order); every lane carries
//   * the record of the link before (the parent of a chain's next link) in registers,
//   * a STACK of the ancestor joints' motion vectors and torque accumulators, indexed by the joint's depth on the path (<= MAXD levels):
//     F_l is added to every ancestor's accumulator when it is formed; a level is written out when another joint takes it,
//   * the records of branch points (links with a child that is not walked right after them) in a per-wave scratch, lanes interleaved
//     (512-byte coalesced lines; 2 records per WALK-MAN sample).
// All lanes of a wave walk the same link at the same time, so every index into the stacks is wave-uniform: the stacks live in
// registers behind scalar branches, never in scratch memory.  The states of a wave's 64 samples are staged through the LDS with
// coalesced loads (a lane reading q[s][d] straight from the row-major arrays touches 64 lines per instruction: 16 x the bytes).
//
// The program (steps, flush lists, slots) is built on the host by fbr_kinid_build -- HIP-free, so that tests/emul runs the same
// program and the same lane body (fbr_kinid_lane) on the CPU (checked there against the CPU restatement of the reference).
#pragma once
#include <vector>

#include "fbr_math.h"
#include "fbr_program.h"

#define FBR_KINID_STEP 8  // ints per step: link, psrc, psave, jtype, dof, level, depth, flushdof
#define FBR_KINID_MAXD 24 // deepest joint path the register-stack instances cover (deeper trees keep the two-kernel path)

#define FBR_KINWRITE_PARTS 4  // waves of a lane-WRITER workgroup = parts the tree is cut into (fbr_kinid_build_parts)

#if defined(__HIPCC__)
// tables every lane of a wave reads at the same index: through the constant address space they are scalar loads
typedef const __attribute__((address_space(4))) long *fbr_clong_ptr;
typedef const __attribute__((address_space(4))) int *fbr_cint_ptr;
typedef const __attribute__((address_space(4))) double *fbr_cdouble_ptr;
#endif

struct FbrKinIdProgram {
    int nsteps = 0, maxlvl = 0, nslots = 0;
    std::vector<int> steps;     // [nsteps][FBR_KINID_STEP]
    std::vector<int> endflush;  // [maxlvl] dof left on level v after the last link (-1: none)
};

// psrc: -1 base link; 0 the parent is the link of the step before (its record is in registers); 1 + b: the parent's record is in slot b.
// psave: slot this link's record is saved to (a later link that is not the next step has it as parent), or -1.
// level: 0-based depth of the link's own joint on its path (-1: fixed joint / base); depth: joints on the link's path (own one included).
// flushdof: the dof that held `level` until now and is complete (every link below it has been walked), or -1.
// keep (optional, [L]): the program walks these links only -- a set closed under parents (fbr_kinid_build_parts).
static inline void fbr_kinid_build(const FbrHostModel &hm, FbrKinIdProgram &p, const std::vector<char> *keep = nullptr)
{
    std::vector<int> ord;
    for (int l : hm.order)
        if (!keep || (*keep)[l]) ord.push_back(l);
    const int L = (int)ord.size();
    p.nsteps = L;
    p.steps.assign((size_t)std::max(L, 1) * FBR_KINID_STEP, -1);
    std::vector<int> lastchild(hm.L, -1);  // the last step whose parent the link is
    for (int k = 0; k < L; k++) {
        const int par = hm.parent[ord[k]];
        if (par >= 0) lastchild[par] = std::max(lastchild[par], k);
    }
    // slots: a link holds one from its own step to the step of its last child whenever that child is not the very next step
    std::vector<int> slot(hm.L, -1), holder;  // holder[b]: link that holds slot b, or -1
    int maxlvl = 0;
    std::vector<int> lvldof(std::max(hm.maxdepth, 1), -1);
    for (int k = 0; k < L; k++) {
        const int l = ord[k], par = hm.parent[l];
        int *st = &p.steps[(size_t)k * FBR_KINID_STEP];
        for (int &h : holder)
            if (h >= 0 && lastchild[h] < k) h = -1;  // (its last child has been walked)
        st[0] = l;
        st[1] = par < 0 ? -1 : (k > 0 && ord[k - 1] == par ? 0 : 1 + slot[par]);
        if (par >= 0 && st[1] != 0 && (slot[par] < 0 || holder[slot[par]] != par)) throw std::runtime_error("fbr_kinid_build: parent record not held");
        st[2] = -1;
        if (lastchild[l] > k + 1) {  // some child comes later than the next step
            int b = 0;
            while (b < (int)holder.size() && holder[b] >= 0) b++;
            if (b == (int)holder.size()) holder.push_back(-1);
            holder[b] = l;
            slot[l] = st[2] = b;
        }
        st[3] = hm.jtype[l];
        st[4] = hm.dof[l];
        const int depth = (int)hm.path[l].size();
        st[5] = (par >= 0 && hm.dof[l] >= 0) ? depth - 1 : -1;
        st[6] = depth;
        st[7] = -1;
        if (st[5] >= 0) {
            // the joint that held this level is complete: every link below it has been walked (a link adds to level v only if its
            // path has more than v joints, i.e. passes through the joint that holds level v at that moment)
            st[7] = lvldof[st[5]];
            lvldof[st[5]] = hm.dof[l];
            maxlvl = std::max(maxlvl, st[5] + 1);
        }
    }
    const int nslots = (int)holder.size();
    p.nslots = nslots;
    p.maxlvl = maxlvl;
    p.endflush.assign(std::max(maxlvl, 1), -1);
    for (int v = 0; v < maxlvl; v++) p.endflush[v] = lvldof[v];
}

// The tree cut into `nparts` programs for waves that share one block of samples (the lane writer: every wave walks ITS links and the
// ancestors they need, and writes the columns of its own links only).  The depth-first order is cut into contiguous ranges of about
// equal cost (cost[l]: what link l costs its owner); own[p][l] = 1: part p owns link l.  Ancestors are walked redundantly (a trunk of a
// few links on WALK-MAN); nothing is exchanged between the waves.
static inline void fbr_kinid_build_parts(const FbrHostModel &hm, const std::vector<double> &cost, int nparts, std::vector<FbrKinIdProgram> &progs,
                                         std::vector<std::vector<char>> &own)
{
    nparts = std::max(1, std::min(nparts, hm.L));
    double total = 0.0;
    for (int l = 0; l < hm.L; l++) total += cost[l];
    progs.assign(nparts, FbrKinIdProgram());
    own.assign(nparts, std::vector<char>(hm.L, 0));
    double acc = 0.0;
    int part = 0;
    for (int k = 0; k < hm.L; k++) {
        const int l = hm.order[k];
        // (a part is closed when its share is reached; the last part takes what is left; every part gets at least one link)
        if (part + 1 < nparts && acc >= total * (part + 1) / nparts && hm.L - k >= nparts - part - 1) part++;
        own[part][l] = 1;
        acc += cost[l];
    }
    for (int p = 0; p < nparts; p++) {
        std::vector<char> keep = own[p];
        for (int l = 0; l < hm.L; l++)
            if (own[p][l])
                for (int a = hm.parent[l]; a >= 0 && !keep[a]; a = hm.parent[a]) keep[a] = 1;
        fbr_kinid_build(hm, progs[p], &keep);
    }
}

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define FBR_UNI(x) __builtin_amdgcn_readfirstlane(x)  // the program is the same for every lane: keep it in scalar registers
#else
#define FBR_UNI(x) (x)
#endif

// ------------------------------------------------------------------------------------------------
// One lane = one sample (or one perturbed evaluation of a sample).  StateFn(d, q, dq, ddq): joint state of dof d; BaseFn(bv6, ba6, rpy3);
// save(b, i, v) / load(b, i): slot b of the branch-point records; ConstFn(l, restR, restp, axis);
// LinkFn(l, depth, rec, Sst, lvd, F): called once per link with its record, the motion vectors Sst[0 .. depth) and dofs lvd[0 .. depth) of
// the joints on its path; with ACC it returns the link's wrench F (frame A), which the lane adds to the base rows and to every ancestor
// joint's torque; Emit(row, value): regressor row `row` of this sample (ACC only; joint rows without friction: the caller adds it).
// `steps` / `endflush` of FbrKinIdProgram; every table access is wave-uniform.
// ------------------------------------------------------------------------------------------------
template <int MAXD, bool ACC, class StateFn, class BaseFn, class SlotSave, class SlotLoad, class LinkFn, class EmitFn, class ConstFn>
FBR_HD void fbr_kinid_lane(int nsteps, int maxlvl, const int *steps, const int *endflush, int floating, const double *g, int fb,
                           StateFn state, BaseFn basest, SlotSave save, SlotLoad load, LinkFn link, EmitFn emit, ConstFn consts)
{
    double P[FBR_LINK_REC];
    double Sst[MAXD][6], tac[MAXD];
    int lvd[MAXD];
    double T[6] = {0, 0, 0, 0, 0, 0};
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < MAXD; j++) {
        tac[j] = 0.0;
        lvd[j] = 0;
        for (int i = 0; i < 6; i++) Sst[j][i] = 0.0;
    }
    for (int k = 0; k < nsteps; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
        const fbr_cint_ptr st = (fbr_cint_ptr)(unsigned long)(steps + k * FBR_KINID_STEP);  // (the step program: scalar loads)
#else
        const int *st = steps + k * FBR_KINID_STEP;
#endif
        const int l = FBR_UNI(st[0]), psrc = FBR_UNI(st[1]), psave = FBR_UNI(st[2]), jt = FBR_UNI(st[3]), d = FBR_UNI(st[4]), lvl = FBR_UNI(st[5]),
                  depth = FBR_UNI(st[6]), fd = FBR_UNI(st[7]);
        double out[FBR_LINK_REC], Sv[6] = {0, 0, 0, 0, 0, 0};
        if (psrc < 0) {
            double v6[6] = {0, 0, 0, 0, 0, 0}, a6[6] = {0, 0, 0, 0, 0, 0}, e3[3] = {0, 0, 0};
            if (floating) basest(v6, a6, e3);
            fbr_kin_base(floating, g, v6, a6, e3, out);
        } else {
            if (psrc > 0)
                for (int i = 0; i < FBR_LINK_REC; i++) P[i] = load(psrc - 1, i);
            double rR[9], rp[3], ax[3];
            consts(l, rR, rp, ax);
            double qv = 0, dqv = 0, ddqv = 0;
            if (d >= 0) state(d, qv, dqv, ddqv);
            fbr_kin_child(P, rR, rp, ax, jt, qv, dqv, ddqv, out, Sv);
        }
        if (psave >= 0)
            for (int i = 0; i < FBR_LINK_REC; i++) save(psave, i, out[i]);
        for (int i = 0; i < FBR_LINK_REC; i++) P[i] = out[i];
        // the joint takes its level: whoever held it is complete
        if (lvl >= 0) {
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < MAXD; j++)
                if (j == lvl) {
                    if (ACC && fd >= 0) emit(fb + fd, tac[j]);
                    tac[j] = 0.0;
                    lvd[j] = d;
                    for (int i = 0; i < 6; i++) Sst[j][i] = Sv[i];
                }
        }
        double F[6] = {0, 0, 0, 0, 0, 0};
        link(l, depth, out, Sst, lvd, F);
        if (ACC) {
            for (int i = 0; i < 6; i++) T[i] += F[i];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < MAXD; j++)
                if (j < depth) tac[j] += fbr_dot6(Sst[j], F);
        }
    }
    if (ACC) {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < MAXD; j++)
            if (j < maxlvl) {
                const int fd = FBR_UNI(endflush[j]);
                if (fd >= 0) emit(fb + fd, tac[j]);
            }
        for (int r = 0; r < fb; r++) emit(r, T[r]);
    }
}

// Score of one regressor evaluation against a weight block (fbr_fd_scores, analyticalGradient.py:92-185):  the contribution of link l's
// inertial columns c = cpl l + p,  sum_r W[r][c] Y[r][c]  with Y's base rows = the unit wrench, joint row of path joint j = S_j . unit wrench.
// Wr(r, c): the weight of regressor row r, column c of this lane's sample.
template <int MAXD, class WFn>
FBR_HD double fbr_kinfd_link_score(int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, int cpl, int fb, WFn Wr)
{
    double acc = 0.0;
#if defined(__HIPCC__)
#pragma unroll  // (the parameter index must be a constant: fbr_unit_wrench indexes small arrays with it)
#endif
    for (int p = 0; p < 10; p++)
        if (p < cpl) {
            const int c = cpl * l + p;
            double w6[6];
            fbr_unit_wrench(rec, p, w6);
            for (int r = 0; r < fb; r++) acc += Wr(r, c) * w6[r];
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < MAXD; j++)
                if (j < depth) acc += Wr(fb + lvd[j], c) * fbr_dot6(Sst[j], w6);
        }
    return acc;
}

#if defined(__HIPCC__)
struct DevKinId {
    int nsteps, maxlvl, nslots, ldn;  // ldn: row stride (doubles, odd) of the staged joint states of one sample
    const int *steps, *endflush;
};
// tables of the lane WRITERS (fbr_kinwrite_kernel below, fbr_kinimg_kernel in fbr_gram64.h)
// destinations arrive as integers: a pointer made from one is GENERIC (flat_store: counted by lgkmcnt as well, so that every wait for a scalar load
// or an LDS read would also wait for the stores in flight) unless it is given the global address space explicitly
typedef __attribute__((address_space(1))) char *fbr_gchar_ptr;
typedef __attribute__((address_space(1))) double *fbr_gdouble_ptr;
struct DevKinWrite {
    const int *lcol10, *colrec;  // lcol10 [parts][10 L]: the columns a part's wave writes; colrec [cols + 1][2]
    const long *dst;
    int ninert, cols, k, has_w;
    int flev;  // (fbr_kinimg_kernel) base rows below this level go through the force-tile words
    int base_only;  // (fbr_kinimg_kernel) the joint rows carry weight 0 in every sample: not produced
    long group_samples;  // (fbr_kinimg_kernel) samples per group of a grouped pass (every group starts a block), 0: one group
    int nparts, part_nsteps[FBR_KINWRITE_PARTS], part_step0[FBR_KINWRITE_PARTS];  // wave w of a workgroup walks steps [step0, step0 + nsteps) of p.steps
};
#endif

#if defined(__HIPCC__) && defined(FBR_KERNELS_CORE)

// mode 0: x = full standard vector (10 per link + friction slots); mode 1: x = identified-parameter vector (cols);
// mode 2: contact wrench -> generalized force J^T w (fbr_contact_torques, model.py:535-555): x = [S][6] wrenches at the frame
// (link `flink`, point fpx/y/z in its axes), every other link contributes nothing, no friction.
// grid-stride over blocks of 64 samples; dynamic LDS: 3 x [64][ldn] doubles (q, dq, ddq of the wave's samples).
// scratch: [gridDim.x][nslots][FBR_LINK_REC][64] doubles.
template <int MAXD>
__global__ __launch_bounds__(64) void fbr_kinid_kernel(DevModel m, DevKinId p, long S, const double *__restrict__ q, const double *__restrict__ dq,
                                                       const double *__restrict__ ddq, const double *__restrict__ bv,
                                                       const double *__restrict__ ba, const double *__restrict__ rpy,
                                                       const double *__restrict__ sign, const double *__restrict__ vel_sign,
                                                       const double *__restrict__ x, int mode, double *__restrict__ tau, double *__restrict__ scratch, int flink, double fpx,
                                                       double fpy, double fpz)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x, n = m.n, ldn = p.ldn;
    double *sq = smem, *sdq = sq + 64 * ldn, *sddq = sdq + 64 * ldn;
    double *scr = scratch + (long)blockIdx.x * p.nslots * FBR_LINK_REC * 64 + lane;
    const long nblk = (S + 63) >> 6;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long base = blk << 6;
        const int valid = (int)min(64L, S - base);
        __syncthreads();  // (the block before has read its states)
        {
            // coalesced copy of the block's q / dq / ddq rows into [sample][ldn]
            const long off = base * n;
            const int cnt = valid * n;
            int sr = lane / n, dc = lane - sr * n;
            const int ds = 64 / n, dd = 64 - ds * n;
            for (int i = lane; i < cnt; i += 64) {
                const double a = q[off + i], b = dq[off + i], c = ddq[off + i];
                sq[sr * ldn + dc] = a;
                sdq[sr * ldn + dc] = b;
                sddq[sr * ldn + dc] = c;
                sr += ds;
                dc += dd;
                if (dc >= n) {
                    dc -= n;
                    sr++;
                }
            }
        }
        __syncthreads();
        const int ls = min(lane, valid - 1);  // lanes behind the last sample repeat it and store nothing
        const long s = base + ls;
        const bool live = lane < valid;
        const double *mysq = sq + ls * ldn, *mysdq = sdq + ls * ldn, *mysddq = sddq + ls * ldn;
        double *ts = tau + s * m.rows;
        auto state = [&](int d, double &a, double &b, double &c) {
            a = mysq[d];
            b = mysdq[d];
            c = mysddq[d];
        };
        auto basest = [&](double *v6, double *a6, double *e3) {
            for (int i = 0; i < 6; i++) {
                v6[i] = bv[s * 6 + i];
                a6[i] = ba[s * 6 + i];
            }
            for (int i = 0; i < 3; i++) e3[i] = rpy[s * 3 + i];
        };
        auto save = [&](int b, int i, double v) { scr[(b * FBR_LINK_REC + i) * 64] = v; };
        auto load = [&](int b, int i) { return scr[(b * FBR_LINK_REC + i) * 64]; };
        auto link = [&](int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, double *F) {
            (void)depth; (void)Sst; (void)lvd;
            if (mode == 2) {
                if (l != flink) return;
                const double *w = x + s * 6;
                const double fp[3] = {fpx, fpy, fpz};
                double t[3], pf[3], f[3] = {w[0], w[1], w[2]}, pxf[3];
                fbr_mv(rec + FBR_OFF_R, fp, t);
                for (int i = 0; i < 3; i++) pf[i] = rec[FBR_OFF_P + i] + t[i];
                fbr_cross(pf, f, pxf);
                for (int i = 0; i < 3; i++) {
                    F[i] = f[i];
                    F[3 + i] = w[3 + i] + pxf[i];  // the wrench about the base origin (frame A)
                }
                return;
            }
            double pi[10];
            if (mode == 0) {
                for (int c = 0; c < 10; c++) pi[c] = x[10 * l + c];
            } else {
                for (int c = 0; c < 10; c++) pi[c] = (c < m.cpl) ? x[m.cpl * l + c] : 0.0;
            }
            fbr_link_wrench(rec, pi, F);
        };
        auto consts = [&](int l, double *rR, double *rp, double *ax) {  // (l is wave-uniform: scalar loads through the constant address space)
            const fbr_cdouble_ptr cR = (fbr_cdouble_ptr)(unsigned long)m.restR, cp = (fbr_cdouble_ptr)(unsigned long)m.restp,
                                  ca = (fbr_cdouble_ptr)(unsigned long)m.axis;
            for (int i = 0; i < 9; i++) rR[i] = cR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = cp[3 * l + i];
                ax[i] = ca[3 * l + i];
            }
        };
        auto emit = [&](int r, double v) {
            if (r >= m.fb && m.fric && mode != 2) {
                const int d = r - m.fb;
                const double dqv = mysdq[d];
                const double sg = sign[s * n + d];
                if (mode == 0) {
                    double t = sg * x[m.fstart + d];
                    if (!m.grav_only) {
                        t += x[m.fstart + n + d] * dqv;
                        const int poff = m.fstart + 2 * n;
                        t += x[poff + d];
                        if (m.stribeck > 0) {
                            const double sgn = (sg > 0) - (sg < 0);
                            t += x[poff + n + d] * exp(-fabs(vel_sign[s * n + d]) / m.stribeck) * sgn;
                        }
                    }
                    v += t;
                } else {
                    for (int c = m.cpl * m.L; c < m.cols; c++) {
                        const int4 cd = m.coldesc[c];
                        if (cd.w == d) v += x[c] * fbr_friction_value(cd.z, dqv, sg, m.stribeck);
                    }
                }
            }
            if (live) ts[r] = v;
        };
        fbr_kinid_lane<MAXD, true>(p.nsteps, p.maxlvl, p.steps, p.endflush, m.floating, m.g, m.fb, state, basest, save, load, link, emit, consts);
    }
}
// Finite-difference sweep (analyticalGradient.py:92-185): one lane per EVALUATION e = s (1 + 3 n) + j -- j = 0 the state of
// sample s itself, 1 + kind n + d the state with +eps on q_d / dq_d / ddq_d -- score[e] = sum_{r,c} W_s[r][c] Y_e[r][c], the regressor
// never stored, no records, no expanded states in memory.  The lanes of a wave share one or two samples: their weights arrive as
// broadcast loads.  (The two-kernel path it replaces evaluates a perturbation's sub-tree columns only but stages every evaluation's
// kinematic records through HBM and the LDS of a whole workgroup: latency bound at 0.01 of its HBM floor.)
template <int MAXD>
__global__ __launch_bounds__(64) void fbr_kinfd_kernel(DevModel m, DevKinId p, long S, int nper, double eps, const double *__restrict__ q,
                                                       const double *__restrict__ dq, const double *__restrict__ ddq, const double *__restrict__ bv,
                                                       const double *__restrict__ ba, const double *__restrict__ rpy, const double *__restrict__ sign,
                                                       const double *__restrict__ W, double *__restrict__ out, double *__restrict__ scratch)
{
    const int lane = threadIdx.x, n = m.n;
    double *scr = scratch + (long)blockIdx.x * p.nslots * FBR_LINK_REC * 64 + lane;
    const long total = S * nper, nblk = (total + 63) >> 6;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long e = min((blk << 6) + lane, total - 1);
        const bool live = (blk << 6) + lane < total;
        const long s = e / nper;
        const int j = (int)(e - s * nper);
        const int kind = (j == 0) ? -1 : (j - 1) / n, dj = (j == 0) ? -1 : (j - 1) % n;
        const double *Ws = W + s * (long)m.rows * m.cols;
        double score = 0.0;
        auto state = [&](int d, double &a, double &b, double &c) {
            a = q[s * n + d] + ((kind == 0 && d == dj) ? eps : 0.0);
            b = dq[s * n + d] + ((kind == 1 && d == dj) ? eps : 0.0);
            c = ddq[s * n + d] + ((kind == 2 && d == dj) ? eps : 0.0);
        };
        auto basest = [&](double *v6, double *a6, double *e3) {
            for (int i = 0; i < 6; i++) {
                v6[i] = bv[s * 6 + i];
                a6[i] = ba[s * 6 + i];
            }
            for (int i = 0; i < 3; i++) e3[i] = rpy[s * 3 + i];
        };
        auto save = [&](int b, int i, double v) { scr[(b * FBR_LINK_REC + i) * 64] = v; };
        auto load = [&](int b, int i) { return scr[(b * FBR_LINK_REC + i) * 64]; };
        auto consts = [&](int l, double *rR, double *rp, double *ax) {  // (l is wave-uniform: scalar loads through the constant address space)
            const fbr_cdouble_ptr cR = (fbr_cdouble_ptr)(unsigned long)m.restR, cp = (fbr_cdouble_ptr)(unsigned long)m.restp,
                                  ca = (fbr_cdouble_ptr)(unsigned long)m.axis;
            for (int i = 0; i < 9; i++) rR[i] = cR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = cp[3 * l + i];
                ax[i] = ca[3 * l + i];
            }
        };
        auto Wr = [&](int r, int c) { return Ws[(long)r * m.cols + c]; };
        auto link = [&](int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, double *F) {
            (void)F;
            score += fbr_kinfd_link_score<MAXD>(l, depth, rec, Sst, lvd, m.cpl, m.fb, Wr);
        };
        auto emit = [&](int, double) {};
        fbr_kinid_lane<MAXD, false>(p.nsteps, p.maxlvl, p.steps, p.endflush, m.floating, m.g, m.fb, state, basest, save, load, link, emit, consts);
        for (int c = m.cpl * m.L; c < m.cols; c++) {  // friction columns: one entry each
            const int4 cd = m.coldesc[c];
            const int jj = cd.w;
            const double dqv = dq[s * n + jj] + ((kind == 1 && jj == dj) ? eps : 0.0);
            score += Ws[(long)(m.fb + jj) * m.cols + c] * fbr_friction_value(cd.z, dqv, sign ? sign[s * n + jj] : 0.0, m.stribeck);
        }
        if (live) out[e] = score;
    }
}
#endif

#if defined(__HIPCC__) && defined(FBR_KERNELS_GROUPS)
// ------------------------------------------------------------------------------------------------
// Regressor WRITER of the TSQR, one lane per sample, kinematics fused in (no records): the chunks the level-0 folds read, written
// COLUMN-major -- element (chunk row o, column c) at A[c * ld + o] with o = slot * Sslot + sample -- so that the 64 samples of a wave
// are 64 consecutive doubles of one column: every store instruction writes 512 contiguous bytes (four whole lines).  It replaces the
// pair fbr_kin_kernel (9.5 KB of records per WALK-MAN sample, partially written lines) + fbr_regressor_groups_kernel (one workgroup
// per sample, ~600 instructions per thread and sample, 8-byte stores at the chunks' row stride): 4.2 + 11.7 ms per 1 M samples.
// What a column writes is resolved on the host into DESTINATIONS (chunk address of sample 0 of the row's slot in the column's position,
// 0: the row's group does not hold the column / the row is switched off), in the order the lane produces the values -- no per-entry decode:
//   colrec[c] = {first destination, number of explicit zeros};  dst[first ..]: [fb base-wrench rows][one per joint of the link's path, root
//   first][the explicit zeros: rows of the groups that hold the column but are not on the link's path, right of their first supported tile]
//   friction column: [its joint's row][zeros];  pseudo-column `cols`: per regressor row its k rhs destinations.
// The tables are read through the scalar cache (constant address space, wave-uniform addresses).  The first version of this kernel decoded
// (row, kind, level, position) entries with per-entry table lookups behind vector loads: 1350 cycles per entry; with scalar loads but a
// level-indexed branch ladder per entry: 650.
// lcol10[10 l + p]: the column of parameter p of link l (-1: not identified / not selected).  Row weights are applied here.
// ------------------------------------------------------------------------------------------------
// A workgroup = nparts waves sharing one block of 64 samples (their states are staged once): wave w walks part w of the tree.
template <int MAXD>
__global__ __launch_bounds__(64 * FBR_KINWRITE_PARTS, MAXD <= 10 ? 2 : 1) void fbr_kinwrite_kernel(DevModel m, DevKinId p, DevKinWrite wr, long S, const double *__restrict__ q,
                                                          const double *__restrict__ dq, const double *__restrict__ ddq, const double *__restrict__ bv,
                                                          const double *__restrict__ ba, const double *__restrict__ rpy, const double *__restrict__ sign,
                                                          const double *__restrict__ rhs, const double *__restrict__ wts, double *__restrict__ scratch)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nth = blockDim.x, tid = threadIdx.x;
    const int n = m.n, ldn = p.ldn, ldw = m.rows | 1;
    double *sq = smem, *sdq = sq + 64 * ldn, *sddq = sdq + 64 * ldn, *sw = sddq + 64 * ldn;  // sw: [64][ldw] row weights (has_w)
    double *scr = scratch + ((long)blockIdx.x * wr.nparts + part) * p.nslots * FBR_LINK_REC * 64 + lane;
    const long nblk = (S + 63) >> 6;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long base = blk << 6;
        const int valid = (int)min(64L, S - base);
        __syncthreads();
        {
            const long off = base * n;
            const int cnt = valid * n;
            int sr = tid / n, dc = tid - sr * n;
            const int ds = nth / n, dd = nth - ds * n;
            for (int i = tid; i < cnt; i += nth) {
                const double a = q[off + i], b = dq[off + i], c = ddq[off + i];
                sq[sr * ldn + dc] = a;
                sdq[sr * ldn + dc] = b;
                sddq[sr * ldn + dc] = c;
                sr += ds;
                dc += dd;
                if (dc >= n) {
                    dc -= n;
                    sr++;
                }
            }
            if (wr.has_w) {
                const int rows = m.rows, cw = valid * rows;
                int wr_ = tid / rows, wc = tid - wr_ * rows;
                const int es = nth / rows, ed = nth - es * rows;
                for (int i = tid; i < cw; i += nth) {
                    sw[wr_ * ldw + wc] = wts[base * rows + i];
                    wr_ += es;
                    wc += ed;
                    if (wc >= rows) {
                        wc -= rows;
                        wr_++;
                    }
                }
            }
        }
        __syncthreads();
        const int ls = min(lane, valid - 1);
        const long s = base + ls;
        const bool live = lane < valid;
        const double *mysq = sq + ls * ldn, *mysdq = sdq + ls * ldn, *mysddq = sddq + ls * ldn, *myw = sw + ls * ldw;
        auto state = [&](int d, double &a, double &b, double &c) {
            a = mysq[d];
            b = mysdq[d];
            c = mysddq[d];
        };
        auto basest = [&](double *v6, double *a6, double *e3) {
            for (int i = 0; i < 6; i++) {
                v6[i] = bv[s * 6 + i];
                a6[i] = ba[s * 6 + i];
            }
            for (int i = 0; i < 3; i++) e3[i] = rpy[s * 3 + i];
        };
        auto save = [&](int b, int i, double v) { scr[(b * FBR_LINK_REC + i) * 64] = v; };
        auto load = [&](int b, int i) { return scr[(b * FBR_LINK_REC + i) * 64]; };
        auto consts = [&](int l, double *rR, double *rp, double *ax) {  // (l is wave-uniform: scalar loads through the constant address space)
            const fbr_cdouble_ptr cR = (fbr_cdouble_ptr)(unsigned long)m.restR, cp = (fbr_cdouble_ptr)(unsigned long)m.restp,
                                  ca = (fbr_cdouble_ptr)(unsigned long)m.axis;
            for (int i = 0; i < 9; i++) rR[i] = cR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = cp[3 * l + i];
                ax[i] = ca[3 * l + i];
            }
        };
        const fbr_clong_ptr cdst = (fbr_clong_ptr)(unsigned long)wr.dst;
        const fbr_cint_ptr crec = (fbr_cint_ptr)(unsigned long)wr.colrec, ccol = (fbr_cint_ptr)(unsigned long)(wr.lcol10 + (long)part * 10 * m.L);
        // one value: to its destination (sample 0 of the chunk) + s; the lanes of a wave write 64 consecutive doubles
        const unsigned sbyte = (unsigned)s << 3;  // (s: sample index inside the chunk -- one launch per chunk; a chunk's column is far below 4 GB)
        auto put = [&](long d0, double v) {
            if (d0 != 0 && live) __builtin_nontemporal_store(v, (fbr_gdouble_ptr)((fbr_gchar_ptr)d0 + sbyte));  // scalar base + 32-bit lane offset
        };
        auto link = [&](int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, double *F) {
            (void)F;
            // the vector loads of the step are waited for here, once: the stores below share their counter, and behind the branches of the
            // column code the compiler would otherwise wait for counter 0 -- the store before -- at every store (see fbr_gram64.h)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
            for (int pp = 0; pp < 10; pp++) {
                const int c = ccol[10 * l + pp];
                if (c < 0) continue;
                const int d0 = crec[2 * c], nz = crec[2 * c + 1];
                if (d0 < 0) continue;  // (the column is not factorised)
                // the record's destinations are requested TOGETHER, before anything is computed (one round trip to the scalar cache / L2 per
                // column instead of one per value; reads past the record's end stay inside the padded table and are never used)
                long db[6], dj[MAXD];
#pragma unroll
                for (int i = 0; i < 6; i++) db[i] = cdst[d0 + i];
#pragma unroll
                for (int j = 0; j < MAXD; j++) dj[j] = cdst[d0 + m.fb + j];
                double w6[6];
                fbr_unit_wrench(rec, pp, w6);
#pragma unroll
                for (int i = 0; i < 6; i++)
                    if (i < m.fb) put(db[i], wr.has_w ? w6[i] * myw[i] : w6[i]);
#pragma unroll
                for (int j = 0; j < MAXD; j++)
                    if (j < depth) {
                        const double v = fbr_dot6(Sst[j], w6);
                        put(dj[j], wr.has_w ? v * myw[m.fb + lvd[j]] : v);
                    }
                for (int z = 0; z < nz; z++) put(cdst[d0 + m.fb + depth + z], 0.0);
            }
        };
        auto emit = [&](int, double) {};
        fbr_kinid_lane<MAXD, false>(wr.part_nsteps[part], p.maxlvl, p.steps + wr.part_step0[part] * FBR_KINID_STEP, p.endflush, m.floating, m.g, m.fb,
                                    state, basest, save, load, link, emit, consts);
        if (part != wr.nparts - 1) continue;  // (friction and rhs columns: the last part, which the cut leaves the lightest)
        // friction columns: one value on the row of the column's joint, explicit zeros on the other rows of the groups that hold it
        for (int c = wr.ninert; c < wr.cols; c++) {
            const int d0 = crec[2 * c], nz = crec[2 * c + 1];
            if (d0 < 0) continue;
            const int4 cd = m.coldesc[c];
            const double fv = fbr_friction_value(cd.z, mysdq[cd.w], sign ? sign[s * n + cd.w] : 0.0, m.stribeck);
            put(cdst[d0], wr.has_w ? fv * myw[m.fb + cd.w] : fv);
            for (int z = 0; z < nz; z++) put(cdst[d0 + 1 + z], 0.0);
        }
        // rhs columns: k destinations per regressor row
        {
            const int d0 = crec[2 * wr.cols];
            for (int r = 0; r < m.rows; r++)
                for (int i = 0; i < wr.k; i++) {
                    const double v = rhs[(s * m.rows + r) * wr.k + i];
                    put(cdst[d0 + r * wr.k + i], wr.has_w ? v * myw[r] : v);
                }
        }
    }
}
#endif
