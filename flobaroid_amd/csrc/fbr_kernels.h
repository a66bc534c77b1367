// fbr_kernels.h -- device tables and HIP kernels of libfbr (gfx950).  Every translation unit of the library includes it with the
// section(s) it launches switched on (FBR_KERNELS_CORE / _GROUPS / _GRAM): a __global__ function is defined in exactly one unit.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "fbr_math.h"
#include "fbr_program.h"

// ------------------------------------------------------------------------------------------------
// device tables
// ------------------------------------------------------------------------------------------------
struct DevModel {
    int L, n, fb, rows, cols, cpl, floating, rec, maxd, nw;  // nw = 32-bit words of the ancestor mask
    int fric, grav_only, fstart;
    double g[3];
    double stribeck;
    const int *order, *parent, *dof;
    const int *jtype;                 // [L] 0 fixed / base, 1 revolute, 2 prismatic
    const double *restR, *restp, *axis;
    const int *pathlen, *pathtab;     // [L], [L*maxd]  movable joints root -> link
    const int *pathpos;               // [L*maxd] packed image row of each joint of the path (FbrHostModel::ppos)
    const unsigned *ancmask;          // [L*nw] bit d set <=> dof d is an ancestor joint of link
    const int4 *coldesc;              // [cols] kind, link, pidx/fkind, joint
    const int *sub_begin, *sub_links; // [n+1], [sum]  links in the subtree of each dof
    const int *dof_link;              // [n] child link of each dof
};

struct DevGram {
    int T, NT, k, Pa, image_doubles, part_image_max, nitems;
    int npw;                  // accumulators per wave of the kernel shape the program was built for (segw * nseg)
    int ks_limit;             // k-steps beyond this one are skipped (base-wrench-only row masks: the joint rows carry weight 0); else large
    int base_ks;              // k-steps of the paired base rows (3 with a floating base, else 0): skipped by the odd sample of a pair
    const int4 *items;        // every real column: image offset, kind, a, b
    const int *slotmeta;      // [T*WPB*NSEG*8] per row segment: [0] = offA/64 | cnt<<10 | kbegin<<18 | (tile I is a chain tile)<<23 ;
                              //   [1+j] = offB_j/64 | lookup_j<<10 | kend_j<<11   (part-local offsets; the pair runs k-steps [kbegin, kend_j))
    const int *piece_begin;   // [T+1]
    const int2 *pieces;       // x = offset in the global image, y = offset in the part image | half<<30  (doubles)
    const int *piece_begin_b; // the same for launches that stop after the base k-steps (ks_limit): one piece per tile, its 8 base rows
    const int2 *pieces_b;
    const int *rid_begin;     // [T+1]
    const int *ridl;          // per part: part-image row -> regressor row (chain tiles)
    const int *slot_tiles;    // [T*WPB*NPW*2] tile I, tile J (or -1)
    // workgroups of one sample group (the deal of fbr_gram_deal), set per launch:
    int wpg;                  // workgroups per sample group
    const int2 *wg_tab;       // [wpg] in dispatch order: x = part, y = index among the part's workgroups | their count << 16
    const int *wg_begin;      // [T+1] first partial-sum block of each part (partial sums are stored sorted by part)
    const int *tilecol;       // [NT*16] augmented column of each slot, -1 = padding
};

#ifdef FBR_KERNELS_CORE  // kinematics, materialising regressor, finite-difference scores, inverse dynamics, contact (fbr_api.hip)
// ------------------------------------------------------------------------------------------------
// K1: link kinematics, one lane per sample, AoS records  rec[s][21*L + 6*n]
// ------------------------------------------------------------------------------------------------
// (256, 5): at most 96 VGPRs (41 spilled), so that a kin wave fits beside the two 173-VGPR Gram waves of a SIMD -- the producer
// kernels run concurrently with the Gram kernel of the previous chunk; +2.7 % on the fused pass
#ifndef FBR_KIN_WAVES
#define FBR_KIN_WAVES 5  // waves per SIMD the kernel is compiled for (<= 96 VGPRs; experiments: 8 = 64 VGPRs)
#endif
// A link whose parent is the link processed just before it (DFS order: every link of a chain but the first) takes the parent's record
// from registers; only the first link after a branch point re-reads it from memory (47 -> 5 dependent 168-byte reads per sample on
// WALK-MAN: fused pass 77.4 -> 75.3 ms, TSQR call -5 ms).  Writing the records through an LDS transposition (coalesced runs of 21
// doubles instead of 8-byte stores at a 9.5 KB stride) was measured on top of that: no difference.
// WAVES: waves per SIMD the instance is compiled for.  FBR_KIN_WAVES (96 VGPRs, spills) for the producer stream of the fused Gram pass;
// 2 (no register cap, no spills) where the kernel runs alone (TSQR, materialised regressor, inverse dynamics, ...)
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void fbr_kin_kernel(DevModel m, long S, const double *__restrict__ q,
                                                       const double *__restrict__ dq, const double *__restrict__ ddq,
                                                       const double *__restrict__ bv, const double *__restrict__ ba,
                                                       const double *__restrict__ rpy, double *rec)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double *r = rec + s * (long)m.rec;
    const double *qs = q + s * m.n, *dqs = dq + s * m.n, *ddqs = ddq + s * m.n;
    double P[FBR_LINK_REC];
    int prev_l = -2;
    for (int k = 0; k < m.L; k++) {
        const int l = m.order[k];
        const int par = m.parent[l];
        double out[FBR_LINK_REC], Sv[6] = {0, 0, 0, 0, 0, 0};
        int d = -1;
        if (par < 0) {
            double v6[6] = {0, 0, 0, 0, 0, 0}, a6[6] = {0, 0, 0, 0, 0, 0}, e3[3] = {0, 0, 0};
            if (m.floating) {
                for (int i = 0; i < 6; i++) {
                    v6[i] = bv[s * 6 + i];
                    a6[i] = ba[s * 6 + i];
                }
                for (int i = 0; i < 3; i++) e3[i] = rpy[s * 3 + i];
            }
            fbr_kin_base(m.floating, m.g, v6, a6, e3, out);
        } else {
            if (par != prev_l)
                for (int i = 0; i < FBR_LINK_REC; i++) P[i] = r[FBR_LINK_REC * par + i];
            d = m.dof[l];
            double rR[9], rp[3], ax[3];
            for (int i = 0; i < 9; i++) rR[i] = m.restR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = m.restp[3 * l + i];
                ax[i] = m.axis[3 * l + i];
            }
            double qv = 0, dqv = 0, ddqv = 0;
            if (d >= 0) {
                qv = qs[d];
                dqv = dqs[d];
                ddqv = ddqs[d];
            }
            fbr_kin_child(P, rR, rp, ax, m.jtype[l], qv, dqv, ddqv, out, Sv);
        }
        if (d >= 0)
            for (int i = 0; i < 6; i++) r[FBR_LINK_REC * m.L + FBR_DOF_REC * d + i] = Sv[i];
        for (int i = 0; i < FBR_LINK_REC; i++) r[FBR_LINK_REC * l + i] = out[i];
        for (int i = 0; i < FBR_LINK_REC; i++) P[i] = out[i];
        prev_l = l;
    }
}

#endif  // FBR_KERNELS_CORE

typedef double fbr_d2 __attribute__((ext_vector_type(2)));
typedef double fbr_d4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the global stores in flight (s_waitcnt vmcnt(0)): in
// the per-sample producer loops below that made every sample wait for the write acknowledgements of the one before.
__device__ __forceinline__ void fbr_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Copy n doubles global -> LDS with all of a thread's loads in flight together (a plain `for (i = tid; ...) dst[i] = src[i]` loop is
// compiled to load / wait / store per element: with 5 elements per thread that was 5 serial memory latencies per sample).
template <int NT_, int U_ = 6> __device__ __forceinline__ void fbr_stage_copy(double *dst, const double *__restrict__ src, int n, int tid)
{
    for (int base = tid; base < n; base += NT_ * U_) {
        double v[U_];
#pragma unroll
        for (int u = 0; u < U_; u++) v[u] = src[min(base + NT_ * u, n - 1)];  // clamped: unconditional loads
#pragma unroll
        for (int u = 0; u < U_; u++)
            if (base + NT_ * u < n) dst[base + NT_ * u] = v[u];
    }
}

#ifdef FBR_KERNELS_CORE
// ------------------------------------------------------------------------------------------------
// K2: materialised standard regressor  Y[s][rows][cols]; one workgroup per sample (grid-stride), one
// thread per column, every row of a sample written as one contiguous, coalesced run of `cols` doubles.
// Bound: HBM write (8*rows*cols bytes per sample).
// ------------------------------------------------------------------------------------------------
// ldy = leading dimension of Y in doubles (>= cols; the TSQR path writes straight into its padded chunk)
__global__ __launch_bounds__(256) void fbr_regressor_kernel(DevModel m, long S, const double *__restrict__ rec,
                                                             const double *__restrict__ dq,
                                                             const double *__restrict__ sign, double *__restrict__ Y, int ldy, long rs_s,
                                                             long rs_r, const int *__restrict__ linkpos, const int *__restrict__ rowfc)
{
    // linkpos (optional): the cpl columns of link l are written at column cpl * linkpos[l] (the TSQR chunks order the links by depth)
    // rowfc (optional): regressor row r is only written from the column tile of rowfc[r] on (the TSQR folds never read left of it)
    // row (s, r) of the output is row s * rs_s + r * rs_r: (rows, 1) = the reference's sample-major stack; (1, S) = row-major by
    // regressor row (the TSQR chunks: all samples' row r together, see fbr_tsqr.h)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *rs = smem;  // [rec]
    const int tid = threadIdx.x;
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        fbr_barrier_lds();
        fbr_stage_copy<256>(rs, rec + s * (long)m.rec, m.rec, tid);
        fbr_barrier_lds();
        double *Ys = Y + s * rs_s * ldy;
        const long rl = rs_r * ldy;
        for (int c = tid; c < m.cols; c += blockDim.x) {
            const int4 cd = m.coldesc[c];
            if (cd.x == 0) {
                const int co = linkpos ? m.cpl * linkpos[cd.y] + (c - m.cpl * cd.y) : c;
                double w6[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
                for (int r = 0; r < m.fb; r++) Ys[r * rl + co] = w6[r];
                for (int d = 0; d < m.n; d++) {
                    const unsigned bit = (m.ancmask[cd.y * m.nw + (d >> 5)] >> (d & 31)) & 1u;
                    double v = 0.0;
                    if (bit) v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * d, w6);
                    if (!rowfc || co >= (rowfc[m.fb + d] & ~15)) Ys[(m.fb + d) * rl + co] = v;
                }
            } else {
                const int j = cd.w;
                const double v = fbr_friction_value(cd.z, dq[s * m.n + j], sign ? sign[s * m.n + j] : 0.0, m.stribeck);
                for (int r = 0; r < m.rows; r++) Ys[r * rl + c] = (r == m.fb + j) ? v : 0.0;
            }
        }
    }
}

// K2b: same result as K2 for an EVEN number of columns: one thread per PAIR of adjacent columns, 16-byte stores
// (1 KiB per wave-instruction), ancestor masks hoisted into registers.  Adjacent inertial columns always belong to
// the same link (10 or 4 columns per link).
// ------------------------------------------------------------------------------------------------
// Finite-difference sweep (excitation/analyticalGradient.py:92-185): per sample the baseline state and
// 3n states perturbed by +eps in q_d, dq_d, ddq_d.  fbr_fd_expand_kernel writes the 1 + 3n states of every sample,
// fbr_score_kernel evaluates score = sum_{r,c} W_s[r][c] * Y[r][c] of each (the regressor block is never stored).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_fd_expand_kernel(long S, int n, int has_base, int has_sign, double eps, const double *__restrict__ q,
                                                             const double *__restrict__ dq, const double *__restrict__ ddq,
                                                             const double *__restrict__ bv, const double *__restrict__ ba,
                                                             const double *__restrict__ rpy, const double *__restrict__ sign,
                                                             double *__restrict__ eq, double *__restrict__ edq, double *__restrict__ eddq,
                                                             double *__restrict__ ebv, double *__restrict__ eba, double *__restrict__ erpy,
                                                             double *__restrict__ esign)
{
    const int nper = 1 + 3 * n;
    const long total = S * nper;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long s = e / nper;
        const int j = (int)(e - s * nper);          // 0 = baseline, 1 + d = q_d, 1 + n + d = dq_d, 1 + 2n + d = ddq_d
        const int kind = (j == 0) ? -1 : (j - 1) / n, d = (j == 0) ? -1 : (j - 1) % n;
        for (int i = 0; i < n; i++) {
            eq[e * n + i] = q[s * n + i] + ((kind == 0 && i == d) ? eps : 0.0);
            edq[e * n + i] = dq[s * n + i] + ((kind == 1 && i == d) ? eps : 0.0);
            eddq[e * n + i] = ddq[s * n + i] + ((kind == 2 && i == d) ? eps : 0.0);
            if (has_sign) esign[e * n + i] = sign[s * n + i];
        }
        if (has_base) {
            for (int i = 0; i < 6; i++) {
                ebv[e * 6 + i] = bv[s * 6 + i];
                eba[e * 6 + i] = ba[s * 6 + i];
            }
            for (int i = 0; i < 3; i++) erpy[e * 3 + i] = rpy[s * 3 + i];
        }
    }
}

// <W_s[:, c], Y_e[:, c]> of one column of one (expanded) sample whose record is staged in rs
__device__ __forceinline__ double fbr_score_column(const DevModel &m, const double *rs, const double *__restrict__ Ws, int c, long e,
                                                   const double *__restrict__ dq, const double *__restrict__ sign)
{
    const int4 cd = m.coldesc[c];
    double acc = 0.0;
    if (cd.x == 0) {
        double w6[6];
        fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
        for (int r = 0; r < m.fb; r++) acc += Ws[(long)r * m.cols + c] * w6[r];
        for (int d = 0; d < m.n; d++) {
            const unsigned bit = (m.ancmask[cd.y * m.nw + (d >> 5)] >> (d & 31)) & 1u;
            if (bit) acc += Ws[(long)(m.fb + d) * m.cols + c] * fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * d, w6);
        }
    } else {
        const int j = cd.w;
        acc = Ws[(long)(m.fb + j) * m.cols + c] * fbr_friction_value(cd.z, dq[e * m.n + j], sign ? sign[e * m.n + j] : 0.0, m.stribeck);
    }
    return acc;
}

// Scores of the finite-difference sweep.  A perturbation of joint d changes the kinematic records of the links BELOW d only, hence only
// their columns of the regressor (and d's own friction columns): the baseline evaluation of a sample (phase 0, one workgroup per
// sample) computes every column once and leaves the partial sums over every joint's sub-tree columns (jcols[jbeg[d] .. jbeg[d+1]))
// in part[s][d]; a perturbed evaluation (phase 1, one workgroup per (sample, perturbation)) computes the sub-tree columns of its joint
// only and returns  baseline - part[s][d] + its own partial sum  (WALK-MAN: 16 % of the columns on average).  Sums run in a fixed order.
__global__ __launch_bounds__(256) void fbr_score_kernel(DevModel m, long S, int nper, int phase, const double *__restrict__ rec,
                                                         const double *__restrict__ dq, const double *__restrict__ sign,
                                                         const double *__restrict__ W, double *__restrict__ out, double *__restrict__ part,
                                                         const int *__restrict__ jbeg, const int *__restrict__ jcols)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *rs = smem;                 // [rec]
    double *red = smem + m.rec;        // [4] reduction
    double *contrib = red + 4;         // [cols] (phase 0)
    const int tid = threadIdx.x;
    const long total = phase == 0 ? S : S * (nper - 1);
    for (long i = blockIdx.x; i < total; i += gridDim.x) {
        const long s = phase == 0 ? i : i / (nper - 1);
        const int j = phase == 0 ? 0 : 1 + (int)(i - s * (nper - 1));
        const long e = s * nper + j;
        fbr_barrier_lds();
        fbr_stage_copy<256>(rs, rec + e * (long)m.rec, m.rec, tid);
        fbr_barrier_lds();
        const double *Ws = W + s * (long)m.rows * m.cols;
        double acc = 0.0;
        int d = -1;
        if (phase == 0) {
            for (int c = tid; c < m.cols; c += blockDim.x) {
                const double v = fbr_score_column(m, rs, Ws, c, e, dq, sign);
                contrib[c] = v;
                acc += v;
            }
        } else {
            d = (j - 1) % m.n;
            for (int q = jbeg[d] + tid; q < jbeg[d + 1]; q += blockDim.x) acc += fbr_score_column(m, rs, Ws, jcols[q], e, dq, sign);
        }
        // deterministic block reduction: wave sums by xor butterflies, then 4 partials in fixed order
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        fbr_barrier_lds();
        const double sum = (red[0] + red[1]) + (red[2] + red[3]);
        if (phase == 0) {
            if (tid == 0) out[e] = sum;
            if (tid < m.n) {
                double p = 0.0;
                for (int q = jbeg[tid]; q < jbeg[tid + 1]; q++) p += contrib[jcols[q]];
                part[s * m.n + tid] = p;
            }
        } else if (tid == 0) {
            out[e] = (out[s * nper] - part[s * m.n + d]) + sum;
        }
    }
}

__global__ __launch_bounds__(256) void fbr_regressor2_kernel(DevModel m, long S, int spb, const double *__restrict__ rec,
                                                              const double *__restrict__ dq,
                                                              const double *__restrict__ sign, double *__restrict__ Y, int ldy, long rs_s,
                                                              long rs_r, const int *__restrict__ linkpos, const int *__restrict__ rowfc)
{
    // spb samples per workgroup pass (small robots: 256 / (cols/2) samples side by side)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int npairs = m.cols >> 1;
    const int ls = tid / npairs, pr = tid - ls * npairs;  // local sample, column pair
    const long ngroups = (S + spb - 1) / spb;
    for (long gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
        const long sb = gi * spb;
        const int ns = (int)min((long)spb, S - sb);
        fbr_barrier_lds();
        fbr_stage_copy<256>(smem, rec + sb * (long)m.rec, ns * m.rec, tid);
        fbr_barrier_lds();
        if (ls >= ns) continue;
        const long s = sb + ls;
        const double *rs = smem + (long)ls * m.rec;
        double *Ys = Y + s * rs_s * ldy;
        for (int prr = pr; prr < npairs; prr += (spb > 1 ? npairs : (int)blockDim.x)) {
            const int c = 2 * prr;
            const int4 ca = m.coldesc[c], cb = m.coldesc[c + 1];
            // (a pair of adjacent inertial columns belongs to one link: it moves with the link's column block, cpl is even)
            const int cout = (linkpos && ca.x == 0) ? m.cpl * linkpos[ca.y] + (c - m.cpl * ca.y) : c;
            fbr_d2 *dst = (fbr_d2 *)(Ys + cout);
            const long rstride = rs_r * (ldy >> 1);  // in double2 units (ldy even)
            if (ca.x == 0) {
                double wa[6], wb[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * ca.y, ca.z, wa);
                fbr_unit_wrench(rs + FBR_LINK_REC * cb.y, cb.z, wb);
                for (int r = 0; r < m.fb; r++) dst[r * rstride] = (fbr_d2){wa[r], wb[r]};
                // same link for both columns: one ancestor mask
                for (int wd = 0; wd < m.nw; wd++) {
                    const unsigned mask = m.ancmask[ca.y * m.nw + wd];
                    const int dmax = min(32, m.n - 32 * wd);
                    for (int b = 0; b < dmax; b++) {
                        const int d = 32 * wd + b;
                        fbr_d2 v = {0.0, 0.0};
                        if ((mask >> b) & 1u) {
                            const double *Sd = rs + FBR_LINK_REC * m.L + FBR_DOF_REC * d;
                            v[0] = fbr_dot6(Sd, wa);
                            v[1] = fbr_dot6(Sd, wb);
                        } else if (rowfc && cout < (rowfc[m.fb + d] & ~15)) {
                            continue;  // structural zero left of the row's first supported column tile: never read by the folds
                        }
                        dst[(m.fb + d) * rstride] = v;
                    }
                }
            } else {
                const double va = fbr_friction_value(ca.z, dq[s * m.n + ca.w], sign ? sign[s * m.n + ca.w] : 0.0, m.stribeck);
                const double vb = fbr_friction_value(cb.z, dq[s * m.n + cb.w], sign ? sign[s * m.n + cb.w] : 0.0, m.stribeck);
                for (int r = 0; r < m.rows; r++)
                    dst[r * rstride] = (fbr_d2){(r == m.fb + ca.w) ? va : 0.0, (r == m.fb + cb.w) ? vb : 0.0};
            }
        }
    }
}

// active[r] = 1 iff some sample gives regressor row r a non-zero weight (row masks such as the base-wrench-only identification,
// identifier.py:629-636, switch whole rows off: their groups / k-steps are skipped)
__global__ __launch_bounds__(256) void fbr_row_active_kernel(const double *__restrict__ w, long S, int rows, int *__restrict__ active)
{
    const long total = S * rows;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        if (w[i] != 0.0) active[i % rows] = 1;  // (benign race: every writer stores 1)
}

#endif  // FBR_KERNELS_CORE
#ifdef FBR_KERNELS_GROUPS  // row-group writers of the tree-structured TSQR (fbr_tsqr_api.hip)
// ------------------------------------------------------------------------------------------------
// K2c: the regressor written as the ROW GROUPS of the tree-structured TSQR (fbr_api.hip: tsqr_group_plan).  A regressor row
// belongs to one group; a group g owns a packed chunk A_g [slot][sample][ld_g] that holds only the columns its rows can touch,
// followed by the rhs columns at psel_g.  One workgroup per sample (grid-stride), one thread per model column / rhs column.  What a
// column thread writes is a host-built entry list (ent[ebeg[c] .. ebeg[c+1])): regressor row | kind << 8 | position in the row's
// group << 10, kind 0 = base-wrench row, 1 = joint row (S_d . w), 2 = explicit zero, 3 = friction value -- rows of groups
// that do not hold the column have no entry, and the zeros left of a row's first supported column tile are left out of the list
// when the folds never read them.  Row weights are applied here.
// Bound: HBM write of the groups' chunks (WALK-MAN: 5.8 k instead of 17.4 k doubles per sample).
// ------------------------------------------------------------------------------------------------
struct FbrDevGroup {
    double *A;
    int ld, psel;
};
// Rows [S, Sslot) of every slot of every group's chunk := 0 (the last chunk of a call is padded to whole fold blocks per slot, so that a
// block never straddles two regressor rows; padded rows are zero rows of the factorisation).  grid = (groups, slots), nrows[g] slots each.
__global__ __launch_bounds__(256) void fbr_groups_clear_pad_kernel(const FbrDevGroup *__restrict__ grp, const int *__restrict__ nrows, long S, long Sslot)
{
    const int g = blockIdx.x, slot = blockIdx.y;
    if (slot >= nrows[g]) return;
    double *p = grp[g].A + ((long)slot * Sslot + S) * grp[g].ld;
    const long cnt = (Sslot - S) * grp[g].ld;
    for (long i = threadIdx.x; i < cnt; i += blockDim.x) p[i] = 0.0;
}
// The same for the COLUMN-major chunks of the lane writer (fbr_kinid.h fbr_kinwrite_kernel): element (slot, sample, column c) at
// A[c * ldc + slot * Sslot + sample], ldc = nrows[g] * Sslot.  Rows [S, Sslot) of every slot := 0 in the columns < pa[g], and the padding
// columns [pa[g], ld) := 0 over all rows.  grid = (groups, maxrows + FBR_CM_PADWG): blocks y < nrows[g] take a slot's padding rows, the
// last FBR_CM_PADWG blocks share the padding columns.
#define FBR_CM_PADWG 64
__global__ __launch_bounds__(256) void fbr_groups_clear_cm_kernel(const FbrDevGroup *__restrict__ grp, const int *__restrict__ nrows, const int *__restrict__ pa,
                                                                  long S, long Sslot, int maxrows)
{
    const int g = blockIdx.x, y = blockIdx.y;
    const long ldc = (long)nrows[g] * Sslot;
    if (y < maxrows) {
        const long padr = Sslot - S;
        if (y >= nrows[g] || padr <= 0) return;
        for (long i = threadIdx.x; i < padr * pa[g]; i += blockDim.x) {
            const long c = i / padr, r = i - c * padr;
            grp[g].A[c * ldc + (long)y * Sslot + S + r] = 0.0;
        }
    } else {
        double *p = grp[g].A + (long)pa[g] * ldc;
        const long cnt = (long)(grp[g].ld - pa[g]) * ldc;
        for (long i = (long)(y - maxrows) * blockDim.x + threadIdx.x; i < cnt; i += (long)FBR_CM_PADWG * blockDim.x) p[i] = 0.0;
    }
}
__global__ __launch_bounds__(256) void fbr_regressor_groups_kernel(DevModel m, long S, const double *__restrict__ rec, const double *__restrict__ dq,
                                                                    const double *__restrict__ sign, const double *__restrict__ rhs, int k,
                                                                    const double *__restrict__ wts, const FbrDevGroup *__restrict__ grp, int ngroups,
                                                                    const int *__restrict__ rowgroup, const int *__restrict__ rowslot,
                                                                    const int *__restrict__ ebeg, const int *__restrict__ ent, long Sslot)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *rs = smem;                                    // [rec]
    double **rowptr = (double **)(smem + ((m.rec + 1) & ~1));  // [rows] chunk row of this sample's regressor row r (its group's chunk)
    const int tid = threadIdx.x;
    // the record of the NEXT sample is fetched into registers while this one is written out (<= 6 doubles per thread)
    constexpr int PU = 6;
    const bool pre = m.rec <= 256 * PU;
    double pv[PU];
    if (pre && (long)blockIdx.x < S)
#pragma unroll
        for (int u = 0; u < PU; u++) pv[u] = rec[blockIdx.x * (long)m.rec + min(tid + 256 * u, m.rec - 1)];
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        fbr_barrier_lds();
        if (pre) {
#pragma unroll
            for (int u = 0; u < PU; u++)
                if (tid + 256 * u < m.rec) rs[tid + 256 * u] = pv[u];
            if (s + gridDim.x < S)
#pragma unroll
                for (int u = 0; u < PU; u++) pv[u] = rec[(s + gridDim.x) * (long)m.rec + min(tid + 256 * u, m.rec - 1)];
        } else {
            fbr_stage_copy<256>(rs, rec + s * (long)m.rec, m.rec, tid);
        }
        if (tid < m.rows) {
            const int g = rowgroup[tid];
            rowptr[tid] = g >= 0 ? grp[g].A + ((long)rowslot[tid] * Sslot + s) * grp[g].ld : nullptr;  // (Sslot >= S: slot stride of the chunk, padded to whole blocks)
        }
        fbr_barrier_lds();
        // rhs columns: one element per thread (a single thread walking all rows would be the critical path of the sample)
        for (int t = tid; t < m.rows * k; t += blockDim.x) {
            const int r = t / k, i = t - r * k;
            const int g = rowgroup[r];
            if (g < 0) continue;
            double v = rhs[(s * m.rows + r) * k + i];
            if (wts) v *= wts[s * m.rows + r];
            rowptr[r][grp[g].psel + i] = v;
        }
        for (int c = tid; c < m.cols; c += blockDim.x) {
            const int e0 = ebeg[c], e1 = ebeg[c + 1];
            if (e0 == e1) continue;
            const int4 cd = m.coldesc[c];
            double w6[6] = {0, 0, 0, 0, 0, 0}, fv = 0.0;
            if (cd.x == 0)
                fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
            else
                fv = fbr_friction_value(cd.z, dq[s * m.n + cd.w], sign ? sign[s * m.n + cd.w] : 0.0, m.stribeck);
            int en_next = ent[e0];
            for (int e = e0; e < e1; e++) {
                const int en = en_next;  // row | kind << 8 | position << 10
                en_next = ent[e + 1 < e1 ? e + 1 : e];  // (prefetched: the loop body does not wait for its own entry)
                const int r = en & 0xff, kind = (en >> 8) & 3, pos = en >> 10;
                double v = 0.0;
                if (kind == 0)
                    v = w6[r];
                else if (kind == 1)
                    v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * (r - m.fb), w6);
                else if (kind == 3)
                    v = fv;
                if (wts) v *= wts[s * m.rows + r];
                __builtin_nontemporal_store(v, rowptr[r] + pos);  // streaming store: the chunk is read back from HBM by the folds (measured -13 %)
            }
        }
    }
}

// K2e (round 5, option tsqr_writer = 32): the same writer with the rows of a sample STAGED IN THE LDS, copied out with 16-byte stores, and
// with no global load between a sample's stores and the next sample's compute phase.  Built to test two explanations of K2c's time
// (10.9 - 12.1 ms per 1 M samples for 22.9 KB per sample = 1.9 TB/s): (i) the width / number of its 8-byte stores, (ii) the
// store-acknowledge latency on every sample's critical path -- on gfx9 loads and stores share one in-order counter (vmcnt) and the compiler
// can only wait for a load with vmcnt(0), so K2c's first global load of a sample (prefetched record, entry lists, rhs) also waits for the
// write acknowledgements of the sample before.  Here
//   * the entry lists are copied into the LDS once per workgroup, column descriptors and list bounds live in registers;
//   * everything per-sample that comes from global memory (record, rhs, weights, dq, sign) is requested for sample i + 1 at the top of
//     iteration i and taken out of its registers AFTER the compute phase of sample i, in front of that sample's copy-out: the stores the
//     wait sits through are those of sample i - 1, a whole compute phase old;  the compute phase reads the LDS only;
//   * a regressor row of a sample is one contiguous run of its group's chunk (ld_g doubles): the column threads write their entries into an
//     LDS image of the sample's rows (rowoff[r]; zero where nothing is ever written: cleared once per workgroup), which the workgroup
//     streams out in 16-byte pieces (structural zeros and padding included: 25.6 instead of 22.9 KB per sample on the regrouped WALK-MAN).
// MEASURED: 10.9 instead of 11.0 - 12.1 ms per 1 M samples (the call 54.8 instead of 55.5 ms) -- neither explanation holds; the kernel's
// ~600 instructions per thread and sample (unit wrench ~100, ~13 entries x ~30: decode, six LDS reads of the joint's motion vector, six
// fused multiply-adds, one store) at 8 waves per SIMD account for its time: it is bound by what it issues.  Kept as an option.
// total = doubles of the image (even); naux = rows k + (wts ? rows : 0) + (fric ? 2 n : 0) <= 512 staged values; nent = list entries.
__global__ __launch_bounds__(256) void fbr_regressor_groups_lds_kernel(DevModel m, long S, const double *__restrict__ rec, const double *__restrict__ dq,
                                                                        const double *__restrict__ sign, const double *__restrict__ rhs, int k,
                                                                        const double *__restrict__ wts, const FbrDevGroup *__restrict__ grp, int ngroups,
                                                                        const int *__restrict__ rowgroup, const int *__restrict__ rowslot,
                                                                        const int *__restrict__ ebeg, const int *__restrict__ ent,
                                                                        const int *__restrict__ rowoff, int total, long Sslot, int nent)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int o_rhs = 0, o_w = m.rows * k, o_dq = o_w + (wts ? m.rows : 0), o_sg = o_dq + (m.fric ? m.n : 0), naux = o_sg + ((m.fric && sign) ? m.n : 0);
    double *rs = smem;                                                  // [rec]
    double *aux = rs + ((m.rec + 1) & ~1);                              // [naux] rhs | weights | dq | sign of the sample
    double **rowptr = (double **)(aux + ((naux + 1) & ~1));             // [rows] chunk row of regressor row r of sample 0 (constant; + s ld per sample)
    double *img = (double *)(rowptr + ((m.rows + 1) & ~1));             // [total] the sample's rows, group by group
    unsigned short *piece_row = (unsigned short *)(img + total);       // [total / 2] regressor row of every 16-byte piece of the image
    int *roff = (int *)(piece_row + ((total / 2 + 3) & ~3));            // [rows] LDS offset of row r (-1: the row belongs to no group)
    int *psel = roff + m.rows;                                          // [rows] position of the first rhs column in row r's group
    int *rld = psel + m.rows;                                           // [rows] leading dimension of row r's group
    int *entl = rld + m.rows;                                           // [nent] the entry lists
    const int tid = threadIdx.x;
    for (int i = tid; i < total; i += 256) img[i] = 0.0;
    for (int i = tid; i < nent; i += 256) entl[i] = ent[i];
    for (int r = tid; r < m.rows; r += 256) {
        const int g = rowgroup[r];
        roff[r] = rowoff[r];
        psel[r] = g >= 0 ? grp[g].psel : 0;
        rld[r] = g >= 0 ? grp[g].ld : 0;
        rowptr[r] = g >= 0 ? grp[g].A + (long)rowslot[r] * Sslot * grp[g].ld : nullptr;
    }
    fbr_barrier_lds();
    for (int r = tid; r < m.rows; r += 256) {
        const int g = rowgroup[r];
        if (g < 0) continue;
        for (int c = roff[r] / 2; c < (roff[r] + grp[g].ld) / 2; c++) piece_row[c] = (unsigned short)r;
    }
    // this thread's columns: descriptor and list bounds (loop invariant; at most two columns per thread: cols <= 512)
    int4 cdv[2];
    int eb0[2], eb1[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int c = tid + 256 * q;
        eb0[q] = eb1[q] = 0;
        cdv[q] = make_int4(0, 0, 0, 0);
        if (c < m.cols) {
            eb0[q] = ebeg[c];
            eb1[q] = ebeg[c + 1];
            cdv[q] = m.coldesc[c];
        }
    }
    constexpr int PU = 6;
    double pv[PU], pa[2];
    auto aux_load = [&](int i, long ss) -> double {  // value i of the staged per-sample inputs of sample ss
        if (i < o_w) return rhs[ss * m.rows * k + i];
        if (i < o_dq) return wts[ss * m.rows + (i - o_w)];
        if (i < o_sg) return dq[ss * m.n + (i - o_dq)];
        return sign[ss * m.n + (i - o_sg)];
    };
    auto request = [&](long ss) {
#pragma unroll
        for (int u = 0; u < PU; u++) pv[u] = rec[ss * (long)m.rec + min(tid + 256 * u, m.rec - 1)];
#pragma unroll
        for (int u = 0; u < 2; u++) pa[u] = (tid + 256 * u < naux) ? aux_load(tid + 256 * u, ss) : 0.0;
    };
    auto deliver = [&]() {
#pragma unroll
        for (int u = 0; u < PU; u++)
            if (tid + 256 * u < m.rec) rs[tid + 256 * u] = pv[u];
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (tid + 256 * u < naux) aux[tid + 256 * u] = pa[u];
    };
    if ((long)blockIdx.x < S) {
        request(blockIdx.x);
        deliver();
        if ((long)blockIdx.x + gridDim.x < S) request(blockIdx.x + gridDim.x);
    }
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        fbr_barrier_lds();  // (record and inputs of this sample in place; the copy-out of the sample before has read the image)
        const double *ws = wts ? aux + o_w : nullptr;
        for (int t = tid; t < m.rows * k; t += blockDim.x) {
            const int r = t / k, i = t - r * k;
            if (roff[r] < 0) continue;
            double v = aux[o_rhs + t];
            if (ws) v *= ws[r];
            img[roff[r] + psel[r] + i] = v;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int e0 = eb0[q], e1 = eb1[q];
            if (e0 == e1) continue;
            const int4 cd = cdv[q];
            double w6[6] = {0, 0, 0, 0, 0, 0}, fv = 0.0;
            if (cd.x == 0)
                fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
            else
                fv = fbr_friction_value(cd.z, aux[o_dq + cd.w], (m.fric && sign) ? aux[o_sg + cd.w] : 0.0, m.stribeck);
            for (int e = e0; e < e1; e++) {
                const int en = entl[e];  // row | kind << 8 | position << 10
                const int r = en & 0xff, kind = (en >> 8) & 3, pos = en >> 10;
                double v = 0.0;
                if (kind == 0)
                    v = w6[r];
                else if (kind == 1)
                    v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * (r - m.fb), w6);
                else if (kind == 3)
                    v = fv;
                if (ws) v *= ws[r];
                img[roff[r] + pos] = v;
            }
        }
        fbr_barrier_lds();
        // the next sample's record and inputs (nobody reads rs / aux any more), and the request for the sample after it
        if (s + gridDim.x < S) {
            deliver();
            if (s + 2 * gridDim.x < S) request(s + 2 * gridDim.x);
        }
        // copy-out: piece p = image doubles [2p, 2p + 2) -> its place in the chunk row of its regressor row
        for (int p = tid; p < total / 2; p += 256) {
            const int r = piece_row[p];
            const fbr_d2 v = *(const fbr_d2 *)(img + 2 * p);
            __builtin_nontemporal_store(v, (fbr_d2 *)(rowptr[r] + s * (long)rld[r] + (2 * p - roff[r])));
        }
    }
}

// K2d: the same writer with one thread per PAIR of adjacent inertial columns and 16-byte stores (a wave instruction covers 1 KiB of a
// chunk row instead of 512 B; the 8-byte form is bound by its store instructions at 2.5 TB/s).  The two columns of a pair belong to one
// link, so they share their entry list: pent[pbeg[pr] .. pbeg[pr+1]) = regressor row | kind << 8 | (even) position of the first column
// in the row's group << 10.  Used when every group keeps the two columns of every pair side by side at an even position (always for the
// whole regressor; a column subset falls back to K2c).  Friction columns take the single-column entries (ebeg / ent) as in K2c.
__global__ __launch_bounds__(256) void fbr_regressor_groups2_kernel(DevModel m, long S, const double *__restrict__ rec, const double *__restrict__ dq,
                                                                     const double *__restrict__ sign, const double *__restrict__ rhs, int k,
                                                                     const double *__restrict__ wts, const FbrDevGroup *__restrict__ grp, int ngroups,
                                                                     const int *__restrict__ rowgroup, const int *__restrict__ rowslot,
                                                                     const int *__restrict__ ebeg, const int *__restrict__ ent,
                                                                     const int *__restrict__ pbeg, const int *__restrict__ pent, int npairs,
                                                                     int ninert, int split, long Sslot)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *rs = smem;                                    // [rec]
    double **rowptr = (double **)(smem + ((m.rec + 1) & ~1));  // [rows] chunk row of this sample's regressor row r (its group's chunk)
    const int tid = threadIdx.x;
    constexpr int PU = 6;
    const bool pre = m.rec <= 256 * PU;
    double pv[PU];
    if (pre && (long)blockIdx.x < S)
#pragma unroll
        for (int u = 0; u < PU; u++) pv[u] = rec[blockIdx.x * (long)m.rec + min(tid + 256 * u, m.rec - 1)];
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        fbr_barrier_lds();
        if (pre) {
#pragma unroll
            for (int u = 0; u < PU; u++)
                if (tid + 256 * u < m.rec) rs[tid + 256 * u] = pv[u];
            if (s + gridDim.x < S)
#pragma unroll
                for (int u = 0; u < PU; u++) pv[u] = rec[(s + gridDim.x) * (long)m.rec + min(tid + 256 * u, m.rec - 1)];
        } else {
            fbr_stage_copy<256>(rs, rec + s * (long)m.rec, m.rec, tid);
        }
        if (tid < m.rows) {
            const int g = rowgroup[tid];
            rowptr[tid] = g >= 0 ? grp[g].A + ((long)rowslot[tid] * Sslot + s) * grp[g].ld : nullptr;  // (Sslot >= S: slot stride of the chunk, padded to whole blocks)
        }
        fbr_barrier_lds();
        for (int t = tid; t < m.rows * k; t += blockDim.x) {
            const int r = t / k, i = t - r * k;
            const int g = rowgroup[r];
            if (g < 0) continue;
            double v = rhs[(s * m.rows + r) * k + i];
            if (wts) v *= wts[s * m.rows + r];
            rowptr[r][grp[g].psel + i] = v;
        }
        // work items: the pairs, then the single columns.  With fewer items than threads an item's entry list is SPLIT over `split`
        // threads (thread t: item t % nitems, entries part, part + split, ...): the sample's critical path is the longest entry walk of
        // a thread, and half a workgroup of idle threads beside 120 busy ones made the 16-byte writer slower than the 8-byte one
        const int nitems = npairs + (m.cols - 2 * npairs);
        for (int t = tid; t < nitems * split; t += blockDim.x) {
            const int part = t / nitems, it = t - part * nitems;
            if (it < npairs) {
                const int pr = it;
                const int e0 = pbeg[pr], e1 = pbeg[pr + 1];
                if (e0 + part >= e1) continue;
                const int4 ca = m.coldesc[2 * pr], cb = m.coldesc[2 * pr + 1];
                double wa[6], wb[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * ca.y, ca.z, wa);
                fbr_unit_wrench(rs + FBR_LINK_REC * ca.y, cb.z, wb);
                int en_next = pent[e0 + part];
                for (int e = e0 + part; e < e1; e += split) {
                    const int en = en_next;
                    en_next = pent[e + split < e1 ? e + split : e];
                    const int r = en & 0xff, kind = (en >> 8) & 3, pos = en >> 10;
                    fbr_d2 v = {0.0, 0.0};
                    if (kind == 0) {
                        v[0] = wa[r];
                        v[1] = wb[r];
                    } else if (kind == 1) {
                        const double *Sd = rs + FBR_LINK_REC * m.L + FBR_DOF_REC * (r - m.fb);
                        v[0] = fbr_dot6(Sd, wa);
                        v[1] = fbr_dot6(Sd, wb);
                    }
                    if (wts) v *= wts[s * m.rows + r];
                    __builtin_nontemporal_store(v, (fbr_d2 *)(rowptr[r] + pos));
                }
                continue;
            }
            // inertial columns without a partner (models with column masks: one per link with an odd column count), then the friction columns
            const int c = 2 * npairs + (it - npairs);
            const int e0 = ebeg[c], e1 = ebeg[c + 1];
            if (e0 + part >= e1) continue;
            const int4 cd = m.coldesc[c];
            if (c < ninert) {
                double w6[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
                for (int e = e0 + part; e < e1; e += split) {
                    const int en = ent[e];
                    const int r = en & 0xff, kind = (en >> 8) & 3, pos = en >> 10;
                    double v = 0.0;
                    if (kind == 0)
                        v = w6[r];
                    else if (kind == 1)
                        v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * (r - m.fb), w6);
                    if (wts) v *= wts[s * m.rows + r];
                    __builtin_nontemporal_store(v, rowptr[r] + pos);
                }
                continue;
            }
            const double fv = fbr_friction_value(cd.z, dq[s * m.n + cd.w], sign ? sign[s * m.n + cd.w] : 0.0, m.stribeck);
            for (int e = e0 + part; e < e1; e += split) {
                const int en = ent[e];
                const int r = en & 0xff, kind = (en >> 8) & 3, pos = en >> 10;
                double v = kind == 3 ? fv : 0.0;
                if (wts) v *= wts[s * m.rows + r];
                __builtin_nontemporal_store(v, rowptr[r] + pos);
            }
        }
    }
}

#endif  // FBR_KERNELS_GROUPS
#ifdef FBR_KERNELS_CORE
// ------------------------------------------------------------------------------------------------
// K3: inverse dynamics / prediction, one wavefront per sample.
//   mode 0: x = full standard vector (10 per link + friction slots), friction model of model.py:299-326
//   mode 1: x = identified-parameter vector (cols): tau = Y_s x
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_id_kernel(DevModel m, long S, const double *__restrict__ rec,
                                                      const double *__restrict__ dq, const double *__restrict__ sign,
                                                      const double *__restrict__ vel_sign,
                                                      const double *__restrict__ x, int mode, double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    double *rs = smem + (long)wave * (m.rec + 6 * m.L);  // per-wave [rec] + F[L][6]
    double *F = rs + m.rec;
    for (long s = (long)blockIdx.x * nwaves + wave; s < S; s += (long)gridDim.x * nwaves) {
        for (int i = lane; i < m.rec; i += 64) rs[i] = rec[s * (long)m.rec + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int l = lane; l < m.L; l += 64) {
            double pi[10];
            if (mode == 0) {
                for (int p = 0; p < 10; p++) pi[p] = x[10 * l + p];
            } else {
                for (int p = 0; p < 10; p++) pi[p] = (p < m.cpl) ? x[m.cpl * l + p] : 0.0;
            }
            double w6[6];
            fbr_link_wrench(rs + FBR_LINK_REC * l, pi, w6);
            for (int i = 0; i < 6; i++) F[6 * l + i] = w6[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int r = lane; r < m.rows; r += 64) {
            double v = 0.0;
            if (r < m.fb) {
                for (int l = 0; l < m.L; l++) v += F[6 * l + r];
            } else {
                const int d = r - m.fb;
                double acc[6] = {0, 0, 0, 0, 0, 0};
                for (int i = m.sub_begin[d]; i < m.sub_begin[d + 1]; i++) {
                    const int l = m.sub_links[i];
                    for (int c = 0; c < 6; c++) acc[c] += F[6 * l + c];
                }
                v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * d, acc);
                if (m.fric) {
                    const double dqv = dq[s * m.n + d];
                    const double sg = sign[s * m.n + d];
                    if (mode == 0) {
                        double t = sg * x[m.fstart + d];
                        if (!m.grav_only) {
                            t += x[m.fstart + m.n + d] * dqv;
                            const int poff = m.fstart + 2 * m.n;
                            t += x[poff + d];
                            if (m.stribeck > 0) {
                                const double sgn = (sg > 0) - (sg < 0);
                                t += x[poff + m.n + d] * exp(-fabs(vel_sign[s * m.n + d]) / m.stribeck) * sgn;
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
            }
            tau[s * m.rows + r] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// K4: contact wrench -> generalized force (J^T w), one lane per sample; uses link records.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_contact_kernel(DevModel m, long S, const double *__restrict__ rec, int flink,
                                                           double fpx, double fpy, double fpz,
                                                           const double *__restrict__ wrench, double *__restrict__ out)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double *r = rec + s * (long)m.rec;
    const double *w = wrench + 6 * s;
    double *o = out + s * m.rows;
    for (int i = 0; i < m.rows; i++) o[i] = 0.0;
    double R[9], pl[3], fp[3] = {fpx, fpy, fpz}, t[3], pf[3];
    for (int i = 0; i < 9; i++) R[i] = r[FBR_LINK_REC * flink + FBR_OFF_R + i];
    for (int i = 0; i < 3; i++) pl[i] = r[FBR_LINK_REC * flink + FBR_OFF_P + i];
    fbr_mv(R, fp, t);
    for (int i = 0; i < 3; i++) pf[i] = pl[i] + t[i];
    double f[3] = {w[0], w[1], w[2]}, nn[3] = {w[3], w[4], w[5]}, pxf[3];
    fbr_cross(pf, f, pxf);
    double w6[6] = {f[0], f[1], f[2], nn[0] + pxf[0], nn[1] + pxf[1], nn[2] + pxf[2]};  // wrench about the base origin
    for (int i = 0; i < m.fb; i++) o[i] = w6[i];
    const int len = m.pathlen[flink];
    for (int j = 0; j < len; j++) {
        const int d = m.pathtab[flink * m.maxd + j];
        double Sv[6];
        for (int i = 0; i < 6; i++) Sv[i] = r[FBR_LINK_REC * m.L + FBR_DOF_REC * d + i];
        o[m.fb + d] = fbr_dot6(Sv, w6);
    }
}

// ------------------------------------------------------------------------------------------------
// K6: Fourier-series joint trajectories of the trajectory optimiser's candidates (excitation/trajectoryGenerator.py:
// OscillationGenerator 411-460, BoundedOscillationGenerator 462-560, the vectorised evaluation of computeTrajectoryDynamics 83-128):
// one thread per (candidate c, sample t, joint j).  With x_l = wf_c * (t / freq * l):
//   classic:  q = sum_l a_l / (wf l) sin x_l - b_l / (wf l) cos x_l + qoff,  dq = sum a_l cos x_l + b_l sin x_l,
//             ddq = sum -a_l wf l sin x_l + b_l wf l cos x_l                                   (qoff = nf * q0)
//   bounded:  raw = sum b_l cos x_l + a_l sin x_l,  q = qoff + qrange tanh(raw)  and its two time derivatives (qoff = q_center)
// coef = [a | b] [C][n][nh] each (harmonics beyond a joint's own nf are zero), so that the candidates never cross PCIe as samples.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_fourier_kernel(int C, long T, int n, int nh, double freq, const double *__restrict__ wf,
                                                          const double *__restrict__ a, const double *__restrict__ b,
                                                          const double *__restrict__ qoff, const double *__restrict__ qrange,
                                                          double *__restrict__ q, double *__restrict__ dq, double *__restrict__ ddq)
{
    const long total = (long)C * T * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % n);
        const long ct = e / n, t = ct % T;
        const int c = (int)(ct / T);
        const double w = wf[c], ts = (double)t / freq;
        const double *aa = a + ((long)c * n + j) * nh, *bb = b + ((long)c * n + j) * nh;
        if (!qrange) {
            double sq = 0.0, sv = 0.0, sa = 0.0;
            for (int l = 1; l <= nh; l++) {
                const double x = w * (ts * (double)l), wl = w * (double)l;
                double sn, cs;
                sincos(x, &sn, &cs);
                const double al = aa[l - 1], bl = bb[l - 1];
                sq += (al / wl) * sn - (bl / wl) * cs;
                sv += al * cs + bl * sn;
                sa += -(al * wl) * sn + (bl * wl) * cs;
            }
            q[e] = sq + qoff[(long)c * n + j];
            dq[e] = sv;
            ddq[e] = sa;
        } else {
            double raw = 0.0, rd = 0.0, rdd = 0.0;
            for (int l = 1; l <= nh; l++) {
                const double x = w * (ts * (double)l), wl = w * (double)l;
                double sn, cs;
                sincos(x, &sn, &cs);
                const double al = aa[l - 1], bl = bb[l - 1];
                raw += bl * cs + al * sn;
                rd += (al * wl) * cs - (bl * wl) * sn;
                rdd += -(al * wl * wl) * sn - (bl * wl * wl) * cs;
            }
            const double th = tanh(raw), sech2 = 1.0 - th * th, qr = qrange[(long)c * n + j];
            q[e] = qoff[(long)c * n + j] + qr * th;
            dq[e] = qr * sech2 * rd;
            ddq[e] = qr * (sech2 * rdd - 2.0 * th * sech2 * rd * rd);
        }
    }
}

#endif  // FBR_KERNELS_CORE
#ifdef FBR_KERNELS_GRAM  // tile-image packer and fused Gram (fbr_gram_api.hip)
// ------------------------------------------------------------------------------------------------
// K5a: packed tile image of [Y_s | rhs_s] (row weights applied), one workgroup per sample (grid-stride),
// one thread per real column.  pimg[s][image_doubles]; structural zeros and padding are never written
// (the buffer is zeroed once when it is allocated).  Bound: HBM write of the non-zero entries.
// ------------------------------------------------------------------------------------------------

struct FbrStage {
    int o_rhs, o_w, o_dq, o_sign, total;
};
__device__ __forceinline__ double fbr_stage_load(const FbrStage &sg, int i, long s, int rec_n, int rows, int k, int n,
                                                 const double *__restrict__ rec, const double *__restrict__ rhs,
                                                 const double *__restrict__ wts, const double *__restrict__ dq,
                                                 const double *__restrict__ sign)
{
    if (i < sg.o_rhs) return rec[s * (long)rec_n + i];
    if (i < sg.o_w) return rhs[s * (long)rows * k + (i - sg.o_rhs)];
    if (i < sg.o_dq) return wts[s * (long)rows + (i - sg.o_w)];
    if (i < sg.o_sign) return dq[s * (long)n + (i - sg.o_dq)];
    if (i < sg.total) return sign[s * (long)n + (i - sg.o_sign)];
    return 0.0;
}

// Floating base: the 6 base-wrench rows of two consecutive samples of a group share 3 MFMA k-steps (FbrHostModel::fbp).  Packed
// base position of row r of a sample: even sample -> r of its own image; odd sample -> r < 2: 6 + r of its partner's image (written
// from this workgroup), else 2 + r of its own.  Dense tiles use the same base rows, joint row j at fbp + j.  The Gram kernel skips
// k-step 0 of the chain tiles of odd samples.  Sg = samples per group of this launch (pairs never straddle groups).
// (256, 7): at most 72 VGPRs, two pack waves fit beside the two 173-VGPR Gram waves of a SIMD (2 x 176 + 2 x 72 <= 512); the rhs
// moments spill under 64.
__global__ __launch_bounds__(256, 7) void fbr_pack_kernel(DevGram g, DevModel m, long S, long Sg, const double *__restrict__ rec,
                                                        const double *__restrict__ dq, const double *__restrict__ sign,
                                                        const double *__restrict__ rhs, const double *__restrict__ wts,
                                                        double *__restrict__ pimg, int base_only, double *__restrict__ mom)
{
    // mom (optional, g.k <= 2 and one work item per thread): the rhs columns have no tiles; this thread's column c accumulates
    // sum_rows Y[r][c] rhs[r][i] over the samples of the workgroup (every entry is in a register when it is stored), thread 255
    // rhs^T rhs -- mom[block][256][4]: [0..1] this thread's column against rhs 0 / 1 ; thread 255: [0] r0.r0, [1] r0.r1, [2] r1.r1
    double macc[3] = {0.0, 0.0, 0.0};
    // base_only: the row weights switch every joint row off (base-wrench-only identification): their image rows are neither computed nor
    // written, the Gram kernel does not run their k-steps (DevGram::ks_limit)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    FbrStage sg;
    sg.o_rhs = m.rec;
    sg.o_w = sg.o_rhs + m.rows * g.k;
    sg.o_dq = sg.o_w + (wts ? m.rows : 0);
    sg.o_sign = sg.o_dq + (m.fric ? m.n : 0);
    sg.total = sg.o_sign + ((m.fric && sign) ? m.n : 0);
    double *rs = smem;                                  // [stage]
    int *plen = (int *)(rs + ((sg.total + 1) & ~1));    // [L]
    int *ptab = plen + m.L;                             // [L*maxd] dof of each path joint
    int *ppos = ptab + m.L * m.maxd;                    // [L*maxd] packed image row of each path joint
    const int tid = threadIdx.x;
    for (int i = tid; i < m.L; i += 256) plen[i] = m.pathlen[i];
    for (int i = tid; i < m.L * m.maxd; i += 256) {
        ptab[i] = m.pathtab[i];
        ppos[i] = m.pathpos[i];
    }
    const double *ws = wts ? rs + sg.o_w : nullptr;
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        fbr_barrier_lds();
        fbr_stage_copy<256>(rs, rec + s * (long)m.rec, m.rec, tid);
        for (int i = sg.o_rhs + tid; i < sg.total; i += 256)  // rhs, weights, dq, sign: usually one element per thread
            rs[i] = fbr_stage_load(sg, i, s, m.rec, m.rows, g.k, m.n, rec, rhs, wts, dq, sign);
        fbr_barrier_lds();
        double *img = pimg + s * (long)g.image_doubles;
        const long local = s % Sg;
        const bool odd = (local & 1) != 0, partner = !odd && local + 1 < Sg;
        const int fbp = m.fb ? 8 : 0;
        // packed base position of base row r and the image it goes to
        auto bpos = [&](int r) { return odd ? (r < 2 ? 6 + r : 2 + r) : r; };
        auto bimg = [&](int r) { return (odd && r < 2) ? img - g.image_doubles : img; };
        // weighted rhs entry (row r, rhs column i) of this sample
        auto rw = [&](int r, int i) { return ws ? rs[sg.o_rhs + r * g.k + i] * ws[r] : rs[sg.o_rhs + r * g.k + i]; };
        if (mom && tid == 255) {
            for (int r = 0; r < (base_only ? m.fb : m.rows); r++) {
                const double a = rw(r, 0), b = g.k > 1 ? rw(r, 1) : 0.0;
                macc[0] += a * a;
                macc[1] += a * b;
                macc[2] += b * b;
            }
        }
        for (int it = tid; it < g.nitems; it += 256) {
            const int4 d = g.items[it];
            if (d.y == 0) {
                double w6[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * d.z, d.w, w6);
                for (int r = 0; r < m.fb; r++) {
                    const double v = ws ? w6[r] * ws[r] : w6[r];
                    bimg(r)[d.x + bpos(r) * FBR_TILE] = v;
                    if (mom) {
                        macc[0] += v * rw(r, 0);
                        if (g.k > 1) macc[1] += v * rw(r, 1);
                    }
                }
                if (m.fb && !odd && !partner) img[d.x + 6 * FBR_TILE] = img[d.x + 7 * FBR_TILE] = 0.0;  // no partner: clear stale ghost rows
                const int len = base_only ? 0 : plen[d.z];
                for (int j = 0; j < len; j++) {
                    const int dd = ptab[d.z * m.maxd + j];
#if defined(FBR_PACK_TIMING_CHEAPDOT)  // timing-only experiments (results wrong): 1 = one multiply and one LDS read per joint row,
                    const double *Sx = rs + FBR_LINK_REC * m.L + FBR_DOF_REC * dd;  // 2 = one multiply, all six LDS reads kept
                    double v = Sx[0] * w6[0];
#if FBR_PACK_TIMING_CHEAPDOT == 2
                    asm volatile("" ::"v"(Sx[1]), "v"(Sx[2]), "v"(Sx[3]), "v"(Sx[4]), "v"(Sx[5]));
#endif
#else
                    double v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * dd, w6);
#endif
                    if (ws) v *= ws[m.fb + dd];
                    img[d.x + ppos[d.z * m.maxd + j] * FBR_TILE] = v;
                    if (mom) {
                        macc[0] += v * rw(m.fb + dd, 0);
                        if (g.k > 1) macc[1] += v * rw(m.fb + dd, 1);
                    }
                }
            } else if (d.y == 1) {
                if (base_only) continue;
                const int r = m.fb + d.z;
                double v = fbr_friction_value(d.w, rs[sg.o_dq + d.z], sign ? rs[sg.o_sign + d.z] : 0.0, m.stribeck);
                if (ws) v *= ws[r];
                img[d.x] = v;  // d.x points at the packed row of the joint
                if (mom) {
                    macc[0] += v * rw(r, 0);
                    if (g.k > 1) macc[1] += v * rw(r, 1);
                }
            } else {
                for (int r = 0; r < (base_only ? m.fb : m.rows); r++) {
                    double v = rs[sg.o_rhs + r * g.k + d.z];
                    if (ws) v *= ws[r];
                    if (r < m.fb)
                        bimg(r)[d.x + bpos(r) * FBR_TILE] = v;
                    else
                        img[d.x + (fbp + r - m.fb) * FBR_TILE] = v;
                }
                // dense x dense pairs run every k-step of every sample: rows this sample does not own must be zero
                if (m.fb && odd)
                    for (int r = 0; r < 4; r++) img[d.x + r * FBR_TILE] = 0.0;
                if (m.fb && !odd && !partner) img[d.x + 6 * FBR_TILE] = img[d.x + 7 * FBR_TILE] = 0.0;
            }
        }
    }
    if (mom && (tid < g.nitems || tid == 255)) {  // (+=: the chunks of a call launch one after the other; the reduction leaves zeros behind)
        double *mo = mom + ((long)blockIdx.x * 256 + tid) * 4;
        mo[0] += macc[0];
        mo[1] += macc[1];
        mo[2] += macc[2];
    }
}

// rhs moments of a call -> G.  Workgroup t = pack thread t (its column itemcol[t], or -1; t = 255: rhs^T rhs): the partial sums of the
// pack workgroups are added in a fixed order (thread j takes workgroups j, j + 256, ..., then a tree over the threads): deterministic.
__global__ __launch_bounds__(256) void fbr_gram_mom_reduce_kernel(int P, int k, int nblocks, const int *__restrict__ itemcol,
                                                                  double *__restrict__ mom, double *__restrict__ G)
{
    // (every partial sum is read by exactly one thread, which puts the zero back: the next call of this parity starts from a clean
    // buffer without a 17 MB fill -- measured at 0.35 ms per call, more than a whole 50 k-sample KUKA pass)
    __shared__ double red[3][256];
    const int Pa = P + k, t = blockIdx.x, j = threadIdx.x;
    const int c = itemcol[t];
    if (c < 0 && t != 255) return;
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = j; b < nblocks; b += 256)
        for (int i = 0; i < 3; i++) {
            s[i] += mom[((long)b * 256 + t) * 4 + i];
            mom[((long)b * 256 + t) * 4 + i] = 0.0;
        }
    for (int i = 0; i < 3; i++) red[i][j] = s[i];
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (j < h)
            for (int i = 0; i < 3; i++) red[i][j] += red[i][j + h];
        __syncthreads();
    }
    if (j != 0) return;
    if (t == 255) {
        G[(long)P * Pa + P] += red[0][0];
        if (k > 1) {
            G[(long)P * Pa + P + 1] += red[1][0];
            G[(long)(P + 1) * Pa + P] += red[1][0];
            G[(long)(P + 1) * Pa + P + 1] += red[2][0];
        }
        return;
    }
    for (int i = 0; i < k; i++) {
        G[(long)c * Pa + P + i] += red[i][0];
        G[(long)(P + i) * Pa + c] += red[i][0];
    }
}

// ------------------------------------------------------------------------------------------------
// K5b: streaming Gram.  Workgroup = (part of the tile-pair list, share of the samples).  Per sample the part's
// tiles are copied global -> LDS by LDS-DMA (global_load_lds, no VGPRs) into one of two buffers while every
// wave runs its <= FBR_NPW accumulators over the other buffer with v_mfma_f64_16x16x4_f64; operands of the
// next k-step (and of the next accumulator's first k-step) are fetched before the current MFMA issues.
// Workgroups are dealt to the parts in proportion to the parts' cost, so that all of them finish together.  Bound: fp64 MFMA.
// TIMING: diagnostic instantiation (s_memtime cycles per phase and wave into dbg[block][wave][8]).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void *fbr_lds_ptr;
typedef const __attribute__((address_space(1))) void *fbr_glb_ptr;

// FBR_SEGW x FBR_NSEG: the wave's accumulator shape (FbrGramConfig, fbr_program.h); (5,2) must fit 128 VGPRs (two workgroups per CU).
template <bool TIMING, int FBR_SEGW, int FBR_NSEG>
__global__ __launch_bounds__(FBR_WPB * 64, (FBR_SEGW * FBR_NSEG <= 10) ? 4 : 2) void fbr_gram_kernel(DevGram g, long S, int NG, const double *__restrict__ pimg,
                                                           double *__restrict__ partial, unsigned long long *__restrict__ dbg, int carry)
{
    // carry: the accumulators start from this workgroup's partial sums of the previous chunk of the same call (same launch shape),
    // so that a call reduces its partial sums ONCE, after its last chunk, instead of once per chunk (fixed order: still deterministic)
    constexpr int FBR_NPW = FBR_SEGW * FBR_NSEG;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *buf0 = smem, *buf1 = smem + g.part_image_max;
    int *ridl = (int *)(smem + 2 * g.part_image_max);   // [part image rows]
    int *mslot = ridl + g.part_image_max / FBR_TILE;    // [WPB*NSEG*8] row-segment metadata of this part
    int *pcs = mslot + FBR_WPB * FBR_NSEG * 8;          // [2*pieces] DMA pieces of this part
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (sample group, part, share of the group's samples): a part's workgroups split the group's samples evenly,
    // the more expensive parts have more workgroups (fbr_gram_deal)
    const int group = blockIdx.x / g.wpg;
    const int2 wt = g.wg_tab[blockIdx.x - group * g.wpg];
    const int part = wt.x, widx = wt.y & 0xffff, wcnt = (int)((unsigned)wt.y >> 16);
    const long gs0 = (S * group) / NG, gs1 = (S * (group + 1)) / NG;
    const long s0 = gs0 + ((gs1 - gs0) * widx) / wcnt, s1 = gs0 + ((gs1 - gs0) * (widx + 1)) / wcnt;
    const int pc0 = g.piece_begin[part], npc = g.piece_begin[part + 1] - pc0;
    const int rb0 = g.rid_begin[part], nrid = g.rid_begin[part + 1] - rb0;

    for (int i = tid; i < 2 * g.part_image_max; i += FBR_WPB * 64) smem[i] = 0.0;
    for (int i = tid; i < nrid; i += FBR_WPB * 64) ridl[i] = g.ridl[rb0 + i];
    for (int i = tid; i < FBR_WPB * FBR_NSEG * 8; i += FBR_WPB * 64) mslot[i] = g.slotmeta[(long)part * FBR_WPB * FBR_NSEG * 8 + i];
    for (int i = tid; i < npc; i += FBR_WPB * 64) {
        const int2 pc = g.pieces[pc0 + i];
        pcs[2 * i] = pc.x;
        pcs[2 * i + 1] = pc.y;
    }

    fbr_d4 acc[FBR_NPW];
    double *pp = partial + ((((long)group * g.wpg + g.wg_begin[part] + widx) * FBR_WPB + wave) * FBR_NPW) * 256;
    if (carry) {
#pragma unroll
        for (int p = 0; p < FBR_NPW; p++) acc[p] = (fbr_d4){pp[p * 256 + lane], pp[p * 256 + 64 + lane], pp[p * 256 + 128 + lane], pp[p * 256 + 192 + lane]};
    } else {
#pragma unroll
        for (int p = 0; p < FBR_NPW; p++) acc[p] = (fbr_d4){0.0, 0.0, 0.0, 0.0};
    }
    const int *wmeta = mslot + wave * FBR_NSEG * 8;
    const int li = lane & 15, kk = lane >> 4;
    unsigned long long tacc[3] = {0, 0, 0}, t0 = 0;

    // LDS-DMA of sample s into buf: wave w issues pieces w, w+WPB, ...; the first FBR_NQ descriptors of the wave are kept
    // in SGPRs for the whole kernel so that the issue is a straight run of s_mov m0 / global_load_lds pairs
    constexpr int FBR_NQ = 10;
    int pgx[FBR_NQ], ply[FBR_NQ];
    __syncthreads();  // tables visible
#pragma unroll
    for (int q = 0; q < FBR_NQ; q++) {
        const int i = wave + FBR_WPB * q;
        pgx[q] = (i < npc) ? __builtin_amdgcn_readfirstlane(pcs[2 * i]) : 0;
        ply[q] = (i < npc) ? __builtin_amdgcn_readfirstlane(pcs[2 * i + 1]) : -1;
    }
    auto dma = [&](long s, double *buf) {
        const double *src = pimg + s * (long)g.image_doubles + 2 * lane;
#pragma unroll
        for (int q = 0; q < FBR_NQ; q++) {
            const int ly = ply[q];
            if (ly < 0) continue;
            const int loff = ly & 0x3fffffff;
            if (ly >> 30) {
                if (lane < 32)
                    __builtin_amdgcn_global_load_lds((fbr_glb_ptr)(src + pgx[q]), (fbr_lds_ptr)(buf + loff), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((fbr_glb_ptr)(src + pgx[q]), (fbr_lds_ptr)(buf + loff), 16, 0, 0);
            }
        }
        for (int i = wave + FBR_WPB * FBR_NQ; i < npc; i += FBR_WPB) {
            const int gx = __builtin_amdgcn_readfirstlane(pcs[2 * i]);
            const int ly = __builtin_amdgcn_readfirstlane(pcs[2 * i + 1]);
            const int loff = ly & 0x3fffffff;
            if (ly >> 30) {
                if (lane < 32)
                    __builtin_amdgcn_global_load_lds((fbr_glb_ptr)(src + gx), (fbr_lds_ptr)(buf + loff), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((fbr_glb_ptr)(src + gx), (fbr_lds_ptr)(buf + loff), 16, 0, 0);
            }
        }
    };
    __syncthreads();  // tables and zeroed buffers visible before the first DMA lands
    if (s0 < s1) dma(s0, buf0);
    if (TIMING) t0 = __builtin_readcyclecounter();
    for (long s = s0; s < s1; s++) {
        double *img = ((s - s0) & 1) ? buf1 : buf0;
        // this sample's image has landed and the other buffer is free.  The wave's LDS-DMA pieces are VMEM operations: every wave must
        // have drained ITS pieces (vmcnt(0)) BEFORE the barrier, or a wave could read rows another wave's DMA has not delivered yet.
        // The wait is explicit: the compiler's own s_waitcnt insertion puts it in front of the wave's first LDS read of the image
        // (enough for the wave's own pieces only) whenever nothing else forces it in front of the barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (TIMING) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[0] += t1 - t0; t0 = t1; }
        if (s + 1 < s1) dma(s + 1, ((s - s0) & 1) ? buf0 : buf1);
        if (TIMING) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[1] += t1 - t0; t0 = t1; }
        const int kskip = ((s - gs0) & 1) ? g.base_ks : 0;  // the base rows of an odd sample sit in its partner's image
        // ---- MFMA phase: per row segment the A fragment of (I, ks) is loaded once and feeds up to SEGW
        //      independent accumulators (tiles J, sorted by k-steps descending)
#pragma unroll
        for (int sgi = 0; sgi < FBR_NSEG; sgi++) {
            const int mv = wmeta[sgi * 8 + (lane & 7)];
            const int m0 = __builtin_amdgcn_readlane(mv, 0);
            const int cnt = (m0 >> 10) & 15;
            if (cnt == 0) continue;
            const int oA = (m0 & 0x3ff) << 6;
            int mj[FBR_SEGW];
#pragma unroll
            for (int j = 0; j < FBR_SEGW; j++) mj[j] = __builtin_amdgcn_readlane(mv, 1 + j);
            // lane-dependent addresses are rebuilt per sample: hoisted out of the sample loop they would cost the VGPRs that
            // keep the kernel at 4 waves per SIMD
            int lane_v = lane;
            asm volatile("" : "+v"(lane_v));
            const double *pa = img + oA + lane_v;
            const int *pr = ridl + (oA >> 4) + (lane_v >> 4);
            {
                // every pair of the segment runs the k-steps [kb, its own end) and the pairs are sorted by their end descending:
                // the pairs that run k-step ks are the first n(ks), n falling.  One branch-free loop per n.  (Friction pairs
                // whose tiles meet in a few rows only start late and end early; k-steps inside the range that are structurally
                // zero run anyway and add zeros.)
                // The row-map entry of k-step ks + 1 (chain x dense pairs) is read during k-step ks: one LDS round trip per k-step.
                // (Reading the operands of ks + 1 between the MFMAs of ks, into the registers just consumed, was measured 10 %
                // slower than this read-all / wait / issue-all order, a second register set for the operands of ks + 1 15 % slower
                // (LDS reads between the MFMAs delay their issue); hand-issued reads with s_waitcnt lgkmcnt(N - j) before MFMA j
                // instead of the compiler's lgkmcnt(0) before the first one: no difference.)
                auto nksteps = [](int mjv) { return (mjv >> 11) & 0xff; };  // one past the last k-step of the pair
                int ks = (m0 >> 18) & 31;
                if ((m0 >> 23) & 1) ks = max(ks, kskip);  // tile I is packed by position: its base k-steps belong to even samples
                int rm = pr[4 * ks];
                auto kstep = [&](auto nc, int ks) {
                    constexpr int N = decltype(nc)::value;
                    const int vlk = rm * FBR_TILE + (lane_v & 15);
                    const int vpos = 64 * ks + lane_v;
                    double b[N];
#pragma unroll
                    for (int j = 0; j < N; j++) b[j] = img[((mj[j] & 0x3ff) << 6) + (((mj[j] >> 10) & 1) ? vlk : vpos)];
                    __builtin_amdgcn_sched_barrier(0);  // keep the A read after the b reads: the wait for `rm` must not cover it
                    const double a = pa[64 * ks];
                    rm = pr[4 * ks + 4];  // may run a few entries past the tile's rows: still inside the LDS tables, never used
#pragma unroll
                    for (int j = 0; j < N; j++)
                        acc[sgi * FBR_SEGW + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[sgi * FBR_SEGW + j], 0, 0, 0);
                };
#define FBR_RUN_PREFIX(N)                                                              \
    if (N <= FBR_SEGW) {                                                               \
        const int kend = min(nksteps(mj[(N) - 1 < FBR_SEGW ? (N) - 1 : 0]), g.ks_limit);    \
        for (; ks < kend; ks++) kstep(std::integral_constant<int, (N) <= FBR_SEGW ? (N) : 1>{}, ks); \
    }
                FBR_RUN_PREFIX(8)
                FBR_RUN_PREFIX(7)
                FBR_RUN_PREFIX(6)
                FBR_RUN_PREFIX(5)
                FBR_RUN_PREFIX(4)
                FBR_RUN_PREFIX(3)
                FBR_RUN_PREFIX(2)
                FBR_RUN_PREFIX(1)
#undef FBR_RUN_PREFIX
            }
        }
        if (TIMING) { const unsigned long long t1 = __builtin_readcyclecounter(); tacc[2] += t1 - t0; t0 = t1; }
    }
    if (TIMING && lane == 0) {
        unsigned long long *d = dbg + ((long)blockIdx.x * FBR_WPB + wave) * 8;
        d[0] = tacc[0]; d[1] = tacc[1]; d[2] = tacc[2];
        d[6] = (unsigned long long)part;
        d[7] = (unsigned long long)(s1 - s0);
    }
    // ---- write this workgroup's accumulators: partial[workgroup][wave][slot][reg][lane]
#pragma unroll
    for (int p = 0; p < FBR_NPW; p++) {
        pp[p * 256 + 0 * 64 + lane] = acc[p][0];
        pp[p * 256 + 1 * 64 + lane] = acc[p][1];
        pp[p * 256 + 2 * 64 + lane] = acc[p][2];
        pp[p * 256 + 3 * 64 + lane] = acc[p][3];
    }
}

// Deterministic reduction over the workgroups of a part + scatter into the symmetric G (augmented column order).
// One workgroup (256 threads = 4 regs x 64 lanes) per accumulator slot.  G must be pre-zeroed or hold the
// running sum: every G entry is touched by exactly one thread.
__global__ __launch_bounds__(256) void fbr_gram_reduce_kernel(DevGram g, const double *__restrict__ partial, double *__restrict__ G)
{
    // blockIdx.y = sample group: the partial sums of its workgroups go to G + group * Pa^2
    const int slot = blockIdx.x;  // (part*WPB + wave)*NPW + p
    const int I = g.slot_tiles[2 * slot], J = g.slot_tiles[2 * slot + 1];
    if (I < 0) return;
    const int t = threadIdx.x, reg = t >> 6, lane = t & 63;
    const int per_wg = FBR_WPB * g.npw;
    const int part = slot / per_wg, rem = slot - part * per_wg;
    const long w0 = (long)blockIdx.y * g.wpg + g.wg_begin[part], w1 = (long)blockIdx.y * g.wpg + g.wg_begin[part + 1];
    // four independent running sums (the loads of a thread are 8 bytes at a 2 KB x slots stride: with one sum the loop is one memory latency
    // per workgroup of the pass), added in a fixed order
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    long w = w0;
    for (; w + 3 < w1; w += 4) {
        v0 += partial[((w * per_wg + rem) << 8) + t];
        v1 += partial[(((w + 1) * per_wg + rem) << 8) + t];
        v2 += partial[(((w + 2) * per_wg + rem) << 8) + t];
        v3 += partial[(((w + 3) * per_wg + rem) << 8) + t];
    }
    for (; w < w1; w++) v0 += partial[((w * per_wg + rem) << 8) + t];
    const double v = (v0 + v1) + (v2 + v3);
    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
    const int ci = g.tilecol[I * FBR_TILE + row], cj = g.tilecol[J * FBR_TILE + col];
    if (ci < 0 || cj < 0) return;
    double *Gg = G + (long)blockIdx.y * g.Pa * g.Pa;
    Gg[(long)ci * g.Pa + cj] += v;
    if (I != J) Gg[(long)cj * g.Pa + ci] += v;
}

#endif  // FBR_KERNELS_GRAM