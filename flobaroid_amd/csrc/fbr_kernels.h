// fbr_kernels.h -- device tables and HIP kernels of libfbr (gfx950).  Included once by fbr_api.hip.
#pragma once
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
    const double *restR, *restp, *axis;
    const int *pathlen, *pathtab;     // [L], [L*maxd]  movable joints root -> link
    const unsigned *ancmask;          // [L*nw] bit d set <=> dof d is an ancestor joint of link
    const int4 *coldesc;              // [cols] kind, link, pidx/fkind, joint
    const int *sub_begin, *sub_links; // [n+1], [sum]  links in the subtree of each dof
    const int *dof_link;              // [n] child link of each dof
};

struct DevGram {
    int T, NT, k, Pa, image_doubles, rid_stride, ntab;
    const int4 *items;      // off, kind, a, b
    const int *item_begin;  // [T+1]
    const int *slotmeta;    // [T*WPB*NPW]: offA/64 | (offB/64)<<10 | common<<20 | lookup<<28   (0 = unused slot)
    const int *rowid;       // [ntab = image rows] image row -> global regressor row (chain tiles)
    const int *slot_tiles;  // [T*WPB*NPW*2] tile I, tile J (or -1)
    const int *tilecol;     // [NT*16] augmented column of each slot, -1 = padding
};

// ------------------------------------------------------------------------------------------------
// K1: link kinematics, one lane per sample, AoS records  rec[s][21*L + 6*n]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_kin_kernel(DevModel m, long S, const double *__restrict__ q,
                                                       const double *__restrict__ dq, const double *__restrict__ ddq,
                                                       const double *__restrict__ bv, const double *__restrict__ ba,
                                                       const double *__restrict__ rpy, double *rec)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double *r = rec + s * (long)m.rec;
    const double *qs = q + s * m.n, *dqs = dq + s * m.n, *ddqs = ddq + s * m.n;
    for (int k = 0; k < m.L; k++) {
        const int l = m.order[k];
        const int par = m.parent[l];
        double out[FBR_LINK_REC];
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
            double P[FBR_LINK_REC], Sv[6];
            for (int i = 0; i < FBR_LINK_REC; i++) P[i] = r[FBR_LINK_REC * par + i];
            const int d = m.dof[l];
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
            fbr_kin_child(P, rR, rp, ax, d >= 0, qv, dqv, ddqv, out, Sv);
            if (d >= 0)
                for (int i = 0; i < 6; i++) r[FBR_LINK_REC * m.L + FBR_DOF_REC * d + i] = Sv[i];
        }
        for (int i = 0; i < FBR_LINK_REC; i++) r[FBR_LINK_REC * l + i] = out[i];
    }
}

// ------------------------------------------------------------------------------------------------
// K2: materialised standard regressor  Y[s][rows][cols]; one workgroup per sample (grid-stride), one
// thread per column, every row of a sample written as one contiguous, coalesced run of `cols` doubles.
// Bound: HBM write (8*rows*cols bytes per sample).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fbr_regressor_kernel(DevModel m, long S, const double *__restrict__ rec,
                                                             const double *__restrict__ dq,
                                                             const double *__restrict__ sign, double *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *rs = smem;  // [rec]
    const int tid = threadIdx.x;
    for (long s = blockIdx.x; s < S; s += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < m.rec; i += blockDim.x) rs[i] = rec[s * (long)m.rec + i];
        __syncthreads();
        double *Ys = Y + s * (long)m.rows * m.cols;
        for (int c = tid; c < m.cols; c += blockDim.x) {
            const int4 cd = m.coldesc[c];
            if (cd.x == 0) {
                double w6[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * cd.y, cd.z, w6);
                for (int r = 0; r < m.fb; r++) Ys[(long)r * m.cols + c] = w6[r];
                for (int d = 0; d < m.n; d++) {
                    const unsigned bit = (m.ancmask[cd.y * m.nw + (d >> 5)] >> (d & 31)) & 1u;
                    double v = 0.0;
                    if (bit) v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * d, w6);
                    Ys[(long)(m.fb + d) * m.cols + c] = v;
                }
            } else {
                const int j = cd.w;
                const double v = fbr_friction_value(cd.z, dq[s * m.n + j], sign ? sign[s * m.n + j] : 0.0, m.stribeck);
                for (int r = 0; r < m.rows; r++) Ys[(long)r * m.cols + c] = (r == m.fb + j) ? v : 0.0;
            }
        }
    }
}

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
// K5: fused regressor -> Gram.  Workgroup = (part of the tile-pair list, slice of the samples).
// Per sample: all 4 waves build the packed tile image of [Y_s | rhs_s] in LDS (VALU), then every wave
// runs its <= FBR_NPW accumulators over that image with v_mfma_f64_16x16x4_f64.  Two workgroups share
// a CU so one's VALU phase overlaps the other's MFMA phase.  Bound: fp64 MFMA.
// ------------------------------------------------------------------------------------------------
typedef double fbr_d4 __attribute__((ext_vector_type(4)));
#define FBR_NPF 6  // record prefetch registers per thread (covers rec <= 1536 doubles)

__global__ __launch_bounds__(256, 2) void fbr_gram_kernel(DevGram g, DevModel m, long S, int NS,
                                                           const double *__restrict__ rec,
                                                           const double *__restrict__ dq,
                                                           const double *__restrict__ sign,
                                                           const double *__restrict__ rhs,
                                                           const double *__restrict__ wts, double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *img = smem;                                   // [image_doubles]
    double *rs = smem + g.image_doubles;                  // [rec]
    int *rid = (int *)(rs + ((m.rec + 1) & ~1));          // [ntab]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = blockIdx.x % g.T, slice = blockIdx.x / g.T;
    const long s0 = (S * slice) / NS, s1 = (S * (slice + 1)) / NS;

    for (int i = tid; i < g.image_doubles; i += 256) img[i] = 0.0;
    for (int i = tid; i < g.ntab; i += 256) rid[i] = g.rowid[i];

    fbr_d4 acc[FBR_NPW];
#pragma unroll
    for (int p = 0; p < FBR_NPW; p++) acc[p] = (fbr_d4){0.0, 0.0, 0.0, 0.0};

    // accumulator-slot metadata (wave-uniform) lives in LDS behind the row map
    int *mslot = rid + g.ntab;  // [WPB*NPW]
    for (int i = tid; i < FBR_WPB * FBR_NPW; i += 256) mslot[i] = g.slotmeta[(long)part * FBR_WPB * FBR_NPW + i];
    const int *mymeta = mslot + wave * FBR_NPW;
    const int it0 = g.item_begin[part], it1 = g.item_begin[part + 1];
    const int li = lane & 15, kk = lane >> 4;

    double pre[FBR_NPF];
    if (s0 < s1) {
#pragma unroll
        for (int j = 0; j < FBR_NPF; j++) {
            const int i = tid + 256 * j;
            pre[j] = (i < m.rec) ? rec[s0 * (long)m.rec + i] : 0.0;
        }
    }
    for (long s = s0; s < s1; s++) {
        // ---- stage this sample's link records (prefetched), then prefetch the next one
#pragma unroll
        for (int j = 0; j < FBR_NPF; j++) {
            const int i = tid + 256 * j;
            if (i < m.rec) rs[i] = pre[j];
        }
        for (int i = tid + 256 * FBR_NPF; i < m.rec; i += 256) rs[i] = rec[s * (long)m.rec + i];
        __syncthreads();  // records visible; every wave is done reading the previous image
        if (s + 1 < s1) {
#pragma unroll
            for (int j = 0; j < FBR_NPF; j++) {
                const int i = tid + 256 * j;
                pre[j] = (i < m.rec) ? rec[(s + 1) * (long)m.rec + i] : 0.0;
            }
        }
        // ---- producer: one real column per item
        const double *ws = wts ? wts + s * m.rows : nullptr;
        for (int it = it0 + tid; it < it1; it += 256) {
            const int4 d = g.items[it];
            if (d.y == 0) {
                double w6[6];
                fbr_unit_wrench(rs + FBR_LINK_REC * d.z, d.w, w6);
                for (int r = 0; r < m.fb; r++) img[d.x + r * FBR_TILE] = ws ? w6[r] * ws[r] : w6[r];
                const int len = m.pathlen[d.z];
                for (int j = 0; j < len; j++) {
                    const int dd = m.pathtab[d.z * m.maxd + j];
                    double v = fbr_dot6(rs + FBR_LINK_REC * m.L + FBR_DOF_REC * dd, w6);
                    if (ws) v *= ws[m.fb + dd];
                    img[d.x + (m.fb + j) * FBR_TILE] = v;
                }
            } else if (d.y == 1) {
                const int r = m.fb + d.z;
                double v = fbr_friction_value(d.w, dq[s * m.n + d.z], sign ? sign[s * m.n + d.z] : 0.0, m.stribeck);
                if (ws) v *= ws[r];
                img[d.x + r * FBR_TILE] = v;
            } else {
                for (int r = 0; r < m.rows; r++) {
                    double v = rhs[(s * m.rows + r) * g.k + d.z];
                    if (ws) v *= ws[r];
                    img[d.x + r * FBR_TILE] = v;
                }
            }
        }
        __syncthreads();  // image complete
        // ---- MFMA phase
#pragma unroll
        for (int p = 0; p < FBR_NPW; p++) {
            const int mt = __builtin_amdgcn_readfirstlane(mymeta[p]);
            const int common = (mt >> 20) & 0xff;
            const int nk4 = (common + 3) >> 2;
            const int oA = (mt & 0x3ff) << 6, oB = ((mt >> 10) & 0x3ff) << 6;
            const double *pa = img + oA + lane;
            if (!((mt >> 28) & 1)) {
                const double *pb = img + oB + lane;
                for (int ks = 0; ks < nk4; ks++) {
                    double a = pa[64 * ks];
                    const double b = pb[64 * ks];
                    a = (4 * ks + kk < common) ? a : 0.0;
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[p], 0, 0, 0);
                }
            } else {
                const int *pr = rid + (oA >> 4) + kk;
                const double *pb = img + oB + li;
                for (int ks = 0; ks < nk4; ks++) {
                    double a = pa[64 * ks];
                    const int posb = pr[4 * ks];
                    const double b = pb[posb * FBR_TILE];
                    a = (4 * ks + kk < common) ? a : 0.0;
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[p], 0, 0, 0);
                }
            }
        }
    }
    // ---- write this workgroup's accumulators: partial[slice][part][wave][slot][reg][lane]
    double *pp = partial + ((((long)slice * g.T + part) * FBR_WPB + wave) * FBR_NPW) * 256;
#pragma unroll
    for (int p = 0; p < FBR_NPW; p++) {
        pp[p * 256 + 0 * 64 + lane] = acc[p][0];
        pp[p * 256 + 1 * 64 + lane] = acc[p][1];
        pp[p * 256 + 2 * 64 + lane] = acc[p][2];
        pp[p * 256 + 3 * 64 + lane] = acc[p][3];
    }
}

// Deterministic reduction over the sample slices + scatter into the symmetric G (augmented column order).
// One workgroup (256 threads = 4 regs x 64 lanes) per accumulator slot.  G must be pre-zeroed or hold the
// running sum: every G entry is touched by exactly one thread.
__global__ __launch_bounds__(256) void fbr_gram_reduce_kernel(DevGram g, int NS, const double *__restrict__ partial,
                                                               double *__restrict__ G)
{
    const int slot = blockIdx.x;  // (part*WPB + wave)*NPW + p
    const int I = g.slot_tiles[2 * slot], J = g.slot_tiles[2 * slot + 1];
    if (I < 0) return;
    const int t = threadIdx.x, reg = t >> 6, lane = t & 63;
    const long per_slice = (long)g.T * FBR_WPB * FBR_NPW * 256;
    double v = 0.0;
    for (int sl = 0; sl < NS; sl++) v += partial[sl * per_slice + (long)slot * 256 + t];
    const int row = (lane >> 4) + 4 * reg, col = lane & 15;
    const int ci = g.tilecol[I * FBR_TILE + row], cj = g.tilecol[J * FBR_TILE + col];
    if (ci < 0 || cj < 0) return;
    G[(long)ci * g.Pa + cj] += v;
    if (I != J) G[(long)cj * g.Pa + ci] += v;
}
