// Where the time of the kinematics kernel goes: the product's fbr_kin_kernel and timing-only variants of it on a
// synthetic 30-body tree (the size of the merged WALK-MAN), 1 M samples.  Variant bits: 1 = no record stores (a checksum instead),
// 2 = joint states from the lane index instead of memory, 4 = sin / cos replaced by two multiplications, 8 = model constants of a link
// read once into registers for ALL lanes by one vector load each instead of scalar loads, 16 = no parent re-read at branch points.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DFBR_KERNELS_CORE -o tools/_build/kin_probe tools/kin_probe.hip && tools/_build/kin_probe
#include "../flobaroid_amd/csrc/fbr_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V, int WAVES> __global__ __launch_bounds__(256, WAVES) void kin_variant(DevModel m, long S, const double *__restrict__ q,
                                                                                       const double *__restrict__ dq, const double *__restrict__ ddq,
                                                                                       const double *__restrict__ bv, const double *__restrict__ ba,
                                                                                       const double *__restrict__ rpy, double *rec, double *chk)
{
    __shared__ double stg[(V & 128) ? 4 * 64 * FBR_LINK_REC : 1];
    const long s0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = (V & 128) ? (s0 < S ? s0 : S - 1) : s0;
    if (s0 >= S && !(V & 128)) return;
    const int lane = threadIdx.x & 63;
    double *sg = stg + ((V & 128) ? (threadIdx.x >> 6) * 64 * FBR_LINK_REC : 0);
    const long wbase = s0 - lane;
    if ((V & 128) && wbase >= S) return;
    double *r = rec + s * (long)m.rec;
    const double *qs = q + s * m.n, *dqs = dq + s * m.n, *ddqs = ddq + s * m.n;
    double P[FBR_LINK_REC], acc = 0.0;
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
            if (!(V & 16) && par != prev_l)
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
                if (V & 2) {
                    qv = 1e-3 * (double)(threadIdx.x + d);
                    dqv = qv * 0.5;
                    ddqv = qv * 0.25;
                } else {
                    if (V & 64) {  // states stored joint-major (coalesced)
                        qv = q[(long)d * S + s];
                        dqv = dq[(long)d * S + s];
                        ddqv = ddq[(long)d * S + s];
                    } else {
                        qv = qs[d];
                        dqv = dqs[d];
                        ddqv = ddqs[d];
                    }
                }
            }
            if (V & 4) {
                // fbr_kin_child with cos / sin replaced (timing only)
                double Rq[9], Rj[9];
                const double c = 1.0 - 0.5 * qv * qv, sn = qv * (1.0 - qv * qv * (1.0 / 6.0)), v = 1.0 - c;
                const double *sx = ax;
                Rq[0] = c + sx[0] * sx[0] * v; Rq[1] = sx[0] * sx[1] * v - sx[2] * sn; Rq[2] = sx[0] * sx[2] * v + sx[1] * sn;
                Rq[3] = sx[1] * sx[0] * v + sx[2] * sn; Rq[4] = c + sx[1] * sx[1] * v; Rq[5] = sx[1] * sx[2] * v - sx[0] * sn;
                Rq[6] = sx[2] * sx[0] * v - sx[1] * sn; Rq[7] = sx[2] * sx[1] * v + sx[0] * sn; Rq[8] = c + sx[2] * sx[2] * v;
                fbr_mm(rR, Rq, Rj);
                fbr_kin_child(P, Rj, rp, ax, 0, 0.0, 0.0, 0.0, out, Sv);  // (fixed-joint arithmetic on the rotated rest frame: same transform work)
                for (int i = 0; i < 3; i++) {
                    out[FBR_OFF_W + i] += ax[i] * dqv;
                    out[FBR_OFF_DW + i] += ax[i] * ddqv;
                }
                fbr_mv(out + FBR_OFF_R, ax, Sv + 3);
                fbr_cross(out + FBR_OFF_P, Sv + 3, Sv);
            } else {
                fbr_kin_child(P, rR, rp, ax, m.jtype[l], qv, dqv, ddqv, out, Sv);
            }
        }
        if (V & 1) {
            for (int i = 0; i < 6; i++) acc += Sv[i];
            for (int i = 0; i < FBR_LINK_REC; i++) acc += out[i];
        } else {
            if (V & 128) {  // the link's records of the wave's 64 samples through the LDS: each store instruction writes 64 consecutive
                            // elements of the [64][21] block = three runs of 168 contiguous bytes
                if (d >= 0 && s0 < S)
                    for (int i = 0; i < 6; i++) r[FBR_LINK_REC * m.L + FBR_DOF_REC * d + i] = Sv[i];
                for (int i = 0; i < FBR_LINK_REC; i++) sg[lane * FBR_LINK_REC + i] = out[i];
                asm volatile("" ::: "memory");
                int smp = lane / FBR_LINK_REC, fld = lane - smp * FBR_LINK_REC;
                double *gp = rec + (wbase + smp) * (long)m.rec + FBR_LINK_REC * l + fld;
#pragma unroll
                for (int i = 0; i < FBR_LINK_REC; i++) {
                    const double v = sg[i * 64 + lane];
                    if (wbase + smp < S) *gp = v;
                    smp += 64 / FBR_LINK_REC;
                    fld += 64 % FBR_LINK_REC;
                    gp += (64 / FBR_LINK_REC) * (long)m.rec + 64 % FBR_LINK_REC;
                    if (fld >= FBR_LINK_REC) {
                        fld -= FBR_LINK_REC;
                        smp += 1;
                        gp += (long)m.rec - FBR_LINK_REC;
                    }
                }
                asm volatile("" ::: "memory");
            } else if (V & 32) {  // records interleaved over the 64 samples of a wave: every store instruction writes 512 contiguous bytes
                constexpr long IL = (V >> 8) ? (V >> 8) : 64;  // interleave factor (bits 8..: 2, 4, 8, 16, 32; default 64)
                double *ri = rec + (s & ~(IL - 1)) * (long)m.rec + (s & (IL - 1));
                if (d >= 0)
                    for (int i = 0; i < 6; i++) ri[(FBR_LINK_REC * m.L + FBR_DOF_REC * d + i) * IL] = Sv[i];
                for (int i = 0; i < FBR_LINK_REC; i++) ri[(FBR_LINK_REC * l + i) * IL] = out[i];
            } else {
                if (d >= 0)
                    for (int i = 0; i < 6; i++) r[FBR_LINK_REC * m.L + FBR_DOF_REC * d + i] = Sv[i];
                for (int i = 0; i < FBR_LINK_REC; i++) r[FBR_LINK_REC * l + i] = out[i];
            }
        }
        for (int i = 0; i < FBR_LINK_REC; i++) P[i] = out[i];
        prev_l = l;
    }
    if (V & 1) chk[s] = acc;
}

// T2: links in index order (parents before children), the record of a sample as ONE stream: the link records are appended to a per-sample
// carry in the LDS and leave as whole 128-byte lines (every store instruction writes four whole lines); what is left of a line stays in the
// LDS for the next link.  The motion vectors S (6 per joint) are a second stream of the same kind.  PF: the joint states of the next link
// are requested before the stores of this one.  soff = offset of the S area (a multiple of 16 doubles), recsz a multiple of 16 too.
template <int PF> __global__ __launch_bounds__(64) void kin_stream(DevModel m, long S, int soff, int recsz, const double *__restrict__ q, const double *__restrict__ dq,
                                                                    const double *__restrict__ ddq, const double *__restrict__ bv, const double *__restrict__ ba,
                                                                    const double *__restrict__ rpy, double *rec)
{
    constexpr int LR = 37, SR = 23;
    __shared__ double ringl[64 * LR], rings[64 * SR];
    const int lane = threadIdx.x;
    const long wbase = (long)blockIdx.x * 64;
    const long s = min(wbase + lane, S - 1);
    const double *qs = q + s * m.n, *dqs = dq + s * m.n, *ddqs = ddq + s * m.n;
    double *rl = ringl + lane * LR, *rsg = rings + lane * SR;
    const int fs = lane >> 4, ff = lane & 15;  // flush: sample 4 i + fs, field ff of the line
    double P[FBR_LINK_REC];
    int cl = 0, cs = 0;        // carried doubles of the two streams (wave-uniform)
    long pl = 0, ps = soff;    // stream positions already written to memory (doubles, multiples of 16)
    constexpr int W = 8, WL = W + 1;
    __shared__ double win[PF == 2 ? 3 * 64 * WL : 1];
    extern __shared__ double dyn_cap[];  // (occupancy cap only)
    int w0 = -1000;
    double qn = 0, dqn = 0, ddqn = 0;
    if (PF == 1 && m.L > 1 && m.dof[1] >= 0) {
        qn = qs[m.dof[1]];
        dqn = dqs[m.dof[1]];
        ddqn = ddqs[m.dof[1]];
    }
    for (int l = 0; l < m.L; l++) {
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
            if (par != l - 1)
                for (int i = 0; i < FBR_LINK_REC; i++) P[i] = rec[s * (long)recsz + FBR_LINK_REC * par + i];
            d = m.dof[l];
            double rR[9], rp[3], ax[3];
            for (int i = 0; i < 9; i++) rR[i] = m.restR[9 * l + i];
            for (int i = 0; i < 3; i++) {
                rp[i] = m.restp[3 * l + i];
                ax[i] = m.axis[3 * l + i];
            }
            double qv = 0, dqv = 0, ddqv = 0;
            if (d >= 0) {
                if (PF == 2) {
                    if (d < w0 || d >= w0 + W) {  // (wave-uniform) the next W joints of the wave's 64 samples: 64-byte runs
                        w0 = d - d % W;
                        asm volatile("" ::: "memory");
                        const int sm0 = lane >> 3, kk = lane & 7;
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int sm = 8 * i + sm0;
                            const long g = min(wbase + sm, S - 1) * m.n + min(w0 + kk, m.n - 1);
                            win[(0 * 64 + sm) * WL + kk] = q[g];
                            win[(1 * 64 + sm) * WL + kk] = dq[g];
                            win[(2 * 64 + sm) * WL + kk] = ddq[g];
                        }
                        asm volatile("" ::: "memory");
                    }
                    qv = win[(0 * 64 + lane) * WL + d - w0];
                    dqv = win[(1 * 64 + lane) * WL + d - w0];
                    ddqv = win[(2 * 64 + lane) * WL + d - w0];
                } else if (PF) {
                    qv = qn; dqv = dqn; ddqv = ddqn;
                } else {
                    qv = qs[d]; dqv = dqs[d]; ddqv = ddqs[d];
                }
            }
            fbr_kin_child(P, rR, rp, ax, m.jtype[l], qv, dqv, ddqv, out, Sv);
        }
        if (PF == 1 && l + 1 < m.L) {
            const int dn = m.dof[l + 1];
            if (dn >= 0) {
                qn = qs[dn]; dqn = dqs[dn]; ddqn = ddqs[dn];
            }
        }
        // ---- link stream
        for (int i = 0; i < FBR_LINK_REC; i++) rl[cl + i] = out[i];
        asm volatile("" ::: "memory");
        {
            const int tot = cl + FBR_LINK_REC, nl = (l + 1 == m.L) ? (tot + 15) >> 4 : tot >> 4;  // (the last link flushes its partial line: the pad is this sample's)
            for (int ln = 0; ln < nl; ln++) {
                double *gp = rec + (wbase + fs) * (long)recsz + pl + ff;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int sm = 4 * i + fs;
                    const double v = ringl[sm * LR + 16 * ln + ff];
                    if (wbase + sm < S) gp[(long)(4 * i) * recsz] = v;
                }
                pl += 16;
            }
            asm volatile("" ::: "memory");
            const int sh = 16 * nl;
            for (int i = 0; i < FBR_LINK_REC; i++)
                if (cl + i - sh >= 0) rl[cl + i - sh] = out[i];
            cl = tot - sh > 0 ? tot - sh : 0;
        }
        // ---- motion-vector stream
        if (d >= 0) {
            for (int i = 0; i < 6; i++) rsg[cs + i] = Sv[i];
            cs += 6;
        }
        asm volatile("" ::: "memory");
        if (cs >= 16 || (l + 1 == m.L && cs > 0)) {
            double *gp = rec + (wbase + fs) * (long)recsz + ps + ff;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int sm = 4 * i + fs;
                const double v = rings[sm * SR + ff];
                if (wbase + sm < S) gp[(long)(4 * i) * recsz] = v;
            }
            ps += 16;
            asm volatile("" ::: "memory");
            const int rem = cs - 16;
            for (int i = 0; i < 6; i++)
                if (i < rem) rsg[i] = rsg[16 + i];
            cs = rem > 0 ? rem : 0;
        }
        for (int i = 0; i < FBR_LINK_REC; i++) P[i] = out[i];
    }
}

// The consumer side of an interleaved record layout: what every per-sample consumer (the tile-image packer, the regressor writers) does
// first -- one workgroup per sample copies the sample's record into the LDS -- (0) from contiguous records, (1) from records interleaved
// over 16 samples ([s / 16][field][s % 16]: 8 bytes per 128-byte line, the sixteen consumers of a tile spread over all XCDs by their
// block index), (2) the same with the sixteen samples of a tile dealt to workgroups of ONE XCD (block index mod 8), so that the tile's
// lines are fetched into one L2 only.
template <int MODE> __global__ __launch_bounds__(256) void consume_kernel(long S, int recsz, const double *__restrict__ rec, double *__restrict__ out)
{
    extern __shared__ double rs[];
    const int tid = threadIdx.x;
    double acc = 0.0;
    const long nb = gridDim.x;
    for (long k = 0;; k++) {
        long s;
        if (MODE == 2) {
            const long xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) + k * (nb >> 3);
            s = ((slot >> 4) * 8 + xcd) * 16 + (slot & 15);
        } else {
            s = blockIdx.x + k * nb;
        }
        if (s >= S) break;
        __syncthreads();
        for (int i = tid; i < recsz; i += 256)
            rs[i] = MODE == 0 ? rec[s * (long)recsz + i] : rec[(s & ~15L) * (long)recsz + (long)i * 16 + (s & 15)];
        __syncthreads();
        acc += rs[(tid * 3) % recsz] + rs[(tid * 7 + 1) % recsz];
    }
    out[(long)blockIdx.x * 256 + tid] = acc;
}

template <typename T> static T *up(const std::vector<T> &v)
{
    T *d;
    hipMalloc(&d, v.size() * sizeof(T));
    hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

static int g_dyn_lds = 0;  // dynamic LDS per workgroup: caps the workgroups per CU (160 KB per CU)
template <int V, int WAVES> static void run(const char *name, DevModel m, long S, double **st, double *rec, double *chk)
{
    hipFuncSetAttribute((const void *)kin_variant<V, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kin_variant<V, WAVES>, 256, g_dyn_lds);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((kin_variant<V, WAVES>), dim3((unsigned)((S + 255) / 256)), dim3(256), g_dyn_lds, 0, m, S, st[0], st[1], st[2], st[3], st[4], st[5], rec, chk);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it) best = ms < best ? ms : best;
    }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)kin_variant<V, WAVES>);
    printf("%-58s %7.3f ms   (%d VGPRs, %d B scratch, %d workgroups of 256 per CU)\n", name, best, fa.numRegs, (int)fa.localSizeBytes, occ);
}

int main()
{
    const int L = 30, n = 29;
    const long S = 1000000;
    srand(3);
    auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
    std::vector<int> order(L), parent(L), dof(L), jtype(L);
    std::vector<double> restR(9 * L, 0.0), restp(3 * L), axis(3 * L);
    // trunk of 3, two arms of 7, two legs of 6 (DFS order = index order)
    int idx = 0;
    auto add = [&](int par) {
        parent[idx] = par; order[idx] = idx; dof[idx] = idx - 1; jtype[idx] = idx ? 1 : 0;
        const double a = rnd(), b = rnd(), c = rnd();  // some rotation: Rz(a) Ry(b) Rx(c)
        const double ca = cos(a), sa = sin(a), cb = cos(b), sb = sin(b), cc = cos(c), sc = sin(c);
        double R[9] = {ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc, sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc, -sb, cb * sc, cb * cc};
        for (int i = 0; i < 9; i++) restR[9 * idx + i] = R[i];
        double ax[3] = {rnd(), rnd(), rnd()}, nn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        for (int i = 0; i < 3; i++) {
            restp[3 * idx + i] = 0.3 * rnd();
            axis[3 * idx + i] = ax[i] / nn;
        }
        return idx++;
    };
    int base = add(-1), t = base;
    for (int i = 0; i < 3; i++) t = add(t);
    for (int arm = 0; arm < 2; arm++) { int p = t; for (int i = 0; i < 7; i++) p = add(p); }
    for (int leg = 0; leg < 2; leg++) { int p = base; for (int i = 0; i < 6; i++) p = add(p); }
    if (idx != L) { printf("tree size %d\n", idx); return 1; }
    DevModel m = {};
    m.L = L; m.n = n; m.floating = 1; m.rec = FBR_LINK_REC * L + FBR_DOF_REC * n;
    m.g[0] = 0; m.g[1] = 0; m.g[2] = -9.81;
    m.order = up(order); m.parent = up(parent); m.dof = up(dof); m.jtype = up(jtype);
    m.restR = up(restR); m.restp = up(restp); m.axis = up(axis);
    std::vector<double> hq((size_t)S * n);
    double *st[6];
    for (int a = 0; a < 3; a++) {
        for (auto &x : hq) x = 6.28 * rnd();
        st[a] = up(hq);
    }
    std::vector<double> hb((size_t)S * 6);
    for (int a = 3; a < 6; a++) {
        for (auto &x : hb) x = rnd();
        st[a] = up(hb);
    }
    double *rec, *chk;
    hipMalloc(&rec, (size_t)S * m.rec * 8);
    hipMalloc(&chk, (size_t)S * 8);
    printf("kinematics of %ld samples, %d bodies, record %d doubles\n", S, L, m.rec);
    run<0, 2>("product kernel body (no register cap)", m, S, st, rec, chk);
    run<0, 5>("  capped at 96 VGPRs", m, S, st, rec, chk);
    run<1, 2>("no record stores", m, S, st, rec, chk);
    run<2, 2>("joint states not loaded", m, S, st, rec, chk);
    run<3, 2>("no stores, no state loads", m, S, st, rec, chk);
    run<4, 2>("sin / cos replaced by two multiplications", m, S, st, rec, chk);
    run<7, 2>("no stores, no state loads, no sin / cos", m, S, st, rec, chk);
    run<16, 2>("no parent re-read at branch points", m, S, st, rec, chk);
    run<23, 2>("none of the four", m, S, st, rec, chk);
    run<128, 2>("link records through an LDS transposition (168-byte runs)", m, S, st, rec, chk);
    run<128 + 16, 2>("  ... no parent re-read", m, S, st, rec, chk);
    run<128 + 2, 2>("  ... joint states not loaded", m, S, st, rec, chk);
    for (int wg = 1; wg <= 2; wg++) {
        g_dyn_lds = wg == 1 ? 100 * 1024 : 60 * 1024;
        printf("at most %d workgroup(s) per CU:\n", wg);
        run<0, 2>("  product kernel body", m, S, st, rec, chk);
        run<128, 2>("  LDS transposition (168-byte runs)", m, S, st, rec, chk);
        run<32 + 16, 2>("  interleaved by 64 samples", m, S, st, rec, chk);
    }
    g_dyn_lds = 0;
    {
        const int soff = (FBR_LINK_REC * L + 15) & ~15, recsz = (soff + FBR_DOF_REC * n + 15) & ~15;
        double *rec2;
        hipMalloc(&rec2, (size_t)S * recsz * 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipFuncSetAttribute((const void *)kin_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int caps[5] = {0, 0, 0, 30 * 1024, 10 * 1024};  // dynamic LDS on top of the static 44 KB: 3, then 2, then 1 wave(s) per CU ... roughly
        for (int pf = 0; pf < 5; pf++) {
            float best = 1e9f;
            for (int it = 0; it < 4; it++) {
                hipEventRecord(a, 0);
                if (pf >= 2)
                    hipLaunchKernelGGL(kin_stream<2>, dim3((unsigned)((S + 63) / 64)), dim3(64), caps[pf], 0, m, S, soff, recsz, st[0], st[1], st[2], st[3], st[4], st[5], rec2);
                else if (pf)
                    hipLaunchKernelGGL(kin_stream<1>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, 0, m, S, soff, recsz, st[0], st[1], st[2], st[3], st[4], st[5], rec2);
                else
                    hipLaunchKernelGGL(kin_stream<0>, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, 0, m, S, soff, recsz, st[0], st[1], st[2], st[3], st[4], st[5], rec2);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (it) best = ms < best ? ms : best;
            }
            printf("whole-line streams through an LDS carry (record %d doubles), variant %d (0 plain, 1 prefetched states, 2.. states through an LDS window, +%d KB LDS): %7.3f ms\n", recsz, pf, caps[pf] / 1024, best);
        }
        // check against the plain kernel's records
        hipLaunchKernelGGL((kin_variant<0, 2>), dim3((unsigned)((S + 255) / 256)), dim3(256), 0, 0, m, S, st[0], st[1], st[2], st[3], st[4], st[5], rec, chk);
        std::vector<double> h1((size_t)4096 * m.rec), h2((size_t)4096 * recsz);
        const long off = S - 4096;
        hipMemcpy(h1.data(), rec + off * m.rec, h1.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), rec2 + off * recsz, h2.size() * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (long sm = 0; sm < 4096; sm++) {
            for (int i = 0; i < FBR_LINK_REC * L; i++) bad += h1[sm * m.rec + i] != h2[sm * recsz + i];
            for (int i = 0; i < FBR_DOF_REC * n; i++) bad += h1[sm * m.rec + FBR_LINK_REC * L + i] != h2[sm * recsz + soff + i];
        }
        printf("  records of the last 4096 samples differing from the plain kernel's: %ld\n", bad);
    }
    run<32 + 16, 2>("records interleaved by 64 samples (coalesced stores)", m, S, st, rec, chk);
    run<32 + 16 + (32 << 8), 2>("  interleaved by 32 samples (256-byte runs)", m, S, st, rec, chk);
    run<32 + 16 + (16 << 8), 2>("  interleaved by 16 samples (whole 128-byte lines)", m, S, st, rec, chk);
    run<32 + 16 + (8 << 8), 2>("  interleaved by 8 samples (64-byte runs)", m, S, st, rec, chk);
    run<32 + 16 + (4 << 8), 2>("  interleaved by 4 samples", m, S, st, rec, chk);
    run<32 + 16 + (2 << 8), 2>("  interleaved by 2 samples", m, S, st, rec, chk);
    run<64, 2>("states joint-major (coalesced loads)", m, S, st, rec, chk);
    run<32 + 16 + 64, 2>("both", m, S, st, rec, chk);
    run<32 + 16 + 64, 3>("both, 3 workgroups per CU asked for", m, S, st, rec, chk);
    {
        const int grid = 256 * 24;
        double *out;
        hipMalloc(&out, (size_t)grid * 256 * 8);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        const char *names[3] = {"contiguous records", "interleaved by 16 samples, consumers dealt by block index", "interleaved by 16 samples, a tile's consumers on one XCD"};
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9f;
            for (int it = 0; it < 4; it++) {
                hipEventRecord(a, 0);
                if (mode == 0) hipLaunchKernelGGL(consume_kernel<0>, dim3(grid), dim3(256), m.rec * 8, 0, S, m.rec, rec, out);
                if (mode == 1) hipLaunchKernelGGL(consume_kernel<1>, dim3(grid), dim3(256), m.rec * 8, 0, S, m.rec, rec, out);
                if (mode == 2) hipLaunchKernelGGL(consume_kernel<2>, dim3(grid), dim3(256), m.rec * 8, 0, S, m.rec, rec, out);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (it) best = ms < best ? ms : best;
            }
            printf("consumer-side stage copy of 1 M records (one workgroup per sample), %-58s %7.3f ms\n", names[mode], best);
        }
    }
    return 0;
}
