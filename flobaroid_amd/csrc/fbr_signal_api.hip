// fbr_signal_api.hip -- the step before the path: signal conditioning of measurement channels (fbr_filtfilt / _medfilt / _central_diff).
#include "fbr_internal.h"
#include "fbr_signal.h"

// ------------------------------------------------------------------------------------------------
// signal conditioning (fbr_signal.h)
// ------------------------------------------------------------------------------------------------
// stage a host array X [S][ld] on the device (or use the device pointer); returns the device pointer
static int sig_stage(fbr_model *m, DevBuf &buf, const double *X, size_t count, int mem, double **dst)
{
    if (mem == FBR_DEVICE) {
        *dst = const_cast<double *>(X);
        return FBR_OK;
    }
    int rc = buf.ensure(std::max<size_t>(count, 1) * sizeof(double));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(buf.p, X, count * sizeof(double), hipMemcpyHostToDevice, m->stream));
    *dst = buf.as<double>();
    return FBR_OK;
}

extern "C" int fbr_filtfilt(fbr_model *m, const double *b, const double *a, int32_t ncoef, double *X, int64_t S, int32_t ncols, int32_t ld, int32_t mem)
{
    if (!m || !b || !a || !X || ncoef < 2 || ncoef > FBR_SIG_MAXC || ncols < 1 || ld < ncols || a[0] == 0.0 || (mem != FBR_HOST && mem != FBR_DEVICE)) {
        set_err("fbr_filtfilt: bad arguments (2 <= ncoef <= 12, a[0] != 0, ld >= ncols)");
        return FBR_E_INVALID;
    }
    const int pad = 3 * ncoef, p = ncoef - 1;
    if (S <= pad) {
        set_err("fbr_filtfilt: the signal must be longer than the padding of 3 * ncoef samples");
        return FBR_E_INVALID;
    }
    if (int rc_enter = enter_blocking(m)) return rc_enter;
    FbrIir f;
    memset(&f, 0, sizeof(f));
    f.nc = ncoef;
    for (int i = 0; i < ncoef; i++) {
        f.b[i] = b[i] / a[0];
        f.a[i] = a[i] / a[0];
    }
    {  // scipy.signal.lfilter_zi: steady state of a unit step
        double bs = 0.0, as = 0.0;
        for (int k = 1; k < ncoef; k++) bs += f.b[k] - f.a[k] * f.b[0];
        for (int k = 0; k < ncoef; k++) as += f.a[k];
        f.zi[0] = bs / as;
        double asum = 1.0, csum = 0.0;
        for (int k = 1; k < p; k++) {
            asum += f.a[k];
            csum += f.b[k] - f.a[k] * f.b[0];
            f.zi[k] = asum * f.zi[0] - csum;
        }
    }
    {  // M^LB of z' = M z + g x (y = z_0 + b_0 x): M[i][0] = -a[i+1], M[i][i+1] = 1
        std::vector<double> M((size_t)p * p, 0.0), R((size_t)p * p, 0.0), Tm((size_t)p * p);
        for (int i = 0; i < p; i++) {
            M[(size_t)i * p] = -f.a[i + 1];
            if (i + 1 < p) M[(size_t)i * p + i + 1] += 1.0;
            R[(size_t)i * p + i] = 1.0;
        }
        auto mul = [&](std::vector<double> &A, const std::vector<double> &B) {  // A = A * B
            for (int i = 0; i < p; i++)
                for (int j = 0; j < p; j++) {
                    double acc = 0.0;
                    for (int k = 0; k < p; k++) acc += A[(size_t)i * p + k] * B[(size_t)k * p + j];
                    Tm[(size_t)i * p + j] = acc;
                }
            A = Tm;
        };
        for (long e = FBR_SIG_LB; e > 0; e >>= 1) {
            if (e & 1) mul(R, M);
            std::vector<double> M2 = M;
            mul(M2, M);
            M = M2;
        }
        for (int i = 0; i < p * p; i++) f.Mp[i] = R[i];
    }
    int rc;
    double *dX = nullptr;
    const size_t xcount = (size_t)(S - 1) * ld + ncols;
    if ((rc = sig_stage(m, m->out_tmp, X, xcount, mem, &dX))) return rc;
    const long Le = S + 2L * pad, nblk = (Le + FBR_SIG_LB - 1) / FBR_SIG_LB;
    const size_t zcount = (size_t)nblk * ncols * (FBR_SIG_MAXC - 1);
    if ((rc = m->st_aux.ensure((size_t)Le * ncols * sizeof(double))) || (rc = m->st_aux2.ensure(2 * zcount * sizeof(double)))) return rc;
    double *Y1 = m->st_aux.as<double>(), *zs = m->st_aux2.as<double>(), *zst = zs + zcount;
    const unsigned grid = (unsigned)((nblk * ncols + 255) / 256);
    for (int dir = 0; dir < 2; dir++) {
        hipLaunchKernelGGL(fbr_sig_iir_kernel, dim3(grid), dim3(256), 0, m->stream, f, 0, dir, dX, (long)S, (int)ncols, (long)ld, pad, Y1, zs, (const double *)zst, nblk);
        hipLaunchKernelGGL(fbr_sig_chain_kernel, dim3((ncols + 63) / 64), dim3(64), 0, m->stream, f, dir, (const double *)dX, (long)S, (int)ncols, (long)ld, pad,
                           (const double *)Y1, (const double *)zs, zst, nblk);
        hipLaunchKernelGGL(fbr_sig_iir_kernel, dim3(grid), dim3(256), 0, m->stream, f, 1, dir, dX, (long)S, (int)ncols, (long)ld, pad, Y1, zs, (const double *)zst, nblk);
        HIPCHK(hipGetLastError());
    }
    return finish_output(m, dX, X, xcount, mem);
}

extern "C" int fbr_medfilt(fbr_model *m, int32_t k, double *X, int64_t S, int32_t ncols, int32_t ld, int32_t mem)
{
    if (!m || !X || k < 1 || k > FBR_SIG_MAXK || (k & 1) == 0 || S < 1 || ncols < 1 || ld < ncols || (mem != FBR_HOST && mem != FBR_DEVICE)) {
        set_err("fbr_medfilt: bad arguments (k odd, 1 <= k <= 31, ld >= ncols)");
        return FBR_E_INVALID;
    }
    if (int rc_enter = enter_blocking(m)) return rc_enter;
    int rc;
    double *dX = nullptr;
    const size_t xcount = (size_t)(S - 1) * ld + ncols;
    if ((rc = sig_stage(m, m->out_tmp, X, xcount, mem, &dX))) return rc;
    if ((rc = m->st_aux.ensure((size_t)S * ncols * sizeof(double)))) return rc;
    const unsigned grid = (unsigned)(((size_t)S * ncols + 255) / 256);
    hipLaunchKernelGGL(fbr_sig_gather_kernel, dim3(grid), dim3(256), 0, m->stream, (const double *)dX, (long)S, (int)ncols, (long)ld, m->st_aux.as<double>());
    hipLaunchKernelGGL(fbr_sig_median_kernel, dim3(grid), dim3(256), 0, m->stream, (int)k, (const double *)m->st_aux.as<double>(), dX, (long)S, (int)ncols, (long)ld);
    HIPCHK(hipGetLastError());
    return finish_output(m, dX, X, xcount, mem);
}

extern "C" int fbr_central_diff(fbr_model *m, const double *A, const double *T, double *D, int64_t S, int32_t ncols, int32_t mem)
{
    if (!m || !A || !T || !D || S < 5 || ncols < 1 || (mem != FBR_HOST && mem != FBR_DEVICE)) {
        set_err("fbr_central_diff: bad arguments (S >= 5)");
        return FBR_E_INVALID;
    }
    if (int rc_enter = enter_blocking(m)) return rc_enter;
    int rc;
    double *dA = nullptr, *dT = nullptr, *dD = D;
    if ((rc = sig_stage(m, m->st_aux, A, (size_t)S * ncols, mem, &dA)) || (rc = sig_stage(m, m->st_aux2, T, (size_t)S, mem, &dT))) return rc;
    if (mem == FBR_HOST) {
        if ((rc = m->out_tmp.ensure((size_t)S * ncols * sizeof(double)))) return rc;
        dD = m->out_tmp.as<double>();
    }
    hipLaunchKernelGGL(fbr_sig_cdiff_kernel, dim3((unsigned)(((size_t)S * ncols + 255) / 256)), dim3(256), 0, m->stream, (const double *)dA, (const double *)dT, dD, (long)S, (int)ncols);
    HIPCHK(hipGetLastError());
    return finish_output(m, dD, D, (size_t)S * ncols, mem);
}
