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
    {  // M^LB of z' = M z + g x (y = z_0 + b_0 x; M[i][0] = -a[i+1], M[i][i+1] = 1): column j = the zero-input recurrence run LB steps
       // from e_j, i.e. the filter's own arithmetic.  NOT by repeated squaring: the companion matrix of a low cut-off is far from normal
       // (poles clustered at z = 1: |M^k| grows like k^(p-1) before it decays), the rounding errors of the squarings are relative to
       // those intermediate norms and do not decay with them -- order 5 at 0.016 of Nyquist (filterLowPass1 = [8 Hz, 5] on a 1 kHz
       // recording) gave an M^LB with entries far above 1 and NaN / 1e190 outputs from the second block on (found by the randomised
       // sweep); in the recurrence the errors made at the peak are carried by the same decaying dynamics.
        for (int j = 0; j < p; j++) {
            std::vector<double> z((size_t)p, 0.0);
            z[(size_t)j] = 1.0;
            for (long t = 0; t < FBR_SIG_LB; t++) {
                const double y = z[0];
                for (int i = 0; i + 1 < p; i++) z[(size_t)i] = z[(size_t)i + 1] - y * f.a[i + 1];
                z[(size_t)p - 1] = -y * f.a[p];
            }
            for (int i = 0; i < p; i++) f.Mp[i * p + j] = z[(size_t)i];
        }
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
