// fbr_tsqr.h -- blocked Householder TSQR fold (placeholder interface; implementation follows).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct FbrTsqrWork {
    void release() {}
};

static thread_local std::string g_tsqr_err;
static inline const char *fbr_tsqr_error() { return g_tsqr_err.c_str(); }

static inline long fbr_tsqr_chunk_samples(int rows, int Pa)
{
    (void)Pa;
    return std::max(1L, (long)(1 << 20) / std::max(rows, 1));
}

static inline int fbr_tsqr_fold(FbrTsqrWork &, hipStream_t, long, int, const double *, int, const double *, const double *,
                                double *, int)
{
    g_tsqr_err = "TSQR not built into this libfbr yet";
    return -4;
}
