// fbr_tsqr_work.h -- host-side state of one TSQR factorisation (working factors, chunk buffer, progress counters): what the model
// handle holds.  The kernels and the functions that drive them are in fbr_tsqr.h (included by fbr_tsqr_api.hip only).
#pragma once
#include <hip/hip_runtime.h>

#define FBR_TSQR_THREADS 512
#define FBR_TSQR_WAVES (FBR_TSQR_THREADS / 64)

// what the model's options (fbr_options.h) say about the kernels of a factorisation
struct FbrTsqrOpts {
    bool narrow = true;        // wave-private kernels for <= 128 columns
    bool tree_one_wg = false;  // merges by one workgroup (bit-identical to the cross-workgroup pipeline, slower)
    bool timing = false;       // diagnostic: cycle counters of the wide level-0 kernel
    bool short_calls = true;   // fewer private factors for calls too short to amortise the merge tree over them
    bool narrow_tall = true;   // 96-row level-0 blocks at one wave per SIMD for long calls over <= 6 column tiles
};

struct FbrTsqrWork {
    FbrTsqrOpts opts;
    double *Rw = nullptr;   // [NW][n][ld]
    double *A = nullptr;    // packed chunk [Mpad][n]
    unsigned *err = nullptr;  // device word: set when a wave gave up waiting on a pipeline flag
    bool own_err = true;      // false: the word belongs to the caller (one per model, cleared once per call and read once at its end)
    size_t rw_bytes = 0, a_bytes = 0;
    int *prog = nullptr;      // progress counters of the cross-workgroup merge pipeline (fbr_tsqr_tree_x_kernel)
    size_t prog_bytes = 0;
    long clean_key = -1;  // (Pa, n) for which the padding columns [Pa, n) of the whole chunk buffer are zero and stay zero (writers that
                          // fill the chunk in place never touch them): the per-chunk tail pass then only clears the rows M..Mpad
    int n = 0, ld = 0, NW = 0, Pa = 0, mb = 0, tpw = 0, sub = 0, waves = FBR_TSQR_WAVES, ttpw = 0;
    bool active = false, narrow = false;
    void release()
    {
        if (Rw) (void)hipFree(Rw);
        if (A) (void)hipFree(A);
        if (prog) (void)hipFree(prog);
        prog = nullptr;
        prog_bytes = 0;
        if (err && own_err) (void)hipFree(err);
        err = nullptr;
        own_err = true;
        Rw = A = nullptr;
        rw_bytes = a_bytes = 0;
        active = false;
    }
};

