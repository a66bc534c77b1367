// fbr_options.h -- per-model options of libfbr, set and read through the C-ABI (fbr_model_set_option / fbr_model_get_option,
// include/fbr.h lists the keys).  Host only.  Nothing in the library reads the process environment: what a call does is decided by its
// arguments and by the options of its model handle.
#pragma once
#include <cstring>

struct FbrOptions {
    // ---- column reductions: which reduced model a Gram / TSQR call runs on
    double link_merge = 1;                  // 0: every reduction runs on all columns of the model itself
    double regroup = 1;                     // 0: fixed links merged only (no revolute regrouping)
    double reduce_min_work = 1e9;           // a Gram call takes the reductions when S (P - P_red) P >= this (0: always)
    double reduce_grouped_min_samples = 512;  // fbr_gram_grouped: reductions for groups of at least this many samples
    // ---- chunking of the passes
    double chunk_samples = 0;               // > 0: samples per chunk (tests force the multi-chunk paths at small sizes); 0: by memory
    double min_chunks = 4;                  // a short fused pass is still cut into this many chunks (the first producer launch is not hidden)
    double h2d_chunked = 1;                 // pinned host inputs are staged chunk by chunk on a copy stream
    // ---- per-sample entry points
    double fused_id = 1;                    // fbr_predict / fbr_inverse_dynamics_batch: kinematics + torques in one kernel, no records in HBM (0: two kernels)
    // ---- fused Gram program
    double gram_lane_waves = 8;             // gram_lane, models of the one-workgroup-per-CU shape: 8 waves of 18 accumulators (16: 16 waves of 10; measured slower)
    double gram_force_tiles = 1;            // gram_lane: the force rows of the base wrench run on tiles of the columns that have a force (fbr_gram64.h)
    double gram_lane = 1;                   // fused Gram over sample-contiguous images with the one-lane-per-sample producer where the model allows (fbr_gram64.h)
    double gram_shape = 0;                  // 0: by model, 1: one workgroup per CU (18 accumulators), 2: two per CU (10)
    double gram_rhs_tile = 0;               // 1: dense rhs tiles even for k <= 2 (default: tau's products come from the pack kernel)
    double gram_orient = 1;                 // pairs turned so that the row segments fill up
    double gram_serial = 0;                 // diagnostic: producer on the main stream (no overlap with the Gram kernel)
    double gram_timing = 0;                 // diagnostic: s_memtime phases of the Gram kernel printed per launch (results still valid)
    // ---- TSQR
    double tsqr_groups = 1;                 // rows grouped along the kinematic tree
    double tsqr_group_min_samples = 24000;  // ... from this many samples on
    double tsqr_reorder = 1;                // columns factorised in link-depth order (single factorisation path)
    double tsqr_tree_one_wg = 0;            // merges by one workgroup instead of pipelined across workgroups (bit-identical, slower)
    double tsqr_narrow = 1;                 // wave-private kernels for <= 128 columns
    double tsqr_narrow_tall = 1;            // ... with 96-row level-0 blocks at one wave per SIMD for long calls over <= 6 column tiles (0: 48-row blocks, two waves)
    double tsqr_force_group = 1;            // row-group TSQR: the force rows of a floating base as a group of their own over the columns that have a force
    double tsqr_lane_writer = 1;            // regressor writer of the TSQR: one lane per sample, kinematics fused, column-major chunks (0: kinematics kernel + workgroup-per-sample writers)
    double tsqr_writer = 0;                 // grouped regressor writer: 0 by work-item count; 8 / 16: store width forced; 32: rows staged in the LDS
    double tsqr_side_trees_beside = 0;      // the side groups' merge trees start beside the main group's last fold (1) or behind it (0)
    double tsqr_prologue_overlap = 1;       // a submission's kinematics / first writer beside the trees of the one before
    double tsqr_timing = 0;                 // diagnostic: per-phase cycle counters of the wide level-0 kernel
    double tsqr_short_call_factors = 1;     // fewer private factors (shallower merge trees) for calls too short to amortise them
};

struct FbrOptionKey {
    const char *name;
    double FbrOptions::*field;
    bool rebuild_programs;  // cached Gram programs depend on it
};

static inline const FbrOptionKey *fbr_option_keys(int *count)
{
    static const FbrOptionKey keys[] = {
        {"link_merge", &FbrOptions::link_merge, false},
        {"regroup", &FbrOptions::regroup, false},
        {"reduce_min_work", &FbrOptions::reduce_min_work, false},
        {"reduce_grouped_min_samples", &FbrOptions::reduce_grouped_min_samples, false},
        {"chunk_samples", &FbrOptions::chunk_samples, false},
        {"min_chunks", &FbrOptions::min_chunks, false},
        {"h2d_chunked", &FbrOptions::h2d_chunked, false},
        {"fused_id", &FbrOptions::fused_id, false},
        {"gram_lane", &FbrOptions::gram_lane, false},
        {"gram_force_tiles", &FbrOptions::gram_force_tiles, false},
        {"gram_lane_waves", &FbrOptions::gram_lane_waves, false},
        {"gram_shape", &FbrOptions::gram_shape, true},
        {"gram_rhs_tile", &FbrOptions::gram_rhs_tile, true},
        {"gram_orient", &FbrOptions::gram_orient, true},
        {"gram_serial", &FbrOptions::gram_serial, false},
        {"gram_timing", &FbrOptions::gram_timing, false},
        {"tsqr_groups", &FbrOptions::tsqr_groups, false},
        {"tsqr_group_min_samples", &FbrOptions::tsqr_group_min_samples, false},
        {"tsqr_reorder", &FbrOptions::tsqr_reorder, false},
        {"tsqr_tree_one_wg", &FbrOptions::tsqr_tree_one_wg, false},
        {"tsqr_narrow", &FbrOptions::tsqr_narrow, false},
        {"tsqr_narrow_tall", &FbrOptions::tsqr_narrow_tall, false},
        {"tsqr_lane_writer", &FbrOptions::tsqr_lane_writer, false},
        {"tsqr_force_group", &FbrOptions::tsqr_force_group, false},
        {"tsqr_writer", &FbrOptions::tsqr_writer, false},
        {"tsqr_side_trees_beside", &FbrOptions::tsqr_side_trees_beside, false},
        {"tsqr_prologue_overlap", &FbrOptions::tsqr_prologue_overlap, false},
        {"tsqr_timing", &FbrOptions::tsqr_timing, false},
        {"tsqr_short_call_factors", &FbrOptions::tsqr_short_call_factors, false},
    };
    *count = (int)(sizeof(keys) / sizeof(keys[0]));
    return keys;
}

static inline const FbrOptionKey *fbr_option_find(const char *name)
{
    int n = 0;
    const FbrOptionKey *k = fbr_option_keys(&n);
    for (int i = 0; i < n; i++)
        if (name && !strcmp(k[i].name, name)) return &k[i];
    return nullptr;
}
