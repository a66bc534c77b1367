// fbr_gram64.h -- the fused Gram over SAMPLE-CONTIGUOUS images (round 6, option "gram_lane").
//
// The contraction G_ab = sum_s sum_r Y_s[r, a] Y_s[r, b] does not care in which order (s, r) is walked.  The first fused pass
// (fbr_kernels.h K5a / K5b) takes ONE sample's image per LDS-DMA and runs the MFMA k-steps over four ROWS of that sample: its producer has to
// lay out a sample's 20 KB contiguously, which a one-lane-per-sample producer can only do with 8-byte stores at a 33 KB stride (partially
// written lines) -- hence the record round trip kinematics kernel -> HBM -> workgroup-per-sample packer, which bounds that pass.
// Here the k-steps run over four SAMPLES of one regressor row:
//   * image of a block of 64 samples: [tile row tr][half][16 columns][32 samples], tile row = (column tile, real row of the tile: the fb
//     base-wrench rows, then the joints of the tile's path); no row padding, no paired base rows.  Inside a 32-sample run sample s of
//     column slot c sits at s ^ 2 c: an operand read of the MFMAs (lane (kk, li) -> column li, sample 4 ks + kk) is served half a wave at a
//     time, and the 32 lanes of a half (16 columns x two values of kk) then fall on the 32 different bank pairs of the LDS (with s ^ 4 (c & 7),
//     round 6's first layout, columns c and c + 8 collided: SQ_LDS_BANK_CONFLICT 42 % of the LDS cycles of the kernel), with the LDS layout
//     EQUAL to the global one (one contiguous 4 KB DMA per slab);
//   * producer fbr_kinimg_kernel: one lane per sample, kinematics fused in, the tree cut into parts for the waves of a workgroup
//     (fbr_kinid.h); every value of a (column, row) goes out as two 256-byte runs per wave;  tau's products with the columns (k <= 1) are
//     accumulated on the way: one six-term dot product per column against the link's t_base + sum_j S_j t_j, added to the lane's own running
//     sum in HBM (no-return atomics, one adder per address: deterministic);
//   * consumer fbr_gram64_kernel: one workgroup of 8 waves per CU, the accumulators of the tile pairs in registers for the whole pass;
//     a stage = (a few consecutive row levels, 32 samples): the slabs of the tiles that have the levels arrive by LDS-DMA into one of two
//     buffers while the MFMAs of the stage before run; a pair takes part in the levels of its range; 8 MFMAs per pair, level and half block;
//   * the force rows of the base wrench (levels 0 .. 2) run on tiles of their own that hold the columns with a force only (fbr_gram64_build).
// Conditions (else the first pass runs): k <= 1 rhs column (or none), a tile program in one part; sample groups (fbr_gram_grouped) for k = 0.
// Friction columns are tiles whose levels are the rows of their own joints.
// Inputs resident in HBM or pinned host memory (staged chunk by chunk); row weights; a base-wrench-only row mask runs the base stages only.
#pragma once
// the swizzle of column slot c inside its 32-sample run (see the image layout above)
#define FBR_G64_SWZ(c) (2 * ((c) & 15))
#include <algorithm>
#include <type_traits>
#include <utility>

#include "fbr_kinid.h"

struct FbrGram64 {  // host program
    int NT = 0;      // column tiles of the tile program ("main" tiles)
    int NF = 0;      // force tiles (below); tiles are numbered main 0 .. NT-1, force NT .. NT+NF-1
    int nlev = 0, fb = 0, flev = 0, ntr = 0, maxact = 0, npw = 0, wpb = 8;  // npw accumulators per wave, wpb waves per workgroup (8 or 16)
    long blk_doubles = 0;
    std::vector<int> trow;       // [NT + NF][nlev] tile-row index or -1
    std::vector<int> slab;       // [nlev][NT + NF] slab index inside the level's stage or -1
    int nstage = 0, base_stages = 0;
    std::vector<int> stage_lev;  // [nstage + 1] first level of each stage
    std::vector<int> lev_begin;  // [nlev + 1] into pieces
    std::vector<int> pieces;     // pairs: global offset (doubles, inside the block image, half 0), LDS offset (doubles, inside a stage buffer)
    std::vector<int> wmeta;      // [wpb waves][npw][3]: tile I (-1: empty slot), tile J, first level | (one past the last level) << 8
    std::vector<int> slot_tiles; // [2][wpb * npw * 2] for the two reductions: main pairs, force pairs (the other kind's slots are -1), in the
                                 // partial-sum order fbr_gram_reduce_kernel walks: [wave & 7][wave >> 3][slot] (16 waves = 8 rows of twice the slots)
    std::vector<int> tilecol;    // [NT + NF][16] column of each tile slot, -1 = padding
    std::vector<int> fcol_tile, fcol_slot;  // per column: its force tile / slot there, or -1
    std::vector<int> tile_lo;    // [NT] first level of every main / friction tile
    long mfma_per_block = 0;
    long busiest = 0, balanced = 0;  // sum over the stages of the busiest wave's pair-levels / of ceil(all pair-levels / 8)
};

// Tiles / pairs of a one-part tile program built WITHOUT rhs tiles (moments); false: the model is outside this pass.
//
// FORCE TILES (floating base).  The first three regressor rows are the force rows of the base wrench, and only the mass and the first
// moments of a link produce a force: in the column tiles of the program most entries of those rows are structural zeros (6 of 10 columns
// of a full link, 5 of 7 of a regrouped one), yet every tile pair pays three levels for them -- a third of all MFMAs of WALK-MAN.  So the
// force rows get tiles of their own: the columns that have a force, 16 to a tile in column order (base rows are common to all columns:
// no path condition), every pair of force tiles runs levels 0 .. 2, and the program's tiles start at level 3.  An entry G_ab with two
// force columns is the sum of its two blocks (two reductions, one after the other).  Used when the extra pairs fit the accumulator slots.
// wide16 (option gram_lane_waves = 16): models of the one-workgroup-per-CU shape (18 accumulators per wave of an 8-wave workgroup) run 16
// waves of 10 accumulators instead -- four waves per SIMD at 128 registers.  Measured: the Gram kernel 9.70 instead of 9.32 ms per 1 M
// WALK-MAN samples (operand reads in half groups, more A reloads, less reuse per wave): not the default.
static inline bool fbr_gram64_build(const FbrHostModel &hm, const FbrGramProgram &gp, FbrGram64 &g, bool force_tiles = true, bool wide16 = false)
{
    if (gp.T != 1 || (gp.k > 0 && gp.rhs_tiles)) return false;
    g.npw = gp.cfg.segw * gp.cfg.nseg;
    g.wpb = FBR_WPB;
    if (wide16 && g.npw > 10) {
        g.wpb = 2 * FBR_WPB;
        g.npw = 10;
    }
    const int W = g.wpb;
    g.NT = gp.NT;
    g.fb = hm.fb;
    g.nlev = 0;
    // levels of a tile: a main tile has the base-wrench rows and the joints of its path; a FRICTION tile (its columns are non-zero on the row
    // of their own joint only) just the levels of its own columns' joints -- a contiguous stretch of its path
    std::vector<int> tlo(gp.NT, 0), thi(gp.NT, 0);
    for (int t = 0; t < gp.NT; t++) {
        const FbrTile &tl = gp.tiles[t];
        if (tl.type != 0) return false;
        thi[t] = hm.fb + (int)tl.tpath.size();
        if (tl.friction) {
            int jmin = (int)tl.tpath.size(), jmax = -1;
            for (int sl = 0; sl < FBR_TILE; sl++) {
                if (tl.col[sl] < 0) continue;
                const int jnt = hm.coldesc[tl.col[sl]].joint;
                for (int j = 0; j < (int)tl.tpath.size(); j++)
                    if (tl.tpath[j] == jnt) jmin = std::min(jmin, j), jmax = std::max(jmax, j);
            }
            if (jmax < 0) return false;
            tlo[t] = hm.fb + jmin;
            thi[t] = hm.fb + jmax + 1;
        }
        g.nlev = std::max(g.nlev, thi[t]);
    }
    if (g.nlev == 0 || g.nlev > 255) return false;
    // the tile pairs of the program: unordered, with their common depth cp = fb + joints both tiles' columns have rows on
    struct Pr {
        int a, b, lo, hi;
    };
    std::vector<Pr> prs;
    for (size_t s = 0; s < gp.slots.size(); s++) {
        const int pi = gp.slots[s].pair;
        if (pi < 0) continue;
        const FbrPair &p = gp.pairs[pi];
        if (p.mode != 0) return false;
        const int cp = hm.fb + FbrGramProgram::common_prefix(gp.tiles[p.I].tpath, gp.tiles[p.J].tpath);
        prs.push_back({p.I, p.J, std::max(tlo[p.I], tlo[p.J]), std::min(cp, std::min(thi[p.I], thi[p.J]))});
    }
    // force columns and their tiles
    g.fcol_tile.assign(hm.cols, -1);
    g.fcol_slot.assign(hm.cols, -1);
    g.tilecol.assign((size_t)g.NT * FBR_TILE, -1);
    std::vector<char> has_tile(hm.cols, 0);
    for (int t = 0; t < g.NT; t++)
        for (int sl = 0; sl < FBR_TILE; sl++) {
            const int c = gp.tiles[t].col[sl];
            if (c >= 0 && c < hm.cols) {
                g.tilecol[(size_t)t * FBR_TILE + sl] = c;
                has_tile[c] = 1;
            }
        }
    g.NF = 0;
    g.flev = 0;
    if (force_tiles && hm.fb == 6) {
        int nf = 0;
        for (int c = 0; c < hm.ninert; c++)
            if (has_tile[c] && hm.coldesc[c].pidx < 4) {
                g.fcol_tile[c] = g.NT + nf / FBR_TILE;
                g.fcol_slot[c] = nf % FBR_TILE;
                nf++;
            }
        const int NF = (nf + FBR_TILE - 1) / FBR_TILE;
        if (NF > 0 && (long)prs.size() + (long)NF * (NF + 1) / 2 <= (long)W * g.npw) {
            g.NF = NF;
            g.flev = 3;
        } else {
            std::fill(g.fcol_tile.begin(), g.fcol_tile.end(), -1);
            std::fill(g.fcol_slot.begin(), g.fcol_slot.end(), -1);
        }
    }
    const int NTT = g.NT + g.NF;
    g.tilecol.resize((size_t)NTT * FBR_TILE, -1);
    for (int c = 0; c < hm.cols; c++)
        if (g.fcol_tile[c] >= 0) g.tilecol[(size_t)g.fcol_tile[c] * FBR_TILE + g.fcol_slot[c]] = c;
    for (int t = 0; t < g.NT; t++)
        if (!gp.tiles[t].friction) tlo[t] = g.flev;  // (the main tiles start behind the force levels)
    for (Pr &p : prs) p.lo = std::max(p.lo, std::max(tlo[p.a], tlo[p.b]));
    prs.erase(std::remove_if(prs.begin(), prs.end(), [](const Pr &p) { return p.lo >= p.hi; }), prs.end());
    for (int f = 0; f < g.NF; f++)
        for (int f2 = f; f2 < g.NF; f2++) prs.push_back({g.NT + f, g.NT + f2, 0, g.flev});
    // tile rows: the force tiles' first (so that "level 0" of a main tile, flev rows in front of its first row, is inside the image), then the
    // rows of every main tile one after the other -- the producer relies on consecutive rows: level lv of a column sits 1024 doubles x lv
    // behind its level 0
    g.trow.assign((size_t)NTT * g.nlev, -1);
    g.ntr = 0;
    for (int f = 0; f < g.NF; f++)
        for (int lv = 0; lv < g.flev; lv++) g.trow[(size_t)(g.NT + f) * g.nlev + lv] = g.ntr++;
    for (int t = 0; t < g.NT; t++) {
        g.ntr = std::max(g.ntr, tlo[t]);  // ("level 0" of the tile, tlo rows in front of its first one, must not fall in front of the image)
        for (int lv = tlo[t]; lv < thi[t]; lv++) g.trow[(size_t)t * g.nlev + lv] = g.ntr++;
    }
    g.tile_lo = tlo;
    g.blk_doubles = (long)g.ntr * 1024;
    // Stages: consecutive levels whose slabs fit one LDS buffer together share a stage --
    // one barrier and one round of LDS-DMA for the three force levels, or for the deep levels only a few tiles reach.
    std::vector<int> nslab(g.nlev, 0);
    int widest = 0;
    for (int lv = 0; lv < g.nlev; lv++) {
        for (int t = 0; t < NTT; t++) nslab[lv] += g.trow[(size_t)t * g.nlev + lv] >= 0;
        widest = std::max(widest, nslab[lv]);
    }
    // (the two-per-CU shape of small models keeps its workgroups below half the LDS: 2 buffers x 8 slabs x 4 KB)
    const int cap = std::max(widest, g.npw <= 10 ? 8 : 16);
    g.stage_lev.assign(1, 0);
    for (int lv = 0, in_stage = 0; lv < g.nlev; lv++) {
        if (lv > g.stage_lev.back() && (in_stage + nslab[lv] > cap || lv == hm.fb)) {  // (the base-wrench rows end a stage: base_stages)
            g.stage_lev.push_back(lv);
            in_stage = 0;
        }
        in_stage += nslab[lv];
    }
    g.stage_lev.push_back(g.nlev);
    g.nstage = (int)g.stage_lev.size() - 1;
    if (g.nstage > 62) return false;  // (the kernel keeps the stages' constants one per lane)
    g.base_stages = 0;  // stages of the base-wrench rows alone: all a call runs whose row weights switch every joint row off
    while (g.base_stages < g.nstage && g.stage_lev[g.base_stages + 1] <= hm.fb) g.base_stages++;
    g.slab.assign((size_t)g.nlev * NTT, -1);
    g.lev_begin.assign(g.nlev + 1, 0);
    g.pieces.clear();
    g.maxact = 0;
    for (int sg = 0; sg < g.nstage; sg++) {
        int idx = 0;
        for (int lv = g.stage_lev[sg]; lv < g.stage_lev[sg + 1]; lv++) {
            g.lev_begin[lv] = (int)g.pieces.size() / 2;
            for (int t = 0; t < NTT; t++) {
                const int tr = g.trow[(size_t)t * g.nlev + lv];
                if (tr < 0) continue;
                g.slab[(size_t)lv * NTT + t] = idx;
                for (int p = 0; p < 4; p++) {
                    g.pieces.push_back(tr * 1024 + p * 128);
                    g.pieces.push_back(idx * 512 + p * 128);
                }
                idx++;
            }
        }
        g.maxact = std::max(g.maxact, idx);
    }
    g.lev_begin[g.nlev] = (int)g.pieces.size() / 2;
    // Pairs -> waves.  All waves of the workgroup meet at every stage (level, half block), so a stage lasts as long as its busiest wave: the
    // cost of a placement is  sum over levels of max over waves of the pairs active at the level,  not the waves' totals (the tile program's
    // own placement, balanced by totals, leaves WALK-MAN at 156 against 126 for a perfect split).  Any pair may sit in any accumulator of
    // any wave (the kernel reloads the A operand when the tile I of the next slot differs).  Deal by decreasing length, then local search:
    // move a pair / swap two while (cost, sum of squared loads) falls.  Deterministic.
    const int NP = (int)prs.size();
    if (NP > W * g.npw) return false;
    std::vector<int> order(NP), wave_of(NP, -1), cnt(W, 0);
    for (int i = 0; i < NP; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return prs[x].hi - prs[x].lo > prs[y].hi - prs[y].lo; });
    std::vector<std::vector<int>> wl(W, std::vector<int>(g.nlev, 0));
    auto shift = [&](int i, int w, int sign) {
        for (int lv = prs[i].lo; lv < prs[i].hi; lv++) wl[w][lv] += sign;
        cnt[w] += sign;
    };
    auto score = [&](long &c, long &sq) {  // (the waves meet once per stage: the loads of a stage's levels add up)
        c = 0, sq = 0;
        for (int sg = 0; sg < g.nstage; sg++) {
            int mx = 0;
            for (int w = 0; w < W; w++) {
                int tot = 0;
                for (int lv = g.stage_lev[sg]; lv < g.stage_lev[sg + 1]; lv++) tot += wl[w][lv];
                mx = std::max(mx, tot);
                sq += (long)tot * tot;
            }
            c += mx;
        }
    };
    for (int r = 0; r < NP; r++) {  // snake deal
        const int round = r / W, pos = r % W, w = (round & 1) ? W - 1 - pos : pos;
        wave_of[order[r]] = w;
        shift(order[r], w, +1);
    }
    long c0, q0;
    score(c0, q0);
    for (int sweep = 0, improved = 1; improved && sweep < 64; sweep++) {
        improved = 0;
        for (int i = 0; i < NP; i++) {
            for (int w2 = 0; w2 < W; w2++) {
                const int w1 = wave_of[i];
                if (w2 == w1 || cnt[w2] >= g.npw) continue;
                shift(i, w1, -1), shift(i, w2, +1);
                long c1, q1;
                score(c1, q1);
                if (c1 < c0 || (c1 == c0 && q1 < q0)) {
                    wave_of[i] = w2, c0 = c1, q0 = q1, improved = 1;
                } else {
                    shift(i, w2, -1), shift(i, w1, +1);
                }
            }
            for (int k2 = i + 1; k2 < NP; k2++) {
                const int w1 = wave_of[i], w2 = wave_of[k2];
                if (w1 == w2 || (prs[i].lo == prs[k2].lo && prs[i].hi == prs[k2].hi)) continue;
                shift(i, w1, -1), shift(k2, w2, -1), shift(i, w2, +1), shift(k2, w1, +1);
                long c1, q1;
                score(c1, q1);
                if (c1 < c0 || (c1 == c0 && q1 < q0)) {
                    wave_of[i] = w2, wave_of[k2] = w1, c0 = c1, q0 = q1, improved = 1;
                } else {
                    shift(i, w2, -1), shift(k2, w1, -1), shift(i, w1, +1), shift(k2, w2, +1);
                }
            }
        }
    }
    g.busiest = c0;
    g.balanced = 0;
    for (int sg = 0; sg < g.nstage; sg++) {
        int tot = 0;
        for (int lv = g.stage_lev[sg]; lv < g.stage_lev[sg + 1]; lv++)
            for (int w = 0; w < W; w++) tot += wl[w][lv];
        g.balanced += (tot + W - 1) / W;
    }
    // slots of a wave: the A operand (tile I) of a slot is kept for the next one when it is the same tile, so the pairs of a wave are ordered
    // by the tile most of them contain (which of a pair's two tiles is "I" is free: the reduction writes the block and its mirror image)
    g.wmeta.assign((size_t)W * g.npw * 3, -1);
    g.slot_tiles.assign((size_t)2 * W * g.npw * 2, -1);
    g.mfma_per_block = 0;
    for (int w = 0; w < W; w++) {
        std::vector<int> mine;
        for (int i = 0; i < NP; i++)
            if (wave_of[i] == w) mine.push_back(i);
        std::vector<char> done(NP, 0);
        int q = 0;
        for (size_t left = mine.size(); left > 0;) {
            std::vector<int> freq(NTT, 0);
            for (int i : mine)
                if (!done[i]) {
                    freq[prs[i].a]++;
                    if (prs[i].b != prs[i].a) freq[prs[i].b]++;
                }
            int T = 0;
            for (int t = 1; t < NTT; t++)
                if (freq[t] > freq[T]) T = t;
            for (int i : mine) {
                if (done[i] || (prs[i].a != T && prs[i].b != T)) continue;
                const int J = prs[i].a == T ? prs[i].b : prs[i].a;
                const size_t s = (size_t)w * g.npw + q;                                                  // wmeta: wave-major
                const size_t sr = ((size_t)(w & 7) * (W / 8) + (size_t)(w >> 3)) * g.npw + q;             // the reduction's order
                g.wmeta[3 * s] = T;
                g.wmeta[3 * s + 1] = J;
                g.wmeta[3 * s + 2] = prs[i].lo | (prs[i].hi << 8);
                const int kind = T >= g.NT ? 1 : 0;
                g.slot_tiles[((size_t)kind * W * g.npw + sr) * 2] = T;
                g.slot_tiles[((size_t)kind * W * g.npw + sr) * 2 + 1] = J;
                g.mfma_per_block += 16L * (prs[i].hi - prs[i].lo);  // 8 MFMAs per level and half
                done[i] = 1;
                q++;
                left--;
            }
        }
    }
    return true;
}

// Producer tables: the tree in parts for the waves of a workgroup (fbr_kinid.h) and, per (part, link), 14 destination words: one per
// parameter, then the FORCE-tile words of parameters 0 .. 3.  A word: byte offset inside an image buffer of (level 0 of the column's tile
// rows, column slot, sample 0) -- a multiple of 256 -- with the slot's swizzle (FBR_G64_SWZ) in its low byte and bit 62 set (0: the part does not write that
// column).  Level lv of the column is 8192 bytes x lv further on (the tile rows of a tile are consecutive; with force tiles "level 0" of a
// main tile is three rows in front of its first row, rows 0 .. 2 of a force column go through its force word).
#define FBR_G64_WORDS 18  // 10 parameters, the force-tile words of parameters 0 .. 3, up to 4 friction columns of the link's joint
#define FBR_G64_FRIC 4
struct FbrGram64Producer {
    int nparts = 1, nslots = 1, step0[FBR_KINWRITE_PARTS] = {0, 0, 0, 0}, nsteps[FBR_KINWRITE_PARTS] = {0, 0, 0, 0};
    std::vector<long long> rel;  // [nparts][L][14]
    std::vector<int> lcol;       // [nparts][10 L] the column (for the rhs moments) or -1, then [nparts][4 L] the friction columns of the link's joint
    std::vector<int> steps;      // the parts' step programs, one after the other
};

static inline bool fbr_gram64_build_producer(const FbrHostModel &hm, const FbrGramProgram &gp, const FbrGram64 &g, FbrGram64Producer &pr)
{
    std::vector<int> tile_of(hm.cols, -1), slot_of(hm.cols, -1);
    for (int t = 0; t < g.NT; t++)
        for (int sl = 0; sl < FBR_TILE; sl++)
            if (gp.tiles[t].col[sl] >= 0 && gp.tiles[t].col[sl] < hm.cols) {
                tile_of[gp.tiles[t].col[sl]] = t;
                slot_of[gp.tiles[t].col[sl]] = sl;
            }
    std::vector<double> lcost(hm.L, 30.0);
    for (int c = 0; c < hm.ninert; c++)
        if (tile_of[c] >= 0) lcost[hm.coldesc[c].link] += 10.0 + (double)(hm.fb + hm.path[hm.coldesc[c].link].size());
    std::vector<FbrKinIdProgram> progs;
    std::vector<std::vector<char>> own;
    try {
        fbr_kinid_build_parts(hm, lcost, FBR_KINWRITE_PARTS, progs, own);
    } catch (const std::exception &) {
        return false;
    }
    pr.nparts = (int)progs.size();
    pr.rel.assign((size_t)pr.nparts * FBR_G64_WORDS * hm.L, 0);
    pr.lcol.assign((size_t)pr.nparts * (10 + FBR_G64_FRIC) * hm.L, -1);
    const int nfric = hm.n > 0 ? (hm.cols - hm.ninert) / hm.n : 0;
    if (nfric > FBR_G64_FRIC) return false;
    auto word = [](long long tile_row, int sl) { return ((tile_row * 1024 + sl * 32) * 8) | (long long)FBR_G64_SWZ(sl) | (1LL << 62); };
    for (int c = 0; c < hm.ninert; c++) {
        const int t = tile_of[c], sl = slot_of[c], l = hm.coldesc[c].link, pidx = hm.coldesc[c].pidx;
        if (t < 0) continue;  // (a column without a tile: structurally zero, e.g. the base link of a fixed base)
        if (hm.path[l].size() > gp.tiles[t].tpath.size()) return false;  // (cannot happen: the tile's path contains the link's)
        const long long tr0 = (long long)g.trow[(size_t)t * g.nlev + g.tile_lo[t]] - g.tile_lo[t];
        if (tr0 < 0) return false;  // (cannot happen: the force tiles' rows come first)
        for (int pq = 0; pq < pr.nparts; pq++)
            if (own[pq][l]) {
                long long *w14 = &pr.rel[((size_t)pq * hm.L + l) * FBR_G64_WORDS];
                w14[pidx] = word(tr0, sl);
                if (g.fcol_tile[c] >= 0) {
                    if (pidx >= 4) return false;
                    w14[10 + pidx] = word(g.trow[(size_t)g.fcol_tile[c] * g.nlev], g.fcol_slot[c]);
                }
                pr.lcol[((size_t)pq * hm.L + l) * 10 + pidx] = c;
            }
    }
    // friction columns: the part that owns the joint's link writes them, on the row of that joint (level fb + position of the joint on the path)
    for (int c = hm.ninert; c < hm.cols; c++) {
        const int t = tile_of[c], sl = slot_of[c], d = hm.coldesc[c].joint, p = (c - hm.ninert) / std::max(hm.n, 1);
        if (t < 0) continue;
        int l = -1;
        for (int x = 0; x < hm.L; x++)
            if (hm.dof[x] == d) l = x;
        if (l < 0 || p >= FBR_G64_FRIC || hm.path[l].empty() || hm.path[l].back() != d) return false;
        const int lv = hm.fb + (int)hm.path[l].size() - 1;
        if (lv < g.tile_lo[t] || g.trow[(size_t)t * g.nlev + lv] < 0) return false;
        const long long tr0 = (long long)g.trow[(size_t)t * g.nlev + g.tile_lo[t]] - g.tile_lo[t];
        if (tr0 < 0) return false;
        for (int pq = 0; pq < pr.nparts; pq++)
            if (own[pq][l]) {
                pr.rel[((size_t)pq * hm.L + l) * FBR_G64_WORDS + 14 + p] = word(tr0, sl);
                pr.lcol[(size_t)pr.nparts * 10 * hm.L + ((size_t)pq * hm.L + l) * FBR_G64_FRIC + p] = c;
            }
    }
    pr.steps.clear();
    pr.nslots = 1;
    for (int pq = 0; pq < pr.nparts; pq++) {
        pr.step0[pq] = (int)(pr.steps.size() / FBR_KINID_STEP);
        pr.nsteps[pq] = progs[pq].nsteps;
        pr.nslots = std::max(pr.nslots, progs[pq].nslots);
        pr.steps.insert(pr.steps.end(), progs[pq].steps.begin(), progs[pq].steps.begin() + (size_t)progs[pq].nsteps * FBR_KINID_STEP);
    }
    return true;
}

#if defined(__HIPCC__) && defined(FBR_KERNELS_GRAM)
struct DevGram64 {
    int NT, nlev, maxact, npieces, nstage;  // NT: main + force tiles
    long blk_doubles;
    const int *slab, *lev_begin, *pieces, *wmeta, *stage_lev;
};

// ------------------------------------------------------------------------------------------------
// Producer: the lane writer of fbr_kinid.h with the image addressing of this pass.  Destination word of a (column, row): address of the
// slab position of column slot c, sample 0 (256-byte aligned) | the slot's swizzle in its low byte; sample slot s of block b goes to
// + b * blk_doubles + (s >> 5) * 512 + ((s & 31) ^ x).  The positions of the lanes behind the last sample of the last block are cleared by the host before the launch (the Gram
// kernel runs whole blocks).  mom (k == 1): [workgroup][cols + 1][64] per-lane running sums of (w Y)^T (w tau) per column and (w tau)^T (w tau), added in block order.
// ------------------------------------------------------------------------------------------------
template <int MAXD, bool HASW>
__global__ __launch_bounds__(64 * FBR_KINWRITE_PARTS, MAXD <= 10 ? 2 : 1) void fbr_kinimg_kernel(DevModel m, DevKinId p, DevKinWrite wr, long S, long blk_doubles,
                                                                              const double *__restrict__ q, const double *__restrict__ dq,
                                                                              const double *__restrict__ ddq, const double *__restrict__ bv,
                                                                              const double *__restrict__ ba, const double *__restrict__ rpy,
                                                                              const double *__restrict__ rhs, const double *__restrict__ wts,
                                                                              double *__restrict__ scratch, double *__restrict__ mom,
                                                                              const double *__restrict__ sign)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nth = blockDim.x, tid = threadIdx.x;
    const int n = m.n, ldn = p.ldn, ldw = m.rows | 1, rows = m.rows;
    double *sq = smem, *sdq = sq + 64 * ldn, *sddq = sdq + 64 * ldn, *sw = sddq + 64 * ldn;  // sw [64][ldw] row weights (has_w)
    double *st = sw + (HASW ? 64 * ldw : 0);                                             // st [64][ldw] w^2 tau (k == 1)
    double *scr = scratch + ((long)blockIdx.x * wr.nparts + part) * p.nslots * FBR_LINK_REC * 64 + lane;
    double *mo = mom ? mom + (long)blockIdx.x * (wr.cols + 1) * 64 : nullptr;  // [cols + 1][64 lanes]
    // sample groups (fbr_gram_grouped): every group starts a block -- block b = (group b / bpg, block b % bpg of the group)
    const long Sg = wr.group_samples > 0 ? wr.group_samples : S, bpg = (Sg + 63) >> 6;
    const long nblk = (S / Sg) * bpg;
    const fbr_clong_ptr cdst = (fbr_clong_ptr)(unsigned long)wr.dst;
    const fbr_cint_ptr ccol = (fbr_cint_ptr)(unsigned long)(wr.lcol10 + (long)part * 10 * m.L);
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long grp = blk / bpg, lb = blk - grp * bpg;
        const long base = grp * Sg + (lb << 6);
        const int valid = (int)min(64L, Sg - (lb << 6));
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
            if (HASW || wr.k) {
                const int cw = valid * rows;
                int wr_ = tid / rows, wc = tid - wr_ * rows;
                const int es = nth / rows, ed = nth - es * rows;
                for (int i = tid; i < cw; i += nth) {
                    const double wv = HASW ? wts[base * rows + i] : 1.0;
                    if (HASW) sw[wr_ * ldw + wc] = wv;
                    if (wr.k) st[wr_ * ldw + wc] = wv * wv * rhs[base * rows + i];
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
        const double *mysq = sq + ls * ldn, *mysdq = sdq + ls * ldn, *mysddq = sddq + ls * ldn, *myw = sw + ls * ldw, *myt = st + ls * ldw;
        const long slot_off = blk * blk_doubles + (long)(lane >> 5) * 512;
        const int s31 = lane & 31;
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
        // byte offset of this lane's sample inside a (tile row, column) run, before the column's swizzle: block, half, sample
        const unsigned vlane = (unsigned)(((unsigned long)slot_off + (unsigned long)s31) << 3);
        auto link = [&](int l, int depth, const double *rec, const double (*Sst)[6], const int *lvd, double *F) {
            (void)F;
            if (wr.base_only) depth = 0;  // (row weights switch every joint row off: identifier.py:629-636 -- only the base-wrench rows are produced)
            // Every vector load of the step (branch records, states) is waited for HERE, once: the stores below share the loads' counter, and
            // behind the branches of the column code the compiler cannot tell how many of them sit in front of a load it still expects --
            // it would wait for counter 0, i.e. for the store before, at every store (measured: 7.4 -> 5.4 ms per 1 M WALK-MAN samples).
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
            long d10[10], dF[4];
#pragma unroll
            for (int pp = 0; pp < 10; pp++) d10[pp] = cdst[((long)part * m.L + l) * FBR_G64_WORDS + pp];
#pragma unroll
            for (int pp = 0; pp < 4; pp++) dF[pp] = cdst[((long)part * m.L + l) * FBR_G64_WORDS + 10 + pp];  // force-tile words (rows 0 .. flev-1)
            // tau's side of the moments: sum_r v_r t_r over the rows of one column = w6 . (t_base + sum_j S_j t_j), t = w^2 tau
            double Ft[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (wr.k) {
#pragma unroll
                for (int i = 0; i < 6; i++)
                    if (i < m.fb) Ft[i] = myt[i];
#pragma unroll
                for (int j = 0; j < MAXD; j++)
                    if (j < depth) {
                        const double tj = myt[m.fb + lvd[j]];
#pragma unroll
                        for (int i = 0; i < 6; i++) Ft[i] += Sst[j][i] * tj;
                    }
            }
            // One value to (column q of the group, level lv).  Scalar base (the column's word, 8192 bytes per level) + this lane's 32-bit
            // offset with the column's swizzle; a column the part does not write (word 0) skips the store only -- the products of a level
            // are computed for the whole group first, branch-free, so that their dependent chains overlap.
            auto store = [&](long d0, int lv, double v) {
                if (d0 == 0) return;
                const unsigned vo = vlane ^ ((unsigned)(d0 & 0xff) << 3);
                __builtin_nontemporal_store(v, (fbr_gdouble_ptr)((fbr_gchar_ptr)(d0 & ~0xffL) + (long)lv * 8192 + vo));
            };
            // mass and first moments: full wrenches, NQ columns at a time
            auto full_group = [&](auto q0c, auto nqc) {
                constexpr int Q0 = decltype(q0c)::value, NQ = decltype(nqc)::value;
                double wA[NQ][6];
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) fbr_unit_wrench(rec, Q0 + qq, wA[qq]);
#pragma unroll
                for (int i = 0; i < 6; i++)
                    if (i < m.fb) {
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) store(i < wr.flev ? dF[Q0 + qq] : d10[Q0 + qq], i, HASW ? wA[qq][i] * myw[i] : wA[qq][i]);
                    }
#pragma unroll
                for (int j = 0; j < MAXD; j++)
                    if (j < depth) {
                        double v[NQ];
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) v[qq] = fbr_dot6(Sst[j], wA[qq]);
                        const double wj = HASW ? myw[m.fb + lvd[j]] : 1.0;
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) store(d10[Q0 + qq], m.fb + j, HASW ? v[qq] * wj : v[qq]);
                    }
                if (wr.k) {
#pragma unroll
                    for (int qq = 0; qq < NQ; qq++)
                        if (d10[Q0 + qq]) unsafeAtomicAdd(mo + (long)ccol[10 * l + Q0 + qq] * 64 + lane, fbr_dot6(Ft, wA[qq]));  // this lane's own running sum
                }
            };
            // inertia entries: pure moments -- the force rows of the base wrench are structural zeros of the image (never written), the joint
            // rows need the moment half of S only
            auto moment_group = [&](auto q0c, auto nqc) {
                constexpr int Q0 = decltype(q0c)::value, NQ = decltype(nqc)::value;
                double nB[NQ][3];
#pragma unroll
                for (int qq = 0; qq < NQ; qq++) fbr_unit_moment3(rec, 4 + Q0 + qq, nB[qq]);
#pragma unroll
                for (int i = 3; i < 6; i++)
                    if (i < m.fb) {
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) store(d10[4 + Q0 + qq], i, HASW ? nB[qq][i - 3] * myw[i] : nB[qq][i - 3]);
                    }
#pragma unroll
                for (int j = 0; j < MAXD; j++)
                    if (j < depth) {
                        double v[NQ];
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) v[qq] = Sst[j][3] * nB[qq][0] + Sst[j][4] * nB[qq][1] + Sst[j][5] * nB[qq][2];
                        const double wj = HASW ? myw[m.fb + lvd[j]] : 1.0;
#pragma unroll
                        for (int qq = 0; qq < NQ; qq++) store(d10[4 + Q0 + qq], m.fb + j, HASW ? v[qq] * wj : v[qq]);
                    }
                if (wr.k) {
#pragma unroll
                    for (int qq = 0; qq < NQ; qq++)
                        if (d10[4 + Q0 + qq])
                            unsafeAtomicAdd(mo + (long)ccol[10 * l + 4 + Q0 + qq] * 64 + lane, Ft[3] * nB[qq][0] + Ft[4] * nB[qq][1] + Ft[5] * nB[qq][2]);
                }
            };
            using std::integral_constant;
            if constexpr (MAXD > 8 && MAXD <= 10) {  // (the instance that has to fit 256 registers for two waves per SIMD: smaller groups)
                full_group(integral_constant<int, 0>{}, integral_constant<int, 2>{});
                full_group(integral_constant<int, 2>{}, integral_constant<int, 2>{});
                moment_group(integral_constant<int, 0>{}, integral_constant<int, 3>{});
                moment_group(integral_constant<int, 3>{}, integral_constant<int, 3>{});
            } else {
                full_group(integral_constant<int, 0>{}, integral_constant<int, 4>{});
                moment_group(integral_constant<int, 0>{}, integral_constant<int, 6>{});
            }
            // friction columns of the link's own joint: one value each, on the row of that joint (the last level of the link's path)
            const int nfr = wr.cols > wr.ninert ? (wr.cols - wr.ninert) / n : 0;
            const int dj = nfr ? m.dof[l] : -1;
            if (dj >= 0 && depth > 0) {
                const fbr_cint_ptr cfr = (fbr_cint_ptr)(unsigned long)(wr.lcol10 + (long)wr.nparts * 10 * m.L + ((long)part * m.L + l) * FBR_G64_FRIC);
#pragma unroll
                for (int pf = 0; pf < FBR_G64_FRIC; pf++)
                    if (pf < nfr) {
                        const long dfw = cdst[((long)part * m.L + l) * FBR_G64_WORDS + 14 + pf];
                        if (dfw == 0) continue;
                        const int c = cfr[pf];
                        const double fv = fbr_friction_value(m.coldesc[c].z, mysdq[dj], sign ? sign[s * n + dj] : 0.0, m.stribeck);
                        store(dfw, m.fb + depth - 1, HASW ? fv * myw[m.fb + dj] : fv);
                        if (wr.k) unsafeAtomicAdd(mo + (long)c * 64 + lane, fv * myt[m.fb + dj]);
                    }
            }
        };
        auto emit = [&](int, double) {};
        // lanes behind the last sample of the last block take no part (EXEC off): their image positions were cleared by the host
        if (live) {
            fbr_kinid_lane<MAXD, false>(wr.part_nsteps[part], p.maxlvl, p.steps + wr.part_step0[part] * FBR_KINID_STEP, p.endflush, m.floating, m.g, m.fb,
                                        state, basest, save, load, link, emit, consts);
            if (wr.k && part == wr.nparts - 1) {  // (w tau)^T (w tau)
                double tt = 0.0;
                for (int r = 0; r < rows; r++) {
                    const double wv = HASW ? myw[r] : 1.0, tv = rhs[s * rows + r] * wv;
                    tt += tv * tv;
                }
                unsafeAtomicAdd(mo + (long)wr.cols * 64 + lane, tt);
            }
        }
    }
}

// The positions of sample slots valid .. 63 of the LAST block of every group (blockIdx.y; bpg blocks per group): groups whose sample count is
// no multiple of 64 end in a block the producer fills partly, and an image buffer is reused from call to call.
__global__ __launch_bounds__(256) void fbr_gram64_tail_zero_kernel(double *__restrict__ img, long blk_doubles, long bpg, int ntr, int valid)
{
    double *blk = img + ((long)blockIdx.y * bpg + bpg - 1) * blk_doubles;
    const int per = 16 * (64 - valid);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)ntr * per; e += (long)gridDim.x * blockDim.x) {
        const int tr = (int)(e / per), r = (int)(e - (long)tr * per), c = r / (64 - valid), sl = valid + r % (64 - valid);
        blk[(long)tr * 1024 + (sl >> 5) * 512 + c * 32 + ((sl & 31) ^ FBR_G64_SWZ(c))] = 0.0;
    }
}

// rhs moments of a call -> G (k == 1): one workgroup per column sums the per-lane running sums of the producer's workgroups in a fixed order
__global__ __launch_bounds__(256) void fbr_gram64_mom_reduce_kernel(int P, int nwg, const double *__restrict__ mom, double *__restrict__ G)
{
    __shared__ double part[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // (independent running sums: the loads are far apart)
    int i = t;
    for (; i + 768 < nwg * 64; i += 1024) {
        s0 += mom[((long)(i >> 6) * (P + 1) + c) * 64 + (i & 63)];
        s1 += mom[((long)((i + 256) >> 6) * (P + 1) + c) * 64 + ((i + 256) & 63)];
        s2 += mom[((long)((i + 512) >> 6) * (P + 1) + c) * 64 + ((i + 512) & 63)];
        s3 += mom[((long)((i + 768) >> 6) * (P + 1) + c) * 64 + ((i + 768) & 63)];
    }
    for (; i < nwg * 64; i += 256) s0 += mom[((long)(i >> 6) * (P + 1) + c) * 64 + (i & 63)];
    part[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t) return;
    const int Pa = P + 1;
    if (c == P) {
        G[(long)P * Pa + P] += part[0];
    } else {
        G[(long)c * Pa + P] += part[0];
        G[(long)P * Pa + c] += part[0];
    }
}

// ------------------------------------------------------------------------------------------------
// Consumer.  One workgroup (8 waves) per CU walks blocks blockIdx.x, + gridDim.x, ...; stage = (block, half, level).  partial:
// [workgroup][wave][slot][4][64] (the layout fbr_gram_reduce_kernel sums); carry: start from it.
// ------------------------------------------------------------------------------------------------
template <int NPW, int WPB>
__global__ __launch_bounds__(WPB * 64, (NPW <= 10) ? 4 : 2) void fbr_gram64_kernel(DevGram64 g, long nblk, const double *__restrict__ img,
                                                                                  double *__restrict__ partial, int carry)
{
    constexpr int MW = 3 * NPW;
    static_assert(MW <= 64, "a wave's slot table is fetched with one load");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int bufd = g.maxact * 512;
    double *buf0 = smem, *buf1 = smem + bufd;
    int *slab = (int *)(smem + 2 * bufd);      // [nlev][NT]
    int *levb = slab + g.nlev * g.NT;          // [nlev + 1]
    int *pcs = levb + g.nlev + 1;              // [npieces][2]
    int *wm = pcs + 2 * g.npieces;             // [8][NPW][3]
    int *stl = wm + WPB * MW;              // [nstage + 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < g.nlev * g.NT; i += WPB * 64) slab[i] = g.slab[i];
    for (int i = tid; i <= g.nlev; i += WPB * 64) levb[i] = g.lev_begin[i];
    for (int i = tid; i < 2 * g.npieces; i += WPB * 64) pcs[i] = g.pieces[i];
    for (int i = tid; i < WPB * MW; i += WPB * 64) wm[i] = g.wmeta[i];
    for (int i = tid; i <= g.nstage; i += WPB * 64) stl[i] = g.stage_lev[i];
    fbr_d4 acc[NPW];
    img += (long)blockIdx.y * nblk * g.blk_doubles;  // blockIdx.y: sample group (nblk blocks each, fbr_gram_grouped)
    // partial sums in the order of fbr_gram_reduce_kernel (8 rows per workgroup): wave w is row w & 7, slots (w >> 3) NPW ...
    double *pp = partial + (((((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (wave & 7)) * (WPB / 8) + (wave >> 3)) * NPW) * 256;
    if (carry) {
#pragma unroll
        for (int q = 0; q < NPW; q++) acc[q] = (fbr_d4){pp[q * 256 + lane], pp[q * 256 + 64 + lane], pp[q * 256 + 128 + lane], pp[q * 256 + 192 + lane]};
    } else {
#pragma unroll
        for (int q = 0; q < NPW; q++) acc[q] = (fbr_d4){0.0, 0.0, 0.0, 0.0};
    }
    __syncthreads();
    const int nmine = (int)((nblk - blockIdx.x + gridDim.x - 1) / gridDim.x);  // blocks of this workgroup
    const long nst = (long)(nmine > 0 ? nmine : 0) * 2 * g.nstage;
    // per-stage constants in registers (lane s: stage s; lane nstage: the end): first level and first DMA piece -- read with a lane select
    // instead of chained LDS look-ups in front of every stage
    const int svl = stl[lane <= g.nstage ? lane : 0], svp = levb[svl];
    // LDS-DMA of step (block-half bh, stage sg) into buffer par: wave w issues pieces w, w + WPB, ... of the stage's levels
    auto dma = [&](int sg, long bh, int par) {
        const long blk = (long)blockIdx.x + (bh >> 1) * gridDim.x;
        const double *src = img + blk * g.blk_doubles + (bh & 1) * 512 + 2 * lane;
        double *buf = par ? buf1 : buf0;
        // (the wave's pieces i0, i0 + WPB, ... of the stage: their table entries are fetched by the lanes in parallel -- one LDS round trip
        // instead of one per piece in front of every stage's first MFMA)
        const int i0 = __builtin_amdgcn_readlane(svp, sg) + wave, i1 = __builtin_amdgcn_readlane(svp, sg + 1);
        const int mine = i0 < i1 ? (i1 - i0 + WPB - 1) / WPB : 0;
        const int il = i0 + (lane < mine ? lane : 0) * WPB;
        const int gxv = mine > 0 ? pcs[2 * il] : 0, lxv = mine > 0 ? pcs[2 * il + 1] : 0;
        for (int j = 0; j < mine; j++) {
            const int gx = __builtin_amdgcn_readlane(gxv, j), lx = __builtin_amdgcn_readlane(lxv, j);
            __builtin_amdgcn_global_load_lds((fbr_glb_ptr)(src + gx), (fbr_lds_ptr)(buf + lx), 16, 0, 0);
        }
    };
    if (nst > 0) dma(0, 0, 0);
    const int li = lane & 15, kk = lane >> 4;
    const int lofs = li * 32, sx = FBR_G64_SWZ(li);
    const int mv = wm[wave * MW + (lane < MW ? lane : 0)];  // this wave's slots: (tile I, tile J, first level | end level << 8)
    const int idxv = (lane < MW && (lane % 3) != 2 && mv >= 0) ? mv : 0;  // the tile this lane looks up per level (lanes 3q, 3q + 1)
    // this lane's operand position inside a slab (doubles) at k-step ks: column li, sample (4 ks + kk) ^ sx = p0 ^ (4 ks)
    const int p0c = lofs | (sx ^ kk);
    int sg = 0;   // stage of step st
    long bh = 0;  // its block-half (of this workgroup's blocks)
    for (long st = 0; st < nst; st++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step st have landed
        __syncthreads();                                  // everyone's have; the other buffer is free
        const int sgn = sg + 1 == g.nstage ? 0 : sg + 1;
        const long bhn = sgn ? bh : bh + 1;
        if (st + 1 < nst) dma(sgn, bhn, (int)((st + 1) & 1));
        const double *buf = (st & 1) ? buf1 : buf0;
        const int lv0 = __builtin_amdgcn_readlane(svl, sg), lv1 = __builtin_amdgcn_readlane(svl, sg + 1);
        sg = sgn;
        bh = bhn;
        for (int lv = lv0; lv < lv1; lv++) {
            const int offv = slab[lv * g.NT + idxv] * 512;  // the slab offsets of all slots of the wave at this level: one LDS read
            int curI = -1;  // the tile whose operand the registers a[] hold (of this level)
            double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NPW; q++) {
                const int tI = __builtin_amdgcn_readlane(mv, 3 * q), rg = __builtin_amdgcn_readlane(mv, 3 * q + 2);
                if (tI < 0 || lv < (rg & 0xff) || lv >= (rg >> 8)) continue;
                const double *pb = buf + __builtin_amdgcn_readlane(offv, 3 * q + 1);
                double b[NPW <= 10 ? 4 : 8];
                int p0 = p0c;
                if constexpr (NPW <= 10) asm volatile("" : "+v"(p0));  // (the 128-register shape: the eight positions are recomputed, not kept)
                if (tI != curI) {
                    const double *pa = buf + __builtin_amdgcn_readlane(offv, 3 * q);
#pragma unroll
                    for (int ks = 0; ks < 8; ks++) a[ks] = pa[p0 ^ (4 * ks)];
                    curI = tI;
                }
                // all operand reads of a group of k-steps are in flight before its first MFMA (the compiler would otherwise pair every MFMA
                // with the read in front of it and wait for each); the 128-register shape reads four k-steps at a time
                constexpr int KG = NPW <= 10 ? 4 : 8;
#pragma unroll
                for (int k0 = 0; k0 < 8; k0 += KG) {
#pragma unroll
                    for (int ks = k0; ks < k0 + KG; ks++) b[ks - k0] = pb[p0 ^ (4 * ks)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = k0; ks < k0 + KG; ks++) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks - k0], acc[q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NPW; q++) {
        pp[q * 256 + 0 * 64 + lane] = acc[q][0];
        pp[q * 256 + 1 * 64 + lane] = acc[q][1];
        pp[q * 256 + 2 * 64 + lane] = acc[q][2];
        pp[q * 256 + 3 * 64 + lane] = acc[q][3];
    }
}
#endif
