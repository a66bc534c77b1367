// fbr_program.h -- host-side (HIP-free) model tables and the tile program of the fused Gram kernel.
//
// Kept free of HIP so that tests can compile it with g++ and emulate the kernels' data flow on the CPU
// (tests/emul/); the product only uses it from fbr_api.hip to fill device tables.
//
// Column tiles: the augmented regressor [Y | rhs] is regrouped into tiles of 16 columns.
//   CHAIN tile: columns of links whose movable-joint paths are nested (one root-to-leaf chain).  Its
//               rows are stored PACKED by path depth: position 0..fb-1 = base-wrench rows, position
//               fb+j = the j-th movable joint on the path.  For two chain tiles the rows that can be
//               non-zero in both are exactly the first `common` packed positions of each, so a tile
//               pair needs ceil(common/4) MFMA k-steps instead of ceil(rows/4).
//   DENSE tile: friction / right-hand-side columns, all `rows` rows in global order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#define FBR_TILE 16
#ifndef FBR_WPB
#define FBR_WPB 8         // waves per workgroup of the Gram kernel (overridable for experiments)
#endif

// Shape of the streaming Gram kernel.  Two shapes are compiled (fbr_kernels.h) and chosen per model:
//   two workgroups per CU:  2 row segments x 5 pairs per wave (10 accumulators, ~106 VGPRs, 4 waves / SIMD), LDS images of
//                           <= 36 KiB -- one workgroup's barrier / DMA phase hides behind the other's MFMAs; best when the
//                           tile images are small enough that the model still splits into few parts (WALK-MAN: 14);
//   one workgroup per CU:   3 row segments x 6 pairs per wave (18 accumulators, ~176 VGPRs), LDS images of <= 74 KiB --
//                           fewer parts and far less image re-reading when the dense (friction) tiles inflate the images
//                           (WALK-MAN with friction: 7 parts instead of 25; 5.5 vs 3.2 M samples/s).
struct FbrGramConfig {
    int segw;        // tile pairs per row segment (same tile I, up to segw different tiles J)
    int nseg;        // row segments per wave
    int img_budget;  // doubles per LDS image buffer of a part
    // cycles per sample of a part's workgroup ~ c0 + cload * (cost units of its most loaded wave) + cmfma * (MFMAs of the part)
    // + cimg * (image doubles): non-negative least squares over the per-part s_memtime phases of the six WALK-MAN layouts
    // (tools/gram_timing_probe.py + tools/fit_gram_cost.py; 3.8 % / 2.8 % rms).  Only ratios matter: the cuts between the parts
    // minimise the summed cost and workgroups are dealt to the parts in proportion to it (fbr_gram_deal).
    double c0, cload, cmfma, cimg;
    int npw() const { return segw * nseg; }  // tile pairs (MFMA accumulators) per wave
    bool operator==(const FbrGramConfig &o) const { return segw == o.segw && nseg == o.nseg && img_budget == o.img_budget; }  // same kernel shape
};
static const FbrGramConfig FBR_CFG_TWO_PER_CU = {5, 2, 4608, 1171.0, 37.2, 22.4, 0.07};
#ifndef FBR_ONE_SEGW  // shape of the one-workgroup-per-CU kernel (overridable for experiments: -DFBR_ONE_SEGW=.. -DFBR_ONE_NSEG=..)
#define FBR_ONE_SEGW 6
#define FBR_ONE_NSEG 3
#endif
#ifndef FBR_ONE_IMG
#define FBR_ONE_IMG 9472
#endif
static const FbrGramConfig FBR_CFG_ONE_PER_CU = {FBR_ONE_SEGW, FBR_ONE_NSEG, FBR_ONE_IMG, 1873.0, 7.7, 18.1, 0.15};
#define FBR_MAX_PARTS_TWO_PER_CU 1   // a model that needs more than one part takes the large-image shape: measured with the branch-free
                                     // kernel (tools/gram_shape_probe.py, profiles/r01n_gram_shapes.txt) the large images win every
                                     // multi-part layout of WALK-MAN by 0-7 % (fewer image re-reads, room for the producer kernels
                                     // beside the Gram kernel), the small ones the single-part robots by 8-12 %
#define FBR_MAX_RHS 16

struct FbrCol {
    int kind;  // 0 inertial, 1 friction
    int link;  // inertial: link index
    int pidx;  // inertial: parameter 0..9 ; friction: fkind (0 Fc,1 Fv,2 Fv+,3 Fv-,4 off,5 Fs)
    int joint; // friction: dof index ; inertial: -1, or -2 for the unpaired column of a link with a column mask (see linkmask)
};

struct FbrHostModel {
    int L = 0, n = 0, fb = 0, rows = 0, cols = 0, cpl = 10;
    // Packed base positions of the tile images.  Floating base: 8 = two MFMA k-steps.  The 6 base-wrench rows of two consecutive
    // samples of a group fill 3 k-steps instead of 2 x 2 with two padding rows each: the even sample's image holds its own
    // rows b0..b5 in positions 0..5 and its partner's b0, b1 in positions 6, 7 ("ghost rows", written by the partner's
    // workgroup); the odd sample's image holds its b2..b5 in positions 4..7 and the Gram kernel skips its k-step 0.
    // Dense (rhs) tiles use the same 8 base rows, then the joint rows.
    int fbp = 0;
    int floating = 0, fric = 0, fric_sym = 0, grav_only = 0;
    double stribeck = 0.0;
    double gravity[3] = {0, 0, -9.81};
    std::vector<int> order, parent, dof;
    std::vector<int> jtype;  // [L] 0 fixed / base, 1 revolute, 2 prismatic
    std::vector<double> restR, restp, axis;
    std::vector<std::vector<int>> path;  // per link: movable joints root -> link
    std::vector<std::vector<int>> ppos;  // per link: PACKED row position of each joint of its path (see FbrTile)
    std::vector<int> pdepth;             // per link: packed rows used (last position + 1)
    std::vector<FbrCol> coldesc;         // identified columns
    int maxdepth = 0;
    // Column masks (internal models of the link-merged / regrouped reductions, fbr_api.hip build_reduction): link l identifies only the
    // parameters whose bit is set in linkmask[l].  The inertial columns are then laid out as [pairs of columns of one link, link by
    // link | the unpaired column of every link with an odd count | friction]: every pair sits at an even column in any selection of
    // whole links (16-byte stores of the grouped TSQR writer).  Without masks: cpl columns per link, link by link.
    bool masked = false;
    int ninert = 0;     // inertial columns
    int npaircols = 0;  // leading inertial columns that form same-link pairs (c, c + 1), c even
    std::vector<std::vector<int>> linkcols;  // per link: its columns

    void build(int L_, int n_, const int32_t *parent_, const int32_t *dof_, const double *restR_, const double *restp_,
               const double *axis_, int floating_, const double *g, int fric_, int fric_sym_, int grav_only_,
               double stribeck_, const unsigned short *linkmask = nullptr, const int32_t *jtype_ = nullptr)
    {
        L = L_; n = n_; floating = floating_ ? 1 : 0; fric = fric_ ? 1 : 0; fric_sym = fric_sym_ ? 1 : 0;
        grav_only = grav_only_ ? 1 : 0; stribeck = stribeck_;
        if (L <= 0 || n < 0) throw std::runtime_error("bad model size");
        fb = floating ? 6 : 0;
        rows = n + fb;
        fbp = fb ? 8 : 0;
        cpl = grav_only ? 4 : 10;
        for (int i = 0; i < 3; i++) gravity[i] = g[i];
        parent.assign(parent_, parent_ + L);
        dof.assign(dof_, dof_ + L);
        restR.assign(restR_, restR_ + 9 * L);
        restp.assign(restp_, restp_ + 3 * L);
        axis.assign(axis_, axis_ + 3 * L);
        jtype.assign(L, 0);
        for (int l = 0; l < L; l++) {
            if (parent_[l] < 0 || dof_[l] < 0) continue;
            const int jt = jtype_ ? jtype_[l] : 1;
            if (jt != 1 && jt != 2) throw std::runtime_error("joint_type of a link with a DOF must be 1 (revolute) or 2 (prismatic)");
            jtype[l] = jt;
        }
        // traversal: stable DFS, parents first
        std::vector<std::vector<int>> children(L);
        int base = -1;
        for (int l = 0; l < L; l++) {
            if (parent[l] < 0) {
                if (base >= 0) throw std::runtime_error("more than one base link");
                base = l;
            } else {
                if (parent[l] >= L) throw std::runtime_error("parent index out of range");
                children[parent[l]].push_back(l);
            }
        }
        if (base < 0) throw std::runtime_error("no base link");
        order.clear();
        std::vector<int> stack{base};
        while (!stack.empty()) {
            int l = stack.back();
            stack.pop_back();
            order.push_back(l);
            for (auto it = children[l].rbegin(); it != children[l].rend(); ++it) stack.push_back(*it);
        }
        if ((int)order.size() != L) throw std::runtime_error("links do not form a single tree");
        std::vector<int> seen(std::max(n, 1), 0);
        path.assign(L, {});
        maxdepth = 0;
        for (int l : order) {
            if (parent[l] >= 0) {
                path[l] = path[parent[l]];
                if (dof[l] >= 0) {
                    if (dof[l] >= n || seen[dof[l]]) throw std::runtime_error("bad dof_index");
                    seen[dof[l]] = 1;
                    path[l].push_back(dof[l]);
                }
            } else if (dof[l] >= 0) {
                throw std::runtime_error("base link cannot have a dof");
            }
            maxdepth = std::max(maxdepth, (int)path[l].size());
        }
        for (int d = 0; d < n; d++)
            if (!seen[d]) throw std::runtime_error("dof without a joint");
        // packed row positions: base rows first, then the joints of the path in order; after every prefix from
        // which two or more different movable joints continue (a branch point of the joint tree) the next position
        // is rounded up to a multiple of 4, so two diverging chains always share a whole number of MFMA k-steps
        // and no operand masking is needed (the rounding gaps are structural zero rows).
        {
            std::vector<int> pdof(std::max(n, 1), -1), nchild(n + 1, 0);  // nchild[n] = children of the empty prefix
            for (int l = 0; l < L; l++) {
                const auto &p = path[l];
                if (!p.empty() && dof[l] == p.back()) pdof[p.back()] = p.size() >= 2 ? p[p.size() - 2] : n;
            }
            for (int d = 0; d < n; d++)
                if (pdof[d] >= 0) nchild[pdof[d]]++;
            ppos.assign(L, {});
            pdepth.assign(L, 0);
            for (int l = 0; l < L; l++) {
                int cur = fbp;
                if (nchild[n] >= 2) cur = (cur + 3) & ~3;
                int depth = fbp;
                for (int d : path[l]) {
                    ppos[l].push_back(cur);
                    depth = cur + 1;
                    cur++;
                    if (nchild[d] >= 2) cur = (cur + 3) & ~3;
                }
                pdepth[l] = depth;
            }
        }
        // identified columns (model.py:134-168, 459-503)
        coldesc.clear();
        linkcols.assign(L, {});
        masked = linkmask != nullptr;
        if (!masked) {
            for (int l = 0; l < L; l++)
                for (int p = 0; p < cpl; p++) {
                    linkcols[l].push_back((int)coldesc.size());
                    coldesc.push_back({0, l, p, -1});
                }
            npaircols = (cpl % 2 == 0) ? cpl * L : 0;
        } else {
            if (grav_only) throw std::runtime_error("column masks and gravity-only models do not combine");
            std::vector<int> single(L, -1);
            for (int l = 0; l < L; l++) {
                std::vector<int> kept;
                for (int p = 0; p < 10; p++)
                    if (linkmask[l] >> p & 1) kept.push_back(p);
                if (kept.size() & 1) {
                    single[l] = kept.back();
                    kept.pop_back();
                }
                for (int p : kept) {
                    linkcols[l].push_back((int)coldesc.size());
                    coldesc.push_back({0, l, p, -1});
                }
            }
            npaircols = (int)coldesc.size();
            for (int l = 0; l < L; l++)
                if (single[l] >= 0) {
                    linkcols[l].push_back((int)coldesc.size());
                    coldesc.push_back({0, l, single[l], -2});
                }
        }
        ninert = (int)coldesc.size();
        if (fric) {
            for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 0, j});
            if (!grav_only) {
                if (fric_sym) {
                    for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 1, j});
                } else {
                    for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 2, j});
                    for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 3, j});
                }
                for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 4, j});
                if (stribeck > 0)
                    for (int j = 0; j < n; j++) coldesc.push_back({1, -1, 5, j});
            }
        }
        cols = (int)coldesc.size();
    }
    int rec_size() const { return FBR_LINK_REC * L + FBR_DOF_REC * n; }
    // standard-vector index of the first friction parameter (model.py:164-168)
    int friction_start() const { return grav_only ? 4 * L : 10 * L; }
};

struct FbrTile {
    int type = 0;                 // 0 chain, 1 dense
    int depth = 0;                // packed rows (chain: fb + path length, dense: rows)
    int off = 0;                  // offset (doubles) of the tile inside the per-sample LDS image
    int col[FBR_TILE];            // augmented column id per slot, -1 = padding
    std::vector<int> rowid;       // chain: global row of each packed position
    int friction = 0;             // chain tile of friction columns
    int okey = 0;                 // traversal position of the tile's last link (tile order)
    std::vector<char> rowreal;    // chain: 1 = the packed position holds a regressor row (0 = alignment gap, always zero)
    std::vector<char> posnz;      // chain: 1 = some column of the tile can be non-zero in that packed position
    std::vector<char> rownz;      // dense: 1 = some column of the tile can be non-zero in that regressor row
    std::vector<int> tpath;       // chain: dof path of the deepest link
    std::vector<int> tpos;        // chain: packed position of each joint of tpath
};
struct FbrPair {
    int I, J, common, mode;  // mode 0: both packed by position; 1: B rows looked up through rowid_I; 2: dense x dense
    unsigned kmask = 0;      // k-step ks (tile-I rows 4ks..4ks+3) can contribute iff bit ks is set: the others are structurally zero
    int kbegin() const { return kmask ? __builtin_ctz(kmask) : 0; }      // first k-step that can contribute
    int nkend() const { return kmask ? 32 - __builtin_clz(kmask) : 0; }  // one past the last one
    // The kernel runs the k-steps [kbegin of the pair's row segment, nkend): a superset of the mask (zeros add nothing).
};
struct FbrItem {  // one producer work item = one real column of one tile
    int off;      // LDS offset of (tile, slot): tile.off + slot
    int kind;     // 0 inertial, 1 friction, 2 rhs
    int a;        // inertial: link ; friction: joint ; rhs: rhs column
    int b;        // inertial: pidx ; friction: fkind
    int col;      // the augmented column it produces (what the packer's rhs moments of this item belong to)
};
struct FbrSlot {  // one accumulator of one wave
    int pair;     // -1 = unused
    int kb;       // first k-step its row segment runs
};

struct FbrPiece {  // one LDS-DMA piece of a part's per-sample image: 128 doubles (1 KiB, whole wave) or 64 (half wave)
    int goff;      // offset (doubles) inside the global per-sample image
    int loff;      // offset (doubles) inside the part-local LDS image
    int half;      // 1: 64 doubles (lanes 0..31 only)
};

struct FbrGramProgram {
    int k = 0, Pa = 0, NT = 0, T = 0;
    int rows_pad = 0;        // rows rounded up to a multiple of 4
    int image_doubles = 0;   // global per-sample image size (all tiles), multiple of 64
    int part_image_max = 0;  // largest part-local LDS image (doubles, incl. one spare k-step of tail padding)
    int maxrow = 0;          // max tile depth
    std::vector<FbrTile> tiles;
    std::vector<FbrPair> pairs;
    std::vector<FbrItem> items;                   // every real column of every tile (pack kernel work list)
    std::vector<FbrSlot> slots;                   // [T][WPB][NPW]
    std::vector<std::vector<int>> part_tiles;     // per part: tiles it needs, in image order
    std::vector<std::vector<int>> part_tile_off;  // per part: local LDS offset (doubles) of each tile, -1 if absent
    std::vector<std::vector<FbrPiece>> pieces;    // per part: DMA pieces
    std::vector<int> part_image;                  // per part: local image size (doubles)
    std::vector<int> part_load;                   // per part: cost units of its most loaded wave (2 per MFMA + fixed costs)
    std::vector<int> part_mfma;                   // per part: MFMAs per sample (all waves)
    std::vector<double> part_cost;                // per part: modelled cycles per sample of one workgroup (FbrGramConfig)
    int base_ks = 0;           // 1 with a floating base: the odd sample of a pair skips k-step 0 (its b0, b1 sit in its partner's image)
    int64_t mfma_per_sample = 0;  // average over an even / odd pair of samples (rounded up)
    int64_t mfma_uniform = 0;  // k-steps that run in row segments whose pairs all share `common` and the operand mode

    static bool nested(const std::vector<int> &a, const std::vector<int> &b)
    {
        size_t m = std::min(a.size(), b.size());
        for (size_t i = 0; i < m; i++)
            if (a[i] != b[i]) return false;
        return true;
    }
    static int common_prefix(const std::vector<int> &a, const std::vector<int> &b)
    {
        size_t m = std::min(a.size(), b.size()), i = 0;
        while (i < m && a[i] == b[i]) i++;
        return (int)i;
    }

    FbrGramConfig cfg = FBR_CFG_TWO_PER_CU;
    // false: the rhs columns get no tiles -- their products Y^T rhs, rhs^T rhs are accumulated by the pack kernel, which holds every
    // regressor entry in a register anyway (fbr_gram_rhs_moments): a dense tile for ONE rhs column costs a full tile's MFMAs against
    // every other tile (the regrouped WALK-MAN: 64 of 336 MFMAs per sample).  Pa, the stride of G, counts the rhs columns either way.
    bool rhs_tiles = true;
    bool orient = false;    // turn chain x chain pairs so that the row segments fill up (build(): orientation); oriented: it was done
    bool oriented = false;
    int block_edge = 0;  // > 0: edge of the square blocks in which the pair triangle is enumerated (fbr_gram_build_best tries several)
    double total_cost() const
    {
        double c = 0.0;
        for (double x : part_cost) c += x;
        return c;
    }

    void build(const FbrHostModel &hm, int k_, const FbrGramConfig &cfg_)
    {
        cfg = cfg_;
        const int FBR_SEGW = cfg.segw, FBR_NSEG = cfg.nseg, FBR_NPW = cfg.npw(), FBR_IMG_BUDGET = cfg.img_budget;
        k = k_;
        if (k < 0 || k > FBR_MAX_RHS) throw std::runtime_error("rhs column count must be 0..16");
        Pa = hm.cols + k;
        rows_pad = (hm.rows + 3) / 4 * 4;
        base_ks = hm.fb ? 1 : 0;
        if (rows_pad > 60) throw std::runtime_error("more than 60 regressor rows per sample (15 MFMA k-steps) are not supported");
        tiles.clear();
        pairs.clear();
        // ---- chain tiles, links in traversal order: first the inertial columns, then (own tiles) the friction columns.
        //      A friction column of joint j is a chain column that is non-zero in one packed row only, the position of j
        //      (model.py:459-503): stored like this it shares the positional addressing of the chain tiles and its
        //      products with links that do not hang below j vanish from the pair list.
        const int F = hm.n > 0 ? (hm.cols - hm.ninert) / hm.n : 0;  // friction columns per joint
        // The order in which the branches of the tree are walked decides how the chains fill their tiles (a tile is closed where the
        // paths stop being nested): the children of every link are tried in index order, largest and smallest sub-tree first, and the
        // walk with the fewest inertial tiles is kept (the regrouped WALK-MAN: base + waist + one arm fill 5 tiles exactly, 16 tiles
        // instead of 17 -- a nearly empty tile costs as many MFMAs as a full one).
        std::vector<int> torder = hm.order;
        {
            std::vector<std::vector<int>> children(hm.L);
            int base = 0;
            for (int l = 0; l < hm.L; l++) (hm.parent[l] < 0 ? (void)(base = l) : children[hm.parent[l]].push_back(l));
            std::vector<long> sub(hm.L, 0);
            for (auto it = hm.order.rbegin(); it != hm.order.rend(); ++it) {
                sub[*it] += (long)hm.linkcols[*it].size();
                if (hm.parent[*it] >= 0) sub[hm.parent[*it]] += sub[*it];
            }
            int best_tiles = -1;
            for (int mode = 0; mode < 3; mode++) {
                std::vector<int> ord, stack{base};
                while (!stack.empty()) {
                    const int l = stack.back();
                    stack.pop_back();
                    ord.push_back(l);
                    std::vector<int> ch = children[l];
                    if (mode == 1) std::stable_sort(ch.begin(), ch.end(), [&](int a, int b) { return sub[a] > sub[b]; });
                    if (mode == 2) std::stable_sort(ch.begin(), ch.end(), [&](int a, int b) { return sub[a] < sub[b]; });
                    for (auto it = ch.rbegin(); it != ch.rend(); ++it) stack.push_back(*it);
                }
                int nt = 0, fill = 0;
                const std::vector<int> *tp = nullptr;
                for (int l : ord)
                    for (size_t p = 0; p < hm.linkcols[l].size(); p++) {
                        if (tp && (fill == FBR_TILE || !nested(*tp, hm.path[l]))) tp = nullptr;
                        if (!tp) {
                            nt++;
                            fill = 0;
                            tp = &hm.path[l];
                        }
                        if (hm.path[l].size() > tp->size()) tp = &hm.path[l];
                        fill++;
                    }
                if (best_tiles < 0 || nt < best_tiles) {
                    best_tiles = nt;
                    torder = ord;
                }
            }
        }
        for (int pass = 0; pass < (F > 0 ? 2 : 1); pass++) {
            FbrTile cur;
            int fill = 0;
            bool open = false;
            auto close = [&]() {
                if (!open) return;
                for (int s = fill; s < FBR_TILE; s++) cur.col[s] = -1;
                tiles.push_back(cur);
                open = false;
            };
            for (size_t oi = 0; oi < torder.size(); oi++) {
                const int l = torder[oi];
                if (pass == 1 && hm.dof[l] < 0) continue;
                const int ncol = pass == 0 ? (int)hm.linkcols[l].size() : F;
                for (int p = 0; p < ncol; p++) {
                    if (open && (fill == FBR_TILE || !nested(cur.tpath, hm.path[l]))) close();
                    if (!open) {
                        cur = FbrTile();
                        cur.type = 0;
                        cur.friction = pass;
                        cur.tpath = hm.path[l];
                        cur.tpos = hm.ppos[l];
                        cur.depth = hm.pdepth[l];
                        fill = 0;
                        open = true;
                    }
                    if (hm.path[l].size() > cur.tpath.size()) {
                        cur.tpath = hm.path[l];
                        cur.tpos = hm.ppos[l];
                        cur.depth = hm.pdepth[l];
                    }
                    cur.col[fill++] = pass == 0 ? hm.linkcols[l][p] : hm.ninert + p * hm.n + hm.dof[l];
                    cur.okey = 2 * (int)oi + pass;
                }
            }
            close();
        }
        // a friction tile follows the inertial tile of the same links: the pairs that survive lie near the diagonal of the
        // (I, J) triangle and a part of the pair list touches few tiles
        std::stable_sort(tiles.begin(), tiles.end(), [](const FbrTile &a, const FbrTile &b) { return a.okey < b.okey; });
        for (auto &t : tiles) {
            // packed position -> regressor row (alignment gaps map to row 0; their image rows are zero)
            t.rowid.assign(t.depth, 0);
            t.rowreal.assign(t.depth, 0);
            for (int i = 0; i < hm.fbp; i++) {  // the dense tiles keep the same 8 base rows
                t.rowid[i] = i;
                t.rowreal[i] = 1;
            }
            for (size_t j = 0; j < t.tpath.size(); j++) {
                t.rowid[t.tpos[j]] = hm.fbp + t.tpath[j];
                t.rowreal[t.tpos[j]] = 1;
            }
            t.posnz = t.rowreal;
            if (t.friction) {  // only the positions of the tile's own joints
                t.posnz.assign(t.depth, 0);
                for (int sl = 0; sl < FBR_TILE; sl++) {
                    if (t.col[sl] < 0) continue;
                    const int jnt = hm.coldesc[t.col[sl]].joint;
                    for (size_t j = 0; j < t.tpath.size(); j++)
                        if (t.tpath[j] == jnt) t.posnz[t.tpos[j]] = 1;
                }
            }
        }
        // ---- dense tiles: rhs columns, all rows in regressor order
        {
            int c = hm.cols;
            while (c < Pa && rhs_tiles) {
                FbrTile t;
                t.type = 1;
                t.depth = hm.fbp + hm.n;  // base rows in the packed order of the chain tiles, then the joint rows
                for (int s = 0; s < FBR_TILE; s++) t.col[s] = (c < Pa) ? c++ : -1;
                t.rownz.assign(t.depth, 1);
                tiles.push_back(t);
            }
        }
        NT = (int)tiles.size();
        int off = 0;
        maxrow = 0;
        for (auto &t : tiles) {
            t.off = off;
            int dp = (t.depth + 3) / 4 * 4;
            off += dp * FBR_TILE;  // multiple of 64 doubles
            maxrow = std::max(maxrow, dp);
        }
        image_doubles = off;
        // ---- pack-kernel work list: every real column of every tile
        items.clear();
        for (int ti = 0; ti < NT; ti++)
            for (int s = 0; s < FBR_TILE; s++) {
                int c = tiles[ti].col[s];
                if (c < 0) continue;
                FbrItem it;
                it.off = tiles[ti].off + s;
                it.col = c;
                if (c >= hm.cols) {
                    it.kind = 2; it.a = c - hm.cols; it.b = 0;
                } else if (hm.coldesc[c].kind == 0) {
                    it.kind = 0; it.a = hm.coldesc[c].link; it.b = hm.coldesc[c].pidx;
                } else {
                    it.kind = 1; it.a = hm.coldesc[c].joint; it.b = hm.coldesc[c].pidx;
                    const FbrTile &t = tiles[ti];  // the one image row of the column: the packed position of its joint
                    for (size_t j = 0; j < t.tpath.size(); j++)
                        if (t.tpath[j] == it.a) it.off += t.tpos[j] * FBR_TILE;
                }
                items.push_back(it);
            }
        // ---- pairs (I <= J), enumerated in square blocks of the (I, J) triangle so that a contiguous chunk
        //      of the list touches few distinct tiles (small part images, little DMA traffic)
        const int PPB = FBR_WPB * FBR_NPW;
        int BE = 1;
        while ((BE + 1) * (BE + 1) <= PPB) BE++;
        BE = std::max(FBR_SEGW, BE / FBR_SEGW * FBR_SEGW);  // block rows split into whole row segments
        if (block_edge > 0) BE = block_edge;
        mfma_per_sample = 0;
        mfma_uniform = 0;
        int64_t mfma_pair = 0;
        const int NB = (NT + BE - 1) / BE;
        for (int bi = 0; bi < NB; bi++)
            for (int bj = bi; bj < NB; bj++)
                for (int I = bi * BE; I < std::min(NT, (bi + 1) * BE); I++)
                    for (int J = std::max(I, bj * BE); J < std::min(NT, (bj + 1) * BE); J++) {
                        const FbrTile &a = tiles[I], &b = tiles[J];
                        FbrPair p{I, J, 0, 0};
                        if (a.type == 0 && b.type == 0) {
                            // nested chains: the shallower tile's rows (the deeper one's extra rows meet zero padding);
                            // diverging chains: the shared prefix, whose end is 4-aligned by construction
                            const int cp = common_prefix(a.tpath, b.tpath);
                            if (cp == (int)std::min(a.tpath.size(), b.tpath.size()))
                                p.common = std::min(a.depth, b.depth);
                            else
                                p.common = a.tpos[cp];
                            p.mode = 0;
                        } else if (a.type == 0 && b.type == 1) {
                            p.common = a.depth;
                            p.mode = 1;
                        } else if (a.type == 1 && b.type == 1) {
                            p.common = a.depth;
                            p.mode = 2;
                        } else {
                            throw std::runtime_error("dense tile before chain tile");
                        }
                        // k-steps that can contribute: the ones holding a row in which both tiles can be non-zero (friction
                        // columns: one row each)
                        for (int ks = 0; 4 * ks < p.common; ks++) {
                            bool on = false;
                            for (int r = 4 * ks; r < std::min(4 * ks + 4, p.common) && !on; r++) {
                                if (p.mode == 0)
                                    on = a.posnz[r] && b.posnz[r];
                                else if (p.mode == 1)
                                    on = a.posnz[r] && b.rownz[a.rowid[r]];
                                else
                                    on = a.rownz[r] && b.rownz[r];
                            }
                            if (on) p.kmask |= 1u << ks;
                        }
                        if (p.kmask == 0) continue;  // structurally zero block (fixed base, disjoint branches, friction of other joints)
                        pairs.push_back(p);
                    }
        // ---- orientation.  Y_I^T Y_J and Y_J^T Y_I hold the same numbers: which of the two tiles of a chain x chain pair is the "row"
        //      (the A operand, shared by the pairs of a row segment) is free.  Listed as I <= J the rows have NT, NT - 1, ..., 1 pairs and
        //      cut into segments of SEGW they leave many short ones (15 tiles: 27 segments for 120 pairs, more than the 24 of one
        //      workgroup: 2 parts).  The pairs are turned so that every row holds a multiple of SEGW pairs (20 full segments: ONE part,
        //      every image streamed once, twice the MFMAs between two barriers): targets by degree, orientation by augmenting paths.
        if (orient && !pairs.empty()) {
            bool ok = true;
            for (const FbrPair &pr : pairs) ok = ok && pr.mode == 0 && pr.kbegin() == pairs[0].kbegin();
            const int npq = (int)pairs.size();
            std::vector<int> deg(NT, 0), target(NT, 0), has_diag(NT, 0);
            for (const FbrPair &pr : pairs) {
                deg[pr.I]++;
                if (pr.J != pr.I) deg[pr.J]++;
                else has_diag[pr.I] = 1;
            }
            int left = npq;
            if (ok) {
                // every tile that has pairs gets one segment's worth (or all it can take), the rest goes out in whole segments to the
                // tiles with the most room, the remainder to one more
                for (int t = 0; t < NT; t++) {
                    target[t] = std::min(deg[t], FBR_SEGW);
                    left -= target[t];
                }
                ok = left >= 0;
                while (ok && left > 0) {
                    int best = -1;
                    for (int t = 0; t < NT; t++)
                        if (deg[t] - target[t] > 0 && (best < 0 || deg[t] - target[t] > deg[best] - target[best])) best = t;
                    if (best < 0) {
                        ok = false;
                        break;
                    }
                    const int give = std::min({left, FBR_SEGW, deg[best] - target[best]});
                    target[best] += give;
                    left -= give;
                }
            }
            if (ok) {
                // owner[p]: the tile whose row holds pair p; capacities = targets (a diagonal pair stays where it is)
                std::vector<int> owner(npq, -1), load(NT, 0);
                for (int i = 0; i < npq; i++)
                    if (pairs[i].I == pairs[i].J) {
                        owner[i] = pairs[i].I;
                        load[pairs[i].I]++;
                    }
                for (int t = 0; t < NT; t++) ok = ok && load[t] <= target[t];
                // augmenting paths: pair i wants an endpoint with room; a full endpoint may push one of its pairs to that pair's other end
                std::vector<char> seen;
                std::function<bool(int)> make_room = [&](int t) -> bool {  // free one unit of capacity at tile t
                    if (load[t] < target[t]) return true;
                    if (seen[t]) return false;
                    seen[t] = 1;
                    for (int i = 0; i < npq; i++) {
                        if (owner[i] != t || pairs[i].I == pairs[i].J) continue;
                        const int other = pairs[i].I == t ? pairs[i].J : pairs[i].I;
                        if (make_room(other)) {
                            owner[i] = other;
                            load[other]++;
                            load[t]--;
                            return true;
                        }
                    }
                    return false;
                };
                for (int i = 0; i < npq && ok; i++) {
                    if (owner[i] >= 0) continue;
                    bool placed = false;
                    for (int e = 0; e < 2 && !placed; e++) {
                        const int t = e == 0 ? pairs[i].I : pairs[i].J;
                        seen.assign(NT, 0);
                        if (make_room(t)) {
                            owner[i] = t;
                            load[t]++;
                            placed = true;
                        }
                    }
                    ok = placed;
                }
                if (ok)
                    for (int i = 0; i < npq; i++)
                        if (owner[i] != pairs[i].I) std::swap(pairs[i].I, pairs[i].J);
            }
            oriented = ok;
        }
        // ---- parts: contiguous chunks of the pair list.  Inside a part the pairs are grouped into ROW SEGMENTS
        //      (same tile I, <= SEGW tiles J, sorted by k-steps descending): a wave loads the A fragment of (I, ks)
        //      once and feeds up to SEGW independent accumulators with it.  A part holds <= WPB*NSEG segments, dealt to
        //      the waves longest-first; two copies of its tile image must fit the LDS budget.
        const int np = (int)pairs.size();
        const int SEGCAP = FBR_WPB * FBR_NSEG;
        const int IMG_BUDGET = FBR_IMG_BUDGET;  // doubles per image buffer (see FbrGramConfig)
        struct Seg { int I; std::vector<int> pr; int w; int kb; };
        struct Plan {
            std::vector<std::vector<Seg>> ws;  // per wave: its row segments
            int load[FBR_WPB];                 // cost units per wave (incl. the late-wave penalty)
            int maxload, mfma, mfma_odd, img;  // mfma: MFMAs of an even sample (all k-steps), mfma_odd: of an odd one
            double cost;
        };
        std::vector<int> stamp(NT, -1);
        int stamp_gen = 0;
        std::vector<std::pair<int, int>> order_buf;
        // plan of the part pairs[b..e): false if it does not fit (materialise: also the wave -> segments -> pairs lists).  Waves 4..7 share their SIMDs with the older waves 0..3, which
        // the arbiter favours: measured, they finish the MFMA phase about 10 cost units later at equal load.
        std::vector<int> sg_begin, sg_w, sg_order;  // row segments of the part being planned: ranges of order_buf, cost units
        auto plan_part = [&](int b, int e, Plan &pl, bool materialise) -> bool {
            stamp_gen++;
            int img = 0;
            auto need = [&](int ti) {
                if (stamp[ti] != stamp_gen) {
                    stamp[ti] = stamp_gen;
                    img += (tiles[ti].depth + 3) / 4 * 4 * FBR_TILE;
                }
            };
            order_buf.clear();
            for (int i = b; i < e; i++) {
                need(pairs[i].I);
                need(pairs[i].J);
                order_buf.emplace_back(pairs[i].I, i);
            }
            if (img + 4 * FBR_TILE > IMG_BUDGET) return false;
            // a row segment: pairs of one tile I that start at the same k-step, sorted by their last k-step descending (the
            // kernel runs k-step ks for the first n(ks) pairs of the segment, n falling)
            std::stable_sort(order_buf.begin(), order_buf.end(), [&](const std::pair<int, int> &x, const std::pair<int, int> &y) {
                if (x.first != y.first) return x.first < y.first;
                const FbrPair &px = pairs[x.second], &py = pairs[y.second];
                if (px.kbegin() != py.kbegin()) return px.kbegin() < py.kbegin();
                if (px.nkend() != py.nkend()) return px.nkend() > py.nkend();
                return px.mode < py.mode;
            });
            sg_begin.clear();
            sg_w.clear();
            pl.mfma = 0;
            pl.mfma_odd = 0;
            for (size_t o = 0; o < order_buf.size();) {
                const int I = order_buf[o].first, kb = pairs[order_buf[o].second].kbegin();
                int c = 3, nkmax = 0, cnt = 0;
                sg_begin.push_back((int)o);
                while (o < order_buf.size() && order_buf[o].first == I && pairs[order_buf[o].second].kbegin() == kb && cnt < FBR_SEGW) {
                    const FbrPair &pr = pairs[order_buf[o].second];
                    c += 2 * (pr.nkend() - kb);  // one MFMA = 2 cost units
                    pl.mfma += pr.nkend() - kb;
                    pl.mfma_odd += std::max(0, pr.nkend() - std::max(kb, base_ks));
                    nkmax = std::max(nkmax, pr.nkend() - kb);
                    cnt++;
                    o++;
                }
                sg_w.push_back(c + 3 * nkmax);  // measured fixed cost: segment preamble + per-k-step A / row-map fetch
            }
            const int nsg = (int)sg_w.size();
            if (nsg > SEGCAP) return false;
            sg_begin.push_back((int)order_buf.size());
            sg_order.resize(nsg);
            for (int i = 0; i < nsg; i++) sg_order[i] = i;
            std::stable_sort(sg_order.begin(), sg_order.end(), [&](int x, int y) { return sg_w[x] > sg_w[y]; });
            int cnt[FBR_WPB] = {0};
            if (materialise) pl.ws.assign(FBR_WPB, {});
            for (int w = 0; w < FBR_WPB; w++) pl.load[w] = w >= FBR_WPB / 2 ? 10 : 0;
            for (int si : sg_order) {  // longest processing time first
                int best = -1;
                for (int w = 0; w < FBR_WPB; w++)
                    if (cnt[w] < FBR_NSEG && (best < 0 || pl.load[w] < pl.load[best])) best = w;
                pl.load[best] += sg_w[si];
                cnt[best]++;
                if (materialise) {
                    Seg sg{order_buf[sg_begin[si]].first, {}, sg_w[si], pairs[order_buf[sg_begin[si]].second].kbegin()};
                    for (int o = sg_begin[si]; o < sg_begin[si + 1]; o++) sg.pr.push_back(order_buf[o].second);
                    pl.ws[best].push_back(std::move(sg));
                }
            }
            pl.maxload = 0;
            for (int w = 0; w < FBR_WPB; w++)
                if (cnt[w]) pl.maxload = std::max(pl.maxload, pl.load[w]);
            pl.img = img;
            pl.cost = cfg.c0 + cfg.cload * pl.maxload + cfg.cmfma * pl.mfma + cfg.cimg * img;
            return true;
        };
        // The cuts minimise the summed modelled cost of the parts (dynamic programme over the cut positions): workgroups are
        // dealt to the parts in proportion to their cost, so the pass time is proportional to that sum.  It trades the
        // number of parts (fixed cost per sample and part), the image volume and the balance of the waves inside a part
        // (the per-sample barrier waits for the most loaded wave).
        std::vector<int> part_begin;
        {
            const double INF = 1e300;
            std::vector<double> best(np + 1, INF);
            std::vector<int> from(np + 1, -1);
            best[0] = 0.0;
            Plan pl;
            for (int i = 1; i <= np; i++)
                for (int j = i - 1; j >= std::max(0, i - PPB); j--) {
                    if (!plan_part(j, i, pl, false)) {
                        if (i - j == 1) throw std::runtime_error("a single tile pair exceeds the LDS image budget (too many rows per sample)");
                        break;  // a longer part ending at i cannot fit either
                    }
                    if (best[j] + pl.cost < best[i]) {
                        best[i] = best[j] + pl.cost;
                        from[i] = j;
                    }
                }
            for (int i = np; i > 0; i = from[i]) part_begin.push_back(i);
            part_begin.push_back(0);
            std::reverse(part_begin.begin(), part_begin.end());
            T = std::max(1, (int)part_begin.size() - 1);
            if (np == 0) part_begin.assign(2, 0);
        }
        // ---- segments and slots: LPT assignment of each part's segments to its waves; part-local images, DMA pieces
        slots.assign((size_t)T * PPB, FbrSlot{-1, 0});
        part_tiles.assign(T, {});
        part_tile_off.assign(T, std::vector<int>(NT, -1));
        pieces.assign(T, {});
        part_image.assign(T, 0);
        part_load.assign(T, 0);
        part_mfma.assign(T, 0);
        part_cost.assign(T, 0.0);
        part_image_max = 0;
        for (int t = 0; t < T; t++) {
            Plan pl;
            if (part_begin[t + 1] > part_begin[t]) {
                if (!plan_part(part_begin[t], part_begin[t + 1], pl, true)) throw std::runtime_error("internal: part does not fit");
                for (int w = 0; w < FBR_WPB; w++)
                    for (int sgi = 0; sgi < (int)pl.ws[w].size(); sgi++) {
                        const Seg &sgm = pl.ws[w][sgi];
                        for (size_t j = 0; j < sgm.pr.size(); j++)
                            slots[((size_t)t * FBR_WPB + w) * FBR_NPW + sgi * FBR_SEGW + j] = FbrSlot{sgm.pr[j], sgm.kb};
                        bool uni = true;
                        for (int pi : sgm.pr)
                            uni = uni && pairs[pi].nkend() == pairs[sgm.pr[0]].nkend() && (pairs[pi].mode == 1) == (pairs[sgm.pr[0]].mode == 1);
                        if (uni) for (int pi : sgm.pr) mfma_uniform += pairs[pi].nkend() - sgm.kb;
                    }
                part_load[t] = pl.maxload;
                part_mfma[t] = pl.mfma;
                mfma_pair += pl.mfma + pl.mfma_odd;
            }
            std::vector<char> need(NT, 0);
            for (int i = part_begin[t]; i < part_begin[t + 1]; i++) need[pairs[i].I] = need[pairs[i].J] = 1;
            int loff = 0;
            for (int ti = 0; ti < NT; ti++) {
                if (!need[ti]) continue;
                part_tiles[t].push_back(ti);
                part_tile_off[t][ti] = loff;
                loff += (tiles[ti].depth + 3) / 4 * 4 * FBR_TILE;
            }
            part_image[t] = loff;
            part_cost[t] = cfg.c0 + cfg.cload * part_load[t] + cfg.cmfma * part_mfma[t] + cfg.cimg * loff;
            part_image_max = std::max(part_image_max, loff + 4 * FBR_TILE);
            // DMA pieces over maximal runs of tiles that are adjacent both in the global and the local image
            size_t i = 0;
            const std::vector<int> &pt = part_tiles[t];
            while (i < pt.size()) {
                size_t j = i;
                int run = 0;
                while (j < pt.size() && (j == i || pt[j] == pt[j - 1] + 1)) {
                    run += (tiles[pt[j]].depth + 3) / 4 * 4 * FBR_TILE;
                    j++;
                }
                int g0 = tiles[pt[i]].off, l0 = part_tile_off[t][pt[i]];
                int o = 0;
                while (run - o >= 128) {
                    pieces[t].push_back({g0 + o, l0 + o, 0});
                    o += 128;
                }
                if (run - o == 64) pieces[t].push_back({g0 + o, l0 + o, 1});
                else if (run - o != 0) throw std::runtime_error("internal: tile size not a multiple of 64 doubles");
                i = j;
            }
        }
        mfma_per_sample = (mfma_pair + 1) / 2;
    }
};


// Deal `slots` workgroups (>= T) to the parts so that the slowest workgroup is as fast as possible: every part gets one, each
// further one goes to the part with the largest cost per workgroup.  Returns n[part]; deterministic.
// base_only: the launch stops every pair after the base k-steps (row masks that switch the joint rows off): a part then costs its fixed
// overhead, its image and base-k-step MFMAs in proportion to its pairs
static inline std::vector<int> fbr_gram_deal(const FbrGramProgram &gp, int slots, bool base_only = false)
{
    std::vector<double> cost = gp.part_cost;
    if (base_only) {
        const int npw = gp.cfg.npw();
        for (int p = 0; p < gp.T; p++) {
            int pairs = 0, wmax = 0;
            for (int w = 0; w < FBR_WPB; w++) {
                int pw = 0;
                for (int i = 0; i < npw; i++) pw += gp.slots[((size_t)p * FBR_WPB + w) * npw + i].pair >= 0;
                pairs += pw;
                wmax = std::max(wmax, pw);
            }
            const double ks = 1.5;  // base k-steps per pair and sample (pairs of samples share three)
            cost[p] = gp.cfg.c0 + gp.cfg.cload * (2.0 * ks * wmax) + gp.cfg.cmfma * (ks * pairs) + gp.cfg.cimg * gp.part_image[p];
        }
    }
    std::vector<int> n(gp.T, 1);
    for (int left = slots - gp.T; left > 0; left--) {
        int best = 0;
        for (int p = 1; p < gp.T; p++)
            if (cost[p] * n[best] > cost[best] * n[p]) best = p;
        n[best]++;
    }
    return n;
}

// Program for the shape that suits the model: small images / two workgroups per CU unless that splits the pairs into too many parts.
// few rhs columns (the reference's one: tau) and one pack thread per column: their moments come from the pack kernel
static inline bool fbr_gram_rhs_moments(const FbrHostModel &hm, int k, bool rhs_tile_forced = false)
{
    return k >= 1 && k <= 2 && hm.cols <= 255 && !rhs_tile_forced;
}

// shape: 0 = by model, 1 = one workgroup per CU, 2 = two per CU (option "gram_shape"); orient: option "gram_orient"
static inline void fbr_gram_build_best(FbrGramProgram &gp, const FbrHostModel &hm, int k, int shape = 0, bool rhs_tiles = true, bool orient = true)
{
    gp.rhs_tiles = rhs_tiles;
    FbrGramConfig two = FBR_CFG_TWO_PER_CU, one = FBR_CFG_ONE_PER_CU;
    // the order of the pair list decides which tiles a contiguous part touches: a few block edges are tried and the one with
    // the lowest modelled cost kept (WALK-MAN: edge 6 instead of 5 saves one part and 7 % of the image traffic)
    auto build_shape_plain = [&](const FbrGramConfig &cfg) {
        gp.block_edge = 0;
        gp.build(hm, k, cfg);
        if (gp.T == 1) return;
        double best = gp.total_cost();
        int best_edge = 0;
        for (int edge = 4; edge <= 8; edge++) {
            gp.block_edge = edge;
            gp.build(hm, k, cfg);
            if (gp.total_cost() < best) {
                best = gp.total_cost();
                best_edge = edge;
            }
        }
        if (gp.block_edge != best_edge) {
            gp.block_edge = best_edge;
            gp.build(hm, k, cfg);
        }
    };
    // a program of several parts is also built with its pairs turned for full row segments, and kept when it needs fewer parts
    auto build_shape = [&](const FbrGramConfig &cfg) {
        gp.orient = false;
        build_shape_plain(cfg);
        if (gp.T <= 1 || !orient) return;
        FbrGramProgram keep = gp;
        gp.orient = true;
        build_shape_plain(cfg);
        if (!gp.oriented || gp.T >= keep.T) gp = keep;
    };
    if (shape == 1) {
        build_shape(one);
        return;
    }
    build_shape(two);
    if (shape != 2 && gp.T > FBR_MAX_PARTS_TWO_PER_CU) build_shape(one);
}

