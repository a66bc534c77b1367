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
#include <stdexcept>
#include <string>
#include <vector>

#define FBR_TILE 16
#define FBR_WPB 4         // waves per workgroup of the Gram kernel
#define FBR_NPW 20        // tile pairs (MFMA accumulators) per wave
#define FBR_MAX_RHS 16

struct FbrCol {
    int kind;  // 0 inertial, 1 friction
    int link;  // inertial: link index
    int pidx;  // inertial: parameter 0..9 ; friction: fkind (0 Fc,1 Fv,2 Fv+,3 Fv-,4 off,5 Fs)
    int joint; // friction: dof index
};

struct FbrHostModel {
    int L = 0, n = 0, fb = 0, rows = 0, cols = 0, cpl = 10;
    int floating = 0, fric = 0, fric_sym = 0, grav_only = 0;
    double stribeck = 0.0;
    double gravity[3] = {0, 0, -9.81};
    std::vector<int> order, parent, dof;
    std::vector<double> restR, restp, axis;
    std::vector<std::vector<int>> path;  // per link: movable joints root -> link
    std::vector<FbrCol> coldesc;         // identified columns
    int maxdepth = 0;

    void build(int L_, int n_, const int32_t *parent_, const int32_t *dof_, const double *restR_, const double *restp_,
               const double *axis_, int floating_, const double *g, int fric_, int fric_sym_, int grav_only_,
               double stribeck_)
    {
        L = L_; n = n_; floating = floating_ ? 1 : 0; fric = fric_ ? 1 : 0; fric_sym = fric_sym_ ? 1 : 0;
        grav_only = grav_only_ ? 1 : 0; stribeck = stribeck_;
        if (L <= 0 || n < 0) throw std::runtime_error("bad model size");
        fb = floating ? 6 : 0;
        rows = n + fb;
        cpl = grav_only ? 4 : 10;
        for (int i = 0; i < 3; i++) gravity[i] = g[i];
        parent.assign(parent_, parent_ + L);
        dof.assign(dof_, dof_ + L);
        restR.assign(restR_, restR_ + 9 * L);
        restp.assign(restp_, restp_ + 3 * L);
        axis.assign(axis_, axis_ + 3 * L);
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
        // identified columns (model.py:134-168, 459-503)
        coldesc.clear();
        for (int l = 0; l < L; l++)
            for (int p = 0; p < cpl; p++) coldesc.push_back({0, l, p, -1});
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
    int rec_size() const { return 21 * L + 6 * n; }
    // standard-vector index of the first friction parameter (model.py:164-168)
    int friction_start() const { return grav_only ? 4 * L : 10 * L; }
};

struct FbrTile {
    int type = 0;                 // 0 chain, 1 dense
    int depth = 0;                // packed rows (chain: fb + path length, dense: rows)
    int off = 0;                  // offset (doubles) of the tile inside the per-sample LDS image
    int col[FBR_TILE];            // augmented column id per slot, -1 = padding
    std::vector<int> rowid;       // chain: global row of each packed position
    std::vector<int> tpath;       // chain: dof path of the deepest link
};
struct FbrPair {
    int I, J, common, mode;  // mode 0: both packed by position; 1: B rows looked up through rowid_I; 2: dense x dense
    int nk4() const { return (common + 3) / 4; }
};
struct FbrItem {  // one producer work item = one real column of one tile
    int off;      // LDS offset of (tile, slot): tile.off + slot
    int kind;     // 0 inertial, 1 friction, 2 rhs
    int a;        // inertial: link ; friction: joint ; rhs: rhs column
    int b;        // inertial: pidx ; friction: fkind
};
struct FbrSlot {  // one accumulator of one wave
    int pair;     // -1 = unused
};

struct FbrGramProgram {
    int k = 0, Pa = 0, NT = 0, T = 0;
    int rows_pad = 0;        // rows rounded up to a multiple of 4
    int image_doubles = 0;   // per-sample LDS image size (all tiles) incl. tail padding
    int maxrow = 0;          // max tile depth
    std::vector<FbrTile> tiles;
    std::vector<FbrPair> pairs;
    std::vector<std::vector<FbrItem>> items;  // per part
    std::vector<FbrSlot> slots;               // [T][WPB][NPW]
    int64_t mfma_per_sample = 0;

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

    void build(const FbrHostModel &hm, int k_)
    {
        k = k_;
        if (k < 0 || k > FBR_MAX_RHS) throw std::runtime_error("rhs column count must be 0..16");
        Pa = hm.cols + k;
        rows_pad = (hm.rows + 3) / 4 * 4;
        tiles.clear();
        pairs.clear();
        // ---- chain tiles over the inertial columns, links in traversal order
        {
            FbrTile cur;
            int fill = 0;
            bool open = false;
            auto close = [&]() {
                if (!open) return;
                for (int s = fill; s < FBR_TILE; s++) cur.col[s] = -1;
                tiles.push_back(cur);
                open = false;
            };
            for (int l : hm.order) {
                for (int p = 0; p < hm.cpl; p++) {
                    if (open && (fill == FBR_TILE || !nested(cur.tpath, hm.path[l]))) close();
                    if (!open) {
                        cur = FbrTile();
                        cur.type = 0;
                        cur.tpath = hm.path[l];
                        fill = 0;
                        open = true;
                    }
                    if (hm.path[l].size() > cur.tpath.size()) cur.tpath = hm.path[l];
                    cur.col[fill++] = hm.cpl * l + p;
                }
            }
            close();
        }
        for (auto &t : tiles) {
            t.depth = hm.fb + (int)t.tpath.size();
            t.rowid.clear();
            for (int i = 0; i < hm.fb; i++) t.rowid.push_back(i);
            for (int d : t.tpath) t.rowid.push_back(hm.fb + d);
        }
        // ---- dense tiles: friction columns then rhs columns
        {
            int c = hm.cpl * hm.L;
            while (c < Pa) {
                FbrTile t;
                t.type = 1;
                t.depth = hm.rows;
                for (int s = 0; s < FBR_TILE; s++) t.col[s] = (c < Pa) ? c++ : -1;
                tiles.push_back(t);
            }
        }
        NT = (int)tiles.size();
        int off = 0;
        maxrow = 0;
        for (auto &t : tiles) {
            t.off = off;
            int dp = (t.depth + 3) / 4 * 4;
            off += dp * FBR_TILE;
            maxrow = std::max(maxrow, dp);
        }
        image_doubles = off + 4 * FBR_TILE;  // tail padding: masked lanes may read one k-step past a tile
        // ---- pairs (I <= J)
        mfma_per_sample = 0;
        for (int I = 0; I < NT; I++)
            for (int J = I; J < NT; J++) {
                const FbrTile &a = tiles[I], &b = tiles[J];
                FbrPair p{I, J, 0, 0};
                if (a.type == 0 && b.type == 0) {
                    p.common = hm.fb + common_prefix(a.tpath, b.tpath);
                    p.mode = 0;
                } else if (a.type == 0 && b.type == 1) {
                    p.common = a.depth;
                    p.mode = 1;
                } else if (a.type == 1 && b.type == 1) {
                    p.common = hm.rows;
                    p.mode = 2;
                } else {
                    throw std::runtime_error("dense tile before chain tile");
                }
                if (p.common == 0) continue;  // structurally zero block (fixed base, disjoint branches)
                pairs.push_back(p);
                mfma_per_sample += p.nk4();
            }
        // ---- parts: contiguous chunks of the row-major pair list, balanced by k-steps
        const int PPB = FBR_WPB * FBR_NPW;
        const int np = (int)pairs.size();
        T = std::max(1, (np + PPB - 1) / PPB);
        std::vector<int> part_begin(T + 1, 0);
        {
            int64_t total = 0;
            for (auto &p : pairs) total += p.nk4();
            int idx = 0;
            int64_t done = 0;
            for (int t = 0; t < T; t++) {
                part_begin[t] = idx;
                int64_t target = (total - done) / (T - t);
                int64_t acc = 0;
                int cnt = 0;
                while (idx < np) {
                    int remaining_after = np - (idx + 1);
                    if (cnt >= PPB) break;
                    if (cnt > 0 && acc >= target && remaining_after + 1 <= (int64_t)(T - t - 1) * PPB) break;
                    acc += pairs[idx].nk4();
                    cnt++;
                    idx++;
                    // must leave no more than what the remaining parts can hold
                    (void)remaining_after;
                }
                // if the remaining pairs do not fit in the remaining parts, keep taking
                while (idx < np && (np - idx) > (int64_t)(T - t - 1) * PPB && cnt < PPB) {
                    acc += pairs[idx].nk4();
                    cnt++;
                    idx++;
                }
                done += acc;
            }
            part_begin[T] = np;
            if (idx != np) throw std::runtime_error("internal: pair partition failed");
        }
        // ---- slots: LPT assignment of each part's pairs to its waves
        slots.assign((size_t)T * PPB, FbrSlot{-1});
        items.assign(T, {});
        for (int t = 0; t < T; t++) {
            std::vector<int> idx;
            for (int i = part_begin[t]; i < part_begin[t + 1]; i++) idx.push_back(i);
            std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return pairs[x].nk4() > pairs[y].nk4(); });
            int load[FBR_WPB] = {0}, cnt[FBR_WPB] = {0};
            for (int i : idx) {
                int best = -1;
                for (int w = 0; w < FBR_WPB; w++)
                    if (cnt[w] < FBR_NPW && (best < 0 || load[w] < load[best])) best = w;
                slots[((size_t)t * FBR_WPB + best) * FBR_NPW + cnt[best]].pair = i;
                cnt[best]++;
                load[best] += pairs[i].nk4();
            }
            // producer items: every real column of every tile this part touches
            std::vector<char> need(NT, 0);
            for (int i = part_begin[t]; i < part_begin[t + 1]; i++) need[pairs[i].I] = need[pairs[i].J] = 1;
            for (int ti = 0; ti < NT; ti++) {
                if (!need[ti]) continue;
                for (int s = 0; s < FBR_TILE; s++) {
                    int c = tiles[ti].col[s];
                    if (c < 0) continue;
                    FbrItem it;
                    it.off = tiles[ti].off + s;
                    if (c >= hm.cols) {
                        it.kind = 2; it.a = c - hm.cols; it.b = 0;
                    } else if (hm.coldesc[c].kind == 0) {
                        it.kind = 0; it.a = hm.coldesc[c].link; it.b = hm.coldesc[c].pidx;
                    } else {
                        it.kind = 1; it.a = hm.coldesc[c].joint; it.b = hm.coldesc[c].pidx;
                    }
                    items[t].push_back(it);
                }
            }
        }
    }
};
