// fbr_tsqr_api.hip -- blocked Householder TSQR of libfbr (fbr_tsqr / _cols / _submit / _merge / _work_info, include/fbr.h; kernels in fbr_tsqr.h).
#define FBR_KERNELS_GROUPS
#include "fbr_internal.h"
#include "fbr_tsqr.h"

// ------------------------------------------------------------------------------------------------
// TSQR (fbr_tsqr.h)
// ------------------------------------------------------------------------------------------------
static FbrTsqrOpts topts(const fbr_model *m)
{
    FbrTsqrOpts o;
    o.narrow = m->opt.tsqr_narrow != 0;
    o.tree_one_wg = m->opt.tsqr_tree_one_wg != 0;
    o.timing = m->opt.tsqr_timing != 0;
    o.short_calls = m->opt.tsqr_short_call_factors != 0;
    o.narrow_tall = m->opt.tsqr_narrow_tall != 0;
    return o;
}
// fbr_tsqr_begin with the model's options
static int tsqr_begin(const fbr_model *m, FbrTsqrWork &wk, hipStream_t st, int Pa, const double *R_in, int num_cus, long rows_hint, unsigned *shared_err = nullptr)
{
    wk.opts = topts(m);
    return fbr_tsqr_begin(wk, st, Pa, R_in, num_cus, rows_hint, shared_err);
}

// first column (in the order of the factorised columns) in which regressor row r can be non-zero: base-wrench rows meet every
// inertial column, the row of joint d the columns of the links below d and its own friction columns; Psel = only the rhs columns
static std::vector<int> tsqr_first_cols(const FbrHostModel &hm, const int32_t *cols, int Psel)
{
    std::vector<int> fc(hm.rows, Psel);
    for (int r = 0; r < hm.rows; r++)
        for (int c = 0; c < Psel; c++) {
            const FbrCol &cd = hm.coldesc[cols ? cols[c] : c];
            bool on;
            if (r < hm.fb)
                on = cd.kind == 0;
            else if (cd.kind == 0)
                on = std::find(hm.path[cd.link].begin(), hm.path[cd.link].end(), r - hm.fb) != hm.path[cd.link].end();
            else
                on = cd.joint == r - hm.fb;
            if (on) {
                fc[r] = c;
                break;
            }
        }
    return fc;
}

// Column order of a factorisation.  R^T R = A^T A holds for any column order of A, and a block of one regressor row is folded from
// the first column it can touch (tsqr_first_cols): with the inertial columns ordered by the DEPTH of their link (number of movable
// joints above it), every joint row starts behind all shallower links.  WALK-MAN: the folds run 0.44 instead of 0.55 of the dense
// tile updates and 0.60 instead of 0.71 of the panel chains.  The factor is computed in that order and brought back to the caller's
// column order by one small re-triangularisation (QR of the column-permuted n x n factor).  Friction columns keep their place behind
// the inertial ones.
struct TsqrPlan {
    int Psel = 0, Pa = 0;
    bool reorder = false;
    std::vector<int> fcols;    // [Psel] regressor column of factor column j
    std::vector<int> perm;     // [Pa]   caller's factor column of internal factor column j (rhs columns: identity)
    std::vector<int> inv;      // [Pa]   internal position of the caller's column j
    std::vector<int> linkpos;  // [L]    (all columns, no subset) block position of every link's columns
    std::vector<int> fc;       // [rows] first supported internal column of every regressor row
};
static long tsqr_plan_work(const std::vector<int> &fc, int n)
{
    long w = 0;
    const int NP = n / 16;
    for (int f : fc) {
        const long np_ = NP - std::min(f, n) / 16;
        w += np_ * (np_ - 1) / 2 + np_;
    }
    return w;
}
static TsqrPlan tsqr_plan(const FbrHostModel &hm, const int32_t *cols, int32_t ncols, int k, long S, bool allow_reorder)
{
    TsqrPlan p;
    p.Psel = cols ? ncols : hm.cols;
    p.Pa = p.Psel + k;
    const int n = (p.Pa + 15) & ~15;
    std::vector<int> ucols(p.Psel);
    for (int j = 0; j < p.Psel; j++) ucols[j] = cols ? cols[j] : j;
    std::vector<int> order(p.Psel);
    for (int j = 0; j < p.Psel; j++) order[j] = j;
    auto depth = [&](int j) { return hm.coldesc[ucols[j]].kind == 0 ? (int)hm.path[hm.coldesc[ucols[j]].link].size() : (1 << 20); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depth(a) < depth(b); });
    std::vector<int> sorted(p.Psel);
    for (int j = 0; j < p.Psel; j++) sorted[j] = ucols[order[j]];
    const std::vector<int> fc_user = tsqr_first_cols(hm, ucols.data(), p.Psel), fc_sorted = tsqr_first_cols(hm, sorted.data(), p.Psel);
    // worth it for wide factors and enough rows to pay for the final n x n re-triangularisation
    p.reorder = allow_reorder && n > 16 * FBR_TSQR_NARROW_MAX_TILES && S * (long)hm.rows >= 64L * n &&
                tsqr_plan_work(fc_sorted, n) * 100 < tsqr_plan_work(fc_user, n) * 97;
    p.perm.resize(p.Pa);
    p.inv.resize(p.Pa);
    for (int j = 0; j < p.Pa; j++) p.perm[j] = (p.reorder && j < p.Psel) ? order[j] : j;
    for (int j = 0; j < p.Pa; j++) p.inv[p.perm[j]] = j;
    p.fcols = p.reorder ? sorted : ucols;
    p.fc = p.reorder ? fc_sorted : fc_user;
    if (!cols && !hm.masked) {
        p.linkpos.assign(hm.L, 0);
        for (int l = 0; l < hm.L; l++) p.linkpos[l] = p.inv[hm.cpl * l] / hm.cpl;
    }
    return p;
}

// ------------------------------------------------------------------------------------------------
// Tree-structured TSQR.  The row of joint d is non-zero only in the columns of the links below d (and its own friction columns), and
// R = qr(A) can be assembled from the factors of any partition of the ROWS.  The rows are therefore grouped along the kinematic
// tree -- the base-wrench rows, and one group per unbranched chain of joints (cut wherever the parent has more than one child joint)
// -- and every group is factorised over the columns its rows can touch only: WALK-MAN's leg joints fold 6 rows x 61 columns, its arm
// joints 7 x 81, the head 2 x 31, the waist 3 x 221 and only the 6 base rows all 481 (0.21 of the dense tile updates instead of the
// 0.44 of one factorisation with depth-ordered columns, and a third of the chunk bytes).  The group factors are embedded into the
// caller's column order and folded into the final factor like data rows.  Within a group the columns are ordered by link depth, so
// a joint row still starts at the first column of its own links.
// ------------------------------------------------------------------------------------------------
struct TsqrGroup {
    std::vector<int> rows;  // regressor rows of the group (slot order)
    std::vector<int> sel;   // factor columns of the group: indices into the caller's selected columns, in the group's order
    std::vector<int> fc;    // per slot: first supported column (group order)
    int Pa = 0;             // sel.size() + k
};
struct TsqrGroupPlan {
    std::vector<TsqrGroup> groups;
    std::vector<int> rowgroup, rowslot;  // per regressor row (-1: the row touches nothing that is factorised)
    bool masked = false;  // some regressor row has weight 0 for every sample and is left out
    int main = -1;  // group whose rows are dense in every factorised column (base-wrench rows): factorised in the caller's column order
                    // straight into the final factor, the other groups' factors are folded into it
};
static TsqrGroupPlan tsqr_group_plan(const FbrHostModel &hm, const int32_t *cols, int32_t ncols, int k, const std::vector<char> *active = nullptr,
                                     bool m_force_group = true)
{
    TsqrGroupPlan gp;
    const int Psel = cols ? ncols : hm.cols;
    // joint tree: parent joint of joint d (-1: hangs off the base), number of child joints of every joint (index 0: the base)
    std::vector<int> pj(hm.n, -1), depth(hm.n, 0), nchild(hm.n + 1, 0);
    for (int l = 0; l < hm.L; l++) {
        const int d = hm.dof[l];
        if (d < 0) continue;
        const std::vector<int> &pa = hm.path[l];
        depth[d] = (int)pa.size();
        pj[d] = pa.size() >= 2 ? pa[pa.size() - 2] : -1;
    }
    for (int d = 0; d < hm.n; d++) nchild[pj[d] + 1]++;
    std::vector<int> jgroup(hm.n, -1);
    int ngroups = 0, base_group = -1;
    if (hm.fb) base_group = ngroups++;
    std::vector<int> order(hm.n);
    for (int d = 0; d < hm.n; d++) order[d] = d;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
    for (int d : order) {
        const int p = pj[d];
        int pg = p < 0 ? base_group : jgroup[p];
        if (nchild[p + 1] == 1 && pg < 0) pg = base_group = ngroups++;  // fixed base, single chain from the root
        jgroup[d] = (nchild[p + 1] == 1) ? pg : ngroups++;
    }
    // The FORCE rows of the base wrench (rows 0 .. 2 of a floating base) are non-zero only in the columns that produce a force -- a link's
    // mass and first moments (an inertia entry is a pure moment) --: a group of their own, factorised over those columns (WALK-MAN, regrouped:
    // 62 of 213), leaves the dense group the three moment rows: the widest group folds half the rows (round 6; option tsqr_force_group)
    const bool force_split = hm.fb == 6 && m_force_group;
    const int force_group = force_split ? ngroups++ : -1;
    std::vector<std::vector<int>> grows(ngroups);
    auto on = [&](int r) { return !active || (*active)[r]; };  // rows switched off by the weights belong to no group
    for (int r = 0; r < hm.fb; r++)
        if (on(r)) grows[(force_split && r < 3) ? force_group : base_group].push_back(r);
    for (int d = 0; d < hm.n; d++)
        if (on(hm.fb + d)) grows[jgroup[d]].push_back(hm.fb + d);
    for (int r = 0; r < hm.rows; r++) gp.masked = gp.masked || !on(r);
    auto touches = [&](int r, int uc) {
        const FbrCol &cd = hm.coldesc[uc];
        if (cd.kind != 0) return cd.joint == r - hm.fb;
        if (force_split && r < 3) return cd.pidx < 4;
        if (r < hm.fb) return true;
        const std::vector<int> &pa = hm.path[cd.link];
        return std::find(pa.begin(), pa.end(), r - hm.fb) != pa.end();
    };
    gp.rowgroup.assign(hm.rows, -1);
    gp.rowslot.assign(hm.rows, -1);
    for (int g = 0; g < ngroups; g++) {
        TsqrGroup G;
        if (grows[g].empty()) continue;
        std::vector<int> inert, fric;
        for (int j = 0; j < Psel; j++) {
            const int uc = cols ? cols[j] : j;
            bool any = false;
            for (int r : grows[g]) any = any || touches(r, uc);
            if (any) (hm.coldesc[uc].kind == 0 ? inert : fric).push_back(j);
        }
        // (the unpaired columns of a model with column masks go behind the paired ones: pairs stay at even positions in every group)
        auto cdepth = [&](int j) {
            const FbrCol &cd = hm.coldesc[cols ? cols[j] : j];
            return (int)hm.path[cd.link].size() + (cd.joint == -2 ? (1 << 16) : 0);
        };
        std::stable_sort(inert.begin(), inert.end(), [&](int a, int b) { return cdepth(a) < cdepth(b); });
        G.sel = inert;
        G.sel.insert(G.sel.end(), fric.begin(), fric.end());
        G.Pa = (int)G.sel.size() + k;
        if (G.Pa == 0) continue;
        // slots: rows with the widest support first (their blocks start at the left-most panels)
        G.rows = grows[g];
        auto first = [&](int r) {
            for (size_t i = 0; i < G.sel.size(); i++)
                if (touches(r, cols ? cols[G.sel[i]] : G.sel[i])) return (int)i;
            return (int)G.sel.size();
        };
        std::stable_sort(G.rows.begin(), G.rows.end(), [&](int a, int b) { return first(a) < first(b); });
        bool dense = (int)G.sel.size() == Psel;
        for (size_t i = 0; i < G.rows.size(); i++) {
            G.fc.push_back(first(G.rows[i]));
            dense = dense && G.fc.back() == 0;
            gp.rowgroup[G.rows[i]] = (int)gp.groups.size();
            gp.rowslot[G.rows[i]] = (int)i;
        }
        if (dense && gp.main < 0) {
            gp.main = (int)gp.groups.size();
            std::sort(G.sel.begin(), G.sel.end());  // = the caller's order
        }
        gp.groups.push_back(std::move(G));
    }
    return gp;
}
// groups pay when the tree branches and there are enough rows to keep every group's workers busy
static bool tsqr_use_groups(const fbr_model *m, const TsqrGroupPlan &gp, long S)
{
    const long min_s = (long)m->opt.tsqr_group_min_samples;  // (tests force the path at small sizes) default 24000: measured on WALK-MAN, groups vs one factorisation: 16 k samples 16 vs 15.8 ms, 32 k 18.5 vs 21.4, 64 k 24 vs 32, 125 k 34 vs 52
    // (a chain on a FIXED base has ONE group and keeps the plain path; on a floating base the force rows are a second group -- tsqr_force_group --
    // and the call takes the row-group path: left arm 500 k samples 3.8 instead of 7.3 ms.  Measured, round 6, with ONE group: sending it through the row-group path for the sake
    // of the lane writer -- column-major chunks -- costs the wave-private level-0 kernel more than the writer saves: left arm 500 k samples
    // 8.5 instead of 7.3 ms (folds 6.6 instead of 5.7 ms: one wave per SIMD cannot hide the 16-lines-per-instruction block loads), KUKA 4.07
    // instead of 4.23)
    return (gp.groups.size() > 1 || (gp.masked && !gp.groups.empty())) && S >= min_s && m->opt.tsqr_groups != 0;
}
// Samples per chunk of a call over S samples, and (*lcm_out) the block granularity: every chunk is a whole number of fold blocks per
// regressor row in every group.  The chunks are cut EVENLY (a call that exceeds the memory-sized chunk by a few samples used to end with
// a chunk of a handful of samples that cost a dozen launches: 0.66 of the 10.3 ms of a 125 k-sample WALK-MAN call), a call up to 5 %
// longer than one chunk stays one chunk, and the last chunk is padded to the granularity with zero rows (tsqr_groups_impl).
static long tsqr_group_chunk_samples(const fbr_model *m, const TsqrGroupPlan &gp, long S, long *lcm_out = nullptr)
{
    double per = 0.0;  // chunk bytes per sample over all groups
    long lcm = 1;
    for (const TsqrGroup &G : gp.groups) {
        FbrTsqrShape sh;
        if (fbr_tsqr_shape(G.Pa, m->num_cus, 1L << 40, &sh, topts(m))) return -1;
        per += 8.0 * (double)G.rows.size() * sh.n;
        lcm = std::lcm(lcm, (long)sh.mb);
    }
    long ch = std::max(1L, (long)(4.0 * 1024 * 1024 * 1024 / per));
    ch = std::min(ch, chunk_size(m, 1L << 40));  // (the memory-sized chunk; chunk_size caps at the call's own length otherwise)
    if (m->opt.chunk_samples >= 1) ch = (long)m->opt.chunk_samples;
    ch = std::max(lcm, ch - ch % lcm);
    if (lcm_out) *lcm_out = lcm;
    if (S <= 0) return ch;
    const long nch = std::max(1L, (long)std::ceil((double)S / (1.05 * (double)ch)));
    const long even = (S + nch - 1) / nch;
    return (even + lcm - 1) / lcm * lcm;  // whole blocks per slot in every group
}

// Device tables of a TSQR call: assembled in pinned host memory that belongs to the call's ticket parity and copied asynchronously on
// `st` -- no host wait, and the tables of the submission before (other parity) stay intact while it is still running.
static int tsqr_upload_tables(fbr_model *m, int par, const std::vector<std::pair<const void *, size_t>> &pieces, const std::vector<size_t> &offs,
                              size_t total, hipStream_t st, const char **dev)
{
    total = std::max<size_t>(total, 16);
    if (m->tsqr_tab_host_bytes[par] < total) {
        if (m->tsqr_tab_host[par]) (void)hipHostFree(m->tsqr_tab_host[par]);
        m->tsqr_tab_host[par] = nullptr;
        m->tsqr_tab_host_bytes[par] = 0;
        HIPCHK(hipHostMalloc(&m->tsqr_tab_host[par], total + total / 2, hipHostMallocDefault));
        m->tsqr_tab_host_bytes[par] = total + total / 2;
    }
    int rc = m->tsqr_tab[par].ensure(total);
    if (rc) return rc;
    for (size_t i = 0; i < pieces.size(); i++)
        if (pieces[i].second) memcpy((char *)m->tsqr_tab_host[par] + offs[i], pieces[i].first, pieces[i].second);
    HIPCHK(hipMemcpyAsync(m->tsqr_tab[par].p, m->tsqr_tab_host[par], total, hipMemcpyHostToDevice, st));
    *dev = (const char *)m->tsqr_tab[par].p;
    return FBR_OK;
}

// overlap: the call follows a TSQR submission that is still running: its prologue (tables, kinematics and the writer of the first chunk)
// goes to the producer stream and waits only for the LAST LEVEL-0 FOLD of that submission -- it runs beside the submission's merge trees,
// which occupy a handful of CUs (7.7 of WALK-MAN's 8.2 ms of trees hide 5.5 + 1.2 ms of kinematics and first writer).
static int tsqr_groups_impl(fbr_model *m, const DevStates &d, const TsqrGroupPlan &gp, const int32_t *cols, int Psel, int k, const double *drhs,
                            const double *dw, const double *Rin_dev, double *R, int par, bool overlap)
{
    const FbrHostModel &hm = m->hm;
    const long S = d.S;
    const int G = (int)gp.groups.size(), Pa = Psel + k;
    int rc;
    auto tsqr_fail = [&](int code, const char *what) {
        set_err(std::string(what) + ": " + fbr_tsqr_error());
        return code == -4 ? FBR_E_UNSUPPORTED : (code == -3 ? FBR_E_HIP : FBR_E_INVALID);
    };
    if ((int)m->tsqr_groups.size() < G) m->tsqr_groups.resize(G);
    long lcm = 1;
    const long ch = tsqr_group_chunk_samples(m, gp, S, &lcm);
    if (ch < 0) return tsqr_fail(-4, "tsqr group shape");
    // device tables: ints [rowgroup | rowslot | entry ranges (cols + 1) x 2 | per group: slot first columns | per group: embedding (Pa)],
    // then the entry lists (int4) and the FbrDevGroup records
    std::vector<int> tab;
    tab.insert(tab.end(), gp.rowgroup.begin(), gp.rowgroup.end());
    tab.insert(tab.end(), gp.rowslot.begin(), gp.rowslot.end());
    // what every model column writes: one entry per row of every group that holds the column (variant 1: without the structural zeros
    // left of the row's first supported column tile)
    std::vector<int> gposv((size_t)G * hm.cols, -1);
    for (int g = 0; g < G; g++)
        for (size_t i = 0; i < gp.groups[g].sel.size(); i++) {
            const int j = gp.groups[g].sel[i];
            gposv[(size_t)g * hm.cols + (cols ? cols[j] : j)] = (int)i;
        }
    std::vector<int> ents[2];
    size_t o_ebeg[2];
    for (int var = 0; var < 2; var++) {
        o_ebeg[var] = tab.size();
        for (int c = 0; c < hm.cols; c++) {
            tab.push_back((int)ents[var].size());
            const FbrCol &cd = hm.coldesc[c];
            for (int r = 0; r < hm.rows; r++) {
                const int g = gp.rowgroup[r];
                if (g < 0) continue;
                const int pos = gposv[(size_t)g * hm.cols + c];
                if (pos < 0) continue;
                int kind;
                if (cd.kind == 0) {
                    if (r < hm.fb)
                        kind = 0;
                    else {
                        const std::vector<int> &pa = hm.path[cd.link];
                        kind = std::find(pa.begin(), pa.end(), r - hm.fb) != pa.end() ? 1 : 2;
                    }
                } else {
                    kind = cd.joint == r - hm.fb ? 3 : 2;
                }
                const int slot = gp.rowslot[r];
                if (var == 1 && kind == 2 && pos < (gp.groups[g].fc[slot] & ~15)) continue;
                ents[var].push_back(r | (kind << 8) | (pos << 10));
            }
        }
        tab.push_back((int)ents[var].size());
    }
    // the same lists per PAIR of adjacent inertial columns (16-byte stores, fbr_regressor_groups2_kernel): possible when both columns
    // of every pair sit side by side at an even position in every group that holds them
    const int npairs = hm.npaircols / 2;
    // threads per work item of the pair writer (fbr_regressor_groups2_kernel: 256 threads, an item's entries dealt to `wsplit` of them)
    const int wsplit = std::max(1, std::min(4, 256 / std::max(1, npairs + (hm.cols - 2 * npairs))));
    // (with fewer work items than half a workgroup -- the regrouped WALK-MAN: 92 pairs + 29 single columns -- the pair writer leaves
    // most threads idle behind twice the work per busy thread: 12.6 ms per 1 M samples with the entries split, 15.9 without, against
    // 11.8 ms of the one-column-per-thread writer)
    bool pairable = npairs > 0 && m->opt.tsqr_writer != 8 && (npairs + (hm.cols - 2 * npairs) >= 128 || m->opt.tsqr_writer == 16);
    std::vector<int> pents[2];
    size_t o_pbeg[2] = {0, 0};
    for (int var = 0; var < 2 && pairable; var++) {
        o_pbeg[var] = tab.size();
        for (int pr = 0; pr < npairs && pairable; pr++) {
            tab.push_back((int)pents[var].size());
            const int c = 2 * pr;
            const int ea = tab[o_ebeg[var] + c], eb = tab[o_ebeg[var] + c + 1], ec = tab[o_ebeg[var] + c + 2];
            pairable = hm.coldesc[c].kind == 0 && hm.coldesc[c + 1].kind == 0 && hm.coldesc[c].link == hm.coldesc[c + 1].link && eb - ea == ec - eb;
            for (int i = 0; i < eb - ea && pairable; i++) {
                const int x = ents[var][ea + i], y = ents[var][eb + i];
                pairable = (x & 0x3ff) == (y & 0x3ff) && (y >> 10) == (x >> 10) + 1 && ((x >> 10) & 1) == 0;
                pents[var].push_back(x);
            }
        }
        tab.push_back((int)pents[var].size());
    }
    // ---- the lane writer (fbr_kinid.h fbr_kinwrite_kernel, option tsqr_lane_writer): one lane per sample, kinematics fused in, chunks
    // written column-major.  Its entries carry the level of the row's joint on the link's path instead of a motion-vector lookup.
    const size_t lane_lds = ((size_t)3 * 64 * (std::max(hm.n, 1) | 1) + (dw ? (size_t)64 * (hm.rows | 1) : 0)) * sizeof(double);
    const bool lane_writer = m->opt.tsqr_lane_writer != 0 && m->opt.tsqr_writer == 0 && m->kinid.nsteps > 0 && lane_lds <= (size_t)(120 << 10) && Pa < 1024;
    // per destination slot of the lane writer: (regressor row, column position in the row's group), or row = -1: absent
    std::vector<std::pair<int, int>> lslots;
    size_t o_lrec = 0, o_lcol = 0, o_lsteps = 0;
    int lane_parts = 1, lane_slots = 1, lane_step0[FBR_KINWRITE_PARTS] = {0, 0, 0, 0}, lane_nsteps[FBR_KINWRITE_PARTS] = {0, 0, 0, 0};
    if (lane_writer) {
        o_lrec = tab.size();
        auto entry_of = [&](int c, int r, int kind, int *pos) {  // the writer entry (variant 1) of column c on row r, if any
            for (int e = tab[o_ebeg[1] + c]; e < tab[o_ebeg[1] + c + 1]; e++)
                if ((ents[1][e] & 0xff) == r && ((ents[1][e] >> 8) & 3) == kind) {
                    *pos = ents[1][e] >> 10;
                    return true;
                }
            return false;
        };
        for (int c = 0; c < hm.cols; c++) {
            const FbrCol &cd = hm.coldesc[c];
            if (tab[o_ebeg[1] + c] == tab[o_ebeg[1] + c + 1]) {  // no group holds the column
                tab.push_back(-1);
                tab.push_back(0);
                continue;
            }
            tab.push_back((int)lslots.size());
            int pos = 0, nz = 0;
            if (cd.kind == 0) {
                for (int r = 0; r < hm.fb; r++) lslots.push_back(entry_of(c, r, 0, &pos) ? std::make_pair(r, pos) : std::make_pair(-1, 0));
                for (int dj : hm.path[cd.link]) lslots.push_back(entry_of(c, hm.fb + dj, 1, &pos) ? std::make_pair(hm.fb + dj, pos) : std::make_pair(-1, 0));
            } else {
                lslots.push_back(entry_of(c, hm.fb + cd.joint, 3, &pos) ? std::make_pair(hm.fb + cd.joint, pos) : std::make_pair(-1, 0));
            }
            for (int e = tab[o_ebeg[1] + c]; e < tab[o_ebeg[1] + c + 1]; e++)
                if (((ents[1][e] >> 8) & 3) == 2) {
                    lslots.push_back({ents[1][e] & 0xff, ents[1][e] >> 10});
                    nz++;
                }
            tab.push_back(nz);
        }
        tab.push_back((int)lslots.size());  // pseudo-column `cols`: k rhs destinations per regressor row
        tab.push_back(0);
        for (int r = 0; r < hm.rows; r++)
            for (int i = 0; i < k; i++)
                lslots.push_back(gp.rowgroup[r] >= 0 ? std::make_pair(r, (int)gp.groups[gp.rowgroup[r]].sel.size() + i) : std::make_pair(-1, 0));
        // the tree in parts: the waves of a workgroup share one block of samples, each walks its links (+ the ancestors they need) and
        // writes the columns of its own links (fbr_kinid_build_parts); cost of a link: its kinematics + what its columns write
        std::vector<double> lcost(hm.L, 30.0);
        for (int c = 0; c < hm.ninert; c++)
            if (tab[o_lrec + 2 * c] >= 0) lcost[hm.coldesc[c].link] += 10.0 + (double)(hm.fb + hm.path[hm.coldesc[c].link].size() + tab[o_lrec + 2 * c + 1]);
        std::vector<FbrKinIdProgram> progs;
        std::vector<std::vector<char>> own;
        fbr_kinid_build_parts(hm, lcost, FBR_KINWRITE_PARTS, progs, own);
        lane_parts = (int)progs.size();
        o_lcol = tab.size();
        tab.resize(tab.size() + (size_t)lane_parts * 10 * hm.L, -1);
        for (int c = 0; c < hm.ninert; c++)
            for (int pq = 0; pq < lane_parts; pq++)
                if (own[pq][hm.coldesc[c].link]) tab[o_lcol + (size_t)pq * 10 * hm.L + 10 * hm.coldesc[c].link + hm.coldesc[c].pidx] = c;
        o_lsteps = tab.size();
        for (int pq = 0; pq < lane_parts; pq++) {
            lane_step0[pq] = (int)((tab.size() - o_lsteps) / FBR_KINID_STEP);
            lane_nsteps[pq] = progs[pq].nsteps;
            lane_slots = std::max(lane_slots, progs[pq].nslots);
            tab.insert(tab.end(), progs[pq].steps.begin(), progs[pq].steps.begin() + (size_t)progs[pq].nsteps * FBR_KINID_STEP);
        }
    }
    // LDS image of one sample's rows (fbr_regressor_groups_lds_kernel): offset of regressor row r, ld_g doubles each -- the padded width
    // of a group is only known once its factorisation has begun (below): the offsets are filled in there
    const size_t o_rowoff = tab.size();
    tab.resize(tab.size() + hm.rows, -1);
    const size_t o_nrows = tab.size();  // slots (regressor rows) of every group
    for (int g = 0; g < G; g++) tab.push_back((int)gp.groups[g].rows.size());
    const size_t o_gpa = tab.size();    // columns (rhs included) of every group
    for (int g = 0; g < G; g++) tab.push_back(gp.groups[g].Pa);
    std::vector<size_t> o_fc(G), o_emb(G);
    for (int g = 0; g < G; g++) {
        o_fc[g] = tab.size();
        tab.insert(tab.end(), gp.groups[g].fc.begin(), gp.groups[g].fc.end());
    }
    for (int g = 0; g < G; g++) {
        // column j of the final factor (caller's order) <- column emb[j] of the group factor, -1: not in the group
        o_emb[g] = tab.size();
        tab.resize(tab.size() + Pa, -1);
        const TsqrGroup &Gg = gp.groups[g];
        for (size_t i = 0; i < Gg.sel.size(); i++) tab[o_emb[g] + Gg.sel[i]] = (int)i;
        for (int i = 0; i < k; i++) tab[o_emb[g] + Psel + i] = (int)Gg.sel.size() + i;
    }
    while (tab.size() & 3) tab.push_back(0);
    const size_t o_ent0 = tab.size() * sizeof(int), o_ent1 = o_ent0 + ents[0].size() * sizeof(int);
    const size_t o_pent0 = o_ent1 + ents[1].size() * sizeof(int), o_pent1 = o_pent0 + (pairable ? pents[0].size() : 0) * sizeof(int);
    const size_t o_grp = (o_pent1 + (pairable ? pents[1].size() : 0) * sizeof(int) + 15) & ~(size_t)15;
    const size_t nlent = lslots.size() + 64;  // (padded: the kernel requests a record's destinations in fixed-size batches)
    const size_t o_lent = (o_grp + (size_t)G * sizeof(FbrDevGroup) + 15) & ~(size_t)15;  // lane writer: two sets of nlent destinations (64-bit)
    // working factors and chunk buffers of the groups
    std::vector<FbrDevGroup> hg(G);
    bool skipzeros = false;
    long mrows = 0;  // rows the final factor folds: the main group's data rows and the other groups' factors
    for (int g = 0; g < G; g++) mrows += g == gp.main ? S * (long)gp.groups[g].rows.size() : gp.groups[g].Pa;
    auto work = [&](int g) -> FbrTsqrWork & { return g == gp.main ? m->tsqr : m->tsqr_groups[g]; };
    for (int g = 0; g < G; g++) {
        const TsqrGroup &Gg = gp.groups[g];
        FbrTsqrWork &wk = work(g);
        if ((rc = tsqr_begin(m, wk, m->stream, Gg.Pa, g == gp.main ? Rin_dev : nullptr, m->num_cus, g == gp.main ? mrows : S * (long)Gg.rows.size(),
                                 m->tsqr_err)))
            return tsqr_fail(rc, "tsqr group begin");
    }
    // prologue stream: everything up to the first chunk's writer
    // (a stream confined to three quarters of the CUs: the prologue's kernels would otherwise fill every CU with their waves, and the
    // tree's eight-wave workgroups -- a whole CU's registers each -- could not be placed until they drain: measured, the first tree
    // level then takes 3.6 instead of 1.0 ms and nothing is gained)
    if (overlap && !m->tsqr_pro_stream) {
        const int words = (m->num_cus + 31) / 32;
        std::vector<uint32_t> mask(words, 0x00ffffffu);
        if (hipExtStreamCreateWithCUMask(&m->tsqr_pro_stream, (uint32_t)words, mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&m->tsqr_pro_stream, hipStreamNonBlocking));
        }
    }
    hipStream_t pst = overlap ? m->tsqr_pro_stream : m->stream;
    if (overlap) HIPCHK(hipStreamWaitEvent(pst, m->ev_tsqr_l0, 0));  // the chunk buffers and the kinematic records are free again
    for (int g = 0; g < G; g++) {
        const TsqrGroup &Gg = gp.groups[g];
        FbrTsqrWork &wk = work(g);
        double *A = nullptr;
        if ((rc = fbr_tsqr_chunk_buffer(wk, ch * (long)Gg.rows.size(), &A)) || (!lane_writer && (rc = fbr_tsqr_chunk_clean(wk, pst))))
            return tsqr_fail(rc, "tsqr group chunk");
        hg[g] = FbrDevGroup{A, wk.n, (int)Gg.sel.size()};
    }
    int img_total = 0;
    for (int r = 0; r < hm.rows; r++)
        if (gp.rowgroup[r] >= 0) {
            tab[o_rowoff + r] = img_total;
            img_total += hg[gp.rowgroup[r]].ld;
        }
    // the LDS-staged writer (option tsqr_writer = 32) needs a sample's rows to fit a third of the LDS beside the record
    const int naux = hm.rows * k + (dw ? hm.rows : 0) + (hm.fric ? hm.n : 0) + ((hm.fric && d.sign) ? hm.n : 0);
    const size_t lds_img = ((size_t)((hm.rec_size() + 1) & ~1) + ((naux + 1) & ~1) + ((hm.rows + 1) & ~1) + img_total) * sizeof(double) +
                           (size_t)((img_total / 2 + 3) & ~3) * sizeof(unsigned short) + ((size_t)3 * hm.rows + ents[1].size()) * sizeof(int);
    // (measured, round 5, regrouped WALK-MAN: the call of 1 M samples 54.95 instead of 55.88 ms, of 125 k samples 9.86 instead of 9.52 ms --
    // the writer is not bound by the width of its stores; the staged writer is therefore an option, not the default)
    const bool lds_writer = m->opt.tsqr_writer == 32 && hm.rows <= 255 && img_total > 0 && lds_img <= 64 * 1024 && hm.rec_size() <= 256 * 6 && naux <= 512 &&
                            hm.cols <= 512;
    if (m->opt.tsqr_writer == 32 && !lds_writer) {
        set_err("tsqr_writer = 32: the rows of a sample do not fit the LDS image of the staged writer");
        return FBR_E_UNSUPPORTED;
    }
    // chunk row count per slot of this call's chunks (every chunk but the last holds ch samples; the last is padded to the granularity):
    // the lane writer's column stride is fixed per chunk, its row pointers are rebuilt per chunk on the device side of the tables below
    // set 0: the full chunks (ch samples per slot), set 1: the last chunk (padded to the granularity): an entry's destination is the address of
    // (its row's slot, sample 0, its column) in the column-major chunk of the row's group
    std::vector<long long> lane_dst(2 * nlent, 0);
    const long last_cs = S > 0 ? S - (S - 1) / ch * ch : 0;
    const long csp_of[2] = {ch, (last_cs + lcm - 1) / lcm * lcm};
    if (lane_writer)
        for (int j = 0; j < 2; j++)
            for (size_t e = 0; e < lslots.size(); e++) {
                const int r = lslots[e].first, pos = lslots[e].second;
                if (r < 0) continue;
                const int g = gp.rowgroup[r];
                const long ldc = (long)gp.groups[g].rows.size() * csp_of[j];
                lane_dst[j * nlent + e] = (long long)(uintptr_t)(hg[g].A + (long)pos * ldc + (long)gp.rowslot[r] * csp_of[j]);
            }
    const char *dtab = nullptr;
    if ((rc = tsqr_upload_tables(m, par,
                                 {{tab.data(), tab.size() * sizeof(int)}, {ents[0].data(), ents[0].size() * sizeof(int)}, {ents[1].data(), ents[1].size() * sizeof(int)},
                                  {pents[0].data(), (pairable ? pents[0].size() : 0) * sizeof(int)}, {pents[1].data(), (pairable ? pents[1].size() : 0) * sizeof(int)},
                                  {hg.data(), G * sizeof(FbrDevGroup)}, {lane_dst.data(), lane_dst.size() * sizeof(long long)}},
                                 {0, o_ent0, o_ent1, o_pent0, o_pent1, o_grp, o_lent}, o_lent + lane_dst.size() * sizeof(long long), pst, &dtab)))
        return rc;
    const int *t = (const int *)dtab;
    const FbrDevGroup *dgrp = (const FbrDevGroup *)(dtab + o_grp);
    const size_t lds = (size_t)((hm.rec_size() + 1) & ~1) * sizeof(double) + (size_t)hm.rows * sizeof(double *);
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_groups_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_groups2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (lds_writer) HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_groups_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_img));
    // Merge trees are latency bound (a level of the full-width tree is 0.93 ms on a handful of workgroups, 8 levels over 256 private
    // factors).  The groups' trees run on side streams beside the main group's.  Their factors, embedded into the caller's column
    // order, are dense rows of the final factorisation: they are folded INSIDE the main tree -- once at most 8 of its factors are
    // alive, one launch deals the embedded rows to those factors (a block or two per workgroup) -- instead of by one workgroup, group
    // after group, behind the tree (round 3: 3.3 ms per call).  Without a dense group the final factor starts from R_in.
    {
        // the side streams get DIFFERENT priority levels: HIP gives a stream of another level a hardware queue of its own, while streams of
        // one level share a few queues round robin -- three trees on two queues were the tail of the call (the legs' tree queued behind
        // the arms')
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int prios[4] = {greatest, least, (least + greatest) / 2, (least + greatest) / 2};
        for (int i = 0; i < (int)(sizeof(m->tsqr_streams) / sizeof(m->tsqr_streams[0])); i++)
            if (!m->tsqr_streams[i]) {
                HIPCHK(hipStreamCreateWithPriority(&m->tsqr_streams[i], hipStreamNonBlocking, prios[i]));
            }
    }
    for (auto &e : m->tsqr_ev)
        if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    constexpr int NSIDE = (int)(sizeof(m->tsqr_streams) / sizeof(m->tsqr_streams[0]));
    size_t rt = 0;
    std::vector<size_t> o_r(G, 0);
    for (int g = 0; g < G; g++) {
        o_r[g] = rt;
        rt += (size_t)gp.groups[g].Pa * gp.groups[g].Pa;
    }
    if ((rc = m->tsqr_rtmp.ensure(rt * sizeof(double)))) return rc;
    double *rtmp = m->tsqr_rtmp.as<double>();
    // the longest trees first, one stream each as far as they go (a short tree queued behind the waist chain's tree was the last to finish)
    std::vector<int> side_order;
    for (int g = 0; g < G; g++)
        if (g != gp.main) side_order.push_back(g);
    std::stable_sort(side_order.begin(), side_order.end(), [&](int a, int b) { return gp.groups[a].Pa > gp.groups[b].Pa; });
    // The side groups' trees are started as soon as their last level-0 fold has been enqueued -- BEFORE the main (base-wrench) group's last
    // fold: they are latency bound (eleven levels over 2048 wave-private factors, a handful of waves each at the end: 1.9 ms for the waist
    // chain's group of a 125 k-sample call) and then run beside the throughput-bound fold of the main group instead of behind it.
    bool side_trees_launched = false;
    auto launch_side_trees = [&]() -> int {
        side_trees_launched = true;
        HIPCHK(hipEventRecord(m->tsqr_ev[NSIDE], m->stream));
        for (int i = 0; i < NSIDE; i++) HIPCHK(hipStreamWaitEvent(m->tsqr_streams[i], m->tsqr_ev[NSIDE], 0));
        // (narrow factors of one shape -- the two arms, the two legs -- share their launches: fbr_tsqr_finish_narrow_batch)
        int nside = 0;
        std::vector<char> finished(G, 0);
        for (int g : side_order) {
            if (finished[g]) continue;
            FbrTsqrWork &wg = m->tsqr_groups[g];
            FbrTsqrWork *batch[FBR_TSQR_NARROW_BATCH];
            double *outs[FBR_TSQR_NARROW_BATCH];
            int nb = 0;
            if (wg.narrow)
                for (int h : side_order)
                    if (!finished[h] && nb < FBR_TSQR_NARROW_BATCH && m->tsqr_groups[h].narrow && m->tsqr_groups[h].n == wg.n && m->tsqr_groups[h].NW == wg.NW &&
                        m->tsqr_groups[h].tpw == wg.tpw) {
                        batch[nb] = &m->tsqr_groups[h];
                        outs[nb++] = rtmp + o_r[h];
                        finished[h] = 1;
                    }
            hipStream_t sst = m->tsqr_streams[nside++ % NSIDE];
            if (nb >= 2) {
                if ((rc = fbr_tsqr_finish_narrow_batch(batch, nb, sst, outs))) return tsqr_fail(rc, "tsqr group finish");
            } else {
                for (int i = 0; i < nb; i++) finished[(int)(batch[i] - &m->tsqr_groups[0])] = 0;  // (a batch of one: the plain path)
                finished[g] = 1;
                if ((rc = fbr_tsqr_finish_async(wg, sst, rtmp + o_r[g]))) return tsqr_fail(rc, "tsqr group finish");
            }
        }
        for (int i = 0; i < NSIDE; i++) HIPCHK(hipEventRecord(m->tsqr_ev[i], m->tsqr_streams[i]));
        return FBR_OK;
    };
    // the kinematic records are produced for several chunks at a time: one lane per sample needs tens of thousands of waves in flight
    // to hide its latencies (1 M samples: 6.4 ms in one launch, 11 ms in twelve)
    const long kin_span = std::max(ch, std::min(S, (long)((size_t)(6ull << 30) / ((size_t)hm.rec_size() * sizeof(double))) / ch * ch));
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        const long k0 = s0 / kin_span * kin_span;
        hipStream_t cst = s0 == 0 ? pst : m->stream;  // the first chunk's kinematics and writer belong to the prologue
        // every slot of the chunk holds csp >= cs rows, a whole number of fold blocks in every group (the last chunk is padded with zero
        // rows): a block never straddles two regressor rows, and the structural zeros left of a row's first supported column tile are
        // never written (the folds do not read them)
        const long csp = (cs + lcm - 1) / lcm * lcm;
        if (lane_writer) {
            // one kernel: kinematics + every entry of the groups' chunks, column-major (512-byte runs); padding rows / columns cleared first
            size_t maxrows = 1;
            for (int g = 0; g < G; g++) maxrows = std::max(maxrows, gp.groups[g].rows.size());
            hipLaunchKernelGGL(fbr_groups_clear_cm_kernel, dim3(G, (unsigned)maxrows + FBR_CM_PADWG), dim3(256), 0, cst, dgrp, t + o_nrows, t + o_gpa, cs, csp, (int)maxrows);
            HIPCHK(hipGetLastError());
            const int set = (csp == csp_of[0] && cs == ch) ? 0 : 1;
            if (csp != csp_of[set]) {
                set_err("internal: chunk stride of the lane writer does not match its row tables");
                return FBR_E_INVALID;
            }
            DevKinId kp;
            kp.nsteps = m->kinid.nsteps;
            kp.maxlvl = m->kinid.maxlvl;
            kp.nslots = lane_slots;
            kp.ldn = std::max(hm.n, 1) | 1;
            kp.steps = t + o_lsteps;
            kp.endflush = m->kinid_endflush;
            DevKinWrite kw;
            kw.nparts = lane_parts;
            for (int pq = 0; pq < FBR_KINWRITE_PARTS; pq++) {
                kw.part_nsteps[pq] = lane_nsteps[pq];
                kw.part_step0[pq] = lane_step0[pq];
            }
            kw.lcol10 = t + o_lcol;
            kw.colrec = t + o_lrec;
            kw.dst = (const long *)(dtab + o_lent) + (size_t)set * nlent;
            kw.ninert = hm.ninert;
            kw.cols = hm.cols;
            kw.k = k;
            kw.has_w = dw ? 1 : 0;
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8 / lane_parts, (size_t)(150 << 10) / std::max<size_t>(lane_lds, 1)));
            const int blocks = (int)std::min<long>((cs + 63) / 64, (long)m->num_cus * per_cu);
            if ((rc = m->kinid_scratch.ensure((size_t)blocks * lane_parts * std::max(kp.nslots, 1) * FBR_LINK_REC * 64 * sizeof(double)))) return rc;
            ProfScope ps(m, FBR_PROF_REGRESSOR, cst);
#define FBR_KINWRITE_LAUNCH(D)                                                                                                                  \
    do {                                                                                                                                        \
        HIPCHK(hipFuncSetAttribute((const void *)fbr_kinwrite_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lane_lds));          \
        hipLaunchKernelGGL(fbr_kinwrite_kernel<D>, dim3(blocks), dim3(64 * lane_parts), lane_lds, cst, m->dm, kp, kw, cs, d.q + s0 * hm.n, d.dq + s0 * hm.n,  \
                           d.ddq + s0 * hm.n, d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr, d.rpy ? d.rpy + s0 * 3 : nullptr, \
                           d.sign ? d.sign + s0 * hm.n : nullptr, drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr,                            \
                           dw ? dw + (size_t)s0 * hm.rows : nullptr, m->kinid_scratch.as<double>());                                            \
    } while (0)
            if (kp.maxlvl <= 4)
                FBR_KINWRITE_LAUNCH(4);
            else if (kp.maxlvl <= 8)
                FBR_KINWRITE_LAUNCH(8);
            else if (kp.maxlvl <= 10)
                FBR_KINWRITE_LAUNCH(10);
            else if (kp.maxlvl <= 12)
                FBR_KINWRITE_LAUNCH(12);
            else
                FBR_KINWRITE_LAUNCH(FBR_KINID_MAXD);
#undef FBR_KINWRITE_LAUNCH
        } else {
        if (s0 == k0 && (rc = run_kin(m, d, k0, std::min(kin_span, S - k0), cst))) return rc;
        const double *recs = m->rec.as<double>() + (size_t)(s0 - k0) * hm.rec_size();
        skipzeros = true;
        if (csp > cs) {
            size_t maxrows = 1;
            for (int g = 0; g < G; g++) maxrows = std::max(maxrows, gp.groups[g].rows.size());
            hipLaunchKernelGGL(fbr_groups_clear_pad_kernel, dim3(G, (unsigned)maxrows), dim3(256), 0, cst, dgrp, t + o_nrows, cs, csp);
            HIPCHK(hipGetLastError());
        }
        {
            ProfScope ps(m, FBR_PROF_REGRESSOR, cst);
            if (lds_writer)
                hipLaunchKernelGGL(fbr_regressor_groups_lds_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds_img, cst, m->dm, cs,
                                   recs, d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr,
                                   drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, k, dw ? dw + (size_t)s0 * hm.rows : nullptr, dgrp, G, t, t + hm.rows,
                                   t + o_ebeg[1], (const int *)(dtab + o_ent1), t + o_rowoff, img_total, csp, (int)ents[1].size());
            else if (pairable)
                hipLaunchKernelGGL(fbr_regressor_groups2_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds, cst, m->dm, cs,
                                   recs, d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr,
                                   drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, k, dw ? dw + (size_t)s0 * hm.rows : nullptr, dgrp, G, t, t + hm.rows,
                                   t + o_ebeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_ent1 : o_ent0)),
                                   t + o_pbeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_pent1 : o_pent0)), npairs, hm.ninert, wsplit, csp);
            else
                hipLaunchKernelGGL(fbr_regressor_groups_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds, cst, m->dm, cs,
                                   recs, d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr,
                                   drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, k, dw ? dw + (size_t)s0 * hm.rows : nullptr, dgrp, G, t, t + hm.rows,
                                   t + o_ebeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_ent1 : o_ent0)), csp);
        }
        }
        HIPCHK(hipGetLastError());
        if (cst != m->stream) {  // the folds (main stream) wait for the prologue
            HIPCHK(hipEventRecord(m->ev_tsqr_pro, cst));
            HIPCHK(hipStreamWaitEvent(m->stream, m->ev_tsqr_pro, 0));
        }
        ProfScope ps(m, FBR_PROF_TSQR);
        auto fold_group = [&](int g) -> int {
            const TsqrGroup &Gg = gp.groups[g];
            FbrTsqrRowOrder ro;
            ro.first_col = t + o_fc[g];
            ro.rows = (int)Gg.rows.size();
            ro.group = csp;
            if (lane_writer) ro.colmajor_ld = csp * (long)Gg.rows.size();
            if ((rc = fbr_tsqr_fold_chunk(work(g), m->stream, csp * (long)Gg.rows.size(), Gg.Pa, 0, nullptr, ro))) return tsqr_fail(rc, "tsqr group fold");
            return FBR_OK;
        };
        for (int g = 0; g < G; g++)
            if (g != gp.main && (rc = fold_group(g))) return rc;
        // (last chunk: option tsqr_side_trees_beside = 1 starts the side groups' trees beside the main group's fold.  That paid while the main
        // group folded all six base-wrench rows; with the force rows in a group of their own its fold is half as long, and the trees' first
        // levels only keep its workgroups off their CUs: 125 k samples 7.3 ... 7.45 ms beside, 6.9 ... 7.0 behind; 1 M samples the same)
        if (s0 + ch >= S && gp.main >= 0 && m->opt.tsqr_side_trees_beside != 0 && (rc = launch_side_trees())) return rc;
        if (gp.main >= 0 && (rc = fold_group(gp.main))) return rc;
    }
    if (!side_trees_launched && (rc = launch_side_trees())) return rc;
    bool l0_recorded = false;
    auto record_l0 = [&]() -> int {  // what a following submission's prologue waits for
        if (!l0_recorded) HIPCHK(hipEventRecord(m->ev_tsqr_l0, m->stream));
        l0_recorded = true;
        m->tsqr_l0_rec = true;
        return FBR_OK;
    };
    // rows of the embedded group factors, stacked: [sum of the groups' Pa][n] in the final factor's column order
    long erows = 0;
    for (int g : side_order) erows += gp.groups[g].Pa;
    const bool inside = gp.main >= 0 && !m->tsqr.narrow && erows > 0;
    auto pack_embedded = [&](FbrTsqrWork &wk, double *dst) -> int {
        long off = 0;
        const long epad = (erows + 15) & ~15L;
        for (size_t i = 0; i < side_order.size(); i++) {
            const int g = side_order[i], Pg = gp.groups[g].Pa;
            const long mp = i + 1 == side_order.size() ? epad - off : Pg;  // (the last one also clears the rows up to the padded count)
            hipLaunchKernelGGL(fbr_tsqr_pack_kernel, dim3(256), dim3(256), 0, m->stream, (long)Pg, mp, Pa, 0, wk.n, rtmp + o_r[g], Pg, t + o_emb[g],
                               (const double *)nullptr, (const double *)nullptr, dst + off * wk.n, 0, 0L);
            HIPCHK(hipGetLastError());
            off += Pg;
        }
        return FBR_OK;
    };
    if (inside) {
        FbrTsqrWork &wk = m->tsqr;
        int alive_stride = 1;  // levels with stride < alive_stride have run
        while ((wk.NW + alive_stride - 1) / alive_stride > 8) alive_stride *= 2;
        // (a following submission's prologue starts behind the two widest tree levels: 128 and 64 workgroups)
        const int s_pro = std::min(4, alive_stride);
        {
            ProfScope ps(m, FBR_PROF_TREE);
            if ((rc = fbr_tsqr_tree_levels(wk, m->stream, 1, s_pro)) || (rc = record_l0()) || (rc = fbr_tsqr_tree_levels(wk, m->stream, s_pro, alive_stride)))
                return tsqr_fail(rc, "tsqr tree");
        }
        for (int i = 0; i < NSIDE; i++) HIPCHK(hipStreamWaitEvent(m->stream, m->tsqr_ev[i], 0));
        if ((rc = m->tsqr_embed.ensure((size_t)((erows + 15) & ~15L) * wk.n * sizeof(double)))) return rc;
        const int alive = (wk.NW + alive_stride - 1) / alive_stride;
        {
            ProfScope ps(m, FBR_PROF_TSQR);
            if ((rc = pack_embedded(wk, m->tsqr_embed.as<double>()))) return rc;
            if ((rc = fbr_tsqr_fold_packed(wk, m->stream, erows, FbrTsqrRowOrder(), m->tsqr_embed.as<double>(), alive_stride, alive)))
                return tsqr_fail(rc, "tsqr embedded group factors");
        }
        ProfScope ps(m, FBR_PROF_TREE);
        if ((rc = fbr_tsqr_tree_levels(wk, m->stream, alive_stride, 1 << 30)) || (rc = fbr_tsqr_copy_out(wk, m->stream, R)))
            return tsqr_fail(rc, "tsqr tree");
        return FBR_OK;  // (the error word of the call is read once, at its end: tsqr_impl)
    }
    // no dense group (fixed base behind a branching first link, masked base rows) or wave-private main kernels: the group factors are
    // folded by one workgroup into a factor seeded with the main group's result / R_in
    if ((rc = record_l0())) return rc;
    ProfScope ps(m, FBR_PROF_TSQR);
    const double *seed = Rin_dev;
    if (gp.main >= 0) {
        if ((rc = fbr_tsqr_finish_async(m->tsqr, m->stream, rtmp + o_r[gp.main]))) return tsqr_fail(rc, "tsqr finish");
        seed = rtmp + o_r[gp.main];
    }
    for (int i = 0; i < NSIDE; i++) HIPCHK(hipStreamWaitEvent(m->stream, m->tsqr_ev[i], 0));
    if ((rc = tsqr_begin(m, m->tsqr, m->stream, Pa, seed, m->num_cus, 1, m->tsqr_err))) return tsqr_fail(rc, "tsqr begin");
    for (int g = 0; g < G; g++) {
        if (g == gp.main) continue;
        const int Pg = gp.groups[g].Pa;
        if ((rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, Pg, Pa, rtmp + o_r[g], 0, nullptr, nullptr, Pg, t + o_emb[g]))) return tsqr_fail(rc, "tsqr group merge");
    }
    if ((rc = fbr_tsqr_finish_async(m->tsqr, m->stream, R))) return tsqr_fail(rc, "tsqr finish");
    return FBR_OK;
}

// async_ticket != nullptr: the factorisation is enqueued and NOT waited for (fbr_tsqr_submit): device-resident inputs and output only.
static int tsqr_impl_inner(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
                           const double *w, const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket)
{
    const bool async = async_ticket != nullptr;
    if (async && (!st || st->mem != FBR_DEVICE || out_mem != FBR_DEVICE)) {
        set_err("fbr_tsqr_submit takes device-resident states, rhs, weights, R_in and R_out");
        return FBR_E_INVALID;
    }
    DevStates d;
    if (m) m->submitting = async;
    int rc = stage_states(m, st, &d);
    if (m) m->submitting = false;
    if (rc) return rc;
    bool overlap = false;
    if (async) {
        // at most two submissions in flight (two sets of tables / error slots / completion events)
        if ((rc = wait_ticket(m, m->next_ticket - 2))) return rc;
        overlap = m->waited_ticket < m->next_ticket - 1 && m->last_submit_kind == 1 && m->tsqr_l0_rec && m->opt.tsqr_prologue_overlap != 0;
    }
    const int par = (int)(m->next_ticket & 1);  // (blocking calls: nothing is in flight, either set is free)
    HIPCHK(hipMemsetAsync(m->tsqr_err, 0, sizeof(unsigned), m->stream));
    if (!R_out || k < 0 || k > FBR_MAX_RHS || (k > 0 && !rhs)) {
        set_err("bad rhs / R_out arguments");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    if (cols) {
        if (ncols <= 0 || ncols > hm.cols) {
            set_err("bad column subset size");
            return FBR_E_INVALID;
        }
        std::vector<char> seen(hm.cols, 0);
        for (int i = 0; i < ncols; i++) {
            if (cols[i] < 0 || cols[i] >= hm.cols || seen[cols[i]]) {
                set_err("column subset entries must be distinct and in range");
                return FBR_E_INVALID;
            }
            seen[cols[i]] = 1;
        }
    }
    const long S = d.S;
    const TsqrPlan plan = tsqr_plan(hm, cols, ncols, k, S, m->opt.tsqr_reorder != 0);
    const int Psel = plan.Psel, Pa = plan.Pa;
    const size_t rcount = (size_t)Pa * Pa;
    const double *drhs = nullptr, *dw = nullptr;
    if ((rc = stage_one(m, m->st_aux, rhs, (size_t)S * hm.rows * k, st->mem, &drhs))) return rc;
    if ((rc = stage_one(m, m->st_aux2, w, (size_t)S * hm.rows, st->mem, &dw))) return rc;
    double *R = R_out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(rcount * sizeof(double)))) return rc;
        R = m->g_tmp.as<double>();
    }
    const double *Rin_dev = nullptr;
    if (R_in) {
        if (out_mem == FBR_HOST) {
            HIPCHK(hipMemcpyAsync(R, R_in, rcount * sizeof(double), hipMemcpyHostToDevice, m->stream));
            Rin_dev = R;
        } else {
            Rin_dev = R_in;
        }
    }
    auto tsqr_fail = [&](int code, const char *what) {
        set_err(std::string(what) + ": " + fbr_tsqr_error());
        return code == -4 ? FBR_E_UNSUPPORTED : (code == -3 ? FBR_E_HIP : FBR_E_INVALID);
    };
    // the end of every path: the call's error word goes to the pinned slot of its parity; a submission returns its ticket, a blocking
    // call waits and looks at the slot
    auto done = [&]() -> int {
        HIPCHK(hipMemcpyAsync(&m->tsqr_err_host[par], m->tsqr_err, sizeof(unsigned), hipMemcpyDeviceToHost, m->stream));
        if (async) {
            const int64_t t = m->next_ticket++;
            m->ticket_kind[t & 1] = 1;
            m->last_submit_kind = 1;
            HIPCHK(hipEventRecord(m->ev_done[t & 1], m->stream));
            *async_ticket = t;
            return FBR_OK;
        }
        int rc2 = finish_output(m, R, R_out, rcount, out_mem);
        if (rc2) return rc2;
        if (m->tsqr_err_host[par]) {
            char hx[16];
            snprintf(hx, sizeof hx, "%08x", m->tsqr_err_host[par]);
            m->tsqr_err_host[par] = 0;
            set_err("TSQR pipeline flag wait timed out (internal error, code " + std::string(hx) + ")");
            return FBR_E_HIP;
        }
        return FBR_OK;
    };
    {
        // (row weights on the device are scanned for switched-off rows: that read-back waits for the stream, i.e. for a submission in flight)
        std::vector<char> act;
        if ((rc = active_rows(m, dw, S, &act))) return rc;
        const TsqrGroupPlan gp = tsqr_group_plan(hm, cols, ncols, k, &act, m->opt.tsqr_force_group != 0);
        if (hm.rows <= 255 && tsqr_use_groups(m, gp, S)) {  // (the writer's entries hold the regressor row in 8 bits)
            if ((rc = tsqr_groups_impl(m, d, gp, cols, Psel, k, drhs, dw, Rin_dev, R, par, overlap))) return rc;
            return done();
        }
    }
    if (hm.masked) return FBR_E_NOT_GROUPED;  // (internal models with column masks factorise by row groups only: the caller takes the merged model)
    // device tables: [fcols (Psel) | perm (Pa) | inv (Pa) | linkpos (L) | row first columns (rows)]
    const int *dcols = nullptr, *dperm = nullptr, *dinv = nullptr, *dlinkpos = nullptr, *dfc = nullptr;
    {
        std::vector<int> tab;
        tab.insert(tab.end(), plan.fcols.begin(), plan.fcols.end());
        tab.insert(tab.end(), plan.perm.begin(), plan.perm.end());
        tab.insert(tab.end(), plan.inv.begin(), plan.inv.end());
        tab.insert(tab.end(), plan.linkpos.begin(), plan.linkpos.end());
        tab.insert(tab.end(), plan.fc.begin(), plan.fc.end());
        const char *dtab = nullptr;
        if ((rc = tsqr_upload_tables(m, par, {{tab.data(), tab.size() * sizeof(int)}}, {0}, tab.size() * sizeof(int), m->stream, &dtab))) return rc;
        const int *t = (const int *)dtab;
        if (cols || (plan.reorder)) dcols = t;  // gather list of the materialised path
        dperm = t + Psel;
        dinv = dperm + Pa;
        if (!cols && plan.reorder) dlinkpos = dinv + Pa;
        dfc = dinv + Pa + plan.linkpos.size();
    }
    // an existing factor seeds working factor 0 directly when the column order is the caller's; in the internal order its rows are
    // folded in like data rows (column gather)
    if ((rc = tsqr_begin(m, m->tsqr, m->stream, Pa, plan.reorder ? nullptr : Rin_dev, m->num_cus, S * (long)hm.rows, m->tsqr_err))) return tsqr_fail(rc, "tsqr begin");
    if (plan.reorder && Rin_dev) {
        ProfScope ps(m, FBR_PROF_TSQR);
        if ((rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, Pa, Pa, Rin_dev, 0, nullptr, nullptr, Pa, dperm))) return tsqr_fail(rc, "tsqr fold R_in");
    }
    if (S > 0) {
        // materialise Y chunk by chunk (K1 + K2) and fold each chunk into the per-workgroup factors.  Without row
        // weights / column subset the regressor kernel writes straight into the padded chunk [Y | rhs | 0] of the
        // factorisation (leading dimension n): no second pass over Y.
        const size_t per = (size_t)hm.rows * hm.cols;
        long ch = fbr_tsqr_chunk_samples(hm.rows, Pa);
        ch = std::min(ch, chunk_size(m, S));
        if (ch > m->tsqr.mb) ch -= ch % m->tsqr.mb;  // whole blocks per regressor row in the row-sorted chunks
        const bool direct = !cols && !dw;
        if (!direct && (rc = m->out_tmp.ensure((size_t)ch * per * sizeof(double)))) return rc;
        for (long s0 = 0; s0 < S; s0 += ch) {
            const long cs = std::min(ch, S - s0);
            if ((rc = run_kin(m, d, s0, cs))) return rc;
            double *dst = m->out_tmp.as<double>();
            int ldy = hm.cols;
            // The chunk is stacked by regressor row (all samples' row r together): R does not depend on the order of the rows,
            // and a 64-row block of one regressor row is zero left of that row's first supported column, so its fold starts
            // there (rows of joints deep in the tree touch a fraction of the panels).
            FbrTsqrRowOrder ro;
            ro.first_col = dfc;
            ro.rows = hm.rows;
            ro.group = cs;
            long rs_s = hm.rows, rs_r = 1;
            if (direct) {
                if ((rc = fbr_tsqr_chunk_buffer(m->tsqr, cs * hm.rows, &dst)) || (k == 0 && (rc = fbr_tsqr_chunk_clean(m->tsqr, m->stream))))
                    return tsqr_fail(rc, "tsqr chunk");
                ldy = m->tsqr.n;
                if (ro.rows) {
                    rs_s = 1;
                    rs_r = cs;
                }
            }
            const int *lp = direct ? dlinkpos : nullptr;  // (the materialised path gathers the columns when it packs the chunk)
            // structural zeros left of a row's first supported column tile are not written when every block holds rows of ONE regressor
            // row (the chunk is a whole number of blocks per row): the folds never read them
            const int *skipfc = (direct && cs % m->tsqr.mb == 0) ? dfc : nullptr;
            if ((rc = launch_regressor(m, d, s0, cs, dst, ldy, rs_s, rs_r, lp, skipfc))) return rc;
            ProfScope ps(m, FBR_PROF_TSQR);
            if (direct)
                rc = fbr_tsqr_fold_chunk(m->tsqr, m->stream, cs * hm.rows, Psel, k, drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, ro);
            else
                rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, cs * hm.rows, Psel, m->out_tmp.as<double>(), k,
                                        drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, dw ? dw + (size_t)s0 * hm.rows : nullptr, hm.cols, dcols, ro);
            if (rc) return tsqr_fail(rc, "tsqr fold");
        }
    }
    {
        ProfScope ps(m, FBR_PROF_TSQR);
        if (!plan.reorder) {
            if ((rc = fbr_tsqr_finish_async(m->tsqr, m->stream, R))) return tsqr_fail(rc, "tsqr finish");
        } else {
            // factor in the internal column order -> the caller's: R = qr(R' [:, inv]) (one workgroup, Pa dense rows)
            if ((rc = m->tsqr_rtmp.ensure(rcount * sizeof(double)))) return rc;
            if ((rc = fbr_tsqr_finish_async(m->tsqr, m->stream, m->tsqr_rtmp.as<double>()))) return tsqr_fail(rc, "tsqr finish");
            if ((rc = tsqr_begin(m, m->tsqr, m->stream, Pa, nullptr, m->num_cus, 1, m->tsqr_err)) ||
                (rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, Pa, Pa, m->tsqr_rtmp.as<double>(), 0, nullptr, nullptr, Pa, dinv)) ||
                (rc = fbr_tsqr_finish_async(m->tsqr, m->stream, R)))
                return tsqr_fail(rc, "tsqr column order");
        }
    }
    return done();
}
// the reduced model a factorisation of every column runs on (-1: the model itself); the regrouped model factorises by row groups only
static int pick_tsqr_reduction(fbr_model *m, long S)
{
    if (!m->opt.link_merge) return -1;
    // (like the Gram pass, pick_gram_reduction: the reduced factorisation costs a second model's launches and one expansion level; a
    // factorisation does about eight times the work of a Gram pass per sample and column pair, hence an eighth of its threshold)
    if (const fbr_model *r0 = m->rdm[0] ? m->rdm[0].get() : m->rdm[1].get();
        r0 && S >= 0 && (double)S * (m->hm.cols - r0->hm.cols) * m->hm.cols < m->opt.reduce_min_work / 8) return -1;
    if (m->rdm[1] && m->opt.regroup && m->opt.tsqr_groups) {
        if (m->rd_grouped < 0) {
            const TsqrGroupPlan gp = tsqr_group_plan(m->rdm[1]->hm, nullptr, 0, 0, nullptr, m->opt.tsqr_force_group != 0);
            m->rd_grouped = gp.groups.size() > 1 && m->rdm[1]->hm.rows <= 255;
        }
        if (m->rd_grouped && S >= (long)m->opt.tsqr_group_min_samples) return 1;
    }
    return m->rdm[0] ? 0 : -1;
}

// The factor through the link-merged model (build_reduction): R_red over the moving bodies' columns, then R = qr([R_in ; R_red E]) --
// the Pra dense rows R_red E become working factor 1 beside R_in (or zero) in working factor 0, and ONE level of the merge tree,
// pipelined across workgroups, folds them (wide factors; narrow ones fold them as ordinary rows).
// cols != NULL: the factor of a COLUMN SUBSET (fbr_tsqr_cols): Y[:, cols] = Y_red E[:, cols], so the reduced factorisation is the same and
// only the expansion takes the subset's columns of E -- the base regressor [YBase | tau] of WALK-MAN (213 of 480 columns, spread over
// every link) costs the regrouped factorisation's 57 ms per 1 M samples instead of 68.
static int tsqr_via_red(fbr_model *m, int which, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k, const double *w,
                        const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket)
{
    fbr_model *r = m->rdm[which].get();
    const bool async = async_ticket != nullptr;
    int rc;
    if ((rc = enter(m))) return rc;
    if ((rc = wait_ticket(m, async ? m->next_ticket - 2 : m->next_ticket - 1))) return rc;
    if (async && out_mem != FBR_DEVICE) {
        set_err("fbr_tsqr_submit takes device-resident states, rhs, weights, R_in and R_out");
        return FBR_E_INVALID;
    }
    r->stream = m->stream;
    r->prof = m->prof;
    const int par = (int)(m->next_ticket & 1), Pa = (cols ? ncols : m->hm.cols) + k, Pra = r->hm.cols + k;
    const size_t cnt = (size_t)Pa * Pa;
    if ((rc = m->red_out[par].ensure((size_t)Pra * Pra * sizeof(double)))) return rc;
    double *Rred = m->red_out[par].as<double>();
    int64_t tr = -1;
    if ((rc = tsqr_impl(r, st, nullptr, 0, rhs, k, w, nullptr, Rred, FBR_DEVICE, async ? &tr : nullptr))) return rc;
    const int *dcolmap = nullptr;  // output column jj of the subset -> column of the augmented full layout (rhs columns behind the identified ones)
    if (cols) {
        std::vector<int> cmap(cols, cols + ncols);
        for (int i = 0; i < k; i++) cmap.push_back(m->hm.cols + i);
        const char *dtab = nullptr;
        if ((rc = tsqr_upload_tables(m, par, {{cmap.data(), cmap.size() * sizeof(int)}}, {0}, cmap.size() * sizeof(int), m->stream, &dtab))) return rc;
        dcolmap = (const int *)dtab;
    }
    double *R = R_out;
    const double *Rin_dev = nullptr;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(cnt * sizeof(double)))) return rc;
        R = m->g_tmp.as<double>();
        if (R_in) {
            HIPCHK(hipMemcpyAsync(R, R_in, cnt * sizeof(double), hipMemcpyHostToDevice, m->stream));
            Rin_dev = R;
        }
    } else {
        Rin_dev = R_in;
    }
    auto fail = [&](int code, const char *what) {
        set_err(std::string(what) + ": " + fbr_tsqr_error());
        return code == -4 ? FBR_E_UNSUPPORTED : (code == -3 ? FBR_E_HIP : FBR_E_INVALID);
    };
    HIPCHK(hipMemsetAsync(m->tsqr_err, 0, sizeof(unsigned), m->stream));
    FbrTsqrShape sh;
    if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh, topts(m))) return fail(-4, "tsqr shape");
    FbrTsqrWork &wk = m->tsqr;
    {
        ProfScope ps(m, FBR_PROF_TREE);
        bool done_wide = false;
        // (the kernels whose merge level takes dense partner rows; a working factor has room for sh.n of them: a column subset narrower
        // than the reduced column set -- Pra > Pa -- takes the row path below)
        if (!sh.narrow && sh.n / 16 > FBR_TSQR_NARROW_MAX_TILES && !m->opt.tsqr_tree_one_wg && Pra <= sh.n) {
            if ((rc = tsqr_begin(m, wk, m->stream, Pa, Rin_dev, m->num_cus, 2L * sh.mb, m->tsqr_err))) return fail(rc, "tsqr begin");
            if (wk.NW == 2) {
                if ((rc = launch_expand_rows(m, which, k, Pra, Rred, wk.Rw + (size_t)wk.n * wk.ld, wk.ld, dcolmap, Pa))) return rc;
                if ((rc = fbr_tsqr_tree_levels(wk, m->stream, 1, 2, Pra)) || (rc = fbr_tsqr_copy_out(wk, m->stream, R))) return fail(rc, "tsqr expansion");
                done_wide = true;
            }
        }
        if (!done_wide) {  // narrow factors: the expanded rows as ordinary data rows of a one-workgroup factorisation
            if ((rc = m->tsqr_embed.ensure((size_t)Pra * Pa * sizeof(double)))) return rc;
            if ((rc = launch_expand_rows(m, which, k, Pra, Rred, m->tsqr_embed.as<double>(), Pa, dcolmap, Pa))) return rc;
            if ((rc = tsqr_begin(m, wk, m->stream, Pa, Rin_dev, m->num_cus, 1, m->tsqr_err)) ||
                (rc = fbr_tsqr_fold_rows(wk, m->stream, Pra, Pa, m->tsqr_embed.as<double>(), 0, nullptr, nullptr, Pa)) ||
                (rc = fbr_tsqr_finish_async(wk, m->stream, R)))
                return fail(rc, "tsqr expansion");
        }
    }
    HIPCHK(hipMemcpyAsync(&m->tsqr_err_host[par], m->tsqr_err, sizeof(unsigned), hipMemcpyDeviceToHost, m->stream));
    if (async) {
        const int64_t t = m->next_ticket++;
        m->ticket_kind[t & 1] = 1;
        m->ticket_via_red[t & 1] = 1 + which;
        m->red_ticket[t & 1] = tr;
        m->last_submit_kind = 1;
        HIPCHK(hipEventRecord(m->ev_done[t & 1], m->stream));
        *async_ticket = t;
        return FBR_OK;
    }
    if ((rc = finish_output(m, R, R_out, cnt, out_mem))) return rc;
    if (m->tsqr_err_host[par]) {
        char hx[16];
        snprintf(hx, sizeof hx, "%08x", m->tsqr_err_host[par]);
        m->tsqr_err_host[par] = 0;
        set_err("TSQR pipeline flag wait timed out (internal error, code " + std::string(hx) + ")");
        return FBR_E_HIP;
    }
    return FBR_OK;
}

int tsqr_impl(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
              const double *w, const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket)
{
    int which = (m && st && R_out && k >= 0 && k <= FBR_MAX_RHS && m->pid == getpid()) ? pick_tsqr_reduction(m, (long)st->num_samples) : -1;
    if (cols && which >= 0) {
        // a column subset goes through the reductions when it is about as wide as the regrouped column set (the base columns: one per
        // direction the regressor can move in) -- a narrow subset is cheaper factorised directly -- and only through the regrouped model
        bool ok = which == 1 && ncols > 0 && ncols <= m->hm.cols && 5L * ncols >= 4L * m->rdm[1]->hm.cols;
        std::vector<char> seen(m->hm.cols, 0);
        for (int i = 0; ok && i < ncols; i++) {  // (invalid lists are reported by the direct path)
            ok = cols[i] >= 0 && cols[i] < m->hm.cols && !seen[cols[i]];
            if (ok) seen[cols[i]] = 1;
        }
        if (!ok) which = -1;
    }
    while (which >= 0) {
        int rc = tsqr_via_red(m, which, st, cols, ncols, rhs, k, w, R_in, R_out, out_mem, async_ticket);
        if (rc == FBR_E_NOT_GROUPED && which == 1) {  // (row weights left the regrouped model without row groups: see tsqr_impl_inner)
            which = (m->rdm[0] && !cols) ? 0 : -1;
            continue;
        }
        if (rc && m->stream) {
            const std::string msg = g_fbr_err;
            drain_after_failed_submit(m);
            set_err(msg);
        }
        return rc;
    }
    int rc = tsqr_impl_inner(m, st, cols, ncols, rhs, k, w, R_in, R_out, out_mem, async_ticket);
    // FBR_E_NOT_GROUPED (a reduced model with column masks whose row weights left it without row groups) is not a failure: only the
    // clearing of the error word and the staging of rhs / weights were enqueued, nothing that reads the caller's buffers stays in
    // flight, and the caller (tsqr_impl of the parent) repeats the call on the merged model -- no drain: it would serialise an
    // asynchronous submission and mark tickets as waited whose error words have not been looked at
    if (rc == FBR_E_NOT_GROUPED) return rc;
    if (rc && m && m->pid == getpid() && m->stream) {  // (blocking calls too: the groups' trees run on side streams)
        const std::string msg = g_fbr_err;
        drain_after_failed_submit(m);
        set_err(msg);
    }
    return rc;
}

extern "C" int fbr_tsqr(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w,
                        const double *R_in, double *R_out, int32_t out_mem)
{
    return tsqr_impl(m, st, nullptr, 0, rhs, k, w, R_in, R_out, out_mem, nullptr);
}

extern "C" int fbr_tsqr_cols(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs,
                             int32_t k, const double *w, const double *R_in, double *R_out, int32_t out_mem)
{
    if (!cols) {
        set_err("cols is NULL");
        return FBR_E_INVALID;
    }
    return tsqr_impl(m, st, cols, ncols, rhs, k, w, R_in, R_out, out_mem, nullptr);
}

extern "C" int fbr_tsqr_submit(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
                               const double *w, const double *R_in, double *R_out, int64_t *ticket)
{
    if (!ticket) {
        set_err("ticket is NULL");
        return FBR_E_INVALID;
    }
    if (cols && ncols <= 0) {
        set_err("bad column subset size");
        return FBR_E_INVALID;
    }
    return tsqr_impl(m, st, cols, cols ? ncols : 0, rhs, k, w, R_in, R_out, FBR_DEVICE, ticket);
}

extern "C" int fbr_tsqr_work_info(fbr_model *m, const int32_t *cols, int32_t ncols, int32_t k, int64_t num_samples, int64_t *mfma_level0,
                                  int64_t *mfma_tree, int32_t *block_rows, int32_t *n_padded)
{
    if (!m || k < 0 || k > FBR_MAX_RHS || num_samples < 0 || (cols && (ncols <= 0 || ncols > m->hm.cols))) {
        set_err("bad arguments");
        return FBR_E_INVALID;
    }
    int which_wi = pick_tsqr_reduction(m, (long)num_samples);
    if (cols && !(which_wi == 1 && 5L * ncols >= 4L * m->rdm[1]->hm.cols)) which_wi = -1;  // (a column subset: the rule of tsqr_impl)
    if (const int which = which_wi; which >= 0) {
        // what fbr_tsqr runs on a link-merged model: the factorisation of the reduced robot, then the Pra expanded rows folded into the
        // final factor by one tree level; block_rows / n_padded describe the FINAL factor (what fbr_tsqr_merge works on)
        int64_t l0 = 0, tr = 0;
        if (int rc = fbr_tsqr_work_info(m->rdm[which].get(), nullptr, 0, k, num_samples, &l0, &tr, nullptr, nullptr)) return rc;
        FbrTsqrShape sh;
        const int Pa = (cols ? ncols : m->hm.cols) + k, Pra = m->rdm[which]->hm.cols + k;
        if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh, topts(m))) {
            set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
            return FBR_E_UNSUPPORTED;
        }
        const long NP = sh.n / 16;
        for (long r0 = 0; r0 < Pra; r0 += sh.tmb) tr += (8L * sh.tsub + 4) * (NP * (NP - 1) / 2);
        if (mfma_level0) *mfma_level0 = l0;
        if (mfma_tree) *mfma_tree = tr;
        if (block_rows) *block_rows = sh.mb;
        if (n_padded) *n_padded = sh.n;
        return FBR_OK;
    }
    const FbrHostModel &hm = m->hm;
    const TsqrPlan plan = tsqr_plan(hm, cols, ncols, k, (long)num_samples, m->opt.tsqr_reorder != 0);
    const int Psel = plan.Psel, Pa = plan.Pa;
    (void)Psel;
    {
        // tree-structured path (tsqr_groups_impl): level 0 of every group over its own chunks, the groups' trees, and the final factor
        // that folds the embedded group factors (dense rows) and runs its own tree
        const TsqrGroupPlan gp = tsqr_group_plan(hm, cols, ncols, k, nullptr, m->opt.tsqr_force_group != 0);
        if (hm.rows <= 255 && tsqr_use_groups(m, gp, (long)num_samples)) {
            long lcm = 1;
            const long ch = tsqr_group_chunk_samples(m, gp, (long)num_samples, &lcm);
            long l0 = 0, tr = 0, mrows = 0;
            FbrTsqrShape sh;
            auto fold_mfma = [&](int first_col) -> long {
                const long np_ = sh.n / 16 - first_col / 16;
                return np_ > 0 ? (8L * sh.sub + 4) * (np_ * (np_ - 1) / 2) : 0;
            };
            auto tree = [&]() {
                long merge = 0, t = 0;
                for (int i0 = 0; i0 < sh.n; i0 += sh.tmb) {
                    const long np_ = sh.n / 16 - i0 / 16;
                    merge += np_ > 0 ? (8L * sh.tsub + 4) * (np_ * (np_ - 1) / 2) : 0;
                }
                for (int stride = 1; stride < sh.NW; stride *= 2)
                    for (long a = 0; a + stride < sh.NW; a += 2L * stride) t += merge;
                return t;
            };
            for (int g = 0; g < (int)gp.groups.size(); g++) mrows += g == gp.main ? num_samples * (long)gp.groups[g].rows.size() : gp.groups[g].Pa;
            for (int g = 0; g < (int)gp.groups.size(); g++) {
                const TsqrGroup &G = gp.groups[g];
                const long ns = (long)G.rows.size();
                if (ch < 0 || fbr_tsqr_shape(G.Pa, m->num_cus, g == gp.main ? mrows : num_samples * ns, &sh, topts(m))) {
                    set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
                    return FBR_E_UNSUPPORTED;
                }
                for (long s0 = 0; s0 < num_samples; s0 += ch) {
                    const long cs = (std::min(ch, (long)num_samples - s0) + lcm - 1) / lcm * lcm, M = cs * ns, Mpad = (M + 15) & ~15L;  // (padded slots)
                    for (long b = 0; b < (Mpad + sh.mb - 1) / sh.mb; b++) {
                        const long r0 = b * sh.mb;
                        int f = sh.n;
                        if (r0 < M)
                            for (long r = r0 / cs; r <= (std::min<long>(r0 + sh.mb, M) - 1) / cs; r++) f = std::min(f, G.fc[r]);
                        l0 += fold_mfma(f);
                    }
                }
                if (g != gp.main) tr += tree();
            }
            if (fbr_tsqr_shape(Pa, m->num_cus, mrows, &sh, topts(m))) {
                set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
                return FBR_E_UNSUPPORTED;
            }
            if (gp.main >= 0) tr += tree();  // the dense group's own tree
            const int main_mb = sh.mb;
            long erows = 0;
            for (int g = 0; g < (int)gp.groups.size(); g++)
                if (g != gp.main) erows += gp.groups[g].Pa;
            if (gp.main >= 0 && !sh.narrow && erows > 0) {
                // the stacked embedded group factors are folded into the factors alive inside the main tree (tsqr_groups_impl)
                for (long r0 = 0; r0 < ((erows + 15) & ~15L); r0 += sh.mb) tr += fold_mfma(0);
            } else {
                if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh, topts(m))) {  // (that factorisation is begun for a handful of rows: tsqr_begin(m, .., 1))
                    set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
                    return FBR_E_UNSUPPORTED;
                }
                for (int g = 0; g < (int)gp.groups.size(); g++)
                    if (g != gp.main)
                        for (long r0 = 0; r0 < ((gp.groups[g].Pa + 15) & ~15); r0 += sh.mb) tr += fold_mfma(0);
            }
            if (mfma_level0) *mfma_level0 = l0;
            if (mfma_tree) *mfma_tree = tr;
            if (block_rows) *block_rows = main_mb;
            if (n_padded) *n_padded = sh.n;
            return FBR_OK;
        }
    }
    FbrTsqrShape sh;
    if (fbr_tsqr_shape(Pa, m->num_cus, num_samples * (long)hm.rows, &sh, topts(m))) {
        set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
        return FBR_E_UNSUPPORTED;
    }
    const std::vector<int> &fc = plan.fc;
    const int NP = sh.n / 16;
    const long per_update = 8L * sh.sub + 4;  // V^T C (4 SUB) + T (4) + C -= V W (4 SUB) MFMAs per (panel, tile right of it)
    auto fold_mfma = [&](int first_col) -> long {
        const long np_ = NP - first_col / 16;
        return np_ > 0 ? per_update * (np_ * (np_ - 1) / 2) : 0;
    };
    long l0 = 0, tr = 0;
    if (num_samples > 0) {
        long ch = std::min(fbr_tsqr_chunk_samples(hm.rows, Pa), chunk_size(m, num_samples));
        if (ch > sh.mb) ch -= ch % sh.mb;
        for (long s0 = 0; s0 < num_samples; s0 += ch) {
            const long cs = std::min(ch, (long)num_samples - s0), M = cs * hm.rows, Mpad = (M + 15) & ~15L;
            const long nblocks = (Mpad + sh.mb - 1) / sh.mb;
            for (long b = 0; b < nblocks; b++) {
                const long r0 = b * sh.mb;
                int f = sh.n;
                if (r0 < M) {
                    const int ra = (int)(r0 / cs), rb = (int)((std::min<long>(r0 + sh.mb, M) - 1) / cs);
                    for (int r = ra; r <= rb; r++) f = std::min(f, fc[r]);
                }
                l0 += fold_mfma(f);
            }
        }
    }
    long merge = 0;  // one node of the tree: the partner's triangular factor folded in block_rows-row pieces
    for (int i0 = 0; i0 < sh.n; i0 += sh.tmb) {
        const long np_ = NP - i0 / 16;
        merge += np_ > 0 ? (8L * sh.tsub + 4) * (np_ * (np_ - 1) / 2) : 0;
    }
    for (int stride = 1; stride < sh.NW; stride *= 2)
        for (long a = 0; a + stride < sh.NW; a += 2L * stride) tr += merge;
    if (plan.reorder) {  // the factor is brought back to the caller's column order: Pa dense rows folded by one workgroup
        FbrTsqrShape s1;
        if (fbr_tsqr_shape(Pa, m->num_cus, 1, &s1, topts(m))) return FBR_E_UNSUPPORTED;
        for (long r0 = 0; r0 < ((Pa + 15) & ~15); r0 += s1.mb) tr += (8L * s1.sub + 4) * ((long)NP * (NP - 1) / 2);
    }
    if (mfma_level0) *mfma_level0 = l0;
    if (mfma_tree) *mfma_tree = tr;
    if (block_rows) *block_rows = sh.mb;
    if (n_padded) *n_padded = sh.n;
    return FBR_OK;
}

extern "C" int fbr_tsqr_merge(fbr_model *m, int32_t n, const double *R_a, const double *R_b, double *R_out, int32_t mem)
{
    if (!m || n <= 0 || !R_a || !R_b || !R_out) {
        set_err("bad arguments");
        return FBR_E_INVALID;
    }
    if (int rc_enter = enter_blocking(m)) return rc_enter;
    const size_t cnt = (size_t)n * n;
    int rc;
    const double *da = nullptr, *db = nullptr;
    if ((rc = stage_one(m, m->st_aux, R_a, cnt, mem, &da))) return rc;
    if ((rc = stage_one(m, m->st_aux2, R_b, cnt, mem, &db))) return rc;
    double *R = R_out;
    if (mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(cnt * sizeof(double)))) return rc;
        R = m->g_tmp.as<double>();
    }
    // one workgroup (narrow factors: one wave) folds the partner's factor, 64 (32) rows at a time, into a working factor seeded with
    // R_a; a block of the triangular R_b is folded from its first non-zero column.  (rows_hint = 1: a single working factor, no tree.)
    // Wide factors: the two triangles become working factors 0 and 1 and ONE level of the merge tree joins them -- pipelined across up to
    // eight workgroups (fbr_tsqr_tree_x_kernel: 0.33 instead of 0.93 ms for WALK-MAN's 496 columns; the same blocks in the same order,
    // bit-identical).  This is the step on the critical path of the TSQR rank tree across GPUs (flobaroid_amd/dist.py: one merge per level).
    FbrTsqrShape sh;
    if (!fbr_tsqr_shape(n, m->num_cus, 1, &sh, topts(m)) && !sh.narrow && sh.n / 16 > FBR_TSQR_NARROW_MAX_TILES && !m->opt.tsqr_tree_one_wg) {
        FbrTsqrWork &wk = m->tsqr;
        if ((rc = tsqr_begin(m, wk, m->stream, n, da, m->num_cus, 2L * sh.mb))) {  // rows for two blocks: two working factors
            set_err(std::string("tsqr merge: ") + fbr_tsqr_error());
            return rc == -4 ? FBR_E_UNSUPPORTED : (rc == -3 ? FBR_E_HIP : FBR_E_INVALID);
        }
        if (wk.NW == 2) {
            hipLaunchKernelGGL(fbr_tsqr_copy_kernel, dim3(256), dim3(256), 0, m->stream, n, db, n, wk.Rw + (size_t)wk.n * wk.ld, wk.ld, wk.n, wk.ld);
            HIPCHK(hipGetLastError());
            if ((rc = fbr_tsqr_finish(wk, m->stream, R))) {
                set_err(std::string("tsqr merge: ") + fbr_tsqr_error());
                return rc == -4 ? FBR_E_UNSUPPORTED : (rc == -3 ? FBR_E_HIP : FBR_E_INVALID);
            }
            return finish_output(m, R, R_out, cnt, mem);
        }
    }
    FbrTsqrRowOrder tri;
    tri.rows = -1;
    if ((rc = tsqr_begin(m, m->tsqr, m->stream, n, da, m->num_cus, 1)) ||
        (rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, n, n, db, 0, nullptr, nullptr, 0, nullptr, tri)) ||
        (rc = fbr_tsqr_finish(m->tsqr, m->stream, R))) {
        set_err(std::string("tsqr merge: ") + fbr_tsqr_error());
        return rc == -4 ? FBR_E_UNSUPPORTED : (rc == -3 ? FBR_E_HIP : FBR_E_INVALID);
    }
    return finish_output(m, R, R_out, cnt, mem);
}
