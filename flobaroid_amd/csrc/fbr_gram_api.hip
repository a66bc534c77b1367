// fbr_gram_api.hip -- the fused regressor -> Gram pass of libfbr (fbr_gram_accumulate / _submit / _grouped, include/fbr.h).
#define FBR_KERNELS_GRAM
#include "fbr_internal.h"

// Device tables of the deal of `wpg` workgroups to the parts (cached per holder).
static int get_deal(GramHolder *h, int wpg, GramHolder::Deal *out, bool base_only = false)
{
    const int key = wpg | (base_only ? 1 << 24 : 0);
    auto it = h->deals.find(key);
    if (it != h->deals.end()) {
        *out = it->second;
        return FBR_OK;
    }
    const std::vector<int> n = fbr_gram_deal(h->prog, wpg, base_only);
    // dispatch order: round robin over the parts.  The SIMD arbiter favours the older waves, so the workgroups dispatched first
    // run ~20 % faster than the ones that arrive second on a CU; every part gets the same mix of both.
    std::vector<int2> tab;
    std::vector<int> begin(h->prog.T + 1, 0), given(h->prog.T, 0);
    for (int p = 0; p < h->prog.T; p++) begin[p + 1] = begin[p] + n[p];
    while ((int)tab.size() < begin[h->prog.T])
        for (int p = 0; p < h->prog.T; p++)
            if (given[p] < n[p]) tab.push_back(make_int2(p, given[p]++ | (n[p] << 16)));
    GramHolder::Deal d;
    int rc;
    if ((rc = upload(h->pool, tab, &d.tab))) return rc;
    if ((rc = upload(h->pool, begin, &d.begin))) return rc;
    h->deals[key] = d;
    *out = d;
    return FBR_OK;
}
// G (+)= E^T W, W = G_red E [Pra x Pa] (fbr_expand_rows_kernel) on the augmented layouts (Pa = cols + k, Pra = cols_red + k); the k rhs
// columns of E sit at E_beg[cols + r]
__global__ __launch_bounds__(256) void fbr_expand_gram_kernel(int cols, int k, int Pra, const int *__restrict__ Eb, const int *__restrict__ Er,
                                                               const double *__restrict__ Ev, const double *__restrict__ Gred, double *__restrict__ G,
                                                               int accumulate)
{
    const int Pa = cols + k;
    Gred += (long)blockIdx.y * Pra * Pa;  // blockIdx.y: sample group (fbr_gram_grouped: one Gram and one W per group)
    G += (long)blockIdx.y * Pa * Pa;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)Pa * Pa; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / Pa), j = (int)(e - (long)i * Pa);
        if (i > j) continue;  // the upper triangle is computed, the lower one mirrored: G is symmetric to the bit, like the fused Gram's
        double acc = 0.0;  // (Gred here: W = G_red E [Pra x Pa], fbr_expand_rows_kernel)
        for (int a = Eb[i]; a < Eb[i + 1]; a++) acc += Ev[a] * Gred[(long)Er[a] * Pa + j];
        G[e] = accumulate ? G[e] + acc : acc;
        if (i != j) G[(long)j * Pa + i] = accumulate ? G[(long)j * Pa + i] + acc : acc;
    }
}

// dst[r][jj] (leading dimension ldd) = (R_red E)[r][j] for r < Pra and the Pout output columns jj: j = colmap[jj] (a column subset of the
// augmented layout, fbr_tsqr_cols through the reductions) or jj itself (colmap == NULL, Pout = cols + k): the rows the final factor folds
__global__ __launch_bounds__(256) void fbr_expand_rows_kernel(int cols, int k, int Pra, const int *__restrict__ Eb, const int *__restrict__ Er,
                                                               const double *__restrict__ Ev, const double *__restrict__ Rred, double *__restrict__ dst,
                                                               int ldd, const int *__restrict__ colmap, int Pout)
{
    Rred += (long)blockIdx.y * Pra * Pra;  // blockIdx.y: sample group (a dense [Pra x ldd] block of dst per group)
    dst += (long)blockIdx.y * Pra * ldd;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)Pra * Pout; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / Pout), jj = (int)(e - (long)r * Pout);
        const int j = colmap ? colmap[jj] : jj;
        double acc = 0.0;
        for (int b = Eb[j]; b < Eb[j + 1]; b++) acc += Ev[b] * Rred[(long)r * Pra + Er[b]];
        dst[(long)r * ldd + jj] = acc;
    }
}

int launch_expand_rows(fbr_model *m, int which, int k, int Pra, const double *Rred, double *dst, int ldd, const int *colmap, int Pout)
{
    hipLaunchKernelGGL(fbr_expand_rows_kernel, dim3(512), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which], m->E_val[which], Rred, dst, ldd,
                       colmap, colmap ? Pout : m->hm.cols + k);
    HIPCHK(hipGetLastError());
    return FBR_OK;
}
// ------------------------------------------------------------------------------------------------
// fused Gram
// ------------------------------------------------------------------------------------------------
static int get_gram(fbr_model *m, int k, GramHolder **out, bool moments = false)
{
    const int key = k + (moments ? 64 : 0);
    auto it = m->gram.find(key);
    if (it != m->gram.end()) {
        *out = it->second.get();
        return FBR_OK;
    }
    std::unique_ptr<GramHolder> h(new GramHolder());
    h->moments = moments;
    try {
        fbr_gram_build_best(h->prog, m->hm, k, (int)m->opt.gram_shape, !moments, m->opt.gram_orient != 0);
    } catch (const std::exception &e) {
        set_err(std::string("gram program: ") + e.what());
        return FBR_E_INVALID;
    }
    FbrGramProgram &gp = h->prog;
    DevGram &dg = h->dev;
    memset(&dg, 0, sizeof(dg));
    dg.T = gp.T; dg.NT = gp.NT; dg.k = gp.k; dg.Pa = gp.Pa; dg.image_doubles = gp.image_doubles;
    dg.part_image_max = gp.part_image_max;
    dg.nitems = (int)gp.items.size();
    if (gp.part_image_max > 1023 * 64 || m->hm.rows > 255) {
        set_err("model too large for the fused Gram tile image");
        return FBR_E_UNSUPPORTED;
    }
    std::vector<int4> items;
    for (auto &it2 : gp.items) items.push_back(make_int4(it2.off, it2.kind, it2.a, it2.b));
    // per part: DMA pieces and the part-image-row -> regressor-row map (identity in dense tiles)
    std::vector<int2> pieces;
    std::vector<int> piece_begin(gp.T + 1, 0), rid_begin(gp.T + 1, 0), ridl;
    for (int t = 0; t < gp.T; t++) {
        piece_begin[t] = (int)pieces.size();
        for (auto &pc : gp.pieces[t]) pieces.push_back(make_int2(pc.goff, pc.loff | (pc.half << 30)));
        rid_begin[t] = (int)ridl.size();
        std::vector<int> rl((size_t)gp.part_image_max / FBR_TILE, 0);
        for (int ti : gp.part_tiles[t])
            for (size_t j = 0; j < gp.tiles[ti].rowid.size(); j++) rl[(size_t)gp.part_tile_off[t][ti] / FBR_TILE + j] = gp.tiles[ti].rowid[j];
        ridl.insert(ridl.end(), rl.begin(), rl.end());
    }
    piece_begin[gp.T] = (int)pieces.size();
    rid_begin[gp.T] = (int)ridl.size();
    // base-wrench-only launches read the first 8 packed rows (one 1 KiB DMA) of every tile only
    std::vector<int2> pieces_b;
    std::vector<int> piece_begin_b(gp.T + 1, 0);
    for (int t = 0; t < gp.T; t++) {
        piece_begin_b[t] = (int)pieces_b.size();
        for (int ti : gp.part_tiles[t]) pieces_b.push_back(make_int2(gp.tiles[ti].off, gp.part_tile_off[t][ti]));
    }
    piece_begin_b[gp.T] = (int)pieces_b.size();
    const int FBR_SEGW = gp.cfg.segw, FBR_NSEG = gp.cfg.nseg, FBR_NPW = gp.cfg.npw();
    dg.npw = FBR_NPW;
    dg.base_ks = gp.base_ks;
    const size_t nslots = gp.slots.size();
    std::vector<int> meta((size_t)gp.T * FBR_WPB * FBR_NSEG * 8, 0);
    std::vector<int> slot_tiles(2 * nslots, -1);
    for (int part = 0; part < gp.T; part++)
        for (int w = 0; w < FBR_WPB; w++)
            for (int sg = 0; sg < FBR_NSEG; sg++) {
                int *mm = &meta[(((size_t)part * FBR_WPB + w) * FBR_NSEG + sg) * 8];
                int cnt = 0, offA = 0, kb = 0, last_nk = 1 << 30, chainA = 0;
                bool sorted = true;
                for (int j = 0; j < FBR_SEGW; j++) {
                    const size_t s = ((size_t)part * FBR_WPB + w) * FBR_NPW + sg * FBR_SEGW + j;
                    const int pi = gp.slots[s].pair;
                    if (pi < 0) continue;
                    const FbrPair &p = gp.pairs[pi];
                    offA = gp.part_tile_off[part][p.I];
                    chainA = gp.tiles[p.I].type == 0;  // packed positions: the odd sample of a pair skips the base k-steps
                    kb = gp.slots[s].kb;
                    const int offB = gp.part_tile_off[part][p.J];
                    mm[1 + j] = (offB / 64) | ((p.mode == 1 ? 1 : 0) << 10) | (p.nkend() << 11);
                    // the kernel relies on: last k-steps falling along the slots, no holes before a slot, one start per segment
                    if (p.nkend() > last_nk || cnt != j || p.kbegin() < kb) sorted = false;
                    last_nk = p.nkend();
                    cnt++;
                    slot_tiles[2 * s] = p.I;
                    slot_tiles[2 * s + 1] = p.J;
                }
                mm[0] = (offA / 64) | (cnt << 10) | (kb << 18) | (chainA << 23);
                if (cnt && !sorted) {
                    set_err("internal: row segment is not sorted by k-steps");
                    return FBR_E_INVALID;
                }
            }
    std::vector<int> tilecol((size_t)gp.NT * FBR_TILE);
    for (int t = 0; t < gp.NT; t++)
        for (int s = 0; s < FBR_TILE; s++) tilecol[(size_t)t * FBR_TILE + s] = gp.tiles[t].col[s];
    int rc;
    h->pool.reserve(16);
    if ((rc = upload(h->pool, items, &dg.items))) return rc;
    if ((rc = upload(h->pool, meta, &dg.slotmeta))) return rc;
    if ((rc = upload(h->pool, piece_begin, &dg.piece_begin))) return rc;
    if ((rc = upload(h->pool, pieces, &dg.pieces))) return rc;
    if ((rc = upload(h->pool, piece_begin_b, &dg.piece_begin_b))) return rc;
    if ((rc = upload(h->pool, pieces_b, &dg.pieces_b))) return rc;
    if ((rc = upload(h->pool, rid_begin, &dg.rid_begin))) return rc;
    if ((rc = upload(h->pool, ridl, &dg.ridl))) return rc;
    if ((rc = upload(h->pool, slot_tiles, &dg.slot_tiles))) return rc;
    if ((rc = upload(h->pool, tilecol, &dg.tilecol))) return rc;
    if (moments) {
        // (the column comes from the item itself: looking it up through the image offset fails for tiles WITHOUT rows -- the base link and
        // the links welded to it on a fixed base -- which share their offset with the next tile: their all-zero items were then taken for
        // that tile's columns, and two reduction workgroups raced on one G entry.  Found in round 5 by the prismatic-joint tests.)
        std::vector<int> itemcol(256, -1);
        for (size_t i = 0; i < gp.items.size() && i < 256; i++) itemcol[i] = gp.items[i].col;
        if ((rc = upload(h->pool, itemcol, &h->itemcol))) return rc;
    }
    size_t max_pieces = 0;
    for (auto &v : gp.pieces) max_pieces = std::max(max_pieces, v.size());
    h->lds_bytes = (size_t)2 * gp.part_image_max * sizeof(double) +
                   ((size_t)gp.part_image_max / FBR_TILE + FBR_WPB * FBR_NSEG * 8 + 2 * max_pieces) * sizeof(int);
    {
        const int stage = m->hm.rec_size() + m->hm.rows * gp.k + m->hm.rows + 2 * m->hm.n;
        h->pack_lds_bytes = (size_t)((stage + 1) & ~1) * sizeof(double) +
                            ((size_t)m->hm.L + (size_t)2 * m->hm.L * std::max(m->hm.maxdepth, 1)) * sizeof(int);
    }
    if (h->lds_bytes > 160 * 1024 || h->pack_lds_bytes > 160 * 1024) {
        set_err("model too large: fused Gram needs more than 160 KiB of LDS");
        return FBR_E_UNSUPPORTED;
    }
    *out = h.get();
    m->gram[key] = std::move(h);
    return FBR_OK;
}

extern "C" int fbr_gram_program_info(const fbr_model *mc, int32_t k, int64_t num_samples, int32_t *num_tiles, int32_t *num_pairs,
                                     int64_t *mfma_per_sample, int32_t *num_parts)
{
    if (!mc) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    fbr_model *m = const_cast<fbr_model *>(mc);
    if (int rc_enter = enter(m)) return rc_enter;
    // what fbr_gram_accumulate runs on a batch of num_samples samples (< 0: a batch large enough for the column reductions)
    if (const int wr = pick_gram_reduction(m, num_samples < 0 ? -1 : (long)num_samples); wr >= 0) m = m->rdm[wr].get();
    GramHolder *h = nullptr;
    int rc = get_gram(m, k, &h, fbr_gram_rhs_moments(m->hm, k, m->opt.gram_rhs_tile != 0));
    if (rc) return rc;
    if (num_tiles) *num_tiles = h->prog.NT;
    if (num_pairs) *num_pairs = (int32_t)h->prog.pairs.size();
    if (mfma_per_sample) *mfma_per_sample = h->prog.mfma_per_sample;
    if (num_parts) *num_parts = h->prog.T;
    return FBR_OK;
}

static int get_gram64(fbr_model *m, GramHolder *h);

extern "C" int fbr_gram_lane_info(const fbr_model *mc, int32_t k, int64_t num_samples, int64_t info[12])
{
    if (!mc || !info) {
        set_err("null model / info");
        return FBR_E_INVALID;
    }
    fbr_model *m = const_cast<fbr_model *>(mc);
    if (int rc_enter = enter(m)) return rc_enter;
    if (const int wr = pick_gram_reduction(m, num_samples < 0 ? -1 : (long)num_samples); wr >= 0) m = m->rdm[wr].get();
    for (int i = 0; i < 12; i++) info[i] = 0;
    const bool moments = fbr_gram_rhs_moments(m->hm, k, m->opt.gram_rhs_tile != 0);
    GramHolder *h = nullptr;
    int rc = get_gram(m, k, &h, moments);
    if (rc) return rc;
    if (!m->opt.gram_lane || k > 1 || (k == 1 && !moments)) return FBR_OK;
    if ((rc = get_gram64(m, h))) return rc;
    if (h->g64_state != 1) return FBR_OK;
    const FbrGram64 &g = h->g64;
    info[0] = 1;
    info[1] = g.ntr;
    info[2] = g.blk_doubles * (int64_t)sizeof(double);
    info[3] = g.mfma_per_block;
    info[4] = g.nlev;
    info[5] = g.maxact;
    info[6] = (int64_t)((size_t)2 * g.maxact * 512 * sizeof(double) +
                        ((size_t)g.nlev * (g.NT + g.NF) + g.nlev + 1 + g.pieces.size() + g.wmeta.size() + g.stage_lev.size()) * sizeof(int));
    info[7] = g.NT;
    info[8] = g.NF;
    info[9] = g.busiest;
    info[10] = g.balanced;
    info[11] = g.nstage;
    return FBR_OK;
}

// G (+)= R^T R for an upper-triangular R (Pa x Pa): the Gram of a robot the fused tile program does not cover, from its TSQR factor
__global__ __launch_bounds__(256) void fbr_rtr_kernel(int Pa, const double *__restrict__ R, double *__restrict__ G, int accumulate)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)Pa * Pa; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / Pa), j = (int)(e % Pa);
        double acc = 0.0;
        for (int r = 0; r <= min(i, j); r++) acc += R[(long)r * Pa + i] * R[(long)r * Pa + j];
        G[e] = accumulate ? G[e] + acc : acc;
    }
}

// Robots with more than 60 regressor rows per sample (54 DOF on a floating base) are outside the fused Gram's tile program (15 MFMA
// k-steps per tile pair).  Their Gram is formed from the Householder factor of the same rows: G = R^T R with R from fbr_tsqr (up to 255
// rows per sample and 768 columns) -- slower than the fused pass, numerically at least as good, and it keeps every caller of
// fbr_gram_accumulate working for any URDF the reference loads (model.py:116-168).
static int gram_via_tsqr(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out, int32_t out_mem,
                         int32_t accumulate)
{
    const int Pa = m->hm.cols + k;
    const size_t cnt = (size_t)Pa * Pa;
    int rc;
    if ((rc = m->gram_r_tmp.ensure(cnt * sizeof(double)))) return rc;
    double *R = m->gram_r_tmp.as<double>();
    if ((rc = tsqr_impl(m, st, nullptr, 0, rhs, k, w, nullptr, R, FBR_DEVICE, nullptr))) return rc;
    double *G = G_out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(cnt * sizeof(double)))) return rc;
        G = m->g_tmp.as<double>();
        if (accumulate) HIPCHK(hipMemcpyAsync(G, G_out, cnt * sizeof(double), hipMemcpyHostToDevice, m->stream));
    }
    hipLaunchKernelGGL(fbr_rtr_kernel, dim3(1024), dim3(256), 0, m->stream, Pa, R, G, accumulate ? 1 : 0);
    HIPCHK(hipGetLastError());
    return finish_output(m, G, G_out, cnt, out_mem);
}
// ------------------------------------------------------------------------------------------------
// The pass over sample-contiguous images (fbr_gram64.h): tables of a holder, built on first use
// ------------------------------------------------------------------------------------------------
static int get_gram64(fbr_model *m, GramHolder *h)
{
    if (h->g64_state) return FBR_OK;
    h->g64_state = -1;
    const FbrHostModel &hm = m->hm;
    if (m->kinid.nsteps <= 0 || !fbr_gram64_build(hm, h->prog, h->g64, m->opt.gram_force_tiles != 0, m->opt.gram_lane_waves >= 16)) return FBR_OK;
    FbrGram64 &g = h->g64;
    const size_t lds = (size_t)2 * g.maxact * 512 * sizeof(double) +
                       ((size_t)g.nlev * (g.NT + g.NF) + g.nlev + 1 + g.pieces.size() + g.wmeta.size() + g.stage_lev.size()) * sizeof(int);
    if (lds > 156 * 1024) return FBR_OK;
    if (!fbr_gram64_build_producer(hm, h->prog, g, h->g64p)) return FBR_OK;
    std::vector<int> wgbegin{0, 0};  // (filled per launch: one part)
    int rc;
    if ((rc = upload(h->pool, g.slab, &h->d64_slab)) || (rc = upload(h->pool, g.lev_begin, &h->d64_levb)) || (rc = upload(h->pool, g.pieces, &h->d64_pieces)) ||
        (rc = upload(h->pool, g.wmeta, &h->d64_wmeta)) || (rc = upload(h->pool, h->g64p.lcol, &h->d64_lcol)) ||
        (rc = upload(h->pool, h->g64p.steps, &h->d64_steps)) ||
        (rc = upload(h->pool, g.slot_tiles, &h->d64_slot_tiles)) || (rc = upload(h->pool, g.tilecol, &h->d64_tilecol)) ||
        (rc = upload(h->pool, g.stage_lev, &h->d64_stagelev)))
        return rc;
    h->g64_state = 1;
    return FBR_OK;
}

// blocks a chunk of the sample-contiguous pass may hold (two image buffers of at most 3 GB)
static long gram64_chunk_blocks(const FbrGram64 &g) { return std::max<long>(1, (long)((size_t)3 * 1024 * 1024 * 1024 / ((size_t)g.blk_doubles * 8))); }

// [0, workgroups]: the one-part workgroup table fbr_gram_reduce_kernel reads, uploaded once per grid size a holder has used
static int gram64_wg_table(GramHolder *h, int wgs, const int **out)
{
    auto it = h->d64_wb.find(wgs);
    if (it == h->d64_wb.end()) {
        std::vector<int> wb{0, wgs};
        const int *dwb = nullptr;
        if (int rc = upload(h->pool, wb, &dwb)) return rc;
        it = h->d64_wb.emplace(wgs, dwb).first;
    }
    *out = it->second;
    return FBR_OK;
}

// the two image buffers of a holder for chunks of chb blocks, and the producer's destination words inside them
static int gram64_ensure_images(fbr_model *m, GramHolder *h, long chb)
{
    const FbrGram64 &g = h->g64;
    int rc;
    if (chb <= h->img64_blocks) return FBR_OK;
    for (int b = 0; b < 2; b++) {
        if ((rc = h->img64[b].ensure((size_t)chb * g.blk_doubles * sizeof(double)))) return rc;
        HIPCHK(hipMemsetAsync(h->img64[b].p, 0, h->img64[b].bytes, m->stream));  // structural zeros and padding slots are never written
        std::vector<long long> dst(h->g64p.rel.size(), 0);
        for (size_t i = 0; i < dst.size(); i++)
            if (h->g64p.rel[i]) dst[i] = (long long)(uintptr_t)h->img64[b].p + (h->g64p.rel[i] & ~(1LL << 62));
        if ((rc = h->dst64[b].ensure(dst.size() * sizeof(long long)))) return rc;
        HIPCHK(hipMemcpyAsync(h->dst64[b].p, dst.data(), dst.size() * sizeof(long long), hipMemcpyHostToDevice, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));  // (dst is a local)
    }
    h->img64_blocks = chb;
    return FBR_OK;
}

// One call of the fused pass through fbr_kinimg_kernel / fbr_gram64_kernel (device-resident inputs, one group, k <= 1); everything on the
// model's stream.  G has been cleared / holds the running sum.
// h2d_chunked: d / drhs / dw are PINNED HOST pointers: every chunk is copied into one of two staging buffers on the copy stream while the
// kernels of the chunk before run (the rate including the transfer of the states).
static int gram64_pass(fbr_model *m, GramHolder *h, const DevStates &d, const double *drhs, const double *dw, int k, double *G, bool base_only,
                       bool h2d_chunked)
{
    const FbrHostModel &hm = m->hm;
    FbrGram64 &g = h->g64;
    const long S = d.S;
    const int Pa = hm.cols + k;
    int rc;
    // blocks per chunk: what the image buffers hold (a call that fits is ONE chunk: a 125 k-sample shard ran 1792 + 162 blocks before);
    // several chunks: whole rounds of the chip (both kernels walk blocks workgroup by workgroup).  Pinned inputs: chunks of seven rounds
    // (measured: 20.7 ms per 1 M-sample step against 21.2 with chunks of four -- with two submissions in flight the first copy of a step
    // hides behind the last kernels of the step before, and fewer chunks mean fewer launches)
    const long nblocks = (S + 63) / 64;
    long chb = gram64_chunk_blocks(g);
    if (h2d_chunked) chb = std::min<long>(chb, 7L * m->num_cus);
    if (m->opt.chunk_samples >= 1) chb = std::max<long>(1, ((long)m->opt.chunk_samples + 63) / 64);  // (tests: the multi-chunk paths at small sizes)
    chb = std::min(nblocks, chb);
    if (chb < nblocks && chb > m->num_cus) chb = chb / m->num_cus * m->num_cus;
    if ((rc = gram64_ensure_images(m, h, chb))) return rc;
    const int ldn = std::max(hm.n, 1) | 1, ldw = hm.rows | 1;
    const size_t plds = ((size_t)3 * 64 * ldn + (dw ? (size_t)64 * ldw : 0) + (k ? (size_t)64 * ldw : 0)) * sizeof(double);
    // producer grid: two workgroups per CU where the kernel instance fits 256 registers (fbr_kinimg_kernel's launch bounds)
    const int pgrid_max = (m->kinid.maxlvl <= 10 ? 2 : 1) * m->num_cus;
    const int pblocks = (int)std::min<long>(chb, (long)pgrid_max);
    const int gwgs = (int)std::min<long>(chb, (long)m->num_cus);
    if ((rc = h->scr64.ensure((size_t)pgrid_max * h->g64p.nparts * std::max(h->g64p.nslots, 1) * FBR_LINK_REC * 64 * sizeof(double)))) return rc;
    if (k) {
        if ((rc = h->mom64.ensure((size_t)pgrid_max * (hm.cols + 1) * 64 * sizeof(double)))) return rc;
        HIPCHK(hipMemsetAsync(h->mom64.p, 0, (size_t)pblocks * (hm.cols + 1) * 64 * sizeof(double), m->stream));  // (the producer grid of this call)
    }
    const int npw = g.npw;
    if ((rc = m->partial.ensure((size_t)m->num_cus * g.wpb * npw * 256 * sizeof(double)))) return rc;
    const size_t glds = (size_t)2 * g.maxact * 512 * sizeof(double) + ((size_t)g.nlev * (g.NT + g.NF) + g.nlev + 1 + g.pieces.size() + g.wmeta.size() + g.stage_lev.size()) * sizeof(int);
    DevKinId kp;
    kp.nsteps = 0;
    kp.maxlvl = m->kinid.maxlvl;
    kp.nslots = h->g64p.nslots;
    kp.ldn = ldn;
    kp.steps = h->d64_steps;
    kp.endflush = m->kinid_endflush;
    DevGram64 dg;
    dg.NT = g.NT + g.NF;
    dg.nlev = g.nlev;
    dg.maxact = g.maxact;
    dg.npieces = (int)g.pieces.size() / 2;
    dg.blk_doubles = g.blk_doubles;
    dg.slab = h->d64_slab;
    dg.lev_begin = h->d64_levb;
    dg.pieces = h->d64_pieces;
    dg.wmeta = h->d64_wmeta;
    dg.nstage = base_only ? g.base_stages : g.nstage;  // (base-wrench-only row masks: the joint levels' stages are not run)
    dg.stage_lev = h->d64_stagelev;
    typedef void (*g64_fn)(DevGram64, long, const double *, double *, int);
    const g64_fn gk = g.wpb == 16 ? fbr_gram64_kernel<10, 16> : (g.npw == 10 ? fbr_gram64_kernel<10, 8> : fbr_gram64_kernel<FBR_ONE_SEGW * FBR_ONE_NSEG, 8>);
    const int vnpw = g.npw * (g.wpb / 8);  // accumulator slots per ROW of the partial sums (fbr_gram_reduce_kernel walks 8 rows per workgroup)
    HIPCHK(hipFuncSetAttribute((const void *)gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
    int launches = 0, first_wgs = 0;
    const size_t stage_per = (size_t)3 * hm.n + (hm.floating ? 15 : 0) + (d.sign ? hm.n : 0) + (size_t)hm.rows * k + (dw ? hm.rows : 0);  // doubles per staged sample
    if (h2d_chunked) {
        for (int b = 0; b < 2; b++)
            if ((rc = m->st_chunk[b].ensure(std::max<size_t>(1, (size_t)chb * 64 * stage_per) * sizeof(double)))) return rc;
        if (!m->copy) {  // (a priority level of its own: see the per-sample-image pass below)
            int least = 0, greatest = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPCHK(hipStreamCreateWithPriority(&m->copy, hipStreamNonBlocking, greatest));
        }
        HIPCHK(hipEventRecord(m->ev_fork, m->stream));
    }
    // the states of chunk c in device memory: the caller's arrays, or staging buffer (c & 1) filled on the copy stream
    struct Staged {
        DevStates dc;
        long o;
        const double *rhs, *w;
    };
    auto stage = [&](long c, Staged &out) -> int {
        const long s0 = c * chb * 64, cs = std::min(chb * 64, S - s0);
        const int b = (int)(c & 1);
        out.dc = d;
        out.o = s0;
        out.rhs = drhs;
        out.w = dw;
        if (!h2d_chunked) return FBR_OK;
        // the buffer's last reader is the producer launch of the chunk two before (or of an earlier call: the event is simply complete then)
        HIPCHK(hipStreamWaitEvent(m->copy, m->ev_pack_rec[b] ? m->ev_pack[b] : m->ev_fork, 0));
        ProfScope ps(m, FBR_PROF_H2D, m->copy);
        double *p = m->st_chunk[b].as<double>();
        auto put = [&](const double *src, size_t per, const double **dst) -> int {
            *dst = nullptr;
            if (!src || per == 0) return FBR_OK;
            HIPCHK(hipMemcpyAsync(p, src + (size_t)s0 * per, (size_t)cs * per * sizeof(double), hipMemcpyHostToDevice, m->copy));
            *dst = p;
            p += (size_t)cs * per;
            return FBR_OK;
        };
        int r3;
        if ((r3 = put(d.q, hm.n, &out.dc.q)) || (r3 = put(d.dq, hm.n, &out.dc.dq)) || (r3 = put(d.ddq, hm.n, &out.dc.ddq)) ||
            (r3 = put(d.bv, 6, &out.dc.bv)) || (r3 = put(d.ba, 6, &out.dc.ba)) || (r3 = put(d.rpy, 3, &out.dc.rpy)) ||
            (r3 = put(d.sign, hm.n, &out.dc.sign)) || (r3 = put(drhs, (size_t)hm.rows * k, &out.rhs)) || (r3 = put(dw, hm.rows, &out.w)))
            return r3;
        out.o = 0;
        HIPCHK(hipEventRecord(m->ev_h2d[b], m->copy));
        return FBR_OK;
    };
    Staged cur, nxt;
    if (S > 0 && (rc = stage(0, nxt))) return rc;
    for (long b0 = 0; b0 * 64 < S; b0 += chb, launches++) {
        const long s0 = b0 * 64, cs = std::min(chb * 64, S - s0), nb = (cs + 63) / 64;
        const int b = launches & 1;
        cur = nxt;
        if ((b0 + chb) * 64 < S && (rc = stage(launches + 1, nxt))) return rc;  // the copy of the next chunk is enqueued before this chunk's kernels
        if (h2d_chunked) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_h2d[b], 0));
        const DevStates &dc = cur.dc;
        const long so = cur.o;
        const double *crhs = cur.rhs, *cw = cur.w;
        DevKinWrite kw;
        kw.lcol10 = h->d64_lcol;
        kw.colrec = nullptr;
        kw.dst = (const long *)h->dst64[b].p;
        kw.ninert = hm.ninert;
        kw.cols = hm.cols;
        kw.k = k;
        kw.has_w = dw ? 1 : 0;
        kw.flev = g.flev;
        kw.base_only = base_only ? 1 : 0;
        kw.group_samples = 0;
        kw.nparts = h->g64p.nparts;
        for (int pq = 0; pq < FBR_KINWRITE_PARTS; pq++) {
            kw.part_nsteps[pq] = h->g64p.nsteps[pq];
            kw.part_step0[pq] = h->g64p.step0[pq];
        }
        if (cs & 63)  // the block the producer fills partly: what its idle lanes would have written (a buffer is reused from chunk to chunk)
            HIPCHK(hipMemsetAsync(h->img64[b].as<double>() + (nb - 1) * g.blk_doubles, 0, (size_t)g.blk_doubles * sizeof(double), m->stream));
        {
            ProfScope ps(m, FBR_PROF_PACK);
#define FBR_KINIMG_LAUNCH2(D, W)                                                                                                                    \
    do {                                                                                                                                         \
        HIPCHK(hipFuncSetAttribute((const void *)fbr_kinimg_kernel<D, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));                 \
        hipLaunchKernelGGL((fbr_kinimg_kernel<D, W>), dim3(pblocks), dim3(64 * h->g64p.nparts), plds, m->stream, m->dm, kp, kw, cs, g.blk_doubles,       \
                           dc.q + so * hm.n, dc.dq + so * hm.n, dc.ddq + so * hm.n, dc.bv ? dc.bv + so * 6 : nullptr, dc.ba ? dc.ba + so * 6 : nullptr, \
                           dc.rpy ? dc.rpy + so * 3 : nullptr, crhs ? crhs + (size_t)so * hm.rows * k : nullptr,                                   \
                           cw ? cw + (size_t)so * hm.rows : nullptr, h->scr64.as<double>(), k ? h->mom64.as<double>() : nullptr,                   \
                           dc.sign ? dc.sign + so * hm.n : nullptr);                                                                              \
    } while (0)
#define FBR_KINIMG_LAUNCH(D)            \
    do {                                \
        if (dw)                         \
            FBR_KINIMG_LAUNCH2(D, true); \
        else                            \
            FBR_KINIMG_LAUNCH2(D, false); \
    } while (0)
            if (kp.maxlvl <= 4)
                FBR_KINIMG_LAUNCH(4);
            else if (kp.maxlvl <= 8)
                FBR_KINIMG_LAUNCH(8);
            else if (kp.maxlvl <= 10)
                FBR_KINIMG_LAUNCH(10);
            else if (kp.maxlvl <= 12)
                FBR_KINIMG_LAUNCH(12);
            else
                FBR_KINIMG_LAUNCH(FBR_KINID_MAXD);
#undef FBR_KINIMG_LAUNCH2
#undef FBR_KINIMG_LAUNCH
            HIPCHK(hipGetLastError());
            if (h2d_chunked) {  // (the staging buffer may be refilled once this launch is through)
                HIPCHK(hipEventRecord(m->ev_pack[b], m->stream));
                m->ev_pack_rec[b] = true;
            }
        }
        // the accumulators are carried from chunk to chunk by workgroup index: every chunk but a shorter last one starts the same grid
        const int wgs = (int)std::min<long>(nb, (long)gwgs);
        if (launches == 0) first_wgs = wgs;
        if (wgs > first_wgs) {
            set_err("internal: Gram launch wider than the first one of the call");
            return FBR_E_INVALID;
        }
        {
            ProfScope ps(m, FBR_PROF_GRAM);
            hipLaunchKernelGGL(gk, dim3(first_wgs), dim3(g.wpb * 64), glds, m->stream, dg, nb, h->img64[b].as<double>(), m->partial.as<double>(), launches > 0 ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
    }
    if (launches > 0) {
        ProfScope ps(m, FBR_PROF_REDUCE);
        DevGram dr = h->dev;  // the reduction of fbr_gram_reduce_kernel: one part of first_wgs workgroups
        dr.wpg = first_wgs;
        dr.npw = vnpw;
        dr.slot_tiles = h->d64_slot_tiles;
        if ((rc = gram64_wg_table(h, first_wgs, &dr.wg_begin))) return rc;
        dr.tilecol = h->d64_tilecol;
        hipLaunchKernelGGL(fbr_gram_reduce_kernel, dim3(FBR_WPB * vnpw, 1), dim3(256), 0, m->stream, dr, m->partial.as<double>(), G);
        HIPCHK(hipGetLastError());
        if (g.NF > 0) {  // the blocks of the force tiles: entries the main blocks have written too, hence a launch of their own
            dr.slot_tiles = h->d64_slot_tiles + (size_t)FBR_WPB * vnpw * 2;
            hipLaunchKernelGGL(fbr_gram_reduce_kernel, dim3(FBR_WPB * vnpw, 1), dim3(256), 0, m->stream, dr, m->partial.as<double>(), G);
            HIPCHK(hipGetLastError());
        }
        if (k) {
            hipLaunchKernelGGL(fbr_gram64_mom_reduce_kernel, dim3(hm.cols + 1), dim3(256), 0, m->stream, hm.cols, pblocks, h->mom64.as<double>(), G);
            HIPCHK(hipGetLastError());
        }
    }
    (void)Pa;
    return FBR_OK;
}

// fbr_gram_grouped through the sample-contiguous pass (k = 0, device-resident inputs): every group starts a block of 64 samples, a group's
// blocks are shared by wpg workgroups whose partial sums one reduction per group adds up; groups are taken a chunk of whole groups at a time.
// G: [ngroups][Pa][Pa], cleared.
static int gram64_grouped_pass(fbr_model *m, GramHolder *h, const DevStates &d, const double *dw, double *G, int ngroups)
{
    const FbrHostModel &hm = m->hm;
    FbrGram64 &g = h->g64;
    const long S = d.S, Sg = S / ngroups, bpg = (Sg + 63) / 64;
    const int Pa = hm.cols, npw = g.npw;
    int rc;
    const int gpc = (int)std::min<long>(ngroups, std::max<long>(1, gram64_chunk_blocks(g) / bpg));  // groups per chunk
    if ((rc = gram64_ensure_images(m, h, (long)gpc * bpg))) return rc;
    const int wpg = (int)std::max<long>(1, std::min<long>(bpg, (long)m->num_cus / std::min(gpc, ngroups)));
    const int ldn = std::max(hm.n, 1) | 1, ldw = hm.rows | 1;
    const size_t plds = ((size_t)3 * 64 * ldn + (dw ? (size_t)64 * ldw : 0)) * sizeof(double);
    const int pgrid_max = (m->kinid.maxlvl <= 10 ? 2 : 1) * m->num_cus;
    if ((rc = h->scr64.ensure((size_t)pgrid_max * h->g64p.nparts * std::max(h->g64p.nslots, 1) * FBR_LINK_REC * 64 * sizeof(double)))) return rc;
    if ((rc = m->partial.ensure((size_t)gpc * wpg * g.wpb * npw * 256 * sizeof(double)))) return rc;
    const size_t glds = (size_t)2 * g.maxact * 512 * sizeof(double) + ((size_t)g.nlev * (g.NT + g.NF) + g.nlev + 1 + g.pieces.size() + g.wmeta.size() + g.stage_lev.size()) * sizeof(int);
    DevKinId kp;
    kp.nsteps = 0;
    kp.maxlvl = m->kinid.maxlvl;
    kp.nslots = h->g64p.nslots;
    kp.ldn = ldn;
    kp.steps = h->d64_steps;
    kp.endflush = m->kinid_endflush;
    DevGram64 dg;
    dg.NT = g.NT + g.NF;
    dg.nlev = g.nlev;
    dg.maxact = g.maxact;
    dg.npieces = (int)g.pieces.size() / 2;
    dg.blk_doubles = g.blk_doubles;
    dg.slab = h->d64_slab;
    dg.lev_begin = h->d64_levb;
    dg.pieces = h->d64_pieces;
    dg.wmeta = h->d64_wmeta;
    dg.nstage = g.nstage;
    dg.stage_lev = h->d64_stagelev;
    typedef void (*g64_fn)(DevGram64, long, const double *, double *, int);
    const g64_fn gk = g.wpb == 16 ? fbr_gram64_kernel<10, 16> : (g.npw == 10 ? fbr_gram64_kernel<10, 8> : fbr_gram64_kernel<FBR_ONE_SEGW * FBR_ONE_NSEG, 8>);
    const int vnpw = g.npw * (g.wpb / 8);  // accumulator slots per ROW of the partial sums (fbr_gram_reduce_kernel walks 8 rows per workgroup)
    HIPCHK(hipFuncSetAttribute((const void *)gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
    const int *wgb = nullptr;
    if ((rc = gram64_wg_table(h, wpg, &wgb))) return rc;
    int launches = 0;
    for (int g0 = 0; g0 < ngroups; g0 += gpc, launches++) {
        const int ng = std::min(gpc, ngroups - g0), b = launches & 1;
        const long s0 = (long)g0 * Sg, cs = (long)ng * Sg, nb = (long)ng * bpg;
        DevKinWrite kw;
        kw.lcol10 = h->d64_lcol;
        kw.colrec = nullptr;
        kw.dst = (const long *)h->dst64[b].p;
        kw.ninert = hm.ninert;
        kw.cols = hm.cols;
        kw.k = 0;
        kw.has_w = dw ? 1 : 0;
        kw.flev = g.flev;
        kw.base_only = 0;
        kw.group_samples = Sg;
        kw.nparts = h->g64p.nparts;
        for (int pq = 0; pq < FBR_KINWRITE_PARTS; pq++) {
            kw.part_nsteps[pq] = h->g64p.nsteps[pq];
            kw.part_step0[pq] = h->g64p.step0[pq];
        }
        const int pblocks = (int)std::min<long>(nb, (long)pgrid_max);
        if (Sg & 63) {  // every group ends in a block the producer fills partly: what its idle lanes would have written
            hipLaunchKernelGGL(fbr_gram64_tail_zero_kernel, dim3(16, ng), dim3(256), 0, m->stream, h->img64[b].as<double>(), g.blk_doubles, bpg, g.ntr, (int)(Sg & 63));
            HIPCHK(hipGetLastError());
        }
        {
            ProfScope ps(m, FBR_PROF_PACK);
#define FBR_KINIMG_GLAUNCH2(D, W)                                                                                                                 \
    do {                                                                                                                                         \
        HIPCHK(hipFuncSetAttribute((const void *)fbr_kinimg_kernel<D, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));                 \
        hipLaunchKernelGGL((fbr_kinimg_kernel<D, W>), dim3(pblocks), dim3(64 * h->g64p.nparts), plds, m->stream, m->dm, kp, kw, cs, g.blk_doubles, \
                           d.q + s0 * hm.n, d.dq + s0 * hm.n, d.ddq + s0 * hm.n, d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr,    \
                           d.rpy ? d.rpy + s0 * 3 : nullptr, (const double *)nullptr, dw ? dw + (size_t)s0 * hm.rows : nullptr,                    \
                           h->scr64.as<double>(), (double *)nullptr, d.sign ? d.sign + s0 * hm.n : nullptr);                                      \
    } while (0)
#define FBR_KINIMG_GLAUNCH(D)             \
    do {                                  \
        if (dw)                           \
            FBR_KINIMG_GLAUNCH2(D, true); \
        else                              \
            FBR_KINIMG_GLAUNCH2(D, false); \
    } while (0)
            if (kp.maxlvl <= 4)
                FBR_KINIMG_GLAUNCH(4);
            else if (kp.maxlvl <= 8)
                FBR_KINIMG_GLAUNCH(8);
            else if (kp.maxlvl <= 10)
                FBR_KINIMG_GLAUNCH(10);
            else if (kp.maxlvl <= 12)
                FBR_KINIMG_GLAUNCH(12);
            else
                FBR_KINIMG_GLAUNCH(FBR_KINID_MAXD);
#undef FBR_KINIMG_GLAUNCH2
#undef FBR_KINIMG_GLAUNCH
            HIPCHK(hipGetLastError());
        }
        {
            ProfScope ps(m, FBR_PROF_GRAM);
            hipLaunchKernelGGL(gk, dim3(wpg, ng), dim3(g.wpb * 64), glds, m->stream, dg, bpg, h->img64[b].as<double>(), m->partial.as<double>(), 0);
            HIPCHK(hipGetLastError());
        }
        {
            ProfScope ps(m, FBR_PROF_REDUCE);
            DevGram dr = h->dev;
            dr.wpg = wpg;
            dr.npw = vnpw;
            dr.slot_tiles = h->d64_slot_tiles;
            dr.wg_begin = wgb;
            dr.tilecol = h->d64_tilecol;
            double *Gc = G + (size_t)g0 * Pa * Pa;
            hipLaunchKernelGGL(fbr_gram_reduce_kernel, dim3(FBR_WPB * vnpw, ng), dim3(256), 0, m->stream, dr, m->partial.as<double>(), Gc);
            if (g.NF > 0) {
                dr.slot_tiles = h->d64_slot_tiles + (size_t)FBR_WPB * vnpw * 2;
                hipLaunchKernelGGL(fbr_gram_reduce_kernel, dim3(FBR_WPB * vnpw, ng), dim3(256), 0, m->stream, dr, m->partial.as<double>(), Gc);
            }
            HIPCHK(hipGetLastError());
        }
    }
    return FBR_OK;
}

// async_ticket != nullptr: the pass is enqueued and NOT waited for (fbr_gram_submit): device-resident inputs and output only.
static int gram_impl_inner(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out,
                           int32_t out_mem, int32_t accumulate, int32_t ngroups, int64_t *async_ticket)
{
    const bool async = async_ticket != nullptr;
    // a submission whose predecessor is still in flight lets its producer start beside the predecessor's last Gram launches
    bool overlap_prev = false;
    // Pinned host inputs are staged chunk by chunk on the producer stream, overlapped with the Gram kernel of the previous chunk
    // (the PCIe-inclusive rate of the pass); pageable ones up front (an asynchronous copy from pageable memory blocks
    // the host thread and was measured slower when interleaved with the launches).
    const bool h2d_chunked = st && st->mem == FBR_HOST && m && m->opt.h2d_chunked && is_pinned_host(st->q) && is_pinned_host(st->dq) &&
                             is_pinned_host(st->ddq) && is_pinned_host(st->base_vel) && is_pinned_host(st->base_acc) &&
                             is_pinned_host(st->base_rpy) && is_pinned_host(st->sign) && is_pinned_host(rhs) && is_pinned_host(w);
    if (async && (!st || out_mem != FBR_DEVICE || (st->mem != FBR_DEVICE && !h2d_chunked))) {
        set_err("fbr_gram_submit takes a device-resident output and device-resident or PINNED host states / rhs / weights");
        return FBR_E_INVALID;
    }
    DevStates d;
    if (m) m->submitting = async;  // (a blocking call first waits for every submission in flight: stage_states)
    int rc = stage_states(m, st, &d, true, h2d_chunked);
    if (m) m->submitting = false;
    if (rc) return rc;
    if (async) {
        // at most two submissions in flight (two tile-image buffers, two completion events): the one before the last must be done
        if ((rc = wait_ticket(m, m->next_ticket - 2))) return rc;
        overlap_prev = m->waited_ticket < m->next_ticket - 1;
    }
    if (!G_out || k < 0 || k > FBR_MAX_RHS || (k > 0 && !rhs)) {
        set_err("bad rhs / G_out arguments");
        return FBR_E_INVALID;
    }
    if (ngroups < 1 || d.S % ngroups != 0) {
        set_err("the number of samples must be a multiple of the number of groups");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    if ((hm.rows + 3) / 4 * 4 > 60) {  // beyond the tile program's 15 k-steps: the Gram from the TSQR factor
        if (async || ngroups != 1) {
            set_err("robots with more than 60 regressor rows per sample (54 DOF on a floating base) are served by the blocking, ungrouped "
                    "fbr_gram_accumulate only (Gram from the TSQR factor): fbr_gram_submit / fbr_gram_grouped are limited to 60 rows");
            return FBR_E_UNSUPPORTED;
        }
        return gram_via_tsqr(m, st, rhs, k, w, G_out, out_mem, accumulate);
    }
    GramHolder *h = nullptr;
    // few rhs columns: their products come from the pack kernel instead of a dense tile (one Gram per call only: a pack workgroup's
    // samples straddle the groups of a grouped launch)
    const bool moments = ngroups == 1 && fbr_gram_rhs_moments(hm, k, m->opt.gram_rhs_tile != 0) && !m->opt.gram_timing;
    if ((rc = get_gram(m, k, &h, moments))) return rc;
    const int Pa = h->prog.Pa;
    const size_t gcount = (size_t)Pa * Pa * ngroups;
    const long S = d.S;
    const double *drhs = nullptr, *dw = nullptr;
    if (h2d_chunked) {
        drhs = rhs;  // host pointers: staged per chunk in produce()
        dw = w;
    } else {
        if ((rc = stage_one(m, m->st_aux, rhs, (size_t)S * hm.rows * k, st->mem, &drhs))) return rc;
        if ((rc = stage_one(m, m->st_aux2, w, (size_t)S * hm.rows, st->mem, &dw))) return rc;
    }
    // row masks that switch every joint row off (base-wrench-only identification, identifier.py:629-636): only the base k-steps run
    bool base_only = false;
    if (w && S > 0 && hm.fb > 0 && hm.rows > hm.fb) {
        // Host weights are looked at on the host, so that pinned and pageable inputs take the same path: ordinary WLS weights show a
        // non-zero joint-row weight in the very first sample and cost nothing; only a vector that starts like a base-wrench mask is
        // scanned to the end.  Device weights: one small scan kernel + a 4-byte-per-row copy.
        if (st->mem == FBR_HOST) {
            base_only = true;
            for (long s = 0; s < S && base_only; s++)
                for (int r = hm.fb; r < hm.rows; r++)
                    if (w[s * hm.rows + r] != 0.0) {
                        base_only = false;
                        break;
                    }
        } else {
            std::vector<char> act;
            if ((rc = active_rows(m, dw, S, &act))) return rc;
            base_only = true;
            for (int r = hm.fb; r < hm.rows; r++) base_only = base_only && !act[r];
        }
    }
    double *G = G_out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(gcount * sizeof(double)))) return rc;
        G = m->g_tmp.as<double>();
        if (accumulate) HIPCHK(hipMemcpyAsync(G, G_out, gcount * sizeof(double), hipMemcpyHostToDevice, m->stream));
    }
    if (!accumulate) HIPCHK(hipMemsetAsync(G, 0, gcount * sizeof(double), m->stream));
    // the pass over sample-contiguous images (fbr_gram64.h) where the call allows it
    bool lane_pass = false;
    if (S > 0 && m->opt.gram_lane != 0 && ngroups == 1 && !m->opt.gram_timing && !m->opt.gram_serial &&
        k <= 1 && (k == 0 || moments) && d.q) {
        if ((rc = get_gram64(m, h))) return rc;
        lane_pass = h->g64_state == 1;
    }
    bool lane_grouped = false;
    if (!lane_pass && S > 0 && m->opt.gram_lane != 0 && ngroups > 1 && k == 0 && !h2d_chunked && !base_only && !m->opt.gram_timing &&
        !m->opt.gram_serial && d.q) {
        if ((rc = get_gram64(m, h))) return rc;
        lane_grouped = h->g64_state == 1 && (S / ngroups + 63) / 64 <= gram64_chunk_blocks(h->g64);
    }
    if (lane_grouped) {
        if ((rc = gram64_grouped_pass(m, h, d, dw, G, ngroups))) return rc;
    } else if (lane_pass) {
        if ((rc = gram64_pass(m, h, d, drhs, dw, k, G, base_only && h->g64.base_stages > 0, h2d_chunked))) return rc;
        if (!async && h2d_chunked && m->copy) HIPCHK(hipStreamSynchronize(m->copy));
    } else if (S > 0) {
        const int T = h->prog.T;
        const bool two_per_cu = h->prog.cfg == FBR_CFG_TWO_PER_CU;
        const int blocks_per_cu = (two_per_cu && h->lds_bytes <= 79 * 1024) ? 2 : 1;
        const int FBR_NPW = h->prog.cfg.npw();
        const bool timing = m->opt.gram_timing != 0;
        typedef void (*gram_fn)(DevGram, long, int, const double *, double *, unsigned long long *, int);
        const gram_fn gram_kernel = two_per_cu ? (timing ? fbr_gram_kernel<true, 5, 2> : fbr_gram_kernel<false, 5, 2>)
                                               : (timing ? fbr_gram_kernel<true, FBR_ONE_SEGW, FBR_ONE_NSEG> : fbr_gram_kernel<false, FBR_ONE_SEGW, FBR_ONE_NSEG>);
        HIPCHK(hipFuncSetAttribute((const void *)gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        HIPCHK(hipFuncSetAttribute((const void *)fbr_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->pack_lds_bytes));
        const size_t img_bytes = (size_t)h->prog.image_doubles * sizeof(double);
        long ch = chunk_size(m, S);
        ch = std::max(1L, std::min(ch, (long)((size_t)4 * 1024 * 1024 * 1024 / img_bytes)));
        if (ngroups == 1 && m->opt.chunk_samples < 1) {
            // a short batch (e.g. one rank's shard of a multi-GPU run) is still cut into several chunks, so that only a small first
            // chunk's producer work runs before the first Gram launch instead of half the batch's
            const long min_chunks = std::max(1L, (long)m->opt.min_chunks);  // measured: 125 k samples 2 / 4 / 8 / 16 chunks = 11.45 / 11.69 / 11.19 / 9.35 M samples/s
            ch = std::max(std::min(ch, 8192L), std::min(ch, (S + min_chunks - 1) / min_chunks));
        }
        // work items: several whole groups per launch, or (groups larger than a chunk) pieces of one group
        struct Item { long s0, cs; int g0, ng; };
        std::vector<Item> items;
        const long Sg = S / ngroups;
        if (Sg <= ch) {
            const int gpc = (int)std::min<long>(ngroups, std::max(1L, ch / Sg));
            for (int g0 = 0; g0 < ngroups; g0 += gpc) {
                const int ng = std::min(gpc, ngroups - g0);
                items.push_back({g0 * Sg, ng * Sg, g0, ng});
            }
            ch = gpc * Sg;
        } else {
            // the producer work of the very first chunk is the only one that nothing hides: it is made smaller (a quarter of a chunk:
            // measured on a 125 k-sample shard, tools/chunk_probe.py)
            for (int g = 0; g < ngroups; g++)
                for (long c0 = 0; c0 < Sg; c0 += ch) items.push_back({g * Sg + c0, std::min(ch, Sg - c0), g, 1});
        }
        const long nchunks = (long)items.size();
        bool fresh_images = false;  // a tile-image buffer was (re)allocated and zeroed on the main stream in this call
        // workgroups per group of a launch: every resident workgroup slot is used (see the launch below)
        auto wpg_of = [&](long cs, int ng) {
            const long spg_max = std::max(1L, cs / ng);
            const int rounds = ng > 1 ? (T > 1 ? 4 : 2) : 1;
            int wpg = std::max(T, (rounds * m->num_cus * blocks_per_cu) / ng);
            if ((long)wpg > (long)T * spg_max) wpg = (int)((long)T * spg_max);
            return std::min(wpg, 0xffff);
        };
        // One reduction per call: when every chunk of a single-group call has the same launch shape, a workgroup carries its partial
        // sums from chunk to chunk (the accumulators start from the partial-sum buffer) and fbr_gram_reduce_kernel runs once, after
        // the last chunk -- 15 of the 16 reductions of a 1 M-sample WALK-MAN pass (72 us each, between two Gram launches) go away.
        bool carry_ok = ngroups == 1 && nchunks > 1 && !timing;
        for (long ci = 1; ci < nchunks && carry_ok; ci++) carry_ok = wpg_of(items[ci].cs, 1) == wpg_of(items[0].cs, 1);
        for (int b = 0; b < (nchunks > 1 ? 2 : 1); b++)
            if ((size_t)ch * img_bytes > h->pimg[b].bytes) {
                if ((rc = h->pimg[b].ensure((size_t)ch * img_bytes))) return rc;
                HIPCHK(hipMemsetAsync(h->pimg[b].p, 0, h->pimg[b].bytes, m->stream));  // structural zeros are never rewritten
                fresh_images = true;
            }
        // producer (kinematics + tile-image packing of chunk i+1) runs on a second stream and shares the CUs with the
        // MFMA-bound Gram kernel of chunk i; the images are double buffered
        HIPCHK(hipEventRecord(m->ev_fork, m->stream));
        // FBR_GRAM_SERIAL (diagnostic): producer on the main stream, i.e. no overlap with the Gram kernel
        hipStream_t side = m->opt.gram_serial ? m->stream : m->side;
        // The producer normally starts after everything enqueued on the main stream so far.  A submission that follows another one
        // (fbr_gram_submit) skips that: its inputs are device resident, and what its first producer launches must wait for is only
        // the tile-image buffer they write (ev_gram below) -- kinematics and packing of its first chunk then run beside the last
        // Gram launches of the submission before, the one piece of producer work nothing else hides.
        const bool cross = overlap_prev && !fresh_images && side != m->stream;
        if (!cross) HIPCHK(hipStreamWaitEvent(side, m->ev_fork, 0));
        // per-sample doubles of one staged chunk (pinned host inputs): q dq ddq [bv ba rpy] [sign] [rhs] [w]
        const size_t stage_per = (size_t)3 * hm.n + (hm.floating ? 15 : 0) + (d.sign ? hm.n : 0) + (size_t)hm.rows * k + (dw ? hm.rows : 0);
        if (h2d_chunked)
            for (int b = 0; b < (nchunks > 1 ? 2 : 1); b++)
                if ((rc = m->st_chunk[b].ensure(std::max<size_t>(1, (size_t)ch * stage_per) * sizeof(double)))) return rc;
        // pack workgroups per CU of a launch's grid (each walks its share of the chunk's samples).  More than are ever resident (7 per CU
        // alone, 2 beside the Gram kernel): with 8 the workgroups of the last, partial round ran on a half-empty chip at the end of every
        // launch (measured per 1 M-sample step, two runs each: 8 -> 24.1, 16 -> 23.5 ... 24.0, 24 -> 23.1 ... 23.4, 32 / 48 -> 23.4)
        const int pack_wgs_per_cu = 24;
        const int pack_blocks_max = m->num_cus * pack_wgs_per_cu;
        auto produce = [&](long ci) -> int {
            const long s0 = items[ci].s0, cs = items[ci].cs;
            const int b = (int)(ci & 1);
            // Gram of chunk ci-2 (or, across submissions, the last Gram launch that read this buffer) is done with it
            if (ci >= 2 || (cross && m->ev_gram_rec[b])) HIPCHK(hipStreamWaitEvent(side, m->ev_gram[b], 0));
            DevStates dc = d;     // what the kernels of this chunk read, and the sample offset into it
            long o = s0;
            const double *crhs = drhs, *cw = dw;
            if (h2d_chunked) {
                // copies run on their own stream so that the copy of this chunk overlaps the kinematics / packing of the one before:
                // they wait for the pack kernel of chunk ci-2 (the last reader of this staging buffer), the producer waits for them
                // (created on first use: HIP maps streams to hardware queues in creation order, and the producer stream's queue must
                // stay what it is for device-resident inputs)
                if (!m->copy) {
                    // a priority level of its own (main stream: normal, producer: lowest, copies: highest), so that the copy stream
                    // never lands on the hardware queue of the Gram stream whatever streams the process created before (seen in
                    // bench.py after the TSQR leg had created two more streams: copies and Gram launches serialised, 78.6 instead
                    // of 74.1 ms per step)
                    int least = 0, greatest = 0;
                    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
                    HIPCHK(hipStreamCreateWithPriority(&m->copy, hipStreamNonBlocking, greatest));
                }
                hipStream_t cps = m->copy ? m->copy : side;
                if (cps != side) {
                    // the staging buffer's last reader is the pack kernel of the chunk two before (or, across submissions, the last
                    // pack launch that used this buffer)
                    if (ci >= 2 || (cross && m->ev_pack_rec[b]))
                        HIPCHK(hipStreamWaitEvent(cps, m->ev_pack[b], 0));
                    else
                        HIPCHK(hipStreamWaitEvent(cps, m->ev_fork, 0));
                }
                ProfScope ps(m, FBR_PROF_H2D, cps);
                double *p = m->st_chunk[b].as<double>();
                auto put = [&](const double *src, size_t per, const double **dst) -> int {
                    *dst = nullptr;
                    if (!src || per == 0) return FBR_OK;
                    HIPCHK(hipMemcpyAsync(p, src + (size_t)s0 * per, (size_t)cs * per * sizeof(double), hipMemcpyHostToDevice, cps));
                    *dst = p;
                    p += (size_t)cs * per;
                    return FBR_OK;
                };
                int r3;
                if ((r3 = put(d.q, hm.n, &dc.q)) || (r3 = put(d.dq, hm.n, &dc.dq)) || (r3 = put(d.ddq, hm.n, &dc.ddq)) ||
                    (r3 = put(d.bv, 6, &dc.bv)) || (r3 = put(d.ba, 6, &dc.ba)) || (r3 = put(d.rpy, 3, &dc.rpy)) ||
                    (r3 = put(d.sign, hm.n, &dc.sign)) || (r3 = put(drhs, (size_t)hm.rows * k, &crhs)) || (r3 = put(dw, hm.rows, &cw)))
                    return r3;
                o = 0;
                if (cps != side) {
                    HIPCHK(hipEventRecord(m->ev_h2d[b], cps));
                    HIPCHK(hipStreamWaitEvent(side, m->ev_h2d[b], 0));
                }
            }
            int rc2 = run_kin(m, dc, o, cs, side, &m->rec2);
            if (rc2) return rc2;
            {
                ProfScope ps(m, FBR_PROF_PACK, side);
                const int blocks = (int)std::min<long>(cs, (long)pack_blocks_max);
                hipLaunchKernelGGL(fbr_pack_kernel, dim3(blocks), dim3(256), h->pack_lds_bytes, side, h->dev, m->dm, cs, cs / items[ci].ng,
                                   m->rec2.as<double>(), dc.dq + o * hm.n, dc.sign ? dc.sign + o * hm.n : nullptr,
                                   crhs ? crhs + (size_t)o * hm.rows * k : nullptr, cw ? cw + (size_t)o * hm.rows : nullptr,
                                   h->pimg[b].as<double>(), base_only ? 1 : 0, moments ? h->mom[(int)(m->next_ticket & 1)].as<double>() : nullptr);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(m->ev_pack[b], side));
            m->ev_pack_rec[b] = true;
            return FBR_OK;
        };
        const int mpar = (int)(m->next_ticket & 1);
        if (moments) {
            const size_t mbytes = (size_t)pack_blocks_max * 256 * 4 * sizeof(double);
            if (h->mom[mpar].bytes < mbytes) h->mom_clean[mpar] = false;
            if ((rc = h->mom[mpar].ensure(mbytes))) return rc;
            if (!h->mom_clean[mpar]) HIPCHK(hipMemsetAsync(h->mom[mpar].p, 0, mbytes, side));
            h->mom_clean[mpar] = false;  // (until this call's reduction has been enqueued)
        }
        if ((rc = produce(0))) return rc;
        for (long ci = 0; ci < nchunks; ci++) {
            const long cs = items[ci].cs;
            const int ng = items[ci].ng;
            const int b = (int)(ci & 1);
            if (ci + 1 < nchunks && (rc = produce(ci + 1))) return rc;
            HIPCHK(hipStreamWaitEvent(m->stream, m->ev_pack[b], 0));
            // every resident workgroup slot is used: the slots of a sample group are dealt to the parts by cost (fbr_gram_deal),
            // a part's workgroups split the group's samples evenly.  Tiny batches: no more workgroups than samples per part.
            // Grouped launches (many short candidates) are oversubscribed: with one round of resident workgroups a group gets too few
            // of them to follow the parts' costs (WALK-MAN, 64 groups x 2000 samples: 5 per group, 15.4 ms; 4 rounds: 11.8 ms; KUKA
            // 0.92 -> 0.90 ms with 2 rounds) and the hardware dispatcher evens out the rest.  Bulk launches lose 17 % when
            // oversubscribed (late workgroups run beside the producer kernels of the next chunk): one round, dealt by cost.
            const int wpg = wpg_of(cs, ng);
            GramHolder::Deal deal;
            if ((rc = get_deal(h, wpg, &deal, base_only))) return rc;
            DevGram dg = h->dev;
            dg.wpg = wpg;
            dg.ks_limit = base_only ? hm.fbp / 4 : (1 << 20);
            if (base_only && hm.fbp == 8) {  // (8 base positions x 16 columns = one full DMA piece per tile)
                dg.pieces = dg.pieces_b;
                dg.piece_begin = dg.piece_begin_b;
            }
            dg.wg_tab = deal.tab;
            dg.wg_begin = deal.begin;
            const int NW = wpg * ng;  // workgroups of this launch
            const size_t pcount = (size_t)NW * FBR_WPB * FBR_NPW * 256;
            if ((rc = m->partial.ensure(pcount * sizeof(double)))) return rc;
            unsigned long long *dbg = nullptr;
            if (timing) {
                if ((rc = m->st_x.ensure((size_t)NW * FBR_WPB * 8 * sizeof(unsigned long long)))) return rc;
                dbg = m->st_x.as<unsigned long long>();
            }
            {
                ProfScope ps(m, FBR_PROF_GRAM);
                hipLaunchKernelGGL(gram_kernel, dim3(NW), dim3(FBR_WPB * 64), h->lds_bytes, m->stream, dg, cs, ng,
                                   h->pimg[b].as<double>(), m->partial.as<double>(), dbg, (carry_ok && ci > 0) ? 1 : 0);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(m->ev_gram[b], m->stream));
            m->ev_gram_rec[b] = true;
            if (timing) {
                std::vector<unsigned long long> hb((size_t)NW * FBR_WPB * 8);
                HIPCHK(hipMemcpyAsync(hb.data(), dbg, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
                HIPCHK(hipStreamSynchronize(m->stream));
                static const char *names[3] = {"wait_dma+barrier", "dma_issue", "mfma"};
                std::vector<double> sum((size_t)T * 3, 0.0), ns(T, 0.0), nw(T, 0.0), wv((size_t)T * FBR_WPB, 0.0);
                for (size_t e = 0; e + 8 <= hb.size(); e += 8) {
                    const int part = (int)hb[e + 6];
                    if (part < 0 || part >= T) continue;
                    for (int i = 0; i < 3; i++) sum[(size_t)part * 3 + i] += (double)hb[e + i];
                    ns[part] += (double)hb[e + 7];
                    nw[part] += 1.0;
                    wv[(size_t)part * FBR_WPB + (e / 8) % FBR_WPB] += (double)hb[e + 2];
                }
                for (int part = 0; part < T; part++) {
                    fprintf(stderr, "[fbr gram timing] part %d (cycles per sample per wave):", part);
                    for (int i = 0; i < 3; i++) fprintf(stderr, " %s=%.0f", names[i], sum[(size_t)part * 3 + i] / std::max(ns[part], 1.0));
                    fprintf(stderr, " | workgroups=%.0f cycles per workgroup=%.0f | mfma phase per wave:", nw[part] / FBR_WPB,
                            (sum[(size_t)part * 3] + sum[(size_t)part * 3 + 1] + sum[(size_t)part * 3 + 2]) / std::max(nw[part], 1.0));
                    for (int w = 0; w < FBR_WPB; w++) fprintf(stderr, " %.0f", wv[(size_t)part * FBR_WPB + w] * FBR_WPB / std::max(ns[part], 1.0));
                    fprintf(stderr, "\n");
                }
            }
            if (!carry_ok || ci + 1 == nchunks) {
                ProfScope ps(m, FBR_PROF_REDUCE);
                hipLaunchKernelGGL(fbr_gram_reduce_kernel, dim3(T * FBR_WPB * FBR_NPW, ng), dim3(256), 0, m->stream, dg,
                                   m->partial.as<double>(), G + (size_t)items[ci].g0 * Pa * Pa);
            }
            HIPCHK(hipGetLastError());
        }
        if (moments) {  // (the main stream has waited for the last pack launch before its last Gram launch)
            ProfScope ps(m, FBR_PROF_REDUCE);
            hipLaunchKernelGGL(fbr_gram_mom_reduce_kernel, dim3(256), dim3(256), 0, m->stream, hm.cols, k, pack_blocks_max, h->itemcol,
                               h->mom[mpar].as<double>(), G);
            HIPCHK(hipGetLastError());
            h->mom_clean[mpar] = true;
        }
        if (!async) {
            HIPCHK(hipStreamSynchronize(side));
            if (h2d_chunked && m->copy) HIPCHK(hipStreamSynchronize(m->copy));
        }
    }
    if (async) {
        const int64_t t = m->next_ticket++;
        m->ticket_kind[t & 1] = 0;
        m->last_submit_kind = 0;
        HIPCHK(hipEventRecord(m->ev_done[t & 1], m->stream));
        *async_ticket = t;
        return FBR_OK;
    }
    return finish_output(m, G, G_out, gcount, out_mem);
}

static int gram_impl(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out, int32_t out_mem,
                     int32_t accumulate, int32_t ngroups, int64_t *async_ticket);

// The Gram through the link-merged model (build_reduction): G_red on the moving bodies' columns, then G (+)= E^T G_red E.
static int gram_via_red(fbr_model *m, int which, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out, int32_t out_mem,
                        int32_t accumulate, int32_t ngroups, int64_t *async_ticket)
{
    fbr_model *r = m->rdm[which].get();
    const bool async = async_ticket != nullptr;
    int rc;
    if ((rc = enter(m))) return rc;
    if ((rc = wait_ticket(m, async ? m->next_ticket - 2 : m->next_ticket - 1))) return rc;
    if (async && out_mem != FBR_DEVICE) {
        set_err("fbr_gram_submit takes a device-resident output and device-resident or PINNED host states / rhs / weights");
        return FBR_E_INVALID;
    }
    r->stream = m->stream;
    r->prof = m->prof;
    const int par = (int)(m->next_ticket & 1), Pa = m->hm.cols + k, Pra = r->hm.cols + k;
    if (ngroups < 1 || (async && ngroups != 1)) {
        set_err("bad number of groups");
        return FBR_E_INVALID;
    }
    const size_t cnt = (size_t)Pa * Pa * ngroups;  // (grouped: one Gram per group of samples, each expanded on its own)
    if ((rc = m->red_out[par].ensure((size_t)Pra * Pra * ngroups * sizeof(double)))) return rc;
    double *Gred = m->red_out[par].as<double>();
    int64_t tr = -1;
    if ((rc = gram_impl(r, st, rhs, k, w, Gred, FBR_DEVICE, 0, ngroups, async ? &tr : nullptr))) return rc;
    double *G = G_out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure(cnt * sizeof(double)))) return rc;
        G = m->g_tmp.as<double>();
        if (accumulate) HIPCHK(hipMemcpyAsync(G, G_out, cnt * sizeof(double), hipMemcpyHostToDevice, m->stream));
    }
    // all groups in two launches (one W = G_red E per group): 128 dependent launches for 64 candidate trajectories were up to two thirds of the
    // grouped call (bench.py other_configs: 12.7 ms on a run whose stream-to-queue mapping made a launch 68 us, 5.2 ms on the next)
    if ((rc = m->red_w.ensure((size_t)Pra * Pa * ngroups * sizeof(double)))) return rc;
    for (int g0 = 0; g0 < ngroups; g0 += 32768) {
        const int ng = std::min(32768, ngroups - g0);
        hipLaunchKernelGGL(fbr_expand_rows_kernel, dim3(ng > 1 ? 64 : 512, ng), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which],
                           m->E_val[which], Gred + (size_t)g0 * Pra * Pra, m->red_w.as<double>() + (size_t)g0 * Pra * Pa, Pa, (const int *)nullptr, Pa);
        hipLaunchKernelGGL(fbr_expand_gram_kernel, dim3(ng > 1 ? 128 : 1024, ng), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which],
                           m->E_val[which], m->red_w.as<double>() + (size_t)g0 * Pra * Pa, G + (size_t)g0 * Pa * Pa, accumulate ? 1 : 0);
    }
    HIPCHK(hipGetLastError());
    if (async) {
        const int64_t t = m->next_ticket++;
        m->ticket_kind[t & 1] = 0;
        m->ticket_via_red[t & 1] = 1 + which;
        m->red_ticket[t & 1] = tr;
        m->last_submit_kind = 0;
        HIPCHK(hipEventRecord(m->ev_done[t & 1], m->stream));
        *async_ticket = t;
        return FBR_OK;
    }
    return finish_output(m, G, G_out, cnt, out_mem);
}

static int gram_impl(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out,
                     int32_t out_mem, int32_t accumulate, int32_t ngroups, int64_t *async_ticket = nullptr)
{
    // (many small groups: two launches per group for the expansion -- worth it while a group's pass is longer than that)
    const bool grouped_ok = st && (ngroups == 1 || (ngroups >= 1 && (double)(st->num_samples / ngroups) >= m->opt.reduce_grouped_min_samples));
    const int which =
        (m && st && ngroups >= 1 && G_out && k >= 0 && k <= FBR_MAX_RHS && m->pid == getpid() && grouped_ok) ? pick_gram_reduction(m, (long)st->num_samples) : -1;
    if (which >= 0) {
        int rc = gram_via_red(m, which, st, rhs, k, w, G_out, out_mem, accumulate, ngroups, async_ticket);
        if (rc && m->stream) {
            const std::string msg = g_fbr_err;
            drain_after_failed_submit(m);
            set_err(msg);
        }
        return rc;
    }
    int rc = gram_impl_inner(m, st, rhs, k, w, G_out, out_mem, accumulate, ngroups, async_ticket);
    // a failed submission issues no ticket, and a blocking call that fails half way may have launched on the producer / copy streams:
    // nothing of either may stay in flight when the error is returned
    if (rc && m && m->pid == getpid() && m->stream) {
        const std::string msg = g_fbr_err;
        drain_after_failed_submit(m);
        set_err(msg);
    }
    return rc;
}
extern "C" int fbr_gram_accumulate(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w,
                                   double *G_out, int32_t out_mem, int32_t accumulate)
{
    return gram_impl(m, st, rhs, k, w, G_out, out_mem, accumulate, 1);
}

extern "C" int fbr_gram_submit(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out,
                               int32_t accumulate, int64_t *ticket)
{
    if (!ticket) {
        set_err("ticket is NULL");
        return FBR_E_INVALID;
    }
    return gram_impl(m, st, rhs, k, w, G_out, FBR_DEVICE, accumulate, 1, ticket);
}
extern "C" int fbr_gram_grouped(fbr_model *m, const fbr_states *st, int32_t ngroups, const double *rhs, int32_t k, const double *w,
                                double *G_out, int32_t out_mem)
{
    return gram_impl(m, st, rhs, k, w, G_out, out_mem, 0, ngroups);
}
