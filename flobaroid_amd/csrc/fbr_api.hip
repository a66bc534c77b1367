// fbr_api.hip -- C-ABI of libfbr (see include/fbr.h): model handle, options, state staging, per-sample entry points.
// gfx950 only; there is deliberately no CPU path in this library.  The fused Gram lives in fbr_gram_api.hip, the TSQR in
// fbr_tsqr_api.hip, the signal conditioning in fbr_signal_api.hip.
#define FBR_KERNELS_CORE
#include "fbr_internal.h"
#include "fbr_reduce.h"

thread_local std::string g_fbr_err;

// ------------------------------------------------------------------------------------------------
// 101 (round 6): fbr_topology.joint_type, the num_samples argument of fbr_gram_program_info / fbr_model_link_merge_info (both added in
// round 5 under 100), option "fused_id"; 102: fbr_gram_lane_info, options "gram_lane" / "gram_force_tiles" / "tsqr_force_group".
// flobaroid_amd/_lib.py refuses a library of another version than the header it was written for.
extern "C" int fbr_version(void) { return FBR_VERSION; }

// (dispatched once on each of a model's two Gram streams at creation: see create_model)
__global__ void fbr_noop_kernel() {}

// The HIP runtime does not survive fork(): a child that inherits an initialised runtime hangs or fails in its first call.  The
// reference's multi-process users build one Model per worker AFTER the fork (analyticalGradient.py:188-210); this records the process
// that first touched HIP through the library so that the other order is reported instead of deadlocking.
static pid_t g_hip_pid = 0;
static bool forked_after_hip_init()
{
    const pid_t me = getpid();
    if (g_hip_pid == 0) g_hip_pid = me;
    if (g_hip_pid == me) return false;
    set_err("HIP was initialised in the parent process before fork(): create the model in the worker (after the fork) or start "
            "workers with the 'spawn' method");
    return true;
}
int enter(fbr_model *m)
{
    if (m->pid != getpid()) {
        set_err("this fbr_model was created in another process (before fork()): create one per process");
        return FBR_E_FORK;
    }
    HIPCHK(hipSetDevice(m->device));
    return FBR_OK;
}

// entry of a blocking call that does not go through stage_states: every asynchronous submission before it has completed
int enter_blocking(fbr_model *m)
{
    if (int rc = enter(m)) return rc;
    return wait_ticket(m, m->next_ticket - 1);
}

extern "C" int fbr_device_count(void)
{
    if (forked_after_hip_init()) return 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *fbr_last_error(void) { return g_fbr_err.c_str(); }

static int build_reduction(fbr_model *m, const fbr_topology *t, int which);
static int create_model(const fbr_topology *t, int device, fbr_model **out, bool allow_merge, const unsigned short *linkmask = nullptr)  // (t->joint_type: NULL or [L])
{
    if (!t || !out) {
        set_err("null argument");
        return FBR_E_INVALID;
    }
    *out = nullptr;
    if (forked_after_hip_init()) return FBR_E_FORK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err("no HIP device available (libfbr has no CPU fallback)");
        return FBR_E_NODEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_err("device index out of range");
        return FBR_E_INVALID;
    }
    std::unique_ptr<fbr_model> m(new fbr_model());
    try {
        m->hm.build(t->num_links, t->num_dofs, t->parent, t->dof_index, t->rest_R, t->rest_p, t->axis, t->floating_base,
                    t->gravity, t->friction, t->friction_symmetric, t->gravity_only, t->stribeck_velocity, linkmask, t->joint_type);
    } catch (const std::exception &e) {
        set_err(std::string("invalid topology: ") + e.what());
        return FBR_E_INVALID;
    }
    m->device = device;
    m->pid = getpid();
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    m->num_cus = prop.multiProcessorCount;
    HIPCHK(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
    m->stream = m->own_stream;
    {
        // The producer stream must not share a hardware queue with the stream the Gram kernel runs on, or the two serialise
        // (seen under torch.distributed, where RCCL's streams shift HIP's round-robin stream -> queue assignment).  A stream of
        // another priority level gets a queue of its own; the producer is the background work, so it takes the lowest.
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, least));
        // HIP binds a stream to a hardware queue at its first dispatch: both streams of the Gram pass dispatch once HERE, so that their queues do not
        // depend on which other streams (TSQR trees, copies, the caller's) were used first -- seen in bench.py, round 6: the grouped Gram after a
        // masked TSQR call 13.3 instead of 5.2 ms when the producer stream's first kernel came after the TSQR's side streams'
        hipLaunchKernelGGL(fbr_noop_kernel, dim3(1), dim3(64), 0, m->own_stream);
        hipLaunchKernelGGL(fbr_noop_kernel, dim3(1), dim3(64), 0, m->side);
        HIPCHK(hipGetLastError());
    }
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_h2d[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_pack[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_gram[i], hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_tsqr_l0, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_tsqr_pro, hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&m->tsqr_err, sizeof(unsigned)));
    HIPCHK(hipMemset(m->tsqr_err, 0, sizeof(unsigned)));
    HIPCHK(hipHostMalloc((void **)&m->tsqr_err_host, 2 * sizeof(unsigned), hipHostMallocDefault));
    m->tsqr_err_host[0] = m->tsqr_err_host[1] = 0;

    const FbrHostModel &hm = m->hm;
    DevModel &dm = m->dm;
    memset(&dm, 0, sizeof(dm));
    dm.L = hm.L; dm.n = hm.n; dm.fb = hm.fb; dm.rows = hm.rows; dm.cols = hm.cols; dm.cpl = hm.cpl;
    dm.floating = hm.floating; dm.rec = hm.rec_size(); dm.maxd = std::max(hm.maxdepth, 1);
    dm.nw = std::max(1, (hm.n + 31) / 32);
    dm.fric = hm.fric; dm.grav_only = hm.grav_only; dm.fstart = hm.friction_start();
    for (int i = 0; i < 3; i++) dm.g[i] = hm.gravity[i];
    dm.stribeck = hm.stribeck;
    std::vector<int> pathlen(hm.L), pathtab((size_t)hm.L * dm.maxd, 0), pathpos((size_t)hm.L * dm.maxd, 0);
    std::vector<unsigned> anc((size_t)hm.L * dm.nw, 0u);
    std::vector<std::vector<int>> sub(std::max(hm.n, 1));
    std::vector<int> dof_link(std::max(hm.n, 1), 0);
    for (int l = 0; l < hm.L; l++) {
        pathlen[l] = (int)hm.path[l].size();
        for (size_t j = 0; j < hm.path[l].size(); j++) {
            int d = hm.path[l][j];
            pathtab[(size_t)l * dm.maxd + j] = d;
            pathpos[(size_t)l * dm.maxd + j] = hm.ppos[l][j];
            anc[(size_t)l * dm.nw + (d >> 5)] |= 1u << (d & 31);
            sub[d].push_back(l);
        }
        if (hm.dof[l] >= 0) dof_link[hm.dof[l]] = l;
    }
    std::vector<int> sub_begin(hm.n + 1, 0), sub_links;
    for (int d = 0; d < hm.n; d++) {
        sub_begin[d] = (int)sub_links.size();
        sub_links.insert(sub_links.end(), sub[d].begin(), sub[d].end());
    }
    sub_begin[hm.n] = (int)sub_links.size();
    std::vector<int4> cd(hm.cols);
    for (int c = 0; c < hm.cols; c++) cd[c] = make_int4(hm.coldesc[c].kind, hm.coldesc[c].link, hm.coldesc[c].pidx, hm.coldesc[c].joint);
    int rc = 0;
    m->tables.reserve(32);
    if ((rc = upload(m->tables, hm.order, &dm.order))) return rc;
    if ((rc = upload(m->tables, hm.parent, &dm.parent))) return rc;
    if ((rc = upload(m->tables, hm.dof, &dm.dof))) return rc;
    if ((rc = upload(m->tables, hm.jtype, &dm.jtype))) return rc;
    if ((rc = upload(m->tables, hm.restR, &dm.restR))) return rc;
    if ((rc = upload(m->tables, hm.restp, &dm.restp))) return rc;
    if ((rc = upload(m->tables, hm.axis, &dm.axis))) return rc;
    if ((rc = upload(m->tables, pathlen, &dm.pathlen))) return rc;
    if ((rc = upload(m->tables, pathtab, &dm.pathtab))) return rc;
    if ((rc = upload(m->tables, pathpos, &dm.pathpos))) return rc;
    if ((rc = upload(m->tables, anc, &dm.ancmask))) return rc;
    if ((rc = upload(m->tables, cd, &dm.coldesc))) return rc;
    if ((rc = upload(m->tables, sub_begin, &dm.sub_begin))) return rc;
    if ((rc = upload(m->tables, sub_links, &dm.sub_links))) return rc;
    if ((rc = upload(m->tables, dof_link, &dm.dof_link))) return rc;
    if (hm.maxdepth <= FBR_KINID_MAXD) {
        try {
            fbr_kinid_build(hm, m->kinid);
        } catch (const std::exception &e) {
            set_err(std::string("internal: ") + e.what());
            return FBR_E_INVALID;
        }
        if ((rc = upload(m->tables, m->kinid.steps, &m->kinid_steps))) return rc;
        if ((rc = upload(m->tables, m->kinid.endflush, &m->kinid_endflush))) return rc;
    }
    if (allow_merge) {  // (always built; the options "link_merge" / "regroup" decide per call whether they are used)
        if ((rc = build_reduction(m.get(), t, 0))) return rc;
        if ((rc = build_reduction(m.get(), t, 1))) return rc;
    }
    *out = m.release();
    return FBR_OK;
}

extern "C" int fbr_model_create(const fbr_topology *t, int device, fbr_model **out) { return create_model(t, device, out, true); }

// ------------------------------------------------------------------------------------------------
// Column reductions (fbr_reduce.h has the mathematics: fixed links merged into the bodies they ride on, revolute links regrouped, and the
// constant matrix E with [Y | rhs] = [Y_red | rhs] E).  The reductions run on the REDUCED robot and are expanded at the end of the call:
// G = E^T G_red E,  R = qr(R_red E).  which = 0: merged (m->rdm[0]); which = 1: merged + regrouped (column masks, m->rdm[1]).
// ------------------------------------------------------------------------------------------------
static int build_reduction(fbr_model *m, const fbr_topology *t, int which)
{
    FbrReducedRobot rr;
    if (!fbr_reduce_robot(m->hm, which, rr)) return FBR_OK;
    fbr_topology tr = *t;
    tr.num_links = rr.Lr;
    tr.parent = rr.parent.data();
    tr.dof_index = rr.dof.data();
    tr.rest_R = rr.restR.data();
    tr.rest_p = rr.restp.data();
    tr.axis = rr.axis.data();
    tr.joint_type = rr.jtype.data();
    fbr_model *red = nullptr;
    if (int rc = create_model(&tr, m->device, &red, false, rr.masked ? rr.masks.data() : nullptr)) return rc;
    m->rdm[which].reset(red);
    red->is_reduction = true;
    std::vector<int> beg, row;
    std::vector<double> val;
    fbr_reduction_matrix(m->hm, rr, red->hm, beg, row, val);
    int rc;
    if ((rc = upload(m->tables, beg, &m->E_beg[which])) || (rc = upload(m->tables, row, &m->E_row[which])) ||
        (rc = upload(m->tables, val, &m->E_val[which])))
        return rc;
    m->hE_beg[which] = beg;
    m->hE_row[which] = row;
    m->hE_val[which] = val;
    return FBR_OK;
}
extern "C" void fbr_model_destroy(fbr_model *m)
{
    if (m && m->pid != getpid()) return;  // a handle inherited through fork(): its device resources belong to the parent, nothing to free here
    delete m;  // ~fbr_model releases the device memory, streams and events
}

// the reduced model a fused Gram pass runs on (-1: the model itself)
int pick_gram_reduction(const fbr_model *m, long S)
{
    const FbrOptions &o = m->opt;
    if (!o.link_merge) return -1;
    // S >= 0: a call over S samples.  The reduced pass costs a second model's launches and two expansion kernels (~0.15 ms): small
    // robots on short batches are faster over all their columns (KUKA, 80 -> 59 columns, 50 k samples: 0.73 against 0.82 ... 1.08 ms)
    if (S >= 0) {
        // (against the reduction that would be taken: the regrouped model within the fused kernel's 60 rows, else the merged one)
        const bool regrouped = m->rdm[1] && o.regroup && (m->hm.rows + 3) / 4 * 4 <= 60;
        const fbr_model *r = regrouped ? m->rdm[1].get() : m->rdm[0].get();
        if (r && (double)S * (m->hm.cols - r->hm.cols) * m->hm.cols < o.reduce_min_work) return -1;
    }
    // (robots beyond the fused kernel's 60 rows take their Gram from a TSQR factor, gram_via_tsqr: every path of the merged model)
    if (m->rdm[1] && o.regroup && (m->hm.rows + 3) / 4 * 4 <= 60) return 1;
    return m->rdm[0] ? 0 : -1;
}

extern "C" int fbr_model_link_merge_info(const fbr_model *m, int64_t num_samples, int32_t *moving_links, int32_t *reduced_cols)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    const int w = pick_gram_reduction(m, num_samples < 0 ? -1 : (long)num_samples);
    if (moving_links) *moving_links = w >= 0 ? m->rdm[w]->hm.L : m->hm.L;
    if (reduced_cols) *reduced_cols = w >= 0 ? m->rdm[w]->hm.cols : m->hm.cols;
    return FBR_OK;
}

static void clear_programs(fbr_model *m)
{
    m->gram.clear();  // (Gram programs and their device tables are rebuilt on the next call)
}

extern "C" int fbr_model_set_option(fbr_model *m, const char *key, double value)
{
    const FbrOptionKey *k = fbr_option_find(key);
    if (!m || !k || !(value == value)) {
        set_err(std::string("fbr_model_set_option: unknown option '") + (key ? key : "(null)") + "' or bad value");
        return FBR_E_INVALID;
    }
    if (m->opt.*(k->field) == value) return FBR_OK;
    // nothing may be in flight when the behaviour of the calls changes (tile programs are freed, chunk sizes move)
    if (int rc = enter_blocking(m)) return rc;
    HIPCHK(hipStreamSynchronize(m->stream));
    m->opt.*(k->field) = value;
    for (auto &r : m->rdm)
        if (r) r->opt = m->opt;
    if (k->rebuild_programs) {
        clear_programs(m);
        for (auto &r : m->rdm)
            if (r) clear_programs(r.get());
    }
    m->rd_grouped = -1;
    return FBR_OK;
}

extern "C" int fbr_model_get_option(const fbr_model *m, const char *key, double *value)
{
    const FbrOptionKey *k = fbr_option_find(key);
    if (!m || !k || !value) {
        set_err(std::string("fbr_model_get_option: unknown option '") + (key ? key : "(null)") + "'");
        return FBR_E_INVALID;
    }
    *value = m->opt.*(k->field);
    return FBR_OK;
}

extern "C" int fbr_model_option_name(int32_t index, const char **name)
{
    int n = 0;
    const FbrOptionKey *k = fbr_option_keys(&n);
    if (!name || index < 0 || index >= n) return FBR_E_INVALID;  // (index == count: the end of the list)
    *name = k[index].name;
    return FBR_OK;
}

extern "C" int fbr_model_dims(const fbr_model *m, int32_t *rows, int32_t *cols)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    if (rows) *rows = m->hm.rows;
    if (cols) *cols = m->hm.cols;
    return FBR_OK;
}

extern "C" int fbr_model_set_stream(fbr_model *m, void *s)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    // submissions in flight were enqueued on the stream used so far (their completion events and Gram launches live there): they
    // are waited for before the switch, so that fbr_wait / the destructor never look at a stream that does not carry them
    const hipStream_t next = s ? (hipStream_t)s : m->own_stream;
    if (next != m->stream) {
        if (int rc = enter_blocking(m)) return rc;
    }
    m->stream = next;
    for (auto &r : m->rdm)
        if (r) r->stream = next;  // (their submissions were waited for through this model's tickets)
    return FBR_OK;
}

extern "C" int fbr_profile_enable(fbr_model *m, int32_t on)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    m->prof = on != 0;
    for (auto &r : m->rdm)
        if (r) r->prof = m->prof;
    return FBR_OK;
}

extern "C" int fbr_profile_get(fbr_model *m, double *ms_out, int64_t *launches_out)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    for (int i = 0; i < FBR_PROF_COUNT; i++) {
        for (auto &r : m->rdm)
            if (r) {  // (the passes that ran on the reduced models)
                m->prof_ms[i] += r->prof_ms[i];
                m->prof_n[i] += r->prof_n[i];
                r->prof_ms[i] = 0;
                r->prof_n[i] = 0;
            }
        if (ms_out) ms_out[i] = m->prof_ms[i];
        if (launches_out) launches_out[i] = m->prof_n[i];
        m->prof_ms[i] = 0;
        m->prof_n[i] = 0;
    }
    return FBR_OK;
}
int stage_one(fbr_model *m, DevBuf &buf, const double *src, size_t count, int mem, const double **dst)
{
    if (!src) {
        *dst = nullptr;
        return FBR_OK;
    }
    if (mem == FBR_DEVICE) {
        *dst = src;
        return FBR_OK;
    }
    int rc = buf.ensure(std::max<size_t>(count, 1) * sizeof(double));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(double), hipMemcpyHostToDevice, m->stream));
    *dst = (const double *)buf.p;
    return FBR_OK;
}

// true iff p is pinned (page-locked / registered) host memory: hipMemcpyAsync from it is asynchronous
bool is_pinned_host(const void *p)
{
    if (!p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// defer_host: leave HOST inputs where they are (d receives the host pointers): the caller stages them chunk by chunk
int stage_states(fbr_model *m, const fbr_states *st, DevStates *d, bool need_vel, bool defer_host)
{
    if (!m || !st) {
        set_err("null argument");
        return FBR_E_INVALID;
    }
    if (st->num_samples < 0 || (st->mem != FBR_HOST && st->mem != FBR_DEVICE)) {
        set_err("bad fbr_states header");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    if (!st->q || (need_vel && (!st->dq || !st->ddq))) {
        set_err("q/dq/ddq must not be NULL");
        return FBR_E_INVALID;
    }
    if (hm.floating && (!st->base_rpy || (need_vel && (!st->base_vel || !st->base_acc)))) {
        set_err("floating base model needs base_vel/base_acc/base_rpy");
        return FBR_E_INVALID;
    }
    if (need_vel && hm.fric && !st->sign) {
        set_err("friction layout needs the Coulomb sign series (fbr_states.sign)");
        return FBR_E_INVALID;
    }
    if (int rc_enter = enter(m)) return rc_enter;
    if (!m->submitting) {  // blocking entry points run after every asynchronous submission before them
        if (int rc_w = wait_ticket(m, m->next_ticket - 1)) return rc_w;
    }
    const size_t S = (size_t)st->num_samples;
    d->S = (long)S;
    int rc;
    if (defer_host && st->mem == FBR_HOST) {
        d->q = st->q;
        d->dq = st->dq;
        d->ddq = st->ddq;
        d->rpy = hm.floating ? st->base_rpy : nullptr;
        d->bv = hm.floating ? st->base_vel : nullptr;
        d->ba = hm.floating ? st->base_acc : nullptr;
        d->sign = hm.fric ? st->sign : nullptr;
        return FBR_OK;
    }
    if ((rc = stage_one(m, m->st_q, st->q, S * hm.n, st->mem, &d->q))) return rc;
    if ((rc = stage_one(m, m->st_dq, st->dq ? st->dq : st->q, S * hm.n, st->mem, &d->dq))) return rc;
    if ((rc = stage_one(m, m->st_ddq, st->ddq ? st->ddq : st->q, S * hm.n, st->mem, &d->ddq))) return rc;
    if (hm.floating) {
        if ((rc = stage_one(m, m->st_rpy, st->base_rpy, S * 3, st->mem, &d->rpy))) return rc;
        // without velocities (contact Jacobian only) the twist inputs are irrelevant: reuse any valid buffer
        if ((rc = stage_one(m, m->st_bv, st->base_vel, S * 6, st->mem, &d->bv))) return rc;
        if ((rc = stage_one(m, m->st_ba, st->base_acc, S * 6, st->mem, &d->ba))) return rc;
    }
    if (hm.fric && st->sign)
        if ((rc = stage_one(m, m->st_sign, st->sign, S * hm.n, st->mem, &d->sign))) return rc;
    return FBR_OK;
}

long chunk_size(const fbr_model *m, long S)
{
    const size_t per = (size_t)m->hm.rec_size() * sizeof(double);
    long ch = (long)((size_t)(768u << 20) / per);
    if (ch < 1024) ch = 1024;
    if (m->opt.chunk_samples >= 1) ch = (long)m->opt.chunk_samples;  // tests: force the multi-chunk paths at small sizes
    return std::min(S, ch);
}

int run_kin(fbr_model *m, const DevStates &d, long s0, long cs, hipStream_t st, DevBuf *recbuf)
{
    const FbrHostModel &hm = m->hm;
    if (!st) st = m->stream;
    if (!recbuf) recbuf = &m->rec;
    int rc = recbuf->ensure((size_t)cs * hm.rec_size() * sizeof(double));
    if (rc) return rc;
    const int threads = 256;
    const int blocks = (int)((cs + threads - 1) / threads);
    ProfScope ps(m, FBR_PROF_KIN, st);
    // (one instance, no register cap: the 96-VGPR instance that once ran beside the Gram kernel spilled 41 registers and, since the column
    // reductions, was the slower one there as well -- kin 6.2 instead of 5.3 ms per 1 M samples: DESIGN.md 10)
    hipLaunchKernelGGL(fbr_kin_kernel<2>, dim3(blocks), dim3(threads), 0, st, m->dm, cs, d.q + s0 * hm.n, d.dq + s0 * hm.n, d.ddq + s0 * hm.n,
                       d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr, d.rpy ? d.rpy + s0 * 3 : nullptr, recbuf->as<double>());
    HIPCHK(hipGetLastError());
    return FBR_OK;
}

int finish_output(fbr_model *m, double *dev_src, double *user_dst, size_t count, int out_mem)
{
    if (out_mem == FBR_HOST)
        HIPCHK(hipMemcpyAsync(user_dst, dev_src, count * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    prof_collect(m);
    return FBR_OK;
}

// ------------------------------------------------------------------------------------------------
int launch_regressor(fbr_model *m, const DevStates &d, long s0, long cs, double *dst, int ldy, long rs_s, long rs_r, const int *linkpos, const int *skipfc)
{
    const FbrHostModel &hm = m->hm;
    const size_t lds = (size_t)hm.rec_size() * sizeof(double);
    const int spb = std::max(1, std::min(16, 256 / std::max(1, hm.cols / 2)));  // samples side by side in one workgroup
    const size_t lds2 = lds * spb;
    ProfScope ps(m, FBR_PROF_REGRESSOR);
    // even column count (and a 16-byte aligned output): paired columns, 16-byte stores
    if ((hm.cols & 1) == 0 && (((uintptr_t)dst) & 15) == 0 && (ldy & 1) == 0) {
        HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(fbr_regressor2_kernel, dim3((unsigned)std::min<long>((cs + spb - 1) / spb, (long)m->num_cus * 8)), dim3(256), lds2, m->stream, m->dm, cs, spb,
                           m->rec.as<double>(), d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, ldy, rs_s, rs_r, linkpos, skipfc);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fbr_regressor_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds, m->stream, m->dm, cs, m->rec.as<double>(),
                           d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, ldy, rs_s, rs_r, linkpos, skipfc);
    }
    HIPCHK(hipGetLastError());
    return FBR_OK;
}

extern "C" int fbr_regressor_batch(fbr_model *m, const fbr_states *st, double *Y_out, int32_t out_mem)
{
    DevStates d;
    int rc = stage_states(m, st, &d);
    if (rc) return rc;
    if (!Y_out) {
        set_err("Y_out is NULL");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    const size_t per = (size_t)hm.rows * hm.cols;
    const long S = d.S;
    if (S == 0) return FBR_OK;
    long ch = chunk_size(m, S);
    if (out_mem == FBR_HOST) {
        // bound the device staging buffer of the output to ~1 GiB
        long och = (long)((size_t)(1u << 30) / (per * sizeof(double)));
        ch = std::max(1L, std::min(ch, och));
        if ((rc = m->out_tmp.ensure((size_t)ch * per * sizeof(double)))) return rc;
    }
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        if ((rc = run_kin(m, d, s0, cs))) return rc;
        double *dst = (out_mem == FBR_HOST) ? m->out_tmp.as<double>() : Y_out + (size_t)s0 * per;
        if ((rc = launch_regressor(m, d, s0, cs, dst, hm.cols, (long)hm.rows, 1L, nullptr, nullptr))) return rc;
        if (out_mem == FBR_HOST) {
            HIPCHK(hipMemcpyAsync(Y_out + (size_t)s0 * per, dst, (size_t)cs * per * sizeof(double), hipMemcpyDeviceToHost,
                                  m->stream));
            HIPCHK(hipStreamSynchronize(m->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    prof_collect(m);
    return FBR_OK;
}

// Fused kinematics + torques (csrc/fbr_kinid.h): one kernel, one lane per sample, the link records stay in registers, branch-point
// records in a per-wave scratch.  mode 0 / 1: x = parameters (device); mode 2: x = [S][6] contact wrenches at frame (flink, fp).
static void kinid_params(const fbr_model *m, DevKinId *kp)
{
    kp->nsteps = m->kinid.nsteps;
    kp->maxlvl = m->kinid.maxlvl;
    kp->nslots = m->kinid.nslots;
    kp->ldn = std::max(m->hm.n, 1) | 1;
    kp->steps = m->kinid_steps;
    kp->endflush = m->kinid_endflush;
}
static int launch_kinid(fbr_model *m, const DevStates &d, long S, const double *dvs, const double *x, int mode, double *dst, int flink, const double *fp)
{
    DevKinId kp;
    kinid_params(m, &kp);
    const size_t lds = (size_t)3 * 64 * kp.ldn * sizeof(double);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(150 << 10) / std::max<size_t>(lds, 1)));
    const long nblk = (S + 63) / 64;
    const int blocks = (int)std::min<long>(nblk, (long)m->num_cus * per_cu);
    if (int rc = m->kinid_scratch.ensure((size_t)blocks * std::max(kp.nslots, 1) * FBR_LINK_REC * 64 * sizeof(double))) return rc;
    const double f0 = fp ? fp[0] : 0.0, f1 = fp ? fp[1] : 0.0, f2 = fp ? fp[2] : 0.0;
    ProfScope ps(m, FBR_PROF_ID);
#define FBR_KINID_LAUNCH(D)                                                                                                              \
    do {                                                                                                                                 \
        HIPCHK(hipFuncSetAttribute((const void *)fbr_kinid_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
        hipLaunchKernelGGL(fbr_kinid_kernel<D>, dim3(blocks), dim3(64), lds, m->stream, m->dm, kp, S, d.q, d.dq, d.ddq, d.bv, d.ba, d.rpy, \
                           d.sign, dvs, x, mode, dst, m->kinid_scratch.as<double>(), flink, f0, f1, f2);                                 \
    } while (0)
    if (kp.maxlvl <= 4)
        FBR_KINID_LAUNCH(4);
    else if (kp.maxlvl <= 8)
        FBR_KINID_LAUNCH(8);
    else if (kp.maxlvl <= 12)
        FBR_KINID_LAUNCH(12);
    else
        FBR_KINID_LAUNCH(FBR_KINID_MAXD);
#undef FBR_KINID_LAUNCH
    HIPCHK(hipGetLastError());
    return FBR_OK;
}

static int run_id(fbr_model *m, const fbr_states *st, const double *x, int nx, const double *vel_sign, int mode,
                  double *tau_out, int32_t out_mem)
{
    // Y x = Y_red (E x): torques are linear in the parameters, and the parameters of a link welded to a moving body are parameters of that
    // body (build_reduction, merged model rdm[0]: every moving link keeps its ten columns).  The kinematics and the per-link wrench loop
    // then run over the moving bodies only (WALK-MAN: 30 of 48 links) -- same rows, same result to rounding.
    if (m && x && st && m->rdm[0] && m->opt.link_merge && m->pid == getpid()) {
        fbr_model *r = m->rdm[0].get();
        const FbrHostModel &hm = m->hm, &rh = r->hm;
        const int ninert = hm.cpl * hm.L, rin = rh.cpl * rh.L;
        const int full = mode == 0 ? ninert : hm.cols;  // mode 0: the inertial block of x_std; mode 1: every identified column
        if (hm.cpl == 10 && nx >= full) {
            std::vector<double> xr((size_t)std::max(rin + (nx - ninert), rh.cols), 0.0);
            const std::vector<int> &eb = m->hE_beg[0], &er = m->hE_row[0];
            const std::vector<double> &ev = m->hE_val[0];
            for (int j = 0; j < ninert; j++)
                for (int e = eb[j]; e < eb[j + 1]; e++) xr[er[e]] += ev[e] * x[j];
            for (int j = ninert; j < nx; j++) xr[rin + (j - ninert)] = x[j];  // friction slots / columns: the same joints in the same layout
            if (int rc = enter_blocking(m)) return rc;
            r->stream = m->stream;
            r->prof = m->prof;
            return run_id(r, st, xr.data(), rin + (nx - ninert), vel_sign, mode, tau_out, out_mem);
        }
    }
    DevStates d;
    int rc = stage_states(m, st, &d);
    if (rc) return rc;
    const FbrHostModel &hm = m->hm;
    if (!x || !tau_out) {
        set_err("null x / tau_out");
        return FBR_E_INVALID;
    }
    const int need = (mode == 0) ? (hm.fric ? hm.friction_start() + (hm.cols - hm.cpl * hm.L) : 10 * hm.L) : hm.cols;
    if (mode == 0 && nx < std::max(need, 10 * hm.L)) {
        set_err("x_std too short for this model layout");
        return FBR_E_INVALID;
    }
    const long S = d.S;
    if (S == 0) return FBR_OK;
    if ((rc = m->st_x.ensure((size_t)std::max(nx, 1) * sizeof(double)))) return rc;
    HIPCHK(hipMemcpyAsync(m->st_x.p, x, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, m->stream));
    const double *dvs = nullptr;
    if (mode == 0 && hm.fric && hm.stribeck > 0) {
        if (!vel_sign) {
            set_err("Stribeck model needs vel_sign");
            return FBR_E_INVALID;
        }
        if ((rc = stage_one(m, m->st_aux, vel_sign, (size_t)S * hm.n, st->mem, &dvs))) return rc;
    }
    double *dst = tau_out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->out_tmp.ensure((size_t)S * hm.rows * sizeof(double)))) return rc;
        dst = m->out_tmp.as<double>();
    }
    if (m->opt.fused_id != 0 && m->kinid.nsteps > 0) {
        if ((rc = launch_kinid(m, d, S, dvs, m->st_x.as<double>(), mode, dst, 0, nullptr))) return rc;
        return finish_output(m, dst, tau_out, (size_t)S * hm.rows, out_mem);
    }
    const int waves = 4;
    const size_t lds = (size_t)waves * (hm.rec_size() + 6 * hm.L) * sizeof(double);
    HIPCHK(hipFuncSetAttribute((const void *)fbr_id_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long ch = chunk_size(m, S);
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        if ((rc = run_kin(m, d, s0, cs))) return rc;
        const int blocks = (int)std::min<long>((cs + waves - 1) / waves, (long)m->num_cus * 8);
        ProfScope ps(m, FBR_PROF_ID);
        hipLaunchKernelGGL(fbr_id_kernel, dim3(blocks), dim3(64 * waves), lds, m->stream, m->dm, cs, m->rec.as<double>(),
                           d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dvs ? dvs + s0 * hm.n : nullptr,
                           m->st_x.as<double>(), mode, dst + (size_t)s0 * hm.rows);
        HIPCHK(hipGetLastError());
    }
    return finish_output(m, dst, tau_out, (size_t)S * hm.rows, out_mem);
}

extern "C" int fbr_inverse_dynamics_batch(fbr_model *m, const fbr_states *st, const double *x_std, int32_t num_x,
                                          const double *vel_sign, double *tau_out, int32_t out_mem)
{
    return run_id(m, st, x_std, num_x, vel_sign, 0, tau_out, out_mem);
}

extern "C" int fbr_predict(fbr_model *m, const fbr_states *st, const double *x, double *tau_out, int32_t out_mem)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    return run_id(m, st, x, m->hm.cols, nullptr, 1, tau_out, out_mem);
}

extern "C" int fbr_contact_torques(fbr_model *m, const fbr_states *st, int32_t link, const double *frame_R,
                                   const double *frame_p, const double *wrench, double *out, int32_t out_mem)
{
    (void)frame_R;
    if (!m || !st) {
        set_err("null argument");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    if (link < 0 || link >= hm.L || !frame_p || !wrench || !out) {
        set_err("bad contact frame / null pointer");
        return FBR_E_INVALID;
    }
    fbr_states s2 = *st;
    // only q and rpy matter for the Jacobian: feed q as velocity placeholders (never read into the result)
    s2.dq = st->q;
    s2.ddq = st->q;
    if (hm.floating) {
        s2.base_vel = nullptr;
        s2.base_acc = nullptr;
    }
    s2.sign = nullptr;
    DevStates d;
    int rc = stage_states(m, &s2, &d, false);
    if (rc) return rc;
    const long S = d.S;
    if (S == 0) return FBR_OK;
    if (hm.floating) {
        // zero twist / acceleration buffers
        if ((rc = m->st_bv.ensure((size_t)S * 6 * sizeof(double)))) return rc;
        HIPCHK(hipMemsetAsync(m->st_bv.p, 0, (size_t)S * 6 * sizeof(double), m->stream));
        d.bv = d.ba = m->st_bv.as<double>();
    }
    const double *dw = nullptr;
    if ((rc = stage_one(m, m->st_aux, wrench, (size_t)S * 6, st->mem, &dw))) return rc;
    double *dst = out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->out_tmp.ensure((size_t)S * hm.rows * sizeof(double)))) return rc;
        dst = m->out_tmp.as<double>();
    }
    if (m->opt.fused_id != 0 && m->kinid.nsteps > 0) {  // the same fused kernel: only the frame's link carries a wrench (fbr_kinid.h, mode 2)
        if ((rc = launch_kinid(m, d, S, nullptr, dw, 2, dst, link, frame_p))) return rc;
        return finish_output(m, dst, out, (size_t)S * hm.rows, out_mem);
    }
    const long ch = chunk_size(m, S);
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        if ((rc = run_kin(m, d, s0, cs))) return rc;
        hipLaunchKernelGGL(fbr_contact_kernel, dim3((unsigned)((cs + 255) / 256)), dim3(256), 0, m->stream, m->dm, cs,
                           m->rec.as<double>(), link, frame_p[0], frame_p[1], frame_p[2], dw + s0 * 6,
                           dst + (size_t)s0 * hm.rows);
        HIPCHK(hipGetLastError());
    }
    return finish_output(m, dst, out, (size_t)S * hm.rows, out_mem);
}
// ngroups > 1: the samples form ngroups consecutive groups of equal size, one Gram per group (G_out [ngroups][Pa][Pa])
// Which regressor rows carry a non-zero weight for at least one sample (device scan of w; all rows when there are no weights).
int active_rows(fbr_model *m, const double *dw, long S, std::vector<char> *act)
{
    const int rows = m->hm.rows;
    act->assign(rows, 1);
    if (!dw || S <= 0) return FBR_OK;
    int rc;
    if ((rc = m->row_flags.ensure((size_t)rows * sizeof(int)))) return rc;
    HIPCHK(hipMemsetAsync(m->row_flags.p, 0, (size_t)rows * sizeof(int), m->stream));
    hipLaunchKernelGGL(fbr_row_active_kernel, dim3(1024), dim3(256), 0, m->stream, dw, S, rows, m->row_flags.as<int>());
    HIPCHK(hipGetLastError());
    std::vector<int> h(rows);
    HIPCHK(hipMemcpyAsync(h.data(), m->row_flags.p, (size_t)rows * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    for (int r = 0; r < rows; r++) (*act)[r] = h[r] != 0;
    return FBR_OK;
}
// Block until the submission with this ticket (and every earlier one) is complete; ticket < 0 or beyond the last one: everything.
int wait_ticket(fbr_model *m, int64_t ticket)
{
    const int64_t last = m->next_ticket - 1;
    if (ticket > last) ticket = last;
    if (ticket <= m->waited_ticket) return FBR_OK;
    // the completion event is what carries the submission (recorded on the stream it ran on, whatever m->stream is by now)
    HIPCHK(hipEventSynchronize(m->ev_done[ticket & 1]));
    if (ticket == last) {
        HIPCHK(hipStreamSynchronize(m->stream));
        HIPCHK(hipStreamSynchronize(m->side));
        if (m->copy) HIPCHK(hipStreamSynchronize(m->copy));
        prof_collect(m);
    }
    const int64_t first = m->waited_ticket + 1;
    m->waited_ticket = ticket;
    for (int64_t t = std::max(first, ticket - 1); t <= ticket; t++)
        if (m->ticket_via_red[t & 1]) {  // the pass ran on a reduced model: its bookkeeping, profile and error word
            fbr_model *r = m->rdm[m->ticket_via_red[t & 1] - 1].get();
            m->ticket_via_red[t & 1] = 0;
            if (int rc = wait_ticket(r, m->red_ticket[t & 1])) return rc;
        }
    for (int64_t t = std::max(first, ticket - 1); t <= ticket; t++)  // (at most two submissions were in flight)
        if (m->ticket_kind[t & 1] == 1 && m->tsqr_err_host && m->tsqr_err_host[t & 1]) {
            char hx[16];
            snprintf(hx, sizeof hx, "%08x", m->tsqr_err_host[t & 1]);
            m->tsqr_err_host[t & 1] = 0;
            set_err("TSQR pipeline flag wait timed out (internal error, code " + std::string(hx) + ") in submission " + std::to_string(t));
            return FBR_E_HIP;
        }
    return FBR_OK;
}
extern "C" int fbr_wait(fbr_model *m, int64_t ticket)
{
    int rc = enter(m);
    if (rc) return rc;
    return wait_ticket(m, ticket < 0 ? m->next_ticket - 1 : ticket);
}

extern "C" int fbr_fd_scores(fbr_model *m, const fbr_states *st, const double *W, double eps, double *out, int32_t out_mem)
{
    DevStates d;
    int rc = stage_states(m, st, &d);
    if (rc) return rc;
    if (!W || !out) {
        set_err("W / out is NULL");
        return FBR_E_INVALID;
    }
    const FbrHostModel &hm = m->hm;
    const long S = d.S;
    const int n = hm.n, nper = 1 + 3 * n;
    const double *dW = nullptr;
    if ((rc = stage_one(m, m->st_aux, W, (size_t)S * hm.rows * hm.cols, st->mem, &dW))) return rc;
    double *dout = out;
    if (out_mem == FBR_HOST) {
        if ((rc = m->g_tmp.ensure((size_t)S * nper * sizeof(double)))) return rc;
        dout = m->g_tmp.as<double>();
    }
    if (S > 0 && m->opt.fused_id != 0 && m->kinid.nsteps > 0 && !hm.masked) {
        // one lane per evaluation, nothing staged (fbr_kinfd_kernel, fbr_kinid.h)
        DevKinId kp;
        kinid_params(m, &kp);
        const long nblk = (S * nper + 63) / 64;
        const int blocks = (int)std::min<long>(nblk, (long)m->num_cus * 8);
        if ((rc = m->kinid_scratch.ensure((size_t)blocks * std::max(kp.nslots, 1) * FBR_LINK_REC * 64 * sizeof(double)))) return rc;
        {
            ProfScope ps(m, FBR_PROF_REGRESSOR);
#define FBR_KINFD_LAUNCH(D)                                                                                                               \
    hipLaunchKernelGGL(fbr_kinfd_kernel<D>, dim3(blocks), dim3(64), 0, m->stream, m->dm, kp, S, nper, eps, d.q, d.dq, d.ddq, d.bv, d.ba, d.rpy, \
                       d.sign, dW, dout, m->kinid_scratch.as<double>())
            if (kp.maxlvl <= 4)
                FBR_KINFD_LAUNCH(4);
            else if (kp.maxlvl <= 8)
                FBR_KINFD_LAUNCH(8);
            else if (kp.maxlvl <= 12)
                FBR_KINFD_LAUNCH(12);
            else
                FBR_KINFD_LAUNCH(FBR_KINID_MAXD);
#undef FBR_KINFD_LAUNCH
        }
        HIPCHK(hipGetLastError());
    } else if (S > 0) {
        // columns that a perturbation of joint d can change: the inertial columns of the links below d and d's own friction columns
        if (m->fd_tab_entries < 0) {
            std::vector<int> tab(n + 1, 0), ent;
            for (int dj = 0; dj < n; dj++) {
                tab[dj] = (int)ent.size();
                for (int c = 0; c < hm.cols; c++) {
                    const FbrCol &cd = hm.coldesc[c];
                    const bool on = cd.kind == 0 ? std::find(hm.path[cd.link].begin(), hm.path[cd.link].end(), dj) != hm.path[cd.link].end() : cd.joint == dj;
                    if (on) ent.push_back(c);
                }
            }
            tab[n] = (int)ent.size();
            tab.insert(tab.end(), ent.begin(), ent.end());
            if ((rc = m->fd_tab.ensure(tab.size() * sizeof(int)))) return rc;
            HIPCHK(hipMemcpyAsync(m->fd_tab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, m->stream));
            HIPCHK(hipStreamSynchronize(m->stream));  // tab is a local
            m->fd_tab_entries = (int)ent.size();
        }
        const int *jbeg = m->fd_tab.as<int>(), *jcols = jbeg + n + 1;
        // chunks of original samples such that the expanded kinematic records stay within the usual chunk
        long ch = std::max(1L, chunk_size(m, S * nper) / nper);
        if ((rc = m->fd_part.ensure((size_t)std::min(ch, S) * std::max(n, 1) * sizeof(double)))) return rc;
        const size_t lds = ((size_t)hm.rec_size() + 4 + hm.cols) * sizeof(double);
        HIPCHK(hipFuncSetAttribute((const void *)fbr_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const size_t cnt[7] = {(size_t)n, (size_t)n, (size_t)n, 6, 6, 3, (size_t)n};
        for (long s0 = 0; s0 < S; s0 += ch) {
            const long cs = std::min(ch, S - s0), ce = cs * nper;
            for (int i = 0; i < 7; i++)
                if ((rc = m->fd[i].ensure((size_t)ce * cnt[i] * sizeof(double)))) return rc;
            hipLaunchKernelGGL(fbr_fd_expand_kernel, dim3((unsigned)std::min<long>((ce + 255) / 256, 4096)), dim3(256), 0, m->stream, cs, n,
                               d.bv ? 1 : 0, d.sign ? 1 : 0, eps, d.q + s0 * n, d.dq + s0 * n, d.ddq + s0 * n, d.bv ? d.bv + s0 * 6 : nullptr,
                               d.ba ? d.ba + s0 * 6 : nullptr, d.rpy ? d.rpy + s0 * 3 : nullptr, d.sign ? d.sign + s0 * n : nullptr,
                               m->fd[0].as<double>(), m->fd[1].as<double>(), m->fd[2].as<double>(), m->fd[3].as<double>(),
                               m->fd[4].as<double>(), m->fd[5].as<double>(), m->fd[6].as<double>());
            HIPCHK(hipGetLastError());
            DevStates de;
            de.S = ce;
            de.q = m->fd[0].as<double>();
            de.dq = m->fd[1].as<double>();
            de.ddq = m->fd[2].as<double>();
            if (d.bv) {
                de.bv = m->fd[3].as<double>();
                de.ba = m->fd[4].as<double>();
                de.rpy = m->fd[5].as<double>();
            }
            if (d.sign) de.sign = m->fd[6].as<double>();
            if ((rc = run_kin(m, de, 0, ce))) return rc;
            {
                ProfScope ps(m, FBR_PROF_REGRESSOR);
                for (int phase = 0; phase < (n > 0 ? 2 : 1); phase++)
                    hipLaunchKernelGGL(fbr_score_kernel, dim3((unsigned)std::min<long>(phase ? cs * (nper - 1) : cs, (long)m->num_cus * 8)), dim3(256), lds,
                                       m->stream, m->dm, cs, nper, phase, m->rec.as<double>(), de.dq, de.sign, dW + (size_t)s0 * hm.rows * hm.cols,
                                       dout + (size_t)s0 * nper, m->fd_part.as<double>(), jbeg, jcols);
            }
            HIPCHK(hipGetLastError());
        }
    }
    return finish_output(m, dout, out, (size_t)S * nper, out_mem);
}
// ------------------------------------------------------------------------------------------------
// Fourier-series states of candidate trajectories, generated on the device (fbr.h)
// ------------------------------------------------------------------------------------------------
extern "C" int fbr_fourier_states(fbr_model *m, int32_t ncand, int64_t T, int32_t nharm, double freq, const double *wf, const double *a, const double *b,
                                  const double *q_offset, const double *q_range, double *q, double *dq, double *ddq, int32_t out_mem)
{
    if (!m || ncand < 1 || T < 1 || nharm < 1 || !(freq > 0) || !wf || !a || !b || !q_offset || !q || !dq || !ddq ||
        (out_mem != FBR_HOST && out_mem != FBR_DEVICE)) {
        set_err("fbr_fourier_states: bad arguments");
        return FBR_E_INVALID;
    }
    if (int rc = enter_blocking(m)) return rc;
    const int n = m->hm.n;
    const size_t nc = (size_t)ncand * n, ncoef = nc * nharm, count = (size_t)ncand * (size_t)T * n;
    // coefficients: [wf (C) | a | b | q_offset | q_range] in one staging buffer (host arrays, a few KB)
    std::vector<double> h;
    h.insert(h.end(), wf, wf + ncand);
    h.insert(h.end(), a, a + ncoef);
    h.insert(h.end(), b, b + ncoef);
    h.insert(h.end(), q_offset, q_offset + nc);
    if (q_range) h.insert(h.end(), q_range, q_range + nc);
    int rc;
    if ((rc = m->st_x.ensure(h.size() * sizeof(double)))) return rc;
    HIPCHK(hipMemcpyAsync(m->st_x.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));  // (h is a local)
    const double *d = m->st_x.as<double>();
    const double *dwf = d, *da = d + ncand, *db = da + ncoef, *doff = db + ncoef, *drng = q_range ? doff + nc : nullptr;
    double *oq = q, *odq = dq, *oddq = ddq;
    if (out_mem == FBR_HOST) {
        if ((rc = m->out_tmp.ensure(3 * count * sizeof(double)))) return rc;
        oq = m->out_tmp.as<double>();
        odq = oq + count;
        oddq = odq + count;
    }
    hipLaunchKernelGGL(fbr_fourier_kernel, dim3((unsigned)std::min<size_t>((count + 255) / 256, (size_t)m->num_cus * 32)), dim3(256), 0, m->stream, (int)ncand, (long)T, n,
                       (int)nharm, freq, dwf, da, db, doff, drng, oq, odq, oddq);
    HIPCHK(hipGetLastError());
    if (out_mem == FBR_HOST) {
        HIPCHK(hipMemcpyAsync(q, oq, count * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipMemcpyAsync(dq, odq, count * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipMemcpyAsync(ddq, oddq, count * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(hipStreamSynchronize(m->stream));
    return FBR_OK;
}

// A submission that fails after work was enqueued has no ticket its caller could wait on: everything in flight is drained before the
// error is returned (the same for the Gram pass, gram_impl below), so that the inputs may be freed and later calls start from a quiet device.
int drain_after_failed_submit(fbr_model *m)
{
    if (!m) return FBR_OK;
    for (auto &r : m->rdm)
        if (r) drain_after_failed_submit(r.get());
    m->ticket_via_red[0] = m->ticket_via_red[1] = 0;
    (void)hipStreamSynchronize(m->stream);
    if (m->side) (void)hipStreamSynchronize(m->side);
    if (m->copy) (void)hipStreamSynchronize(m->copy);
    if (m->tsqr_pro_stream) (void)hipStreamSynchronize(m->tsqr_pro_stream);
    for (auto &s2 : m->tsqr_streams)
        if (s2) (void)hipStreamSynchronize(s2);
    m->waited_ticket = m->next_ticket - 1;
    m->ev_gram_rec[0] = m->ev_gram_rec[1] = m->ev_pack_rec[0] = m->ev_pack_rec[1] = false;
    m->tsqr_l0_rec = false;
    return FBR_OK;
}
