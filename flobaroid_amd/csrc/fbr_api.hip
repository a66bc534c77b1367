// fbr_api.hip -- C-ABI of libfbr (see include/fbr.h).  Host side: device tables, workspace, launches.
// gfx950 only; there is deliberately no CPU path in this library.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <numeric>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fbr.h"
#include "fbr_kernels.h"
#include "fbr_tsqr.h"
#include "fbr_reduce.h"
#include "fbr_signal.h"

static thread_local std::string g_err;
static void set_err(const std::string &s) { g_err = s; }

#define HIPCHK(call)                                                                            \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            set_err(std::string(#call) + ": " + hipGetErrorString(e__));                        \
            return FBR_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

struct DevBuf {  // owning device allocation (move-only)
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes)
    {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    int ensure(size_t need)
    {
        if (need <= bytes) return FBR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            set_err(std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
            return FBR_E_HIP;
        }
        bytes = need;
        return FBR_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() { return (T *)p; }
};

template <class T> static int upload(std::vector<DevBuf> &pool, const std::vector<T> &v, const T **out)
{
    pool.emplace_back();
    DevBuf &b = pool.back();
    size_t n = std::max<size_t>(v.size(), 1) * sizeof(T);
    int rc = b.ensure(n);
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)b.p;
    return FBR_OK;
}

struct GramHolder {
    FbrGramProgram prog;
    DevGram dev;
    std::vector<DevBuf> pool;
    DevBuf pimg[2];           // packed tile images of one chunk of samples, double buffered (zeroed when (re)allocated)
    bool moments = false;     // the rhs columns have no tiles: their products come from the pack kernel (fbr_gram_rhs_moments)
    DevBuf mom[2];            // [pack workgroups][256][4] partial rhs moments of a call, by ticket parity
    bool mom_clean[2] = {false, false};  // the buffer holds zeros (left by the reduction of the call before; false after a failed call)
    const int *itemcol = nullptr;  // [256] regressor column of pack thread t (-1: none)
    size_t lds_bytes = 0;     // streaming Gram kernel
    size_t pack_lds_bytes = 0;
    struct Deal { const int2 *tab; const int *begin; };
    std::map<int, Deal> deals;  // workgroups per sample group -> device tables of fbr_gram_deal (at most one per count)
};

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

struct fbr_model {
    FbrHostModel hm;
    DevModel dm;
    int device = 0;
    pid_t pid = 0;                              // process that created the handle (HIP state does not survive fork())
    int num_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side = nullptr;                 // producer stream: kinematics + tile-image packing of the next chunk
    hipStream_t copy = nullptr;                 // staging stream: host -> device copies of the chunk after next (pinned host inputs)
    hipEvent_t ev_pack[2] = {nullptr, nullptr}, ev_gram[2] = {nullptr, nullptr}, ev_fork = nullptr, ev_h2d[2] = {nullptr, nullptr};
    // asynchronous submissions (fbr_gram_submit / fbr_wait): completion event of the submission with ticket t is ev_done[t & 1]
    hipEvent_t ev_done[2] = {nullptr, nullptr};
    int64_t next_ticket = 0;       // ticket of the next submission
    int64_t waited_ticket = -1;    // every ticket <= this one is known complete
    bool submitting = false;       // inside fbr_gram_submit
    bool ev_gram_rec[2] = {false, false};
    bool ev_pack_rec[2] = {false, false};  // ev_gram[b] has been recorded at least once (a later producer may have to wait for it)
    DevBuf rec2;
    std::vector<DevBuf> tables;
    std::map<int, std::unique_ptr<GramHolder>> gram;
    // workspace
    DevBuf st_q, st_dq, st_ddq, st_bv, st_ba, st_rpy, st_sign, st_aux, st_aux2, st_x;
    DevBuf rec, partial, out_tmp, g_tmp;
    DevBuf st_chunk[2];       // per-chunk staging of pinned host inputs (fused Gram pass), double buffered with the tile images
    DevBuf fd[7];             // expanded states of the finite-difference sweep (q, dq, ddq, base_vel, base_acc, rpy, sign)
    DevBuf row_flags;         // active_rows(): per regressor row, does any sample weight it
    DevBuf fd_tab, fd_part;   // sub-tree column lists of every joint [n + 1 | entries] (built on first use), baseline partial sums [S][n]
    int fd_tab_entries = -1;
    FbrTsqrWork tsqr;
    std::vector<FbrTsqrWork> tsqr_groups;  // one factorisation per row group of the tree-structured TSQR (tsqr_group_plan)
    hipStream_t tsqr_streams[4] = {nullptr, nullptr, nullptr, nullptr};  // the groups' merge trees run beside the final factor's (created on first use)
    hipEvent_t tsqr_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [i]: side stream i is done; [4]: fork point on the main stream
    DevBuf tsqr_rtmp;         // factor in the internal column order before it is brought back to the caller's
    DevBuf tsqr_embed;        // stacked rows of the embedded group factors (tree-structured TSQR)
    DevBuf gram_r_tmp;        // factor of gram_via_tsqr (robots beyond the fused Gram's 60 rows per sample)
    // Link merging (build_reduction): the same robot with every FIXED link merged into the moving body it is attached to.  The
    // regressor columns of a fixed link are exact linear combinations of its body's columns (Y_c = Y_a T, T the 10 x 10 change of
    // frame of the inertial parameters), so the reductions run on the moving bodies' columns only and are expanded at the end:
    // G = E^T G_red E,  R = qr(R_red E).  `red` has its own workspaces and runs on this model's stream.
    // Regrouping (second reduction): a revolute joint lets three more parameter directions of its link -- the mass, the first moment
    // along the axis and the inertia 1 - a a^T -- act exactly like parameters of the parent body (they are invariant under the joint's
    // rotation), the classical base-parameter regrouping.  The second reduced model computes 7 instead of 10 columns for every link
    // behind a joint (column masks, link frames turned so that the joint axis is z: m, h_z and I_yy dropped) and E grows accordingly.
    // rdm[0]: fixed links merged (every entry point works on it); rdm[1]: merged + regrouped (fused Gram and the row-group TSQR only).
    std::unique_ptr<fbr_model> rdm[2];
    const int *E_beg[2] = {nullptr, nullptr}, *E_row[2] = {nullptr, nullptr};  // CSC of the augmented E [(cols_red + 16) x (cols + 16)]: column j
    const double *E_val[2] = {nullptr, nullptr};  // of the full layout = sum of E_val[e] x (reduced column E_row[e]), e in [E_beg[j], E_beg[j+1])
    DevBuf red_out[2];        // G_red / R_red of a pass, by ticket parity
    DevBuf red_w;             // G_red E (Gram expansion, second half: E^T (G_red E))
    int64_t red_ticket[2] = {-1, -1};  // the reduced model's ticket behind this model's ticket of that parity
    int ticket_via_red[2] = {0, 0};    // 0: the pass ran on this model; 1 + i: on rdm[i]
    bool is_reduction = false;         // this model is some model's rdm[i]
    int rd_grouped = -1;               // rdm[1]'s factorisations take the row-group path given enough samples (-1: not looked at yet)
    // per-call state of the TSQR entry points, double buffered by the parity of the call's ticket so that a submission (fbr_tsqr_submit)
    // can be enqueued while the one before is still running
    DevBuf tsqr_tab[2];                        // device tables (index lists, entry lists, group records)
    void *tsqr_tab_host[2] = {nullptr, nullptr};  // their pinned host staging (the copy is asynchronous: the source must outlive it)
    size_t tsqr_tab_host_bytes[2] = {0, 0};
    unsigned *tsqr_err = nullptr;              // device word every factorisation of a call reports into (pipeline flag time-out)
    unsigned *tsqr_err_host = nullptr;         // pinned [2]: its value at the end of the call with that ticket parity
    int ticket_kind[2] = {0, 0};               // what the submission with that parity was: 0 = Gram pass, 1 = TSQR
    int last_submit_kind = 0;
    hipEvent_t ev_tsqr_l0 = nullptr;           // the last level-0 fold of the latest TSQR call has been enqueued behind this event
    hipEvent_t ev_tsqr_pro = nullptr;          // prologue (kinematics + first chunk's writer on the producer stream) of a submission
    hipStream_t tsqr_pro_stream = nullptr;     // the stream it runs on (created on first use, confined to part of the CUs)
    bool tsqr_l0_rec = false;
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<std::pair<int, int>> ev_used;  // (class, pool index)
    double prof_ms[FBR_PROF_COUNT] = {0};
    int64_t prof_n[FBR_PROF_COUNT] = {0};

    fbr_model() = default;
    fbr_model(const fbr_model &) = delete;
    fbr_model &operator=(const fbr_model &) = delete;
    // Releases what the handle owns besides its DevBufs (also on the error paths of fbr_model_create, through unique_ptr).
    ~fbr_model()
    {
        fbr_model *m = this;
        (void)hipSetDevice(m->device);
        // submissions still in flight (fbr_gram_submit without fbr_wait) read the workspaces freed below
        if (m->stream) (void)hipStreamSynchronize(m->stream);
        if (m->side) (void)hipStreamSynchronize(m->side);
        if (m->copy) (void)hipStreamSynchronize(m->copy);
        m->tsqr.release();
        for (auto &g : m->tsqr_groups) g.release();
        for (auto &h : m->tsqr_tab_host)
            if (h) (void)hipHostFree(h);
        if (m->tsqr_err) (void)hipFree(m->tsqr_err);
        if (m->tsqr_err_host) (void)hipHostFree(m->tsqr_err_host);
        if (m->ev_tsqr_l0) (void)hipEventDestroy(m->ev_tsqr_l0);
        if (m->ev_tsqr_pro) (void)hipEventDestroy(m->ev_tsqr_pro);
        if (m->tsqr_pro_stream) {
            (void)hipStreamSynchronize(m->tsqr_pro_stream);
            (void)hipStreamDestroy(m->tsqr_pro_stream);
        }
        for (auto &st : m->tsqr_streams)
            if (st) (void)hipStreamDestroy(st);
        for (auto &e : m->tsqr_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &e : m->ev_pool) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (m->side) (void)hipStreamDestroy(m->side);
        if (m->copy) (void)hipStreamDestroy(m->copy);
        for (int i = 0; i < 2; i++) {
            if (m->ev_done[i]) (void)hipEventDestroy(m->ev_done[i]);
            if (m->ev_h2d[i]) (void)hipEventDestroy(m->ev_h2d[i]);
            if (m->ev_pack[i]) (void)hipEventDestroy(m->ev_pack[i]);
            if (m->ev_gram[i]) (void)hipEventDestroy(m->ev_gram[i]);
        }
        if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
        if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    }
};

// Bracket a launch with events (no-op unless profiling is on).
struct ProfScope {
    fbr_model *m;
    int idx = -1;
    hipStream_t st;
    ProfScope(fbr_model *m_, int cls, hipStream_t st_ = nullptr) : m(m_), st(st_ ? st_ : m_->stream)
    {
        if (!m->prof) return;
        size_t i = m->ev_used.size();
        if (i >= m->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            m->ev_pool.emplace_back(a, b);
        }
        idx = (int)i;
        m->ev_used.emplace_back(cls, idx);
        (void)hipEventRecord(m->ev_pool[idx].first, st);
    }
    ~ProfScope()
    {
        if (idx >= 0) (void)hipEventRecord(m->ev_pool[idx].second, st);
    }
};
static void prof_collect(fbr_model *m)
{
    for (auto &u : m->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev_pool[u.second].first, m->ev_pool[u.second].second) == hipSuccess) {
            m->prof_ms[u.first] += ms;
            m->prof_n[u.first] += 1;
        }
    }
    m->ev_used.clear();
}

// ------------------------------------------------------------------------------------------------
extern "C" int fbr_version(void) { return 100; }

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
static int enter(fbr_model *m)
{
    if (m->pid != getpid()) {
        set_err("this fbr_model was created in another process (before fork()): create one per process");
        return FBR_E_FORK;
    }
    HIPCHK(hipSetDevice(m->device));
    return FBR_OK;
}

static int wait_ticket(fbr_model *m, int64_t ticket);
// entry of a blocking call that does not go through stage_states: every asynchronous submission before it has completed
static int enter_blocking(fbr_model *m)
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

extern "C" const char *fbr_last_error(void) { return g_err.c_str(); }

static int build_reduction(fbr_model *m, const fbr_topology *t, int which);
#define FBR_E_NOT_GROUPED (-1000)  // internal: a model with column masks was asked for a factorisation its row-group path does not take

static int create_model(const fbr_topology *t, int device, fbr_model **out, bool allow_merge, const unsigned short *linkmask = nullptr)
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
                    t->gravity, t->friction, t->friction_symmetric, t->gravity_only, t->stribeck_velocity, linkmask);
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
        const char *pe = getenv("FBR_SIDE_PRIORITY");  // experiments: "high" / "none"
        if (pe && pe[0] == 'n')
            HIPCHK(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
        else
            HIPCHK(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, (pe && pe[0] == 'h') ? greatest : least));
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
    if (allow_merge && !getenv("FBR_NO_LINK_MERGE")) {
        if ((rc = build_reduction(m.get(), t, 0))) return rc;
        if (!getenv("FBR_NO_REGROUP") && (rc = build_reduction(m.get(), t, 1))) return rc;
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
    return FBR_OK;
}

// G (+)= E^T W, W = G_red E [Pra x Pa] (fbr_expand_rows_kernel) on the augmented layouts (Pa = cols + k, Pra = cols_red + k); the k rhs
// columns of E sit at E_beg[cols + r]
__global__ __launch_bounds__(256) void fbr_expand_gram_kernel(int cols, int k, int Pra, const int *__restrict__ Eb, const int *__restrict__ Er,
                                                               const double *__restrict__ Ev, const double *__restrict__ Gred, double *__restrict__ G,
                                                               int accumulate)
{
    const int Pa = cols + k;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)Pa * Pa; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / Pa), j = (int)(e - (long)i * Pa);
        if (i > j) continue;  // the upper triangle is computed, the lower one mirrored: G is symmetric to the bit, like the fused Gram's
        double acc = 0.0;  // (Gred here: W = G_red E [Pra x Pa], fbr_expand_rows_kernel)
        for (int a = Eb[i]; a < Eb[i + 1]; a++) acc += Ev[a] * Gred[(long)Er[a] * Pa + j];
        G[e] = accumulate ? G[e] + acc : acc;
        if (i != j) G[(long)j * Pa + i] = accumulate ? G[(long)j * Pa + i] + acc : acc;
    }
}

// dst[r][j] (leading dimension ldd) = (R_red E)[r][j] for r < Pra, j < Pa: the rows the final factor of a TSQR folds
__global__ __launch_bounds__(256) void fbr_expand_rows_kernel(int cols, int k, int Pra, const int *__restrict__ Eb, const int *__restrict__ Er,
                                                               const double *__restrict__ Ev, const double *__restrict__ Rred, double *__restrict__ dst,
                                                               int ldd)
{
    const int Pa = cols + k;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)Pra * Pa; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / Pa), j = (int)(e - (long)r * Pa);
        double acc = 0.0;
        for (int b = Eb[j]; b < Eb[j + 1]; b++) acc += Ev[b] * Rred[(long)r * Pra + Er[b]];
        dst[(long)r * ldd + j] = acc;
    }
}

extern "C" void fbr_model_destroy(fbr_model *m)
{
    if (m && m->pid != getpid()) return;  // a handle inherited through fork(): its device resources belong to the parent, nothing to free here
    delete m;  // ~fbr_model releases the device memory, streams and events
}

// the reduced model a fused Gram pass runs on (-1: the model itself)
static int pick_gram_reduction(const fbr_model *m, long S = -1)
{
    if (getenv("FBR_NO_LINK_MERGE")) return -1;
    // S >= 0: a call over S samples.  The reduced pass costs a second model's launches and two expansion kernels (~0.15 ms): small
    // robots on short batches are faster over all their columns (KUKA, 80 -> 59 columns, 50 k samples: 0.73 against 0.82 ... 1.08 ms)
    if (S >= 0) {
        const fbr_model *r = m->rdm[1] ? m->rdm[1].get() : m->rdm[0].get();
        if (r && (double)S * (m->hm.cols - r->hm.cols) * m->hm.cols < 1e9 && !getenv("FBR_REDUCE_ALWAYS")) return -1;
    }
    // (robots beyond the fused kernel's 60 rows take their Gram from a TSQR factor, gram_via_tsqr: every path of the merged model)
    if (m->rdm[1] && !getenv("FBR_NO_REGROUP") && (m->hm.rows + 3) / 4 * 4 <= 60) return 1;
    return m->rdm[0] ? 0 : -1;
}

extern "C" int fbr_model_link_merge_info(const fbr_model *m, int32_t *moving_links, int32_t *reduced_cols)
{
    if (!m) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    const int w = pick_gram_reduction(m);
    if (moving_links) *moving_links = w >= 0 ? m->rdm[w]->hm.L : m->hm.L;
    if (reduced_cols) *reduced_cols = w >= 0 ? m->rdm[w]->hm.cols : m->hm.cols;
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

// ------------------------------------------------------------------------------------------------
// state staging
// ------------------------------------------------------------------------------------------------
static int wait_ticket(fbr_model *m, int64_t ticket);

struct DevStates {
    long S = 0;
    const double *q = nullptr, *dq = nullptr, *ddq = nullptr, *bv = nullptr, *ba = nullptr, *rpy = nullptr, *sign = nullptr;
};

static int stage_one(fbr_model *m, DevBuf &buf, const double *src, size_t count, int mem, const double **dst)
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
static bool is_pinned_host(const void *p)
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
static int stage_states(fbr_model *m, const fbr_states *st, DevStates *d, bool need_vel = true, bool defer_host = false)
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

static long chunk_size(const fbr_model *m, long S)
{
    const size_t per = (size_t)m->hm.rec_size() * sizeof(double);
    long ch = (long)((size_t)(768u << 20) / per);
    if (ch < 1024) ch = 1024;
    if (const char *e = getenv("FBR_CHUNK_SAMPLES")) ch = std::max(1L, atol(e));  // tests: force the multi-chunk paths at small sizes
    return std::min(S, ch);
}

// beside_gram: the launch shares the CUs with the Gram kernel (producer stream of the fused pass): the register-capped instance
static int run_kin(fbr_model *m, const DevStates &d, long s0, long cs, hipStream_t st = nullptr, DevBuf *recbuf = nullptr, bool beside_gram = false)
{
    const FbrHostModel &hm = m->hm;
    if (!st) st = m->stream;
    if (!recbuf) recbuf = &m->rec;
    int rc = recbuf->ensure((size_t)cs * hm.rec_size() * sizeof(double));
    if (rc) return rc;
    const int threads = 256;
    const int blocks = (int)((cs + threads - 1) / threads);
    ProfScope ps(m, FBR_PROF_KIN, st);
    // (the instance that fits beside the Gram kernel's waves on the producer stream of the fused pass; the uncapped one everywhere else)
    // (the 96-VGPR instance spills 41 registers and, like the 128-VGPR one, fits ONE wave beside the two Gram waves of a SIMD: since the
    // column reductions the uncapped instance is the faster one there as well -- kin 5.3 instead of 6.2 ms per 1 M samples, step -1.5 %;
    // FBR_KIN_CAPPED=1 brings the capped one back)
    if (beside_gram && getenv("FBR_KIN_CAPPED"))
        hipLaunchKernelGGL(fbr_kin_kernel<FBR_KIN_WAVES>, dim3(blocks), dim3(threads), 0, st, m->dm, cs, d.q + s0 * hm.n,
                           d.dq + s0 * hm.n, d.ddq + s0 * hm.n, d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr,
                           d.rpy ? d.rpy + s0 * 3 : nullptr, recbuf->as<double>());
    else if (getenv("FBR_KIN_CAPPED"))
        hipLaunchKernelGGL(fbr_kin_kernel<FBR_KIN_WAVES>, dim3(blocks), dim3(threads), 0, st, m->dm, cs, d.q + s0 * hm.n,
                           d.dq + s0 * hm.n, d.ddq + s0 * hm.n, d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr,
                           d.rpy ? d.rpy + s0 * 3 : nullptr, recbuf->as<double>());
    else
        hipLaunchKernelGGL(fbr_kin_kernel<2>, dim3(blocks), dim3(threads), 0, st, m->dm, cs, d.q + s0 * hm.n,
                           d.dq + s0 * hm.n, d.ddq + s0 * hm.n, d.bv ? d.bv + s0 * 6 : nullptr, d.ba ? d.ba + s0 * 6 : nullptr,
                           d.rpy ? d.rpy + s0 * 3 : nullptr, recbuf->as<double>());
    HIPCHK(hipGetLastError());
    return FBR_OK;
}

static int finish_output(fbr_model *m, double *dev_src, double *user_dst, size_t count, int out_mem)
{
    if (out_mem == FBR_HOST)
        HIPCHK(hipMemcpyAsync(user_dst, dev_src, count * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    prof_collect(m);
    return FBR_OK;
}

// ------------------------------------------------------------------------------------------------
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
    const size_t lds = (size_t)hm.rec_size() * sizeof(double);
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int spb = std::max(1, std::min(16, 256 / std::max(1, hm.cols / 2)));  // samples side by side in one workgroup
    const size_t lds2 = lds * spb;
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        if ((rc = run_kin(m, d, s0, cs))) return rc;
        double *dst = (out_mem == FBR_HOST) ? m->out_tmp.as<double>() : Y_out + (size_t)s0 * per;
        const int blocks = (int)std::min<long>(cs, (long)m->num_cus * 8);
        {
            ProfScope ps(m, FBR_PROF_REGRESSOR);
            // even column count (and a 16-byte aligned output): paired columns, 16-byte stores
            if ((hm.cols & 1) == 0 && (((uintptr_t)dst) & 15) == 0)
                hipLaunchKernelGGL(fbr_regressor2_kernel, dim3((unsigned)std::min<long>((cs + spb - 1) / spb, (long)m->num_cus * 8)), dim3(256), lds2, m->stream, m->dm, cs, spb,
                                   m->rec.as<double>(), d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, hm.cols, (long)hm.rows, 1L, (const int *)nullptr, (const int *)nullptr);
            else
                hipLaunchKernelGGL(fbr_regressor_kernel, dim3(blocks), dim3(256), lds, m->stream, m->dm, cs, m->rec.as<double>(),
                                   d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, hm.cols, (long)hm.rows, 1L, (const int *)nullptr, (const int *)nullptr);
        }
        HIPCHK(hipGetLastError());
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

static int run_id(fbr_model *m, const fbr_states *st, const double *x, int nx, const double *vel_sign, int mode,
                  double *tau_out, int32_t out_mem)
{
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
        fbr_gram_build_best(h->prog, m->hm, k, getenv("FBR_GRAM_SHAPE"), !moments);  // "one" / "two": force a kernel shape
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
        std::vector<int> itemcol(256, -1);
        for (size_t i = 0; i < gp.items.size() && i < 256; i++) {
            const int off = gp.items[i].off;
            for (int t = 0; t < gp.NT; t++) {  // (a friction item's offset points at the image row of its joint)
                const int end = t + 1 < gp.NT ? gp.tiles[t + 1].off : gp.image_doubles;
                if (off >= gp.tiles[t].off && off < end) itemcol[i] = gp.tiles[t].col[(off - gp.tiles[t].off) % FBR_TILE];
            }
        }
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

extern "C" int fbr_gram_program_info(const fbr_model *mc, int32_t k, int32_t *num_tiles, int32_t *num_pairs,
                                     int64_t *mfma_per_sample, int32_t *num_parts)
{
    if (!mc) {
        set_err("null model");
        return FBR_E_INVALID;
    }
    fbr_model *m = const_cast<fbr_model *>(mc);
    if (int rc_enter = enter(m)) return rc_enter;
    if (const int wr = pick_gram_reduction(m); wr >= 0) m = m->rdm[wr].get();  // what fbr_gram_accumulate runs: the program of the reduced model
    GramHolder *h = nullptr;
    int rc = get_gram(m, k, &h, fbr_gram_rhs_moments(m->hm, k));  // (what fbr_gram_accumulate / fbr_gram_submit run)
    if (rc) return rc;
    if (num_tiles) *num_tiles = h->prog.NT;
    if (num_pairs) *num_pairs = (int32_t)h->prog.pairs.size();
    if (mfma_per_sample) *mfma_per_sample = h->prog.mfma_per_sample;
    if (num_parts) *num_parts = h->prog.T;
    return FBR_OK;
}

// ngroups > 1: the samples form ngroups consecutive groups of equal size, one Gram per group (G_out [ngroups][Pa][Pa])
// Which regressor rows carry a non-zero weight for at least one sample (device scan of w; all rows when there are no weights).
static int active_rows(fbr_model *m, const double *dw, long S, std::vector<char> *act)
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

static int drain_after_failed_submit(fbr_model *m);
static int tsqr_impl(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k, const double *w,
                     const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket);

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
    const bool h2d_chunked = st && st->mem == FBR_HOST && !getenv("FBR_NO_CHUNKED_H2D") && is_pinned_host(st->q) && is_pinned_host(st->dq) &&
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
    const bool moments = ngroups == 1 && fbr_gram_rhs_moments(hm, k) && !getenv("FBR_GRAM_TIMING");
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
    if (w && S > 0 && hm.fb > 0 && hm.rows > hm.fb && !getenv("FBR_GRAM_NO_MASK_SKIP")) {
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
    if (S > 0) {
        const int T = h->prog.T;
        const bool two_per_cu = h->prog.cfg == FBR_CFG_TWO_PER_CU;
        const int blocks_per_cu = (two_per_cu && h->lds_bytes <= 79 * 1024) ? 2 : 1;
        const int FBR_NPW = h->prog.cfg.npw();
        const bool timing = getenv("FBR_GRAM_TIMING") != nullptr;
        typedef void (*gram_fn)(DevGram, long, int, const double *, double *, unsigned long long *, int);
        const gram_fn gram_kernel = two_per_cu ? (timing ? fbr_gram_kernel<true, 5, 2> : fbr_gram_kernel<false, 5, 2>)
                                               : (timing ? fbr_gram_kernel<true, FBR_ONE_SEGW, FBR_ONE_NSEG> : fbr_gram_kernel<false, FBR_ONE_SEGW, FBR_ONE_NSEG>);
        HIPCHK(hipFuncSetAttribute((const void *)gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        HIPCHK(hipFuncSetAttribute((const void *)fbr_pack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->pack_lds_bytes));
        const size_t img_bytes = (size_t)h->prog.image_doubles * sizeof(double);
        long ch = chunk_size(m, S);
        ch = std::max(1L, std::min(ch, (long)((size_t)4 * 1024 * 1024 * 1024 / img_bytes)));
        if (ngroups == 1 && !getenv("FBR_CHUNK_SAMPLES")) {
            // a short batch (e.g. one rank's shard of a multi-GPU run) is still cut into several chunks, so that only a small first
            // chunk's producer work runs before the first Gram launch instead of half the batch's
            static const long min_chunks = getenv("FBR_MIN_CHUNKS") ? std::max(1L, atol(getenv("FBR_MIN_CHUNKS"))) : 4;  // measured: 125 k samples 2 / 4 / 8 / 16 chunks = 11.45 / 11.69 / 11.19 / 9.35 M samples/s
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
            static const long first_div = getenv("FBR_FIRST_CHUNK_DIV") ? std::max(1L, atol(getenv("FBR_FIRST_CHUNK_DIV"))) : 1;
            for (int g = 0; g < ngroups; g++) {
                long c0 = 0;
                if (g == 0 && first_div > 1) {
                    const long f = std::max(2L, (ch / first_div) & ~1L);
                    items.push_back({0, std::min(f, Sg), 0, 1});
                    c0 = std::min(f, Sg);
                }
                for (; c0 < Sg; c0 += ch) items.push_back({g * Sg + c0, std::min(ch, Sg - c0), g, 1});
            }
        }
        const long nchunks = (long)items.size();
        bool fresh_images = false;  // a tile-image buffer was (re)allocated and zeroed on the main stream in this call
        // workgroups per group of a launch: every resident workgroup slot is used (see the launch below)
        auto wpg_of = [&](long cs, int ng) {
            static const int oversub_env = getenv("FBR_GROUP_OVERSUB") ? std::max(1, atoi(getenv("FBR_GROUP_OVERSUB"))) : 0;
            const long spg_max = std::max(1L, cs / ng);
            const int rounds = ng > 1 ? (oversub_env ? oversub_env : (T > 1 ? 4 : 2)) : 1;
            int wpg = std::max(T, (rounds * m->num_cus * blocks_per_cu) / ng);
            if ((long)wpg > (long)T * spg_max) wpg = (int)((long)T * spg_max);
            return std::min(wpg, 0xffff);
        };
        // One reduction per call: when every chunk of a single-group call has the same launch shape, a workgroup carries its partial
        // sums from chunk to chunk (the accumulators start from the partial-sum buffer) and fbr_gram_reduce_kernel runs once, after
        // the last chunk -- 15 of the 16 reductions of a 1 M-sample WALK-MAN pass (72 us each, between two Gram launches) go away.
        bool carry_ok = ngroups == 1 && nchunks > 1 && !timing && !getenv("FBR_GRAM_NO_CARRY");
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
        hipStream_t side = getenv("FBR_GRAM_SERIAL") ? m->stream : m->side;
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
        static const int pack_wgs_per_cu = getenv("FBR_PACK_WGS_PER_CU") ? std::max(1, atoi(getenv("FBR_PACK_WGS_PER_CU"))) : 24;
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
                if (!m->copy && !getenv("FBR_H2D_ON_SIDE")) {
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
            int rc2 = run_kin(m, dc, o, cs, side, &m->rec2, true);
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
    if ((rc = m->red_w.ensure((size_t)Pra * Pa * sizeof(double)))) return rc;
    for (int g = 0; g < ngroups; g++) {
        hipLaunchKernelGGL(fbr_expand_rows_kernel, dim3(512), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which],
                           m->E_val[which], Gred + (size_t)g * Pra * Pra, m->red_w.as<double>(), Pa);
        hipLaunchKernelGGL(fbr_expand_gram_kernel, dim3(1024), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which],
                           m->E_val[which], m->red_w.as<double>(), G + (size_t)g * Pa * Pa, accumulate ? 1 : 0);
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
    const bool grouped_ok = st && (ngroups == 1 || (ngroups >= 1 && st->num_samples / ngroups >= 512 && !getenv("FBR_NO_GROUPED_REDUCTION")));
    const int which =
        (m && st && ngroups >= 1 && G_out && k >= 0 && k <= FBR_MAX_RHS && m->pid == getpid() && grouped_ok) ? pick_gram_reduction(m, (long)st->num_samples) : -1;
    if (which >= 0) {
        int rc = gram_via_red(m, which, st, rhs, k, w, G_out, out_mem, accumulate, ngroups, async_ticket);
        if (rc && m->stream) {
            const std::string msg = g_err;
            drain_after_failed_submit(m);
            set_err(msg);
        }
        return rc;
    }
    int rc = gram_impl_inner(m, st, rhs, k, w, G_out, out_mem, accumulate, ngroups, async_ticket);
    // a failed submission issues no ticket, and a blocking call that fails half way may have launched on the producer / copy streams:
    // nothing of either may stay in flight when the error is returned
    if (rc && m && m->pid == getpid() && m->stream) {
        const std::string msg = g_err;
        drain_after_failed_submit(m);
        set_err(msg);
    }
    return rc;
}

// Block until the submission with this ticket (and every earlier one) is complete; ticket < 0 or beyond the last one: everything.
static int wait_ticket(fbr_model *m, int64_t ticket)
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

extern "C" int fbr_wait(fbr_model *m, int64_t ticket)
{
    int rc = enter(m);
    if (rc) return rc;
    return wait_ticket(m, ticket < 0 ? m->next_ticket - 1 : ticket);
}

extern "C" int fbr_gram_grouped(fbr_model *m, const fbr_states *st, int32_t ngroups, const double *rhs, int32_t k, const double *w,
                                double *G_out, int32_t out_mem)
{
    return gram_impl(m, st, rhs, k, w, G_out, out_mem, 0, ngroups);
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
    if (S > 0) {
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
// TSQR (fbr_tsqr.h)
// ------------------------------------------------------------------------------------------------
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
static TsqrPlan tsqr_plan(const FbrHostModel &hm, const int32_t *cols, int32_t ncols, int k, long S)
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
    p.reorder = !getenv("FBR_TSQR_NO_REORDER") && n > 16 * FBR_TSQR_NARROW_MAX_TILES && S * (long)hm.rows >= 64L * n &&
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
static TsqrGroupPlan tsqr_group_plan(const FbrHostModel &hm, const int32_t *cols, int32_t ncols, int k, const std::vector<char> *active = nullptr)
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
    std::vector<std::vector<int>> grows(ngroups);
    auto on = [&](int r) { return !active || (*active)[r]; };  // rows switched off by the weights belong to no group
    for (int r = 0; r < hm.fb; r++)
        if (on(r)) grows[base_group].push_back(r);
    for (int d = 0; d < hm.n; d++)
        if (on(hm.fb + d)) grows[jgroup[d]].push_back(hm.fb + d);
    for (int r = 0; r < hm.rows; r++) gp.masked = gp.masked || !on(r);
    auto touches = [&](int r, int uc) {
        const FbrCol &cd = hm.coldesc[uc];
        if (cd.kind != 0) return cd.joint == r - hm.fb;
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
static bool tsqr_use_groups(const TsqrGroupPlan &gp, long S, const double *R_in_unused = nullptr)
{
    (void)R_in_unused;
    const char *e = getenv("FBR_TSQR_GROUP_MIN_SAMPLES");  // (tests force the path at small sizes)
    const long min_s = e ? atol(e) : 24000;  // measured on WALK-MAN, groups vs one factorisation: 16 k samples 16 vs 15.8 ms, 32 k 18.5 vs 21.4, 64 k 24 vs 32, 125 k 34 vs 52
    return (gp.groups.size() > 1 || (gp.masked && !gp.groups.empty())) && S >= min_s && !getenv("FBR_TSQR_NO_GROUPS");
}
static long tsqr_group_chunk_samples(const fbr_model *m, const TsqrGroupPlan &gp, long S)
{
    double per = 0.0;  // chunk bytes per sample over all groups
    long lcm = 1;
    for (const TsqrGroup &G : gp.groups) {
        FbrTsqrShape sh;
        if (fbr_tsqr_shape(G.Pa, m->num_cus, 1L << 40, &sh)) return -1;
        per += 8.0 * (double)G.rows.size() * sh.n;
        lcm = std::lcm(lcm, (long)sh.mb);
    }
    long ch = std::max(1L, (long)(4.0 * 1024 * 1024 * 1024 / per));
    ch = std::min(ch, chunk_size(m, S));
    if (ch > lcm) ch -= ch % lcm;  // whole blocks per slot in every group
    return ch;
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
    const long ch = tsqr_group_chunk_samples(m, gp, S);
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
    const int wsplit = getenv("FBR_TSQR_WRITER_SPLIT") ? std::max(1, atoi(getenv("FBR_TSQR_WRITER_SPLIT")))
                                                       : std::max(1, std::min(4, 256 / std::max(1, npairs + (hm.cols - 2 * npairs))));
    // (with fewer work items than half a workgroup -- the regrouped WALK-MAN: 92 pairs + 29 single columns -- the pair writer leaves
    // most threads idle behind twice the work per busy thread: 12.6 ms per 1 M samples with the entries split, 15.9 without, against
    // 11.8 ms of the one-column-per-thread writer)
    bool pairable = npairs > 0 && !getenv("FBR_TSQR_WRITER8") && (npairs + (hm.cols - 2 * npairs) >= 128 || getenv("FBR_TSQR_WRITER16"));
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
    // working factors and chunk buffers of the groups
    std::vector<FbrDevGroup> hg(G);
    bool skipzeros = false;
    long mrows = 0;  // rows the final factor folds: the main group's data rows and the other groups' factors
    for (int g = 0; g < G; g++) mrows += g == gp.main ? S * (long)gp.groups[g].rows.size() : gp.groups[g].Pa;
    auto work = [&](int g) -> FbrTsqrWork & { return g == gp.main ? m->tsqr : m->tsqr_groups[g]; };
    for (int g = 0; g < G; g++) {
        const TsqrGroup &Gg = gp.groups[g];
        FbrTsqrWork &wk = work(g);
        if ((rc = fbr_tsqr_begin(wk, m->stream, Gg.Pa, g == gp.main ? Rin_dev : nullptr, m->num_cus, g == gp.main ? mrows : S * (long)Gg.rows.size(),
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
        if (getenv("FBR_TSQR_PROLOGUE_NOMASK") || hipExtStreamCreateWithCUMask(&m->tsqr_pro_stream, (uint32_t)words, mask.data()) != hipSuccess) {
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
        if ((rc = fbr_tsqr_chunk_buffer(wk, std::min(ch, S) * (long)Gg.rows.size(), &A)) || (rc = fbr_tsqr_chunk_clean(wk, pst)))
            return tsqr_fail(rc, "tsqr group chunk");
        hg[g] = FbrDevGroup{A, wk.n, (int)Gg.sel.size()};
    }
    const char *dtab = nullptr;
    if ((rc = tsqr_upload_tables(m, par,
                                 {{tab.data(), tab.size() * sizeof(int)}, {ents[0].data(), ents[0].size() * sizeof(int)}, {ents[1].data(), ents[1].size() * sizeof(int)},
                                  {pents[0].data(), (pairable ? pents[0].size() : 0) * sizeof(int)}, {pents[1].data(), (pairable ? pents[1].size() : 0) * sizeof(int)},
                                  {hg.data(), G * sizeof(FbrDevGroup)}},
                                 {0, o_ent0, o_ent1, o_pent0, o_pent1, o_grp}, o_grp + G * sizeof(FbrDevGroup), pst, &dtab)))
        return rc;
    const int *t = (const int *)dtab;
    const FbrDevGroup *dgrp = (const FbrDevGroup *)(dtab + o_grp);
    const size_t lds = (size_t)((hm.rec_size() + 1) & ~1) * sizeof(double) + (size_t)hm.rows * sizeof(double *);
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_groups_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_groups2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // the kinematic records are produced for several chunks at a time: one lane per sample needs tens of thousands of waves in flight
    // to hide its latencies (1 M samples: 6.4 ms in one launch, 11 ms in twelve)
    const long kin_span = std::max(ch, std::min(S, (long)((size_t)(6ull << 30) / ((size_t)hm.rec_size() * sizeof(double))) / ch * ch));
    for (long s0 = 0; s0 < S; s0 += ch) {
        const long cs = std::min(ch, S - s0);
        const long k0 = s0 / kin_span * kin_span;
        hipStream_t cst = s0 == 0 ? pst : m->stream;  // the first chunk's kinematics and writer belong to the prologue
        if (s0 == k0 && (rc = run_kin(m, d, k0, std::min(kin_span, S - k0), cst))) return rc;
        const double *recs = m->rec.as<double>() + (size_t)(s0 - k0) * hm.rec_size();
        // structural zeros left of a row's first supported column tile are skipped when every block holds rows of one slot
        skipzeros = true;
        for (int g = 0; g < G; g++) skipzeros = skipzeros && cs % work(g).mb == 0;
        {
            ProfScope ps(m, FBR_PROF_REGRESSOR, cst);
            if (pairable)
                hipLaunchKernelGGL(fbr_regressor_groups2_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds, cst, m->dm, cs,
                                   recs, d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr,
                                   drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, k, dw ? dw + (size_t)s0 * hm.rows : nullptr, dgrp, G, t, t + hm.rows,
                                   t + o_ebeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_ent1 : o_ent0)),
                                   t + o_pbeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_pent1 : o_pent0)), npairs, hm.ninert, wsplit);
            else
                hipLaunchKernelGGL(fbr_regressor_groups_kernel, dim3((unsigned)std::min<long>(cs, (long)m->num_cus * 8)), dim3(256), lds, cst, m->dm, cs,
                                   recs, d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr,
                                   drhs ? drhs + (size_t)s0 * hm.rows * k : nullptr, k, dw ? dw + (size_t)s0 * hm.rows : nullptr, dgrp, G, t, t + hm.rows,
                                   t + o_ebeg[skipzeros ? 1 : 0], (const int *)(dtab + (skipzeros ? o_ent1 : o_ent0)));
        }
        HIPCHK(hipGetLastError());
        if (cst != m->stream) {  // the folds (main stream) wait for the prologue
            HIPCHK(hipEventRecord(m->ev_tsqr_pro, cst));
            HIPCHK(hipStreamWaitEvent(m->stream, m->ev_tsqr_pro, 0));
        }
        ProfScope ps(m, FBR_PROF_TSQR);
        for (int g = 0; g < G; g++) {
            const TsqrGroup &Gg = gp.groups[g];
            FbrTsqrRowOrder ro;
            ro.first_col = t + o_fc[g];
            ro.rows = (int)Gg.rows.size();
            ro.group = cs;
            if ((rc = fbr_tsqr_fold_chunk(work(g), m->stream, cs * (long)Gg.rows.size(), Gg.Pa, 0, nullptr, ro))) return tsqr_fail(rc, "tsqr group fold");
        }
    }
    bool l0_recorded = false;
    auto record_l0 = [&]() -> int {  // what a following submission's prologue waits for
        if (!l0_recorded) HIPCHK(hipEventRecord(m->ev_tsqr_l0, m->stream));
        l0_recorded = true;
        m->tsqr_l0_rec = true;
        return FBR_OK;
    };
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
                if (getenv("FBR_TSQR_SIDE_SAME_PRIORITY"))
                    HIPCHK(hipStreamCreateWithFlags(&m->tsqr_streams[i], hipStreamNonBlocking));
                else
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
    HIPCHK(hipEventRecord(m->tsqr_ev[NSIDE], m->stream));
    for (int i = 0; i < NSIDE; i++) HIPCHK(hipStreamWaitEvent(m->tsqr_streams[i], m->tsqr_ev[NSIDE], 0));
    // the longest trees first, one stream each as far as they go (a short tree queued behind the waist chain's tree was the last to finish)
    std::vector<int> side_order;
    for (int g = 0; g < G; g++)
        if (g != gp.main) side_order.push_back(g);
    std::stable_sort(side_order.begin(), side_order.end(), [&](int a, int b) { return gp.groups[a].Pa > gp.groups[b].Pa; });
    // (narrow factors of one shape -- the two arms, the two legs -- share their launches: fbr_tsqr_finish_narrow_batch)
    int nside = 0;
    std::vector<char> finished(G, 0);
    for (int g : side_order) {
        if (finished[g]) continue;
        FbrTsqrWork &wg = m->tsqr_groups[g];
        FbrTsqrWork *batch[FBR_TSQR_NARROW_BATCH];
        double *outs[FBR_TSQR_NARROW_BATCH];
        int nb = 0;
        if (wg.narrow && !getenv("FBR_TSQR_NO_NARROW_BATCH"))
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
    // rows of the embedded group factors, stacked: [sum of the groups' Pa][n] in the final factor's column order
    long erows = 0;
    for (int g : side_order) erows += gp.groups[g].Pa;
    const bool inside = gp.main >= 0 && !m->tsqr.narrow && erows > 0 && !getenv("FBR_TSQR_EMBED_AFTER");
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
    if ((rc = fbr_tsqr_begin(m->tsqr, m->stream, Pa, seed, m->num_cus, 1, m->tsqr_err))) return tsqr_fail(rc, "tsqr begin");
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
        overlap = m->waited_ticket < m->next_ticket - 1 && m->last_submit_kind == 1 && m->tsqr_l0_rec && !getenv("FBR_TSQR_NO_PROLOGUE_OVERLAP");
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
    const TsqrPlan plan = tsqr_plan(hm, cols, ncols, k, S);
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
        const TsqrGroupPlan gp = tsqr_group_plan(hm, cols, ncols, k, &act);
        if (hm.rows <= 255 && tsqr_use_groups(gp, S)) {  // (the writer's entries hold the regressor row in 8 bits)
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
    if ((rc = fbr_tsqr_begin(m->tsqr, m->stream, Pa, plan.reorder ? nullptr : Rin_dev, m->num_cus, S * (long)hm.rows, m->tsqr_err))) return tsqr_fail(rc, "tsqr begin");
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
        const size_t lds = (size_t)hm.rec_size() * sizeof(double);
        HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int spb = std::max(1, std::min(16, 256 / std::max(1, hm.cols / 2)));
        const size_t lds2 = lds * spb;
        HIPCHK(hipFuncSetAttribute((const void *)fbr_regressor2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        const bool direct = !cols && !dw;
        if (!direct && (rc = m->out_tmp.ensure((size_t)ch * per * sizeof(double)))) return rc;
        for (long s0 = 0; s0 < S; s0 += ch) {
            const long cs = std::min(ch, S - s0);
            if ((rc = run_kin(m, d, s0, cs))) return rc;
            const int blocks = (int)std::min<long>(cs, (long)m->num_cus * 8);
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
            {
                ProfScope ps(m, FBR_PROF_REGRESSOR);
                if ((hm.cols & 1) == 0)
                    hipLaunchKernelGGL(fbr_regressor2_kernel, dim3((unsigned)std::min<long>((cs + spb - 1) / spb, (long)m->num_cus * 8)), dim3(256), lds2,
                                       m->stream, m->dm, cs, spb, m->rec.as<double>(), d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, ldy, rs_s, rs_r, lp, skipfc);
                else
                    hipLaunchKernelGGL(fbr_regressor_kernel, dim3(blocks), dim3(256), lds, m->stream, m->dm, cs, m->rec.as<double>(),
                                       d.dq + s0 * hm.n, d.sign ? d.sign + s0 * hm.n : nullptr, dst, ldy, rs_s, rs_r, lp, skipfc);
            }
            HIPCHK(hipGetLastError());
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
            if ((rc = fbr_tsqr_begin(m->tsqr, m->stream, Pa, nullptr, m->num_cus, 1, m->tsqr_err)) ||
                (rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, Pa, Pa, m->tsqr_rtmp.as<double>(), 0, nullptr, nullptr, Pa, dinv)) ||
                (rc = fbr_tsqr_finish_async(m->tsqr, m->stream, R)))
                return tsqr_fail(rc, "tsqr column order");
        }
    }
    return done();
}

// A submission that fails after work was enqueued has no ticket its caller could wait on: everything in flight is drained before the
// error is returned (the same for the Gram pass, gram_impl below), so that the inputs may be freed and later calls start from a quiet device.
static int drain_after_failed_submit(fbr_model *m)
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

// the reduced model a factorisation of every column runs on (-1: the model itself); the regrouped model factorises by row groups only
static int pick_tsqr_reduction(fbr_model *m, long S)
{
    if (getenv("FBR_NO_LINK_MERGE")) return -1;
    if (m->rdm[1] && !getenv("FBR_NO_REGROUP") && !getenv("FBR_TSQR_NO_GROUPS")) {
        if (m->rd_grouped < 0) {
            const TsqrGroupPlan gp = tsqr_group_plan(m->rdm[1]->hm, nullptr, 0, 0);
            m->rd_grouped = gp.groups.size() > 1 && m->rdm[1]->hm.rows <= 255;
        }
        const char *e = getenv("FBR_TSQR_GROUP_MIN_SAMPLES");
        if (m->rd_grouped && S >= (e ? atol(e) : 24000)) return 1;
    }
    return m->rdm[0] ? 0 : -1;
}

// The factor through the link-merged model (build_reduction): R_red over the moving bodies' columns, then R = qr([R_in ; R_red E]) --
// the Pra dense rows R_red E become working factor 1 beside R_in (or zero) in working factor 0, and ONE level of the merge tree,
// pipelined across workgroups, folds them (wide factors; narrow ones fold them as ordinary rows).
static int tsqr_via_red(fbr_model *m, int which, const fbr_states *st, const double *rhs, int32_t k, const double *w, const double *R_in, double *R_out,
                        int32_t out_mem, int64_t *async_ticket)
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
    const int par = (int)(m->next_ticket & 1), Pa = m->hm.cols + k, Pra = r->hm.cols + k;
    const size_t cnt = (size_t)Pa * Pa;
    if ((rc = m->red_out[par].ensure((size_t)Pra * Pra * sizeof(double)))) return rc;
    double *Rred = m->red_out[par].as<double>();
    int64_t tr = -1;
    if ((rc = tsqr_impl(r, st, nullptr, 0, rhs, k, w, nullptr, Rred, FBR_DEVICE, async ? &tr : nullptr))) return rc;
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
    if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh)) return fail(-4, "tsqr shape");
    FbrTsqrWork &wk = m->tsqr;
    {
        ProfScope ps(m, FBR_PROF_TREE);
        bool done_wide = false;
        if (!sh.narrow && !getenv("FBR_TSQR_TREE_ONE_WG") && !getenv("FBR_LINK_MERGE_ROWS")) {
            if ((rc = fbr_tsqr_begin(wk, m->stream, Pa, Rin_dev, m->num_cus, 2L * sh.mb, m->tsqr_err))) return fail(rc, "tsqr begin");
            if (wk.NW == 2) {
                hipLaunchKernelGGL(fbr_expand_rows_kernel, dim3(512), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which], m->E_val[which], Rred,
                                   wk.Rw + (size_t)wk.n * wk.ld, wk.ld);
                HIPCHK(hipGetLastError());
                if ((rc = fbr_tsqr_tree_levels(wk, m->stream, 1, 2, Pra)) || (rc = fbr_tsqr_copy_out(wk, m->stream, R))) return fail(rc, "tsqr expansion");
                done_wide = true;
            }
        }
        if (!done_wide) {  // narrow factors: the expanded rows as ordinary data rows of a one-workgroup factorisation
            if ((rc = m->tsqr_embed.ensure((size_t)Pra * Pa * sizeof(double)))) return rc;
            hipLaunchKernelGGL(fbr_expand_rows_kernel, dim3(512), dim3(256), 0, m->stream, m->hm.cols, k, Pra, m->E_beg[which], m->E_row[which], m->E_val[which], Rred,
                               m->tsqr_embed.as<double>(), Pa);
            HIPCHK(hipGetLastError());
            if ((rc = fbr_tsqr_begin(wk, m->stream, Pa, Rin_dev, m->num_cus, 1, m->tsqr_err)) ||
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

static int tsqr_impl(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
                     const double *w, const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket)
{
    int which = (m && st && !cols && R_out && k >= 0 && k <= FBR_MAX_RHS && m->pid == getpid()) ? pick_tsqr_reduction(m, (long)st->num_samples) : -1;
    while (which >= 0) {
        int rc = tsqr_via_red(m, which, st, rhs, k, w, R_in, R_out, out_mem, async_ticket);
        if (rc == FBR_E_NOT_GROUPED && which == 1) {  // (row weights left the regrouped model without row groups: nothing was enqueued)
            which = m->rdm[0] ? 0 : -1;
            continue;
        }
        if (rc && m->stream) {
            const std::string msg = g_err;
            drain_after_failed_submit(m);
            set_err(msg);
        }
        return rc;
    }
    int rc = tsqr_impl_inner(m, st, cols, ncols, rhs, k, w, R_in, R_out, out_mem, async_ticket);
    if (rc && m && m->pid == getpid() && m->stream) {  // (blocking calls too: the groups' trees run on side streams)
        const std::string msg = g_err;
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
    if (const int which = cols ? -1 : pick_tsqr_reduction(m, (long)num_samples); which >= 0) {
        // what fbr_tsqr runs on a link-merged model: the factorisation of the reduced robot, then the Pra expanded rows folded into the
        // final factor by one tree level; block_rows / n_padded describe the FINAL factor (what fbr_tsqr_merge works on)
        int64_t l0 = 0, tr = 0;
        if (int rc = fbr_tsqr_work_info(m->rdm[which].get(), nullptr, 0, k, num_samples, &l0, &tr, nullptr, nullptr)) return rc;
        FbrTsqrShape sh;
        const int Pa = m->hm.cols + k, Pra = m->rdm[which]->hm.cols + k;
        if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh)) {
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
    const TsqrPlan plan = tsqr_plan(hm, cols, ncols, k, (long)num_samples);
    const int Psel = plan.Psel, Pa = plan.Pa;
    (void)Psel;
    {
        // tree-structured path (tsqr_groups_impl): level 0 of every group over its own chunks, the groups' trees, and the final factor
        // that folds the embedded group factors (dense rows) and runs its own tree
        const TsqrGroupPlan gp = tsqr_group_plan(hm, cols, ncols, k);
        if (hm.rows <= 255 && tsqr_use_groups(gp, (long)num_samples)) {
            const long ch = tsqr_group_chunk_samples(m, gp, (long)num_samples);
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
                if (ch < 0 || fbr_tsqr_shape(G.Pa, m->num_cus, g == gp.main ? mrows : num_samples * ns, &sh)) {
                    set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
                    return FBR_E_UNSUPPORTED;
                }
                for (long s0 = 0; s0 < num_samples; s0 += ch) {
                    const long cs = std::min(ch, (long)num_samples - s0), M = cs * ns, Mpad = (M + 15) & ~15L;
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
            if (fbr_tsqr_shape(Pa, m->num_cus, mrows, &sh)) {
                set_err(std::string("tsqr shape: ") + fbr_tsqr_error());
                return FBR_E_UNSUPPORTED;
            }
            if (gp.main >= 0) tr += tree();  // the dense group's own tree
            const int main_mb = sh.mb;
            long erows = 0;
            for (int g = 0; g < (int)gp.groups.size(); g++)
                if (g != gp.main) erows += gp.groups[g].Pa;
            if (gp.main >= 0 && !sh.narrow && erows > 0 && !getenv("FBR_TSQR_EMBED_AFTER")) {
                // the stacked embedded group factors are folded into the factors alive inside the main tree (tsqr_groups_impl)
                for (long r0 = 0; r0 < ((erows + 15) & ~15L); r0 += sh.mb) tr += fold_mfma(0);
            } else {
                if (fbr_tsqr_shape(Pa, m->num_cus, 1, &sh)) {  // (that factorisation is begun for a handful of rows: fbr_tsqr_begin(.., 1))
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
    if (fbr_tsqr_shape(Pa, m->num_cus, num_samples * (long)hm.rows, &sh)) {
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
        if (fbr_tsqr_shape(Pa, m->num_cus, 1, &s1)) return FBR_E_UNSUPPORTED;
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
    if (!fbr_tsqr_shape(n, m->num_cus, 1, &sh) && !sh.narrow && !getenv("FBR_TSQR_TREE_ONE_WG")) {
        FbrTsqrWork &wk = m->tsqr;
        if ((rc = fbr_tsqr_begin(wk, m->stream, n, da, m->num_cus, 2L * sh.mb))) {  // rows for two blocks: two working factors
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
    if ((rc = fbr_tsqr_begin(m->tsqr, m->stream, n, da, m->num_cus, 1)) ||
        (rc = fbr_tsqr_fold_rows(m->tsqr, m->stream, n, n, db, 0, nullptr, nullptr, 0, nullptr, tri)) ||
        (rc = fbr_tsqr_finish(m->tsqr, m->stream, R))) {
        set_err(std::string("tsqr merge: ") + fbr_tsqr_error());
        return rc == -4 ? FBR_E_UNSUPPORTED : (rc == -3 ? FBR_E_HIP : FBR_E_INVALID);
    }
    return finish_output(m, R, R_out, cnt, mem);
}

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
    {  // M^LB of z' = M z + g x (y = z_0 + b_0 x): M[i][0] = -a[i+1], M[i][i+1] = 1
        std::vector<double> M((size_t)p * p, 0.0), R((size_t)p * p, 0.0), Tm((size_t)p * p);
        for (int i = 0; i < p; i++) {
            M[(size_t)i * p] = -f.a[i + 1];
            if (i + 1 < p) M[(size_t)i * p + i + 1] += 1.0;
            R[(size_t)i * p + i] = 1.0;
        }
        auto mul = [&](std::vector<double> &A, const std::vector<double> &B) {  // A = A * B
            for (int i = 0; i < p; i++)
                for (int j = 0; j < p; j++) {
                    double acc = 0.0;
                    for (int k = 0; k < p; k++) acc += A[(size_t)i * p + k] * B[(size_t)k * p + j];
                    Tm[(size_t)i * p + j] = acc;
                }
            A = Tm;
        };
        for (long e = FBR_SIG_LB; e > 0; e >>= 1) {
            if (e & 1) mul(R, M);
            std::vector<double> M2 = M;
            mul(M2, M);
            M = M2;
        }
        for (int i = 0; i < p * p; i++) f.Mp[i] = R[i];
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
