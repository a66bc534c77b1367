// fbr_internal.h -- what the translation units of libfbr share: the model handle, device buffers, error plumbing, staging and the
// helpers one unit offers the others.  Not part of the C-ABI (include/fbr.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/fbr.h"
#include "fbr_options.h"
#include "fbr_kernels.h"
#include "fbr_kinid.h"
#include "fbr_gram64.h"
#include "fbr_tsqr_work.h"

extern thread_local std::string g_fbr_err;
static inline void set_err(const std::string &s) { g_fbr_err = s; }

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

template <class T> static inline int upload(std::vector<DevBuf> &pool, const std::vector<T> &v, const T **out)
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
    // ---- the pass over sample-contiguous images (fbr_gram64.h, option gram_lane); built on first use, -1: the model is outside it
    int g64_state = 0;        // 0 not looked at, 1 ready, -1 not applicable
    FbrGram64 g64;
    FbrGram64Producer g64p;   // producer tables (parts, destination words relative to an image buffer)
    const int *d64_slab = nullptr, *d64_levb = nullptr, *d64_pieces = nullptr, *d64_wmeta = nullptr, *d64_lcol = nullptr,
              *d64_steps = nullptr, *d64_slot_tiles = nullptr, *d64_tilecol = nullptr, *d64_stagelev = nullptr;
    DevBuf img64[2], dst64[2], mom64, scr64;
    long img64_blocks = 0;    // capacity of the image buffers in blocks of 64 samples
    std::map<int, const int *> d64_wb;  // [0, workgroups] tables of the reduction, one per grid a call has used
    struct Deal { const int2 *tab; const int *begin; };
    std::map<int, Deal> deals;  // workgroups per sample group -> device tables of fbr_gram_deal (at most one per count)
};

struct fbr_model {
    FbrOptions opt;  // fbr_model_set_option; the reduced models rdm[] share their parent's values
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
    FbrKinIdProgram kinid;    // program of the fused kinematics + torque kernel (fbr_kinid.h); nsteps == 0: not available for this tree
    const int *kinid_steps = nullptr, *kinid_endflush = nullptr;
    DevBuf kinid_scratch;     // branch-point records of the waves in flight
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
    const double *E_val[2] = {nullptr, nullptr};
    std::vector<int> hE_beg[2], hE_row[2];  // the same CSC on the host: x_red = E x of the streaming prediction / inverse dynamics (run_id)
    std::vector<double> hE_val[2];  // of the full layout = sum of E_val[e] x (reduced column E_row[e]), e in [E_beg[j], E_beg[j+1])
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
        // the reduced models run on THIS model's stream (r->stream = m->stream, possibly own_stream): they go first, while every stream
        // they synchronise in their own destructors still exists
        for (auto &r : m->rdm) r.reset();
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
static inline void prof_collect(fbr_model *m)
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

// ---- fbr_api.hip -------------------------------------------------------------------------------------------------------------
struct DevStates {
    long S = 0;
    const double *q = nullptr, *dq = nullptr, *ddq = nullptr, *bv = nullptr, *ba = nullptr, *rpy = nullptr, *sign = nullptr;
};
#define FBR_E_NOT_GROUPED (-1000)  // internal: a model with column masks was asked for a factorisation its row-group path does not take
int enter(fbr_model *m);
int enter_blocking(fbr_model *m);  // entry of a blocking call that does not go through stage_states: every submission before it has completed
int wait_ticket(fbr_model *m, int64_t ticket);
int drain_after_failed_submit(fbr_model *m);
int stage_one(fbr_model *m, DevBuf &buf, const double *src, size_t count, int mem, const double **dst);
bool is_pinned_host(const void *p);
int stage_states(fbr_model *m, const fbr_states *st, DevStates *d, bool need_vel = true, bool defer_host = false);
long chunk_size(const fbr_model *m, long S);
int run_kin(fbr_model *m, const DevStates &d, long s0, long cs, hipStream_t st = nullptr, DevBuf *recbuf = nullptr);
int finish_output(fbr_model *m, double *dev_src, double *user_dst, size_t count, int out_mem);
int active_rows(fbr_model *m, const double *dw, long S, std::vector<char> *act);
int pick_gram_reduction(const fbr_model *m, long S = -1);
// the materialising regressor kernel of samples [s0, s0 + cs) into dst (leading dimension ldy, row strides rs_s / rs_r, optional link
// positions / skipped leading zeros of the TSQR chunk layout): fbr_regressor_batch and the single-factorisation TSQR path
int launch_regressor(fbr_model *m, const DevStates &d, long s0, long cs, double *dst, int ldy, long rs_s, long rs_r, const int *linkpos, const int *skipfc);
// ---- fbr_gram_api.hip --------------------------------------------------------------------------------------------------------
// dst[r][j] (leading dimension ldd) = (R_red E)[r][j] on the model's stream (Gram expansion, TSQR expansion)
int launch_expand_rows(fbr_model *m, int which, int k, int Pra, const double *Rred, double *dst, int ldd, const int *colmap = nullptr, int Pout = 0);
// ---- fbr_tsqr_api.hip --------------------------------------------------------------------------------------------------------
int tsqr_impl(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k, const double *w,
              const double *R_in, double *R_out, int32_t out_mem, int64_t *async_ticket);
