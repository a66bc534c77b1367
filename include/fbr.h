/*
 * fbr.h -- C-ABI of libfbr (FloBaRoID regressor hot path on MI355X / gfx950).
 *
 * The reference has no FFI of its own for this path: its per-sample arithmetic is reached through
 * iDynTree's SWIG bindings and its reductions through NumPy/SciPy.  Every entry point below names
 * the reference call site(s) it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; int return: 0 = OK, negative = error (fbr_last_error()).
 *   - All numeric data is IEEE fp64, C-contiguous, row-major.
 *   - Every data pointer carries a memory-space flag: FBR_HOST (pageable/pinned host memory; the call
 *     stages it through the device) or FBR_DEVICE (HIP device memory on the model's device).
 *   - Calls are blocking: results are complete (and the model's stream synchronised) on return.
 *   - A model handle may be used from one host thread at a time; handles are per-process and must be
 *     created after fork() (the reference's multiprocessing users build one Model per worker,
 *     excitation/analyticalGradient.py:188-210).
 *   - There is NO CPU fallback: without a usable HIP device every compute call fails with FBR_E_NODEVICE.
 */
#ifndef FBR_H
#define FBR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBR_OK 0
#define FBR_E_INVALID (-1)   /* bad argument */
#define FBR_E_NODEVICE (-2)  /* no HIP device / HIP runtime failure at init */
#define FBR_E_HIP (-3)       /* HIP runtime error during the call */
#define FBR_E_UNSUPPORTED (-4)
#define FBR_E_FORK (-5)      /* handle / HIP runtime inherited through fork(): create the model in the process that uses it */

#define FBR_HOST 0
#define FBR_DEVICE 1

typedef struct fbr_model fbr_model; /* opaque */

/*
 * Kinematic tree + identified-column layout.  Replaces what the reference pulls out of
 * iDynTree::ModelLoader / Model (identification/model.py:60-68,112-131) and the layout logic of
 * Model.__init__ (model.py:134-168).  Links and DOFs are indexed as the caller serialises them (any order: parents
 * need not precede children); flobaroid_amd/topology.py hands over iDynTree's traversal order, the one the reference's
 * linkNames / jointNames follow (model.py:86-94,121-127).  Columns of the regressor are link-major, 10 per link
 * [m, m*cx, m*cy, m*cz, Ixx, Ixy, Ixz, Iyy, Iyz, Izz] (model.py:220-231), followed by the friction
 * blocks Fc | Fv (or Fv+,Fv-) | off | Fs (model.py:459-503).
 */
typedef struct {
    int32_t num_links;
    int32_t num_dofs;
    const int32_t *parent;    /* [L] parent link, -1 for the base */
    const int32_t *dof_index; /* [L] DOF of the joint to the parent, -1 when fixed / base */
    const double *rest_R;     /* [L][9] child orientation in the parent frame at q = 0 */
    const double *rest_p;     /* [L][3] child origin in the parent frame */
    const double *axis;       /* [L][3] unit joint axis in the child frame (rotation axis / sliding direction) */
    int32_t floating_base;    /* opt['floatingBase']: rows per sample = num_dofs + 6 */
    double gravity[3];        /* reference: (0, 0, -9.81), model.py:182-187 */
    int32_t friction;           /* opt['identifyFrictionSimultaneously'] */
    int32_t friction_symmetric; /* opt['identifySymmetricVelFriction'] */
    int32_t gravity_only;       /* opt['identifyGravityParamsOnly'] (keeps 4 columns per link) */
    double stribeck_velocity;   /* opt['stribeckVelocity'] (> 0 adds the Fs block) */
    const int32_t *joint_type;  /* [L] type of the joint to the parent for links with a DOF: 1 revolute / continuous, 2 prismatic (other
                                   entries ignored); NULL: every DOF is revolute.  iDynTree's loader takes any URDF (model.py:60-67) */
} fbr_topology;

/*
 * A batch of trajectory samples = what Model.computeRegressors reads per sample_index
 * (model.py:374-394, 425-439, 462).  All arrays share the same memory space.
 */
typedef struct {
    int64_t num_samples;
    int32_t mem;            /* FBR_HOST or FBR_DEVICE */
    const double *q;        /* [S][n] positions */
    const double *dq;       /* [S][n] velocities */
    const double *ddq;      /* [S][n] accelerations */
    const double *base_vel; /* [S][6] base twist [lin; ang], mixed representation (floating base) */
    const double *base_acc; /* [S][6] its time derivative */
    const double *base_rpy; /* [S][3] rpy with world_T_base = Transform(RPY(rpy),0).inverse() */
    const double *sign;     /* [S][n] Coulomb sign term (helpers.getFrictionSignSeries); NULL unless friction */
} fbr_states;

/* ---- library / device ------------------------------------------------------------------------- */
/* Version of THIS header.  fbr_version() returns the value the loaded library was built with: a caller compares the two before its first
 * call (flobaroid_amd/_lib.py load_library refuses a mismatch), because the C-ABI has grown in place -- 101: fbr_topology.joint_type,
 * the num_samples argument of fbr_gram_program_info / fbr_model_link_merge_info, option "fused_id"; 102: fbr_gram_lane_info, options
 * "gram_lane" / "gram_force_tiles". */
#define FBR_VERSION 102
int fbr_version(void);
int fbr_device_count(void);        /* number of visible HIP devices (0 if none / no runtime) */
const char *fbr_last_error(void);  /* thread-local message of the last failing call */

/* ---- model ------------------------------------------------------------------------------------ */
/* Builds device tables for `topo` on HIP device `device`.  Replaces ModelLoader.loadModelFromFile +
   KinDynComputations.loadRobotModel (model.py:60-68). */
int fbr_model_create(const fbr_topology *topo, int device, fbr_model **out);
void fbr_model_destroy(fbr_model *m);
/* rows per sample (N_OUT, model.py:102-105) and identified columns (num_identified_params, model.py:138-168) */
int fbr_model_dims(const fbr_model *m, int32_t *rows_per_sample, int32_t *num_cols);
/* Run subsequent calls on this hipStream_t (e.g. torch's current stream); NULL = the model's own stream. */
int fbr_model_set_stream(fbr_model *m, void *hip_stream);

/* ---- per-sample kernels ----------------------------------------------------------------------- */
/*
 * Stacked standard regressor, Y_out [S*rows][cols] -- the reference's regressor_stack
 * (model.py:349-354, 435-523: setRobotState + inverseDynamicsInertialParametersRegressor + friction
 * column blocks + gravity-only column deletion, one Python iteration per sample).
 */
int fbr_regressor_batch(fbr_model *m, const fbr_states *st, double *Y_out, int32_t out_mem);

/*
 * Generalized torques [base wrench(6); joint torques] per sample, tau_out [S][rows], from the full
 * standard parameter vector x_std (host pointer; 10 per link + friction slots laid out as in
 * model.py:134-168) -- Model.simulateDynamicsIDynTree (model.py:239-331): KinDynComputations.
 * inverseDynamics plus the friction model sign*Fc + Fv*dq + off (+ Stribeck with vel_sign, may be NULL).
 */
int fbr_inverse_dynamics_batch(fbr_model *m, const fbr_states *st, const double *x_std, int32_t num_x,
                               const double *vel_sign, double *tau_out, int32_t out_mem);

/*
 * tau_out [S][rows] = Y_s . x for an identified-parameter vector x (host, length cols) WITHOUT
 * materialising Y -- Identification.estimateRegressorTorques' np.dot(YStd, x) (identifier.py:135-141).
 */
int fbr_predict(fbr_model *m, const fbr_states *st, const double *x, double *tau_out, int32_t out_mem);

/*
 * out [S][rows] = J_frame^T w per sample for a frame rigidly attached to `link` at (frame_R, frame_p)
 * (host, 9 and 3 doubles); wrench [S][6] in st->mem.  getFrameFreeFloatingJacobian + J^T w
 * (model.py:542-555).  Uses st->q and st->base_rpy only.
 */
int fbr_contact_torques(fbr_model *m, const fbr_states *st, int32_t link, const double *frame_R,
                        const double *frame_p, const double *wrench, double *out, int32_t out_mem);

/* ---- fused reductions (Y is never materialised) ----------------------------------------------- */
/*
 * G_out [(cols+k)][(cols+k)] (+)= [Y | rhs]^T diag(w)^2 [Y | rhs], the raw (un-normalised) Gram of the
 * stacked regressor augmented with k right-hand-side columns rhs [S*rows][k] (tau, contact forces, ...),
 * row weights w [S*rows] optional (NULL = 1; a 0/1 mask selects e.g. the base-wrench rows of
 * identifier.py:617-681).  accumulate != 0 adds to the existing content of G_out.
 * Serves: R += A^T A of getRandomRegressor (model.py:803-806), YBase^T YBase = G[ic,ic]
 * (identifier.py:361, trajectoryOptimizer.py:263), YBase^T tau, and the OLS normal equations.
 * rhs / w live in st->mem; G_out in out_mem.
 */
int fbr_gram_accumulate(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w,
                        double *G_out, int32_t out_mem, int32_t accumulate);

/*
 * Asynchronous form of fbr_gram_accumulate for callers that keep everything in HBM (states, rhs, w and G_out device resident): the
 * pass is enqueued and the call returns; *ticket identifies it.  Up to TWO submissions are in flight (a third one first waits for
 * the oldest); the kinematics / packing of a submission's first chunk then run beside the last Gram launches of the one before --
 * the only producer work of a pass that nothing else hides (a sample loop split over several calls, model.py:370, or one rank's short
 * shard per step).  G_out must not be read, and the inputs not rewritten, before fbr_wait(m, ticket) has returned.  Every blocking
 * entry point of the same model first waits for all submissions.  fbr_wait: ticket < 0 waits for everything submitted so far.
 */
int fbr_gram_submit(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w, double *G_out,
                    int32_t accumulate, int64_t *ticket);
int fbr_wait(fbr_model *m, int64_t ticket);

/*
 * The same reduction for ngroups consecutive, equally sized groups of samples in ONE pass: G_out [ngroups][(cols+k)][(cols+k)],
 * group g = samples [g*S/ngroups, (g+1)*S/ngroups) (num_samples must be a multiple of ngroups).  Serves the
 * trajectory optimiser's inner loop (excitation/trajectoryOptimizer.py:248-272: YBase^T YBase of every candidate
 * trajectory) and finite-difference sweeps, where one Gram per small trajectory would be launch-bound.
 */
int fbr_gram_grouped(fbr_model *m, const fbr_states *st, int32_t ngroups, const double *rhs, int32_t k, const double *w,
                     double *G_out, int32_t out_mem);

/*
 * Finite-difference sweep of a weighted regressor score (the per-sample worker of the analytical trajectory gradient,
 * excitation/analyticalGradient.py:92-185): out [num_samples][1 + 3 n] with
 *   out[s][0]        = sum_{r,c} W_s[r][c] * Y_s[r][c]                     (baseline state of sample s)
 *   out[s][1 + d]    = the same sum for the state with q_d   + eps,
 *   out[s][1+n + d]  =                                    dq_d  + eps,
 *   out[s][1+2n + d] =                                    ddq_d + eps      (d = 0..n-1),
 * W [num_samples*rows][cols] the weight blocks (st->mem).  (out[s][1+j] - out[s][0]) / eps are the reference's
 * sens_q / sens_dq / sens_ddq; the 1 + 3 n regressor blocks per sample are evaluated on the fly, never stored.
 */
int fbr_fd_scores(fbr_model *m, const fbr_states *st, const double *W, double eps, double *out, int32_t out_mem);

/*
 * Joint states of ncand candidate trajectories, T samples each, generated ON THE DEVICE from their Fourier coefficients -- the
 * parametrisation the trajectory optimiser searches over (excitation/trajectoryGenerator.py: OscillationGenerator 411-460,
 * BoundedOscillationGenerator 462-560; evaluated per candidate by computeTrajectoryDynamics 83-128 before every objective call,
 * trajectoryOptimizer.py:240-272).  q / dq / ddq [ncand * T][n] (out_mem space) are laid out like fbr_states of ncand groups, i.e. ready
 * for fbr_gram_grouped: a candidate crosses PCIe as 2 n nharm + n + 1 coefficients instead of 3 n T samples.
 *   wf [ncand], a / b [ncand][n][nharm] (harmonics beyond a joint's own nf: 0), q_offset [ncand][n], q_range [ncand][n] or NULL: host arrays.
 *   q_range == NULL (classic, Swevers 1997):  dq = sum_l a_l cos(wf l t) + b_l sin(wf l t), q its integral + q_offset (= nf q0), ddq its derivative
 *   q_range != NULL (bounded):                 q = q_offset + q_range tanh(sum_l a_l sin(wf l t) + b_l cos(wf l t)) (q_offset = q_center), dq, ddq by the chain rule
 * t = sample index / freq (radians; useDeg of the reference converts on the host before / after).
 */
int fbr_fourier_states(fbr_model *m, int32_t ncand, int64_t T, int32_t nharm, double freq, const double *wf, const double *a, const double *b,
                       const double *q_offset, const double *q_range, double *q, double *dq, double *ddq, int32_t out_mem);

/*
 * R_out [(cols+k)][(cols+k)] upper triangular with R^T R = [Y|rhs]^T [Y|rhs], by blocked Householder
 * TSQR over sample blocks (no Gram squaring of the condition number; on branched robots the rows are grouped along
 * the kinematic tree and every group is factorised over the columns it can touch -- same R).  If R_in != NULL it is an
 * existing triangular factor (same shape, out_mem space) that is folded in first (streaming / tree
 * reduction across calls and ranks).  Serves la.qr(YBase) (sdp.py:470), sla.qr(YStd, pivoting) column
 * norms / pivots (model.py:841) and lstsq (identifier.py:712) through the small host problems.
 */
int fbr_tsqr(fbr_model *m, const fbr_states *st, const double *rhs, int32_t k, const double *w,
             const double *R_in, double *R_out, int32_t out_mem);

/*
 * Same as fbr_tsqr for a COLUMN SUBSET of the regressor: R_out [(ncols+k)][(ncols+k)] with
 * R^T R = [Y[:, cols] | rhs]^T [Y[:, cols] | rhs].  cols (host, int32, ncols entries, any order, no repeats) are
 * typically Model.independent_cols, which makes this la.qr(YBase) of sdp.py:470 / the lstsq of identifier.py:712
 * at (nb+k)^2 instead of (P+k)^2 cost.  R_in, if given, is a factor of the same subset.
 */
int fbr_tsqr_cols(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
                  const double *w, const double *R_in, double *R_out, int32_t out_mem);

/*
 * Asynchronous form of fbr_tsqr / fbr_tsqr_cols (cols == NULL: every identified column) for callers that keep everything in HBM: the
 * factorisation is enqueued and the call returns; *ticket identifies it, fbr_wait(m, ticket) completes it (tickets are shared with
 * fbr_gram_submit: at most TWO submissions of either kind are in flight, a third one first waits for the oldest).  When the submission
 * before is a TSQR as well, the kinematics and the first chunk's regressor writer of this one run beside its merge trees -- the part of
 * a call that executes no MFMA and that the (latency-bound, few-workgroup) trees leave the CUs free for: a streamed sequence of calls
 * (la.qr of a regressor that arrives in pieces, sdp.py:470-487 over several measurement files; one rank's shard per step) pays the
 * trees once.  R_out must not be read, and the inputs / R_in not rewritten, before fbr_wait has returned; a pipeline error of the
 * kernels is reported by fbr_wait.  Row weights w are scanned for switched-off rows on the host: with w != NULL the call first waits
 * for the stream (correct, not overlapped).
 */
int fbr_tsqr_submit(fbr_model *m, const fbr_states *st, const int32_t *cols, int32_t ncols, const double *rhs, int32_t k,
                    const double *w, const double *R_in, double *R_out, int64_t *ticket);

/* R_out = R factor of [R_a; R_b] (both n x n upper triangular, row-major) -- one node of the TSQR tree. */
int fbr_tsqr_merge(fbr_model *m, int32_t n, const double *R_a, const double *R_b, double *R_out, int32_t mem);

/* ---- the step before the path: signal conditioning of measurement channels ---------------------------- */
/*
 * The array operations of Data.preprocess (identification/data.py:369-619) on channel arrays X [S][ld] (row-major, the first ncols
 * columns are processed, in place), in `mem` space; `m` only supplies the device and the stream.
 *   fbr_filtfilt      scipy.signal.filtfilt(b, a, X, axis=0) with its defaults (odd extension of 3*ncoef samples, steady-state
 *                     initial conditions): the zero-phase Butterworth low-passes of positions / velocities / torques / IMU / FT
 *                     channels (data.py:430-436, 470-480, 505-520, 600-615).  ncoef = order + 1 <= 12, S > 3*ncoef.
 *   fbr_medfilt       scipy.signal.medfilt(X, (k, 1)) (zero padding at the ends), k odd <= 31 (data.py:466, 485, 500, 598).
 *   fbr_central_diff  D [S][ncols] = 4th-order central difference of A [S][ncols] (dense) over the time stamps T [S] with the
 *                     reference's edge rules (data.py:396-418).  S >= 5.
 */
int fbr_filtfilt(fbr_model *m, const double *b, const double *a, int32_t ncoef, double *X, int64_t S, int32_t ncols, int32_t ld, int32_t mem);
int fbr_medfilt(fbr_model *m, int32_t k, double *X, int64_t S, int32_t ncols, int32_t ld, int32_t mem);
int fbr_central_diff(fbr_model *m, const double *A, const double *T, double *D, int64_t S, int32_t ncols, int32_t mem);

/* ---- profiling ---------------------------------------------------------------------------------- */
#define FBR_PROF_KIN 0       /* link kinematics kernel */
#define FBR_PROF_REGRESSOR 1 /* materialising regressor kernel */
#define FBR_PROF_GRAM 2      /* fused regressor->Gram MFMA kernel */
#define FBR_PROF_REDUCE 3    /* slice reduction / scatter of the Gram partials */
#define FBR_PROF_ID 4        /* inverse dynamics / predict kernel */
#define FBR_PROF_TSQR 5      /* TSQR fold kernels (level 0 per chunk, merge tree) */
#define FBR_PROF_PACK 6      /* tile-image packing kernel of the fused Gram pass (producer stream) */
#define FBR_PROF_H2D 7       /* host -> device staging copies of the chunked fused pass (pinned host inputs) */
#define FBR_PROF_TREE 8      /* TSQR merge tree of the main row group on the call's stream (the latency-bound tail of a call; the other
                                groups' trees run beside it on side streams and are not counted) */
#define FBR_PROF_COUNT 9
/* When enabled, every kernel launch is bracketed by hipEvents on the launch stream; fbr_profile_get
   returns, per kernel class, the summed device time in ms and the launch count since the last reset
   (the counters are reset by the call).  Used by bench.py for the live roofline figures. */
int fbr_profile_enable(fbr_model *m, int32_t on);
int fbr_profile_get(fbr_model *m, double *ms_out /*[FBR_PROF_COUNT]*/, int64_t *launches_out /*[FBR_PROF_COUNT]*/);

/* ---- introspection (tests, tooling) ------------------------------------------------------------ */
/*
 * MFMA instructions (v_mfma_f64_16x16x4_f64, 2 * 16 * 16 * 4 = 2048 flop each) that fbr_tsqr / fbr_tsqr_cols executes for num_samples
 * samples and k rhs columns: the level-0 folds of the row-sorted chunks of every row group (a block is folded from the first
 * column its rows can touch) and the merge trees over the per-workgroup factors -- counted on the host from the same chunking and fold
 * rules the kernels use, so bench.py can report executed (not dense-model) flops.  cols == NULL: every identified column.
 * block_rows / n_padded (optional): rows per fold and padded factor width.
 */
int fbr_tsqr_work_info(fbr_model *m, const int32_t *cols, int32_t ncols, int32_t k, int64_t num_samples, int64_t *mfma_level0,
                       int64_t *mfma_tree, int32_t *block_rows, int32_t *n_padded);

/* Tile program of the fused Gram kernel that fbr_gram_accumulate runs for a batch of num_samples samples (< 0: a batch large enough
   for the column reductions below): counts of padded column tiles / tile pairs / MFMA k-steps per sample, so tests and bench.py can
   report executed vs algorithmic flops of the program that was actually executed. */
int fbr_gram_program_info(const fbr_model *m, int32_t k, int64_t num_samples, int32_t *num_tiles, int32_t *num_pairs,
                          int64_t *mfma_per_sample, int32_t *num_parts);

/* The sample-contiguous Gram pass (option "gram_lane", csrc/fbr_gram64.h) for the same batch: info[0] = 1 when the model qualifies and the
   option is on (device-resident inputs, k <= 1), else every entry is 0; [1] tile rows of a 64-sample block image, [2] its bytes, [3] MFMA
   instructions per block, [4] row levels, [5] slabs of the widest stage, [6] LDS bytes of the Gram kernel, [7] column tiles, [8] force
   tiles (option "gram_force_tiles"), [9] sum over the levels of the busiest wave's active tile pairs, [10] the same for a perfect split
   over the 8 waves, [11] stages (barriers) per half block.  The image is written once and read once per pass: 2 * info[2] / 64 bytes of HBM traffic per sample. */
int fbr_gram_lane_info(const fbr_model *m, int32_t k, int64_t num_samples, int64_t info[12]);

/*
 * Column reductions (options "link_merge" / "regroup" / "reduce_min_work" below).  The regressor columns of a link attached by a FIXED
 * joint are an exact, constant linear combination of the columns of the moving body it rides on (the 10 x 10 change of frame of the
 * inertial parameters), and three parameter directions of a link behind a revolute joint act like parameters of its parent.
 * fbr_gram_accumulate / fbr_gram_submit / fbr_tsqr / fbr_tsqr_submit therefore reduce over a full-rank subset of the columns where that
 * pays and expand the small result with a constant matrix E: G = E^T G_red E, R = qr([R_in ; R_red E]) -- the same G and the same R^T R
 * to rounding (1e-15 relative), same layout, same arguments.  moving_links / reduced_cols: what a Gram pass over num_samples samples
 * runs on (< 0: a batch large enough for the reductions; == the model's own counts when nothing is reduced).
 */
int fbr_model_link_merge_info(const fbr_model *m, int64_t num_samples, int32_t *moving_links, int32_t *reduced_cols);

/* ---- options ------------------------------------------------------------------------------------ */
/*
 * Per-model switches and thresholds.  The library never reads the process environment: what a call does is decided by its arguments
 * and by these options.  Setting an option first waits for every submission in flight; unknown keys fail with FBR_E_INVALID.
 *   "link_merge"                 1     column reductions on (0: every reduction runs over all columns of the model)
 *   "regroup"                    1     ... including the revolute regrouping (0: fixed links merged only)
 *   "reduce_min_work"            1e9   a Gram call takes the reductions when S (P - P_red) P >= this, a factorisation from an eighth of
 *                                      it on (0: always; the reduced pass costs a second model's launches and the expansion kernels)
 *   "reduce_grouped_min_samples" 512   fbr_gram_grouped takes them for groups of at least this many samples
 *   "chunk_samples"              0     > 0: samples per chunk of every pass (0: sized by memory)
 *   "min_chunks"                 4     a short fused pass is still cut into this many chunks
 *   "h2d_chunked"                1     pinned host inputs staged chunk by chunk on a copy stream, overlapped with the kernels
 *   "fused_id"                   1     fbr_predict / fbr_inverse_dynamics_batch run kinematics and torques in ONE kernel, the link records
 *                                      stay in registers (0: kinematics kernel + torque kernel with the records staged through HBM)
 *   "gram_lane"                  1     fused Gram over SAMPLE-contiguous images (MFMA k-steps over four samples of one regressor row) with a
 *                                      one-lane-per-sample producer, where the call allows: at most one rhs column, a tile program in one part
 *                                      (friction columns included); inputs resident in HBM, pageable or pinned (staged chunk by chunk);
 *                                      fbr_gram_grouped without rhs columns (0: always the per-sample images of the kinematics + packer kernels)
 *   "gram_lane_waves"            8     gram_lane, models whose tile pairs need the one-workgroup-per-CU shape: workgroups of 8 waves with 18
 *                                      accumulators each (two waves per SIMD); 16: 16 waves with 10 accumulators (four per SIMD at 128
 *                                      registers: measured 4 % slower on WALK-MAN)
 *   "gram_force_tiles"           1     gram_lane: the three force rows of the base wrench run on tiles of their own that hold only the columns
 *                                      with a force (mass, first moments), when the extra tile pairs fit the accumulators (0: every tile pair
 *                                      pays the three levels)
 *   "gram_shape"                 0     fused Gram kernel shape: 0 by model, 1 one workgroup per CU, 2 two per CU
 *   "gram_rhs_tile"              0     1: dense tiles for the rhs columns even for k <= 2 (default: their products come from the packer)
 *   "gram_orient"                1     tile pairs turned so that the row segments fill up
 *   "tsqr_groups"                1     rows grouped along the kinematic tree ...
 *   "tsqr_group_min_samples"     24000 ... from this many samples on
 *   "tsqr_reorder"               1     columns factorised in link-depth order (single-factorisation path)
 *   "tsqr_narrow"                1     wave-private kernels for <= 128 columns
 *   "tsqr_narrow_tall"           1     ... whose level-0 folds take 96-row blocks at one wave per SIMD for long calls over <= 6 column tiles
 *                                      (0: 48-row blocks, two waves per SIMD)
 *   "tsqr_lane_writer"           1     regressor writer of the TSQR: one lane per sample with the kinematics fused in (no records in HBM), chunks
 *                                      written column-major in 512-byte runs (0: kinematics kernel + the workgroup-per-sample writers below)
 *   "tsqr_force_group"           1     row groups of a floating base: the three FORCE rows of the base wrench form a group of their own, factorised
 *                                      over the columns that produce a force (a link's mass and first moments); the dense group keeps the three
 *                                      moment rows (0: one dense group of six rows)
 *   "tsqr_writer"                0     grouped regressor writer: 0 by work-item count; 8 / 16: one thread per column / column pair, stores of that
 *                                      width; 32: rows staged in the LDS and streamed out in 16-byte pieces (measured: no faster)
 *   "tsqr_tree_one_wg"           0     merges by one workgroup instead of pipelined across workgroups (bit-identical, slower)
 *   "tsqr_side_trees_beside"     0     row-group TSQR: 1 starts the side groups' merge trees beside the main group's last level-0 fold (the
 *                                      default of rounds 3 - 5); 0 behind it (round 6: with the force rows in a group of their own the main
 *                                      fold is half as long and a 125 k-sample call is 0.45 ms shorter this way)
 *   "tsqr_prologue_overlap"      1     a submission's kinematics / first writer beside the merge trees of the one before
 *   "tsqr_short_call_factors"    1     fewer private factors (shallower merge trees) for calls too short to amortise them
 *   "gram_serial", "gram_timing", "tsqr_timing"  0   diagnostics (producer on the main stream; cycle counters printed to stderr)
 * fbr_model_option_name enumerates the keys (index 0 .. until it fails).
 */
int fbr_model_set_option(fbr_model *m, const char *key, double value);
int fbr_model_get_option(const fbr_model *m, const char *key, double *value);
int fbr_model_option_name(int32_t index, const char **name);


#ifdef __cplusplus
}
#endif
#endif /* FBR_H */
