/*
 * gelato_amd.h -- C-ABI of the MI355X-native LGR defect-residual / FD-Jacobian engine.
 *
 * This is the drop-in boundary for GELATO's hot path.  In the reference that
 * path is reached from pyoptsparse's objfunc/sens callbacks
 * (Trajectory_Optimization.py:194-312) through eight Python functions of
 * lib/con_dynamics.py and, below them, the pybind11 module `dynamics_c`
 * (src/pybind_dynamics.cpp:108-114) plus NumPy's D.dot(X).  Each entry point
 * below names the reference interface it replaces.  Plain pointers and sizes
 * only; the caller owns every buffer it passes; the engine owns device memory.
 * One handle = one HIP device + one HIP stream; a handle is not thread-safe.
 *
 * Packed decision vector x (all normalised by `units`, doubles) -- exactly the
 * concatenation of the reference's xdict arrays
 * (Trajectory_Optimization.py:318-352):
 *     [ mass M | position 3M | velocity 3M | quaternion 4M | u 2N | t S+1 ]
 * with S phases, N = sum(num_nodes), M = N + S.
 *
 * Residual vector (11N doubles): [ eqcon_dyn_mass N | eqcon_dyn_pos 3N |
 * eqcon_dyn_vel 3N | eqcon_dyn_quat 4N ], each exactly as
 * lib/con_dynamics.py:63,152,289,533 returns it.
 *
 * Jacobian values: 13 COO blocks in the order
 *   group 0 eqcon_dyn_mass : mass, t
 *   group 1 eqcon_dyn_pos  : position, velocity, t
 *   group 2 eqcon_dyn_vel  : mass, position, velocity, quaternion, t
 *   group 3 eqcon_dyn_quat : quaternion, u, t
 * each in the reference's exact emission order (lib/con_dynamics.py:66-113,
 * 155-213,292-496,536-632).  "full" = all 13 blocks
 * concatenated; "compact" = the DISTINCT x-dependent values (everything that is
 * not a constant D / 0 / +-1 / massflow entry; a tf column that is the exact
 * negative of its t0 column, and the node-uniform pos/velocity diagonal value,
 * are held once), in the engine's coalesced device order, mapped into "full" by
 * the gather map gel_full_source().
 *
 * Status codes: 0 = ok; GEL_NONFINITE (1) = the evaluation completed but some
 * output is NaN/Inf (the Python shim maps it to pyoptsparse's fail=True; the
 * reference itself hard-codes fail=False, Trajectory_Optimization.py:240,311);
 * negative = error (bad argument, HIP failure).  gel_last_error() gives text.
 */
#ifndef GELATO_AMD_H_
#define GELATO_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEL_OK 0
#define GEL_NONFINITE 1
#define GEL_ERR_ARG (-1)
#define GEL_ERR_HIP (-2)
#define GEL_ERR_ALLOC (-3)

#define GEL_DEVICE_NONE (-1) /* gel_problem_desc.device: host-only handle (pattern / dims / LGR), cannot evaluate */

/* gel_problem_desc.flags */
#define GEL_FLAG_DX_MFMA 1 /* force the D.X product onto v_mfma_f64_16x16x4_f64 */
#define GEL_FLAG_DX_VALU 2 /* force wavefront dot-products (VALU FMAs) */
#define GEL_FLAG_NO_PACK 4 /* matrix-pipe form of a problem whose phases all have <= 32 nodes: one decision vector per
                              wavefront (32 idle lanes) instead of two -- for A/B measurements and parity tests */
#define GEL_FLAG_ITEM_MAJOR 16 /* cooperative launches in work-item major order also when every phase has at most 32 nodes (the default there is
                              vector-group major, XCD-aware: see gel_eval_kernel.h) -- for A/B measurements */
#define GEL_FLAG_FD_RECOMPUTE 8 /* every finite-difference sweep re-runs the reference's chain on the perturbed input
                              (lib/con_dynamics.py:381-400,452-480), as rounds 1-2 did.  Default (flag clear): the three position
                              sweeps form the CHANGE of altitude / atmosphere / wind from algebraic difference identities of the
                              reference's formulas (accurate to ~1e-12 of the change, where a recomputation carries the
                              reference's own 1e-8 .. 1e-4 finite-difference noise), and the t0 / tf columns of aerodynamic
                              phases are written in closed form +-f_c unit_t/2 (the RHS does not depend on t; the reference's two
                              sweeps only add rounding noise, <= 2e-6 measured).  Both forms meet the stated tolerance against
                              the reference's values; with the flag set the compact layout keeps separate t0 / tf slots. */

#define GEL_NUM_GROUPS 4
#define GEL_NUM_BLOCKS 13

typedef struct gel_problem gel_problem; /* opaque */

/* Static problem = the hot path's view of pdict + unitdict
 * (Trajectory_Optimization.py:116-167): per-phase event parameters
 * (lib/con_dynamics.py:249-252,46-61,521), units, dx, wind and CA tables. */
typedef struct {
  int32_t num_sections;          /* S  (pdict["num_sections"]) */
  const int32_t* num_nodes;      /* [S] LGR nodes per phase (events["num_nodes"]) */
  const double* thrust;          /* [S] vacuum thrust, N          (params[i]["thrust"]) */
  const double* massflow;        /* [S] kg/s                      (params[i]["massflow"]) */
  const double* reference_area;  /* [S] m^2; == 0 selects the NoAir RHS (con_dynamics.py:257) */
  const double* nozzle_area;     /* [S] m^2                       (params[i]["nozzle_area"]) */
  const int32_t* engine_on;      /* [S] params[i]["engineOn"] */
  const int32_t* attitude_hold;  /* [S] 1 iff params[i]["attitude"] in ("hold","vertical") */
  double unit_mass, unit_position, unit_velocity, unit_u, unit_t; /* unitdict */
  double dx;                     /* pdict["dx"], 1e-8 in the reference */
  double barC20;                 /* normalised C20 of the J2 gravity; 0 -> -0.484165371736e-3 (src/gravity.cpp:18) */
  int32_t wind_rows;             /* K */
  const double* wind_table;      /* [K][3] altitude, wind_n, wind_e (pdict["wind_table"]) */
  int32_t ca_rows;
  const double* ca_table;        /* [K][2] Mach, CA (pdict["ca_table"]) */
  const double* D;               /* optional: per-phase D (n x (n+1), row-major) concatenated, i.e.
                                    pdict["ps_params"].D(i); NULL -> generated by gel_lgr_diffmat */
  const double* tau;             /* optional: per-phase tau concatenated; NULL -> gel_lgr_nodes */
  int32_t device;                /* HIP device ordinal */
  int32_t flags;                 /* 0 or GEL_FLAG_* */
} gel_problem_desc;

typedef struct {
  int32_t S, N, M;
  int32_t num_vars;              /* 11M + 2N + S + 1 */
  int32_t num_rows[GEL_NUM_GROUPS]; /* N, 3N, 3N, 4N */
  int64_t block_nnz[GEL_NUM_BLOCKS];
  int64_t block_shape[GEL_NUM_BLOCKS][2];
  int64_t total_nnz;             /* "full" length */
  int64_t num_var_entries;       /* "compact" length V */
  int64_t algorithmic_bytes;     /* SURVEY.md 8(d) A_min per eval: 8*(num_vars + 11N + V_ref), V_ref = every x-dependent
                                    value the reference computes (9n + 39n|30n + 32n per phase) */
  int64_t stored_bytes;          /* what one eval writes: 8*(11N + V) */
} gel_dims;

/* ---- LGR transcription on the host (replaces lib/PSfunctions.py:149-168,182-208,
 *      i.e. nodes_LGR(n) and differentiation_matrix_LGR(n), reverse=True) ---- */
int gel_lgr_nodes(int32_t n, double* tau /* [n] */);
int gel_lgr_diffmat(int32_t n, double* D /* [n][n+1] */);

/* ---- problem life cycle (replaces building pdict["ps_params"] + the static
 *      part of every con_dynamics call) ---- */
int gel_problem_create(const gel_problem_desc* desc, gel_problem** out);
int gel_problem_destroy(gel_problem* p);
int gel_problem_dims(const gel_problem* p, gel_dims* out);
int gel_problem_D(const gel_problem* p, int32_t phase, double* D);     /* PSparams.D(i),   SectionParameters.py:46-49 */
int gel_problem_tau(const gel_problem* p, int32_t phase, double* tau); /* PSparams.tau(i), SectionParameters.py:41-44 */

/* ---- fixed sparsity pattern (what the first sens() call hands to
 *      addConGroup(jac=...), Trajectory_Optimization.py:354-355,400-407) ---- */
int gel_pattern(const gel_problem* p, int32_t block, int32_t* rows, int32_t* cols); /* block 0..12 */
int gel_pattern_all(const gel_problem* p, int32_t* rows_full, int32_t* cols_full);  /* all 13 blocks concatenated ("full"), one pass */
int gel_const_values(const gel_problem* p, double* vals_full); /* constants filled, x-dependent entries 0 */
int gel_var_index(const gel_problem* p, int64_t* idx /* [V] compact slot -> index into full (its first plain use) */);
/* the gather map full <- compact: src[i] = -1: constant (gel_const_values); s >= 0: compact[s]; s <= -2: -compact[-2 - s] */
int gel_full_source(const gel_problem* p, int32_t* src /* [total_nnz] */);

/* ---- evaluation, host buffers (B = 1; the pyoptsparse callback path) ----
 * gel_eval_residual : equality_dynamics_{mass,position,velocity,quaternion}
 *                     (lib/con_dynamics.py:34,116,216,499) in one launch.
 * gel_eval_jacobian : equality_jac_dynamics_{...} (lib/con_dynamics.py:66,155,292,536)
 *                     in one launch.  vals_full must be a buffer of total_nnz
 *                     doubles; if fill_constants != 0 the constant entries are
 *                     (re)written too, otherwise only the x-dependent entries are
 *                     touched (the buffer is assumed to still hold the constants).
 * gel_eval          : both at once (one objfunc + one sens share of the hot path). */
int gel_eval_residual(gel_problem* p, const double* x, double* res);
int gel_eval_jacobian(gel_problem* p, const double* x, double* vals_full, int32_t fill_constants);
int gel_eval(gel_problem* p, const double* x, double* res, double* vals_full, int32_t fill_constants);

/* ---- evaluation, batched over B decision vectors ----
 * host buffers: x [B][num_vars], res [B][11N], jvar [B][V] (compact). */
int gel_eval_batch(gel_problem* p, int32_t B, const double* x, double* res, double* jvar);
/* device buffers (inputs already resident in HBM; asynchronous on `stream`,
 * which is a hipStream_t or NULL for the handle's own stream).  res may be NULL
 * (Jacobian only is not offered: the centre RHS is shared); jvar may be NULL
 * (residual only).  Returns after enqueueing; call gel_sync() for the status. */
int gel_eval_batch_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, void* stream);
/* materialise every COO value like the reference does: d_jfull [B][total_nnz],
 * from d_jvar [B][V] (compact -> full expansion kernel). */
int gel_expand_full_device(gel_problem* p, int32_t B, const double* d_jvar, double* d_jfull, void* stream);
/* The handle's own pinned host buffers for the one-vector calls (gel_eval_residual, gel_eval, gel_eval_jacobian,
 * gel_eval_callback): *res [11N] and *vals_full [total_nnz, the constants already in place].  A caller that passes THESE pointers
 * as its res / vals_full gets its results without a host copy: the kernel writes the residual rows and every block of the value
 * vector whose entries are all x-dependent straight to their places (COO-direct output, gelato_amd/csrc/gel_eval_kernel.h), and
 * the host only scatters the entries that sit alone between constants.  The buffers belong to the handle and are rewritten by the
 * next such call (the reference returns fresh arrays, lib/con_dynamics.py:108-113; pyoptsparse copies what it is given at once).
 * *x0, *x1 [num_vars]: two pinned decision-vector buffers; a one-vector call whose x is one of them reads it in place (two, so that a
 * caller can keep the previous vector for comparison while it fills the next).  Any of the four out-pointers may be NULL. */
/* One-vector calls (gel_eval, gel_eval_jacobian, gel_eval_callback) return when the kernel itself has told the host that its results
 * are in the pinned buffers (its last workgroup stores a sequence number there after every workgroup's system-scope release), 4-5 us
 * before the runtime's end-of-kernel signal; the handle's stream may still hold the kernel's tail, and whatever is launched on it
 * next is ordered behind.  Environment GEL_DONE_FLAG=0: wait for the runtime's signal instead. */
int gel_pinned_buffers(gel_problem* p, double** res, double** vals_full, double** x0, double** x1);
/* The same result without rewriting the constants (the reference rebuilds every COO value per call,
 * lib/con_dynamics.py:108-111,491-494,627-630): gel_fill_full_device lays the constant template into d_jfull [B][total_nnz]
 * ONCE, gel_update_full_device then writes only the x-dependent entries (4 % of the values at 6 x 64) from d_jvar [B][V] after
 * every evaluation.  Bit-identical to gel_expand_full_device as long as nothing else writes the buffer. */
int gel_fill_full_device(gel_problem* p, int32_t B, double* d_jfull, void* stream);
int gel_update_full_device(gel_problem* p, int32_t B, const double* d_jvar, double* d_jfull, void* stream);
/* One evaluation with EVERY COO value valid in HBM afterwards: the fused launch (d_res may be NULL) followed by the update of
 * d_jfull [B][total_nnz] (laid down once by gel_fill_full_device) from the compact values it has just written to d_jvar [B][V].
 * A launch whose output fits the Infinity Cache writes the compact values with ordinary instead of non-temporal stores, so that
 * the update finds them there (6 x 64, B = 1024: 0.26 -> 0.165 ms, 6.2 M evals/s).  Same bits as gel_eval_batch_device +
 * gel_update_full_device. */
int gel_eval_full_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, double* d_jfull, void* stream);
/* multi-GPU sharding of ONE batch (the defect path is block-diagonal per phase, lib/con_dynamics.py:46,132,237,320,512,554,
 * and its forward-difference columns are independent).  A work item = one 64-node chunk of one phase;
 * unit = 4 * work_item + part, part 0 = everything of the work item except its three position
 * sweeps, parts 1..3 = one position sweep each (empty for phases without aerodynamics), i.e. the forward-
 * difference COLUMNS of lib/con_dynamics.py:381-400 dealt to different GPUs (BASELINE.json configs[3]).  A unit
 * writes only its own entries of d_res / d_jvar; disjoint unit ranges compose to the full evaluation.  d_jvar is
 * required; d_res may be NULL. */
int gel_eval_shard_units_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar,
                                int32_t unit_begin, int32_t unit_count, void* stream);
/* which unit writes which output entry: res_owner [11N], jvar_owner [V] (unit ids as above).  With it ranks owning disjoint
 * unit ranges exchange exactly their own entries (one all-gather, gelato_amd/parallel.py) instead of reducing full buffers. */
int gel_unit_owner(const gel_problem* p, int32_t* res_owner, int32_t* jvar_owner);
/* Packed exchange of a unit-sharded evaluation: zero pack / unpack launches around the one collective.  gel_shard_plan fixes the
 * layout for `nranks` ranks holding the contiguous unit ranges [unit_begin[r], unit_begin[r+1]) (unit_begin[0] = 0,
 * unit_begin[nranks] = 4 * work items): ONE buffer out [nranks][B][width]; rank r's entries of vector b are the contiguous block
 * out[r][b][0 .. its share), every unit's entries one contiguous run inside it.  res_pos [11N] / jvar_pos [V] (either may be NULL)
 * = rank * width + offset of every entry of the ordinary res / compact-value layouts: entry i of vector b sits at
 * out[(pos / width) * B * width + b * width + pos % width].  gel_eval_shard_packed_device makes rank `rank`'s kernel write its
 * entries STRAIGHT into its slice out[rank] (all B vectors), so an in-place all-gather over the slices (send = out[rank],
 * receive = out) completes the buffer on every rank; a consumer reads it through the map, or asks
 * gel_shard_unpack_device for the ordinary layouts (one gather launch; d_res or d_jvar may be NULL).  A host-only handle can plan
 * (the CPU tests do); the plan is per handle and replaced by the next call -- which is why the two device calls take the
 * (nranks, width) the caller's buffer was sized for and return GEL_ERR_ARG when the handle holds another plan by now.
 * (lib/con_dynamics.py:46,132,237,320,512,554: per-phase independence; :381-400: independent forward-difference columns.) */
int gel_shard_plan(gel_problem* p, int32_t nranks, const int32_t* unit_begin /* [nranks + 1] */, int64_t* width,
                   int64_t* res_pos /* [11N] or NULL */, int64_t* jvar_pos /* [V] or NULL */);
int gel_eval_shard_packed_device(gel_problem* p, int32_t B, const double* d_x, double* d_out /* [nranks][B][width] */, int32_t rank,
                                 int32_t nranks, int64_t width, void* stream);
int gel_shard_unpack_device(gel_problem* p, int32_t B, const double* d_out, double* d_res, double* d_jvar, int32_t nranks,
                            int64_t width, void* stream);
int gel_num_chunks(const gel_problem* p, int32_t* nchunks);
int gel_chunk_phase(const gel_problem* p, int32_t* phase /* [nchunks] */);
/* which form of the fused kernel a launch of B vectors takes: info = {jacobian, D.X on the matrix pipe, split
 * latency form, wavefronts launched, two decision vectors per wavefront} -- so that a measurement can name the kernel
 * it timed. */
int gel_launch_info(const gel_problem* p, int32_t B, int32_t want_res, int32_t want_jac, int32_t* info /* [5] */);
int gel_sync(gel_problem* p, void* stream); /* waits; returns GEL_NONFINITE if any eval since the last sync produced NaN/Inf */

/* ---- generic column-batched dense forward difference (replaces lib/jac_fd.py:29-62
 *      applied to the four defect residuals; used as the cross-check of the
 *      structured Jacobians).  J [num_rows[group]][num_vars], row-major. ---- */
int gel_jac_fd(gel_problem* p, int32_t group, const double* x, double* J);
/* The same quotients without the zeros.  Each phase's rows see only the phase's own 13 n + 13 columns (every other column of
 * the reference's dense result, lib/jac_fd.py:54-60, is an exact zero): block i = [rows[i] = w n_i][cols[i] = 13 n_i + 13],
 * row-major, at offset[i] doubles of `blocks` (offset[S] = total), its first row at row row0[i] of the group, its local column c
 * at global column gel_jac_fd_block_cols(phase)[c] (local layout [mass n+1 | position 3(n+1) | velocity 3(n+1) | quaternion
 * 4(n+1) | u 2n | t0 tf]).  6 x 64, velocity rows: 7.8 MB instead of 46.7 MB across PCIe. */
int gel_jac_fd_block_dims(const gel_problem* p, int32_t group, int64_t* rows /* [S] or NULL */, int64_t* cols /* [S] or NULL */,
                          int64_t* row0 /* [S] or NULL */, int64_t* offset /* [S + 1] or NULL */);
int gel_jac_fd_block_cols(const gel_problem* p, int32_t phase, int32_t* cols /* [13 n + 13] */);
int gel_jac_fd_blocks(gel_problem* p, int32_t group, const double* x, double* blocks /* [offset[S]] */);
/* Device-resident form: x and the result stay in HBM (blocks = 0: dense [num_rows[group]][num_vars]; 1: the blocks), launches
 * only, on `stream` (NULL = the handle's); NaN / Inf is reported by the next gel_sync.  One call in flight per handle (the
 * perturbed vectors and their residuals live in the handle). */
int gel_jac_fd_device(gel_problem* p, int32_t group, const double* d_x, double* d_J, int32_t blocks, void* stream);

/* ---- aero path constraints (replace lib/con_aero.py:90-252 inequality_max_alpha /
 *      _q / _qalpha, :254-309 inequality_length_*, :311-756 inequality_jac_max_*).
 *      kind 0 = AOA_max, 1 = dynamic_pressure_max, 2 = Q_alpha_max.  A spec = one entry of
 *      condition[...]: the constrained phase (increasing, < num_sections - 1), range "all" (every state node,
 *      n + 1 rows) or "initial" (1 row), and limit = units[3] (value*pi/180 for kinds 0 and 2, value for 1).
 *      con [B][nrows] = 1 - f/limit.  jac_vals [B][sum nnz4]: the position | velocity | quaternion | t
 *      blocks of the reference's COO values (already negated, con_aero.py:443-463), in its emission order;
 *      gel_aero_pattern gives the matching rows / cols per block (var 0..3). ---- */
int gel_aero_configure(gel_problem* p, int32_t kind, int32_t nspec, const int32_t* phase, const int32_t* range_all,
                       const double* limit);
int gel_aero_dims(const gel_problem* p, int32_t kind, int32_t* nrows, int64_t* nnz4 /* [4] */);
int gel_aero_pattern(const gel_problem* p, int32_t kind, int32_t var, int32_t* rows, int32_t* cols);
int gel_eval_aero(gel_problem* p, int32_t kind, int32_t B, const double* x, double* con, double* jac_vals /* or NULL */);
/* all three kinds in ONE launch (a state node constrained by several kinds runs the air-velocity chain once): con[kind] /
 * jac[kind] as above, NULL = not wanted (jac may be NULL altogether).  Host buffers, or device buffers with the inputs
 * already resident (asynchronous on `stream`; status through gel_sync). */
int gel_eval_aero_all(gel_problem* p, int32_t B, const double* x, double* const* con /* [3] */, double* const* jac /* [3] or NULL */);
int gel_eval_aero_all_device(gel_problem* p, int32_t B, const double* d_x, double* const* d_con /* [3] */,
                             double* const* d_jac /* [3] or NULL */, void* stream);
/* Defect groups AND aero path constraints of a resident batch in one call [r6] -- the hot-path share of objfunc + sens with the aero
 * rows riding on it (lib/con_dynamics.py:216-496 and lib/con_aero.py:89-248,311-371 evaluate the same geodetic -> atmosphere -> wind
 * chain at the same nodes; src/pybind_dynamics.cpp:42-59 / src/wrapper_utils.hpp:89-206).  d_res [B][11 N] and d_jvar [B][V] as
 * gel_eval_batch_device writes them; d_aero [B][width]: ONE record per decision vector in two parts, each
 *   [con alpha | con q | con q-alpha | jac alpha | jac q | jac q-alpha]
 * (part B laid out like gel_eval_aero_all's arrays for its rows, part A spec-major): part A = state nodes 1 .. n of the aerodynamic phases' "all nodes"
 * specs (the rows a lane of the fused kernel has: a spec's row of a column is n doubles -- whole 64-byte lines per store), part
 * B = every other row (state node 0 of a phase, phases without aerodynamics, "initial" specs).  gel_aero_record_layout: width and
 * the twelve section offsets (off_con / off_jac [2][3]: part, kind; every section on a multiple of eight doubles);
 * gel_aero_record_map: the record index of every entry of gel_eval_aero_all's arrays (var = -1: the constraint vector; 0..3: the
 * position / velocity / quaternion / t block in gel_aero_pattern's order) -- the gather a consumer applies, like gel_full_source
 * for the compact Jacobian values; index -1 = an exact zero that is not stored (the t0 / tf columns of part A: the air-relative
 * velocity does not depend on the Earth angle; problems created with GEL_FLAG_FD_RECOMPUTE run those sweeps and store them).
 * Where the launch takes the throughput form with one decision vector per wavefront (not: a handful of vectors, meshes of phases
 * of at most 32 nodes, GEL_FLAG_FD_RECOMPUTE), the lanes of an aerodynamic phase write part A themselves, from the centre
 * evaluation and the position sweeps they run anyway; otherwise -- and with GEL_AERO_FUSED=0 in the environment -- aero_kernel
 * writes it in a launch of its own.  Part B is always a small second launch.  Every value is the same bit for bit as
 * gel_eval_batch_device's and gel_eval_aero_all_device's either way.  Asynchronous on `stream`; status through gel_sync. */
int gel_aero_record_layout(const gel_problem* p, int64_t* width, int64_t* off_con /* [2][3] */, int64_t* off_jac /* [2][3] */);
int gel_aero_record_map(const gel_problem* p, int32_t kind, int32_t var, int64_t* idx);
int gel_eval_batch_aero_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, double* d_aero,
                               void* stream);

/* ---- knot / terminal / user rows.
 *  Linear rows: value = (coef0 * x[idx0] + coef1 * x[idx1]) + c0 (idx1 < 0: one term) over the packed decision vector --
 *    what equality_init, equality_time, inequality_time and equality_knot_LGR compute
 *    (lib/con_init_terminal_knot.py:39-52,118-141,174-252,422-436); with coefficients +-1 it rounds like the
 *    reference's (a - b) - c.  Their Jacobians are these coefficients: constant COO blocks the caller lays out.
 *  Node-function rows: a function f of ONE knot state (r = position * unit_position, v = velocity * unit_velocity of state
 *    node `node`; for fn >= 9 also the knot time t = x_t[tcol] * unit_t) and its forward difference over the seven
 *    columns the row can see (position xyz, velocity xyz of that node, then the knot time), formed in the kernel.
 *      mode & 3: value = f / p[0] - p[1] (0) or (f - p[1]) / p[0] (1); mode & 8: negated ("max" rows).
 *      mode & 4 clear: difference of the value itself, (value(x + dx e_c) - value(x)) / dx -- what lib/jac_fd.py and
 *        equality_jac_6DoF_LGR_terminal do; set: ((f(x + dx e_c) - f(x)) / dx) / p[0] (negated with mode & 8) -- what
 *        lib/con_waypoint.py does.
 *      fn 0 orbit energy, 1 angular momentum, 2 inclination [rad]  -> equality_6DoF_LGR_terminal and its Jacobian
 *        (lib/con_init_terminal_knot.py:329-405, src/wrapper_coordinate.hpp:222-250);
 *      fn 3 a, 4 e, 5 a(1-e), 6 a(1+e) of the orbital elements (src/Coordinate.cpp:197-245), 7 |r|, 8 |v|
 *        -> the shipped user constraint example/user_constraints.py:120-139 (fn 5, p = {6378137, 1}), whose generic
 *        lib/jac_fd.py:29-62 loop perturbs every column of x and gets an exact zero for all but these;
 *      fn 9 / 10 / 11 geodetic latitude / longitude [deg] / altitude [m] at the knot time -> equality_posLLH,
 *        inequality_posLLH (lib/con_waypoint.py:507-560,717-784); fn 12 / 13 latitude / longitude [deg] of the
 *        instantaneous impact point (FAA algorithm, lib/IIP.py:30-135) -> equality_IIP, inequality_IIP (:164-207,330-381);
 *        fn 14 sine of the elevation above an antenna's horizon, p[2..4] = antenna ECEF position, p[5..7] = its local
 *        vertical -> inequality_antenna (:45-51,70-105); fn 15 downrange [m]: Vincenty distance (lib/downrange.py:32-111)
 *        from the launch point p[2] = latitude, p[3] = longitude [deg] to the position's geodetic latitude / longitude
 *        -> the "downrange" rows of equality_posLLH / inequality_posLLH (:531-534,551-554,742,771-778) and their gradient
 *        (downrange_gradient, :583-607).
 *  con [B][nlin + nfn] (linear rows first); jfn [B][nfn][7]. ---- */
typedef struct { int32_t idx0, idx1; double coef0, coef1, c0; } gel_linear_row;
typedef struct { int32_t fn, node, tcol, mode; double p[8]; } gel_nodefn_row;
int gel_rows_configure(gel_problem* p, int32_t nlin, const gel_linear_row* lin, int32_t nfn, const gel_nodefn_row* fn);
int gel_rows_dims(const gel_problem* p, int32_t* nlin, int32_t* nfn);
int gel_rows_eval(gel_problem* p, int32_t B, const double* x, double* con, double* jfn /* or NULL */);
int gel_rows_eval_device(gel_problem* p, int32_t B, const double* d_x, double* d_con, double* d_jfn /* or NULL */, void* stream);

/* ---- one optimiser callback = one device round trip: the four defect groups, the knot / terminal / user row table and the
 *      aero path constraints of ONE decision vector launched back to back on the handle's stream, one synchronise
 *      (what objfunc / sens of Trajectory_Optimization.py:194-312 need from the device).  Every output pointer may be
 *      NULL (= not wanted); vals_full as in gel_eval_jacobian. ---- */
typedef struct {
  double* res;            /* [11N] */
  double* vals_full;      /* [total_nnz] */
  int32_t fill_constants;
  double* rows_con;       /* [nlin + nfn] */
  double* rows_jfn;       /* [nfn][7] */
  double* aero_con[3];    /* per kind */
  double* aero_jac[3];
} gel_callback_io;
int gel_eval_callback(gel_problem* p, const double* x, const gel_callback_io* io);

/* ---- post-processing table (replaces the per-node loop of output_result.py:121-262): for every
 *      state node of the decision vector x, at its time tx_res [M] (seconds: (tau_x (tf - to) / 2 + (tf + to) / 2) unit_t,
 *      Trajectory_Optimization.py:476-491), the derived quantities of the reference's output table, one device thread per
 *      node: out [M][GEL_OUTPUT_COLUMNS] in the order of gel_output_column.  The columns the reference copies from x (time,
 *      mass, position, velocity, quaternion, interpolated body rates) and its text columns (event, stage) stay with the
 *      caller (gelato_amd/output_result.py).  lat_IIP / lon_IIP are NaN where the FAA algorithm has no solution
 *      (posLLH_IIP_FAA(.., fill_na = False)).  launch_*: pdict["LaunchCondition"]["lat" / "lon"] for the downrange. ---- */
#define GEL_OUTPUT_COLUMNS 34
typedef enum {
  GEL_OUT_THRUST = 0, GEL_OUT_LAT, GEL_OUT_LON, GEL_OUT_LAT_IIP, GEL_OUT_LON_IIP, GEL_OUT_DOWNRANGE, GEL_OUT_ALTITUDE,
  GEL_OUT_ALTITUDE_APOGEE, GEL_OUT_ALTITUDE_PERIGEE, GEL_OUT_INCLINATION, GEL_OUT_ARGUMENT_PERIGEE, GEL_OUT_LON_ASCENDING_NODE,
  GEL_OUT_TRUE_ANOMALY, GEL_OUT_VEL_GROUND_NED_X, GEL_OUT_VEL_GROUND_NED_Y, GEL_OUT_VEL_GROUND_NED_Z, GEL_OUT_ACCEL_BODY_X,
  GEL_OUT_AERO_BODY_X, GEL_OUT_HEADING_NED2BODY, GEL_OUT_PITCH_NED2BODY, GEL_OUT_ROLL_NED2BODY,
  GEL_OUT_FLIGHTPATH_VEL_INERTIAL_GEOCENTRIC, GEL_OUT_AZIMUTH_VEL_INERTIAL_GEOCENTRIC, GEL_OUT_THRUST_DIRECTION_ECI_X,
  GEL_OUT_THRUST_DIRECTION_ECI_Y, GEL_OUT_THRUST_DIRECTION_ECI_Z, GEL_OUT_VEL_GROUND, GEL_OUT_VEL_AIR, GEL_OUT_AOA_TOTAL,
  GEL_OUT_AOA_PITCH, GEL_OUT_AOA_YAW, GEL_OUT_DYNAMIC_PRESSURE, GEL_OUT_Q_ALPHA, GEL_OUT_MACH
} gel_output_column;
int gel_output_table(gel_problem* p, const double* x, const double* tx_res /* [M] */, double launch_lat_deg,
                     double launch_lon_deg, double* out /* [M][GEL_OUTPUT_COLUMNS] */);

/* ---- from-file initial guess on the host (replaces initialize.py:322-409 initialize_xdict_6DoF_from_file, LGR mode;
 *      SURVEY.md 8f row f-3): linear interpolation (scipy interp1d with fill_value="extrapolate": the end intervals extend
 *      beyond the table) of a reference trajectory at the state-node times [-1, tau] and the control-node times tau of
 *      every phase, normalised by the handle's units.  table [nref][13] = mass | pos_ECI xyz | vel_ECI xyz | quat_ECI2BODY
 *      wxyz | rate_BODY_Y, rate_BODY_Z.  x = the packed decision vector.  Host arithmetic; works on host-only handles. ---- */
int gel_initial_guess(const gel_problem* p, int32_t nref, const double* t_ref, const double* table /* [nref][13] */,
                      const double* knot_times /* [S+1], seconds */, double* x /* [num_vars] */);

/* ---- node-batched RHS, host buffers (replace dynamics_c.dynamics_velocity,
 *      dynamics_velocity_NoAir, dynamics_quaternion; src/pybind_dynamics.cpp:30,73,94) ---- */
int gel_dynamics_velocity(int32_t n, const double* mass_e, const double* pos_eci_e, const double* vel_eci_e,
                          const double* quat_eci2body, const double* t, const double* param /* [5] */,
                          const double* wind_table, int32_t wind_rows, const double* ca_table, int32_t ca_rows,
                          const double* units /* [3] */, double barC20, double* acc_out /* [n][3] */);
int gel_dynamics_velocity_NoAir(int32_t n, const double* mass_e, const double* pos_eci_e,
                                const double* quat_eci2body, const double* param, const double* units,
                                double barC20, double* acc_out);
int gel_dynamics_quaternion(int32_t n, const double* quat_eci2body, const double* u_e, double unit_u,
                            double* dquat_out /* [n][4] */);

/* ---- device point functions (parity hooks for SURVEY.md 8a rows a4..a9).
 *  kind 0: geometric alt -> [geopotential alt, T, P, rho, a]   in [n]      out [n][5]  (src/Air.cpp:47-111)
 *  kind 1: ECEF xyz      -> [lat deg, lon deg, alt m]          in [n][3]   out [n][3]  (wrapper_coordinate.hpp:105-111)
 *  kind 2: pos           -> gravity                            in [n][3]   out [n][3]  (src/gravity.cpp:11-57), aux[0]=barC20
 *  kind 3: (pos,t,wn,we) -> quatrot(quat_nedg2eci(pos,t),(wn,we,0)) in [n][6] out [n][3] (src/Coordinate.cpp:104-110,
 *                                                                                    src/pybind_dynamics.cpp:51-52)
 *  kind 4: (vel,pos,t)   -> vel_eci2ecef                       in [n][7]   out [n][3]  (src/Coordinate.cpp:69-73)
 *  kind 5: altitude      -> wind_ned, table in aux [K][3]      in [n]      out [n][3]  (wrapper_utils.hpp:82-87)
 *  kind 6: x             -> interp(x, aux[:,0], aux[:,1])      in [n]      out [n]     (wrapper_utils.hpp:51-80)
 *  kind 7: (a, b)        -> [fsqrt(a), sqrt(a), fdiv(a,b), a/b]  in [n][2]   out [n][4]  self-check of the path's guard-free
 *                           fp64 sqrt/division (csrc/gel_physics.h) against the compiler's sequences
 *  kind 8: (angle, ratio) -> [fsin, fcos, sin, cos, flog, log]       in [n][2]   out [n][6]  the path's sincos / log
 *                           (csrc/gel_physics.h: fsincos, flog_ratio) beside the library's
 *  kind 9: 8 abscissae     -> kind 6 of each, looked up one after the other through the table interval kept from the
 *                           previous lookup (what the fused kernel does across a node's sweeps)   in [n][8]  out [n][8]
 *  kind 10: 8 altitudes    -> kind 5 (wn, we) of each, likewise                                    in [n][8]  out [n][16]
 *  kind 11: (q, p)       -> quatmult(q, p)                      in [n][8]   out [n][4]  (src/wrapper_coordinate.hpp:50-57)
 *  kind 12: (q, v)       -> quatrot(q, v) = vec(conj(q) (0, v) q)  in [n][7]  out [n][3]  (:70-78)
 *  kind 13: q            -> [conj(q), thrust direction quatrot(conj(q), (1, 0, 0))]  in [n][4]  out [n][7]  (:59-61,
 *           src/pybind_dynamics.cpp:62-63)
 *  kind 14: x            -> [fexp(x), exp(x)]                    in [n][1]   out [n][2]  self-check of the path's exp
 */
int gel_point_eval(int32_t kind, int32_t n, const double* in, const double* aux, int32_t aux_rows, double* out);

const char* gel_last_error(void);
const char* gel_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GELATO_AMD_H_ */
