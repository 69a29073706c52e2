/*
 * cwr_transport.h -- C ABI of the MI355X-native transport engine that drops in behind
 * clearwater_riverine's ClearwaterRiverine.update() time-step loop.
 *
 * The reference has no FFI: the seam is the Python method
 *   ClearwaterRiverine.update(update_concentration)      src/clearwater_riverine/transport.py:201-276
 * whose per-step work is LHS.update_values (linalg.py:34-156) -> csr_matrix (transport.py:215-218)
 * -> per constituent RHS.update_values (linalg.py:177-201) -> scipy spsolve (transport.py:249)
 * -> write-back (transport.py:252-264) -> _mass_flux (transport.py:406-429).
 * Each entry point below names the reference interface it replaces; INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary; every call returns an int status
 *     (0 = CWR_OK, < 0 = error; message via cwr_last_error()).
 *   - all array arguments are HOST pointers to C-contiguous buffers owned by the caller; the engine
 *     owns all device memory.  One engine = one GPU = one host thread at a time.
 *   - cell ids and face ("edge") ids are the reference's ids: real cells 0..nreal, ghost cells
 *     > nreal (io/hdf.py:257-269), faces in HDF order.  Concentration layout is x[cell * K + k]
 *     (constituents are the inner, coalesced dimension), float64.
 *   - time level t uses advection_coeff[t], coeff_to_diffusion[t], volume[t+1] on the left-hand
 *     side and volume[t], boundary terms of level t+1 on the right-hand side, exactly as
 *     linalg.py:61-66,84-89 and linalg.py:238-240,274 do.
 */
#ifndef CWR_TRANSPORT_H
#define CWR_TRANSPORT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cwr_engine cwr_engine;

enum {
  CWR_OK = 0,
  CWR_ERR_BAD_ARG = -1,        /* python: TypeError / ValueError raised by the wrapper's own checks   */
  CWR_ERR_HIP = -2,            /* a HIP runtime call failed                                          */
  CWR_ERR_NOT_CONVERGED = -3,  /* iterative solve did not reach tol in max_iter (spsolve has no such) */
  CWR_ERR_GHOST_COEFF = -4,    /* active ghost face with a zero coefficient: the reference raises a
                                  shape-mismatch ValueError at linalg.py:349-351                      */
  CWR_ERR_RCCL = -5,           /* an RCCL call failed (partitioned engines only)                     */
  CWR_ERR_STATE = -6,          /* call out of order (no flow field loaded, t out of range, ...)      */
  CWR_ERR_NONFINITE = -7       /* NaN/Inf met in the solve (spsolve would return NaNs silently)      */
};

/* flags for cwr_step() */
enum {
  CWR_STEP_MASS_FLUX = 1,      /* also evaluate the three per-face mass-flux arrays (transport.py:406-429) */
  CWR_STEP_PROFILE = 2,        /* bracket every operator launch with HIP events (see cwr_profile_read)     */
  CWR_STEP_FORCE_BICGSTAB = 4, /* skip the Jacobi fast path                                                */
  CWR_STEP_FORCE_JACOBI = 8,   /* never switch to BiCGSTAB (fails with CWR_ERR_NOT_CONVERGED instead)      */
  CWR_STEP_MASS_BALANCE = 16,  /* add this step's boundary-line mass fluxes to the device ledger (cwr_set_boundary_lines) */
  CWR_STEP_DETERMINISTIC = 32  /* this step's passes ping-pong between two vectors (nothing is relaxed in place): results
                                  are bitwise reproducible from run to run, as the reference's spsolve is (transport.py:249); the
                                  default chained in-place passes agree with them to <= 1e-10 but not bit for bit.  The engine
                                  still walks its tile chains (a tile takes its predecessor's rows from LDS; ranks of a partition
                                  too), and with up to 8 constituents it takes these passes whether the flag is given or not
                                  (1-4 % slower than in place there, as at 12-16 constituents, where the flag decides; more on
                                  very stiff steps).  Partitioned runs: every rank must give the same value; bitwise
                                  reproducible for the same partition (the tiling, hence the arithmetic, follows the ranks) */
};

typedef struct cwr_step_info {
  int32_t iterations;          /* BiCGSTAB iterations (0 when the Jacobi sweeps alone converged) */
  int32_t sweeps;              /* fused Jacobi sweeps */
  int32_t restarts;            /* BiCGSTAB restarts (true-residual verification rounds that went on iterating) */
  int32_t status;              /* CWR_OK or the error code also returned */
  int32_t operator_launches;   /* face-flux operator launches in this step */
  int32_t solver;              /* 0 = Jacobi sweeps only, 1 = BiCGSTAB only, 2 = sweeps then BiCGSTAB */
  double max_rel_residual;     /* max over constituents of ||D^-1 (b - A x)||_2 / ||D^-1 b||_2 */
  double solve_ms;             /* host wall time of the step, for information only */
  int32_t sweep_kernel;        /* which Jacobi kernel ran: 4 plain sweep, 5 J^2 pass, 6 tiled J^2 pass (x tile in LDS),
                                  7 one-launch solver of meshes up to 24 576 cells (k_small_jacobi: one workgroup per constituent up to
                                  4 096 cells, several with exchanged halo layers above), 0 none (BiCGSTAB only) */
  int32_t flags;               /* CWR_INFO_* bits: tolerance decisions that were NOT met exactly (0 in a clean step) */
  int32_t exchanges;           /* partitioned engines: neighbour halo exchanges of this step (0 on one GPU) */
  int32_t overlapped;          /* ... of which ran on the communication stream beside interior tiles */
  int32_t checks;              /* convergence checks = blocking host round trips (one all-reduce each when partitioned) */
  int32_t local_reps;          /* tile-local J^2 applications per visit the passes of this step used (0: no tiled pass) */
  int32_t chained;             /* 1: the passes relaxed in place along tile chains (not bitwise reproducible run to run);
                                  2: the passes walked the tile chains between two vectors (CWR_STEP_DETERMINISTIC, or <= 8 constituents);
                                  0: ping-pong passes in tile order, plain sweeps, the small-mesh solver or BiCGSTAB (0, 2: deterministic) */
} cwr_step_info;

/* bits of cwr_step_info.flags */
enum {
  CWR_INFO_LOOSE_RESIDUAL = 1,     /* BiCGSTAB stagnated within 100 x tol after 6 verified restarts and was accepted */
  CWR_INFO_ELEMENTWISE_MISSED = 2, /* the element-wise rule |x'-x| <= 1e6 tol |x| + tol max|x| (scaled, see cwr_step) was still
                                      violated after 3 tightened BiCGSTAB rounds; the norm criterion holds */
  CWR_INFO_ELEMENTWISE_CLAMPED = 4, /* the a-posteriori error factor F of this step (cwr_get_error_factors: the row-wise bound
                                      max((I - J)^-1 1) - 1, or ||J||_inf / (1 - ||J||_inf) where that is smaller) is so large -- or no
                                      bound at all -- that the step's max-norm forward-error bound F (ew_rel + ew_abs) max|x| exceeds
                                      (1e6 tol + tol) max|x|.  Since round 6 only the ABSOLUTE part of the element-wise rule is floored
                                      at s = 1e-3 (it is at rounding size there); the relative part follows s = 0.3 / F down to
                                      1e-13: at tol = 1e-12 the bit is raised for F > ~7e8, or for a level without any factor
                                      (until round 5: for every F > 300, e.g. river-band meshes with a dry cell at dt = 3600 s) */
  CWR_INFO_SMALL_FALLBACK = 8      /* a part of the one-launch solver for meshes of 4 097 .. 24 576 cells waited for another longer
                                      than its bound once (CU shortage): the state was restored, the step -- and every later one,
                                      all carrying this bit -- taken by the multi-launch passes */
};

int32_t cwr_abi_version(void);
/* Rows of one tile of the engine's dominant sweep kernel for K constituents (64 at K = 16, 256 at K = 1, ...).  A host
 * wrapper that chooses the cell numbering it creates the engine with (engine.py's cell_order: a space-filling curve) may
 * arrange the cells of every tile-sized window by their work (ordering.balance_windows sorts them by the number of cells
 * within two face steps): the kernel's waves loop to the longest of their rows.  0 = no arrangement wanted.  Purely a speed
 * matter. */
int32_t cwr_tile_rows(int32_t n_constituents);
/* Rows from which an engine with K constituents links its tiles into chains along the flow and relaxes in place (see "tiling and
 * the chained passes" below): 1.75 tiles per block of its persistent grid (CWR_CHAIN_MIN_TILES), four resident blocks per CU.  A
 * host wrapper that chooses the cell numbering before it creates the engine -- lanes along the flow for engines that chain, an
 * isotropic space-filling curve for those that do not -- asks this, so that numbering and engine follow one threshold.  No handle,
 * no GPU and no HIP call (the MI355X's 256 CUs are assumed).  Purely a speed matter. */
int32_t cwr_chain_min_rows(int32_t n_constituents);

/* ---- construction -------------------------------------------------------------------------------
 * Replaces LHS.__init__ (linalg.py:18-32: internal / real face index sets) and RHS.__init__
 * (linalg.py:159-175: ghost face set).  Local cell numbering:
 *   [0, n_owned)                      real cells whose rows this engine solves
 *   [n_owned, n_owned + n_halo)       real cells owned by other ranks (n_halo = 0 on one GPU)
 *   [n_owned + n_halo, n_cells)       ghost (boundary) cells
 * On one GPU n_owned = nreal + 1 and the numbering is the reference's own.
 * face1/face2: (n_edges) int32, face1 must be a real cell (as in every HEC-RAS file the reference reads).
 * n_constituents = K of every (.., K) array at this boundary.  Internally the engine may carry more columns: K = 3, 5, 7 run as
 * 4, 6, 8 and K > 8 as the next multiple of 4 (zero columns behind the caller's, which solve to zero and are stripped at every
 * read-out: the kernels' wide mappings need an even K, best a multiple of 4, and the reference's cost is linear in K,
 * transport.py:231; CWR_K_PAD=0 switches it off).  Visible only through cwr_state_device_ptr: cwr_state_row_stride.
 */
int32_t cwr_create(int32_t n_owned, int32_t n_halo, int32_t n_cells, int32_t n_edges,
                   int32_t n_constituents, const int32_t* face1, const int32_t* face2,
                   int32_t device, cwr_engine** out);
void cwr_destroy(cwr_engine* e);
const char* cwr_last_error(const cwr_engine* e);   /* e may be NULL: last error of a failed cwr_create */

/* ---- flow field (inputs of the path) ----------------------------------------------------------
 * cwr_load_flow_field derives the coefficients ON DEVICE, replacing
 * WQVariableCalculator.calculate else-branch (utilities.py:513-541):
 *   advection_coeff = face_flow * sign(|edge_velocity|)            (float32)
 *   edge_vertical_area = advection_coeff / edge_velocity, NaN -> 0  (float32)
 *   coeff_to_diffusion = float32(area * D) / face_to_face_dist       (float64)
 * face_flow, edge_velocity: (T, n_edges) float32; volume: (T, n_cells) float32;
 * dt: (T) float64 (seconds, last entry unused); face_to_face_dist: (n_edges) float64
 * (utilities.py:261-278, computed by the host wrapper); D = mesh.attrs['diffusion_coefficient'].
 */
int32_t cwr_load_flow_field(cwr_engine* e, int32_t n_times, const float* face_flow,
                            const float* edge_velocity, const float* volume, const double* dt,
                            const double* face_to_face_dist, double diffusion_coefficient);
/* Same, from the reference's already-derived Dataset variables 'advection_coeff' (T,E) f32 and
 * 'coeff_to_diffusion' (T,E) f64 (variables.py:30-33): the drop-in route from an existing xarray mesh. */
int32_t cwr_load_coefficients(cwr_engine* e, int32_t n_times, const float* advection_coeff,
                              const double* coeff_to_diffusion, const float* edge_velocity,
                              const float* volume, const double* dt, double diffusion_coefficient);
/* Windowed residency ("time-series streaming"): the device holds a RING of window_levels levels instead of all
 * n_times (~37 MB per level at 1 M cells: a 10 801-stamp file as the reference's tests/data/simple_test_cases/plan01_10x5 does
 * not fit whole; the reference's reader windows a file by datetime_range, io/hdf.py:149-191, and derives per level,
 * utilities.py:513-541).  Single engines, and (ABI 7) the ranks of a partition: every rank opens the same window and loads the same levels
 * (its slices of them) at the same steps; the zero-coefficient flag and ||J||_inf of an arriving level are all-reduced on the
 * communication stream where the level is loaded (one sum all-reduce of world x 3 doubles per level, beside the steps), the row-wise
 * error factor is taken where the step runs (collectively).  Attach the communicator BEFORE the first level is sent to the ring.
 * cwr_flow_window_open: n_times levels in the run, a ring of window_levels >= 2 of them (level t lives in slot t % window_levels);
 *   dt (n_times), face_to_face_dist (n_edges) and D as for cwr_load_flow_field.  Replaces a loaded flow field.
 * cwr_flow_window_load: levels t0 .. t0 + n_levels - 1 (face_flow, edge_velocity: (n_levels, n_edges) f32; volume: (n_levels,
 *   n_cells) f32) into their slots, replacing what those held.  Only NOTED by this call: the next cwr_step sends the levels to a stream
 *   of the engine's own -- behind its batch of passes, while the host would otherwise wait for the convergence check, unless the step
 *   itself needs them (then first) -- where upload, coefficient derivation, the zero-coefficient flag of every level and ||J||_inf of
 *   every step the arrived levels complete run beside the steps (behind every kernel already enqueued that may still read the levels
 *   being replaced).  Any other call that needs a level (cwr_get_coefficients, cwr_apply, ...) and cwr_synchronize send them at once.
 *   The host arrays must stay untouched until a cwr_step that reads the levels, or cwr_synchronize, has returned, and only
 *   page-locked arrays (cwr_host_register) make the upload itself asynchronous.  cwr_step(t) needs levels t and t + 1 in the ring (CWR_ERR_STATE otherwise) and waits for them on the
 *   device; results are those of the all-resident engine, bit for bit with CWR_STEP_DETERMINISTIC.  The row-wise error factor of a
 *   step (cwr_get_error_factors) is taken when the step runs, and only where its norm form exceeds 4. */
int32_t cwr_flow_window_open(cwr_engine* e, int32_t n_times, int32_t window_levels, const double* dt,
                             const double* face_to_face_dist, double diffusion_coefficient);
int32_t cwr_flow_window_load(cwr_engine* e, int32_t t0, int32_t n_levels, const float* face_flow,
                             const float* edge_velocity, const float* volume);
/* Read back the device-resident coefficients of level t (parity check of the on-device derivation). */
int32_t cwr_get_coefficients(cwr_engine* e, int32_t t, float* advection_coeff, double* coeff_to_diffusion);

/* ---- boundary values ----------------------------------------------------------------------------
 * Ghost-cell columns of Constituent.input_array (constituents.py:31,153-164), all time levels:
 * ghost_conc is (T, n_ghost, K) float64 with n_ghost = n_cells - n_owned - n_halo; 0 = "no boundary
 * value" (the reference's sentinel, transport.py:258-264). */
int32_t cwr_load_boundary(cwr_engine* e, int32_t n_times, const double* ghost_conc);
/* (ABI 7) ghost_conc == NULL: n_times levels of zeros ("no boundary value") are allocated and the values arrive later, a few levels at
 * a time -- a run that streams its flow field (cwr_flow_window_load) need not hold all levels of its boundary values on the host. */
/* One level only (blocking upload). */
int32_t cwr_set_boundary_level(cwr_engine* e, int32_t t, const double* ghost_conc_level);
/* (ABI 7) Levels t0 .. t0 + n_levels - 1, (n_levels, n_ghost, K) float64, into their rows of the array cwr_load_boundary allocated
 * (input_array[t, ghost cells], constituents.py:153-164; the reference holds all T levels in RAM, constituents.py:39-48).  On an engine
 * with a flow-field window the call only NOTES the pointer, exactly as cwr_flow_window_load does: the next cwr_step enqueues the copy on
 * the engine's flow stream (first where the step reads level t + 1 of them, otherwise behind its batch of passes) and the step that
 * reads the rows waits for them on the device.  The host array must stay untouched until a cwr_step that reads the levels, or
 * cwr_synchronize, has returned; page-locked arrays (cwr_host_register) make the copy asynchronous.  Other engines: a blocking upload. */
int32_t cwr_boundary_window_load(cwr_engine* e, int32_t t0, int32_t n_levels, const double* ghost_conc);

/* ---- state --------------------------------------------------------------------------------------
 * cwr_set_state: concentrations of the owned real cells at the current level, (n_owned, K).  Used for
 * the initial condition (constituents.py:94-98) and for the per-step override of update()'s
 * `update_concentration` argument (transport.py:233-236).
 * cwr_get_state: full (n_cells, K) row of mesh[name][t+1] after a step: solved real cells, ghost cells
 * = boundary value where non-zero, NaN elsewhere (transport.py:252-264, constituents.py:39-48);
 * halo rows hold the neighbours' values of the last exchange. */
int32_t cwr_set_state(cwr_engine* e, const double* conc_owned);
int32_t cwr_get_state(cwr_engine* e, double* conc_all_cells);
/* Non-zero entries of input_array[level >= 1] on REAL cells (point sources / fixed concentrations inside the domain):
 * the reference overwrites the solved level t+1 with them before the mass fluxes are taken (transport.py:258-264) and
 * uses them as x_t of the next step (linalg.py:199-200).  Sparse triplets sorted by level: level[i], row[i] (an owned real
 * cell), values[i*K .. i*K+K) with 0 = "no input for this constituent".  Replaces every earlier call.
 * Partitioned engines: COLLECTIVE -- every rank calls it with the entries of the cells it owns (also with none): the ranks
 * agree on the levels that carry inputs anywhere, because such a step's closing kernels, which include a halo exchange, must
 * be taken on the same path by all of them. */
int32_t cwr_load_real_inputs(cwr_engine* e, int32_t n_entries, const int32_t* level, const int32_t* row,
                             const double* values);

/* ---- device-resident reaction hook (callers: examples/02_...tsm.ipynb cell[39] run_n_timesteps) -----
 * The reference's coupling loop overrides c[t, 0:n] per constituent from a host reaction model before update()
 * (transport.py:233-236).  These two entry points keep that step in HBM:
 * cwr_react_linear: c[cell, :] <- M c[cell, :] on every owned cell, M (K, K) row-major host array (first-order
 *   decay on the diagonal, pairwise exchange off it) -- the built-in stand-in for a TSM/NSM kinetics kernel;
 * cwr_state_device_ptr: the device pointer of the (n_cells, K) float64 state and the engine's hipStream_t, for a
 *   caller-supplied HIP reaction kernel launched between two cwr_step() calls (row stride: cwr_state_row_stride).  Rows are in the cell numbering this
 *   engine was CREATED with: rows [0, n_owned) are the real cells face1/face2 of cwr_create refer to (a host wrapper that
 *   renumbers cells before cwr_create -- engine.py's cell_order -- must hand its row map to the kernel's author:
 *   TransportEngine.state_row_order()).  The pointer never changes; once it has been handed out the engine assumes the
 *   state may have been rewritten before every later step (partitioned engines then never skip the start-of-step halo
 *   exchange).  Work enqueued on the returned stream is ordered with the engine's own kernels. */
int32_t cwr_react_linear(cwr_engine* e, const double* reaction_matrix);
int32_t cwr_state_device_ptr(cwr_engine* e, void** state, void** stream);
/* Doubles per row of that state: n_constituents, or the padded count the engine runs with (cwr_create); columns >= n_constituents
 * hold zeros (NaN on ghost rows) and must be left alone. */
int32_t cwr_state_row_stride(const cwr_engine* e);

/* ---- the face-flux operator (exported for parity tests and roofline timing) ---------------------
 * y = A x with A the matrix LHS.update_values(mesh, t) + csr_matrix build (linalg.py:34-156,
 * transport.py:215-218), evaluated matrix-free per cell over the CSR face adjacency.
 * x: (n_owned + n_halo, K), y: (n_owned, K). */
int32_t cwr_apply(cwr_engine* e, int32_t t, const double* x, double* y);
/* b of RHS.update_values(solution=x_t, mesh, t) (linalg.py:177-201,262-275,354-406). x_t, b: (n_owned, K). */
int32_t cwr_rhs(cwr_engine* e, int32_t t, const double* x_t, double* b);

/* ---- one time step ------------------------------------------------------------------------------
 * Replaces the body of ClearwaterRiverine.update() for all constituents at once
 * (transport.py:209-273): operator set-up for level t, right-hand side, implicit solve converging to the
 * spsolve solution (fully fused Jacobi sweeps while their measured contraction is fast enough, otherwise
 * Jacobi-scaled BiCGSTAB; the K systems share A), write-back of the real cells and of the ghost cells,
 * optional mass flux.  State advances from level t to t+1.
 * tol: target for ||D^-1 (b - A x)||_2 / ||D^-1 b||_2 per constituent (e.g. 1e-12); max_iter bounds
 * sweeps and BiCGSTAB iterations each.  info may be NULL.
 * On top of the norm criterion every cell and constituent must satisfy |x'_i - x_i| <= s (1e6 tol |x'_i| + tol max|x'|)
 * for one more Jacobi sweep x -> x', with s = 0.3 / F (at most 0.1) and F the a-posteriori factor of this step's Jacobi
 * iteration matrix J, ||x* - x'||_inf <= F ||x' - x||_inf (cwr_get_error_factors): the ROW-WISE bound max((I - J)^-1 1) - 1, taken
 * from a few sweeps of the Neumann series when the flow field is loaded (partitioned engines: over the ranks, so that every rank
 * holds the factor of the global matrix), or the norm form ||J||_inf / (1 - ||J||_inf) where that is smaller.  That gives a
 * RIGOROUS max-norm forward error of 0.3 (1e6 tol + tol) max|x| (3e-7 of the largest concentration at tol = 1e-12) -- also on
 * meshes with dry or nearly dry cells, whose worst row sum (> 1 beside a dry cell) admits no norm bound; that every cell is also
 * within 1e-6 of ITS OWN value down to the 1e-12 max|x| floor -- plume fronts many decades below the peak, which a 2-norm cannot
 * see -- is what the per-cell form of the rule buys empirically (tests: element-wise against spsolve output up to CFL 180 and on
 * river-band meshes with dry cells at dt = 3600 / 14 400 s).  Floors (round 6): the relative part s 1e6 tol is followed down to
 * 1e-13 (a converged sweep repeats itself to a few 1e-16 of a cell's own size), the absolute part s tol max|x'| is held at
 * s = 1e-3 (rounding size); the bound becomes (0.3 * 1e6 tol + 1e-3 F tol) max|x| and CWR_INFO_ELEMENTWISE_CLAMPED is set only when
 * it exceeds (1e6 tol + tol) max|x| (F > ~7e8) or the level has no factor.  CWR_EW_SPLIT=0: both parts floored at s = 1e-3, the
 * bit set for every F > 300, as until round 5.
 * A step that fails (CWR_ERR_NOT_CONVERGED, CWR_ERR_NONFINITE, CWR_ERR_GHOST_COEFF) leaves the state exactly as it
 * found it: it may be retried with another tolerance, iteration budget or solver.
 * The call returns as soon as convergence is known: the ghost write-back and flux kernels that close the step may still
 * be running on the engine's stream.  Every read-out (cwr_get_state, cwr_get_mass_flux, ...), every later step and
 * cwr_synchronize are ordered behind them; cwr_step_info.solve_ms is the time until convergence was known. */
int32_t cwr_step(cwr_engine* e, int32_t t, double tol, int32_t max_iter, int32_t flags,
                 cwr_step_info* info);
/* The three (n_edges, K) arrays of the last step taken with CWR_STEP_MASS_FLUX
 * (advection, diffusion, total: transport.py:419-429).  Any pointer may be NULL. */
int32_t cwr_get_mass_flux(cwr_engine* e, double* advection, double* diffusion, double* total);
/* ||J||_inf of the Jacobi iteration matrix J = I - D^-1 A of every step the loaded flow field allows: norms[t] for step t
 * (t = 0 .. T-2; norms[T-1] = 0), the largest row sum of |offd| / diag over this engine's computed rows -- evaluated on the
 * device when the flow field is loaded.  It scales the element-wise stopping rule of cwr_step (see there).  A partitioned
 * engine holds the MAXIMUM over the ranks (one all-reduce when the communicator is attached / a flow field is loaded with
 * a communicator attached: collective calls), so that all ranks apply the rule of the global matrix -- the one a single
 * engine would apply.  cwr_set_jacobi_norms overrides the values (a caller with bounds of its own).
 * (No reference counterpart: spsolve is direct, transport.py:249.) */
int32_t cwr_get_jacobi_norms(cwr_engine* e, int32_t n_times, double* norms);
int32_t cwr_set_jacobi_norms(cwr_engine* e, int32_t n_times, const double* norms);
/* The factor the element-wise rule of cwr_step is scaled by: factors[t] = F_t with ||x* - x'||_inf <= F_t ||x' - x||_inf for
 * a Jacobi sweep x -> x' of step t (x*: the solution spsolve returns, transport.py:249).  ||J||_inf / (1 - ||J||_inf) where that is
 * finite, replaced by the row-wise bound max((I - J)^-1 1) - 1 <= max(w_m - 1) / (1 - ||J^m 1||_inf), taken from a few sweeps of the
 * Neumann series when the flow field is loaded, where that is smaller: meshes with dry or nearly dry cells, whose worst row sum
 * (> 1 beside a dry cell) says nothing about the error of a sweep.  Partitioned engines run those sweeps over the ranks (one halo
 * exchange per `exchange_every` sweeps, one all-reduce per check -- at load / attach time only: both calls are collective) and
 * hold the factors of the GLOBAL matrix, the ones a single engine holds.  cwr_set_jacobi_norms resets the factors to the norm
 * form of the caller's values. */
int32_t cwr_get_error_factors(cwr_engine* e, int32_t n_times, double* factors);

/* ---- measurement --------------------------------------------------------------------------------
 * cwr_time_apply: `reps` back-to-back launches of the operator of level t on device-resident vectors,
 * timed with HIP events on the engine's own stream; variant 0 = the sweep kernel the last step used (the plain
 * Jacobi sweep, or the J^2 double sweep), 1 = scatter form with global float64 atomics (A/B only),
 * 2 = BiCGSTAB's first product.  avg_us = mean launch duration.
 * cwr_profile_read: event-timed totals of the operator launches made by steps run with
 * CWR_STEP_PROFILE since the last call: number of launches and their summed duration. */
int32_t cwr_time_apply(cwr_engine* e, int32_t t, int32_t variant, int32_t reps, double* avg_us);
int32_t cwr_profile_read(cwr_engine* e, int64_t* launches, double* total_us);
/* (ABI 7) The communication side of the steps taken with CWR_STEP_PROFILE since the last call (partitioned engines; zeros otherwise):
 * out[0..1] exchanges with nothing beside them: count, microseconds on the communication stream (grouped send / receive + unpack: a
 * peer's lateness included); out[2..3] exchanges that ran beside compute; out[4..5] all-reduces; out[6..7] convergence checks: count,
 * microseconds of host wall time inside them.  The reference has nothing to mirror (serial K-loop on one host, transport.py:231-249):
 * this is what bench.py --gpus N reports per rank so that a scaling run explains itself. */
int32_t cwr_comm_profile_read(cwr_engine* e, double out[8]);
int32_t cwr_synchronize(cwr_engine* e);
/* Algorithmic bytes of one operator launch (read, written), as DESIGN.md defines them. */
int32_t cwr_apply_bytes(const cwr_engine* e, int64_t* bytes_read, int64_t* bytes_written);

/* ---- tiling and the chained passes (diagnostics / tuning; no reference counterpart: spsolve is direct, transport.py:249) ----
 * The dominant sweep kernel walks tiles of cwr_tile_rows(K) rows with a persistent grid.  From 1.75 tiles per block up
 * (CWR_CHAIN_MIN_TILES; cwr_tiling_info returns tiles and blocks: a host wrapper that chooses the cell numbering asks it) the
 * engine links the tiles into chains along the flow of the level being solved and relaxes IN PLACE along them (block
 * Gauss-Seidel along the flow without any block waiting for another; what consecutive tiles of a block's list share is
 * carried over in LDS; re-derived every CWR_CHAIN_REFRESH = 64 levels; CWR_NO_CHAINS=1: ping-pong passes in tile order).
 * cwr_tiling_info: out = {tiled pass available, tiles, blocks of its grid, rows per tile}.
 * cwr_set_tile_schedule: install a caller's schedule instead: sched[it * n_lists + b] = it-th tile of block b, -1 = end of
 *   the list; n_lists must equal the grid, every tile must appear exactly once (checked).  depth = 0: back to the engine's own.
 * cwr_get_tile_schedule: info = {depth, n_lists, level the engine built it for (-1: none, or the caller's), schedules built
 *   so far}; out (may be NULL) receives the depth * n_lists entries. */
int32_t cwr_tiling_info(cwr_engine* e, int32_t out[4]);
int32_t cwr_set_tile_schedule(cwr_engine* e, int32_t n_lists, int32_t depth, const int32_t* sched);
int32_t cwr_get_tile_schedule(cwr_engine* e, int32_t info[4], int32_t* out, int64_t out_cap);

/* ---- domain decomposition (one process per GPU, RCCL over xGMI) --------------------------------
 * A partitioned engine is created with n_owned = the rows it COMPUTES (its own core range plus, with deep
 * halos, the inner halo layers it replays) and n_halo = the outermost, read-only halo layer.
 * cwr_comm_unique_id: rank 0 creates the 128-byte RCCL unique id, the host broadcasts it.
 * cwr_attach_comm: joins the communicator and installs the halo exchange of this rank:
 *   n_core                        rows [0, n_core) are the rank's own cells: only they enter inner products,
 *                                 are sent to neighbours, and own faces in the mass-flux output
 *   exchange_every                Jacobi sweeps between two exchanges (= halo depth; 1 = before every sweep)
 *   peers[i]                      rank of the i-th neighbour
 *   send_ptr[i] .. send_ptr[i+1]  slice of send_cells (local core ids) packed for peers[i]
 *   recv_ptr[i] .. recv_ptr[i+1]  slice of recv_cells (local ids in [n_core, n_owned + n_halo)) filled by peers[i]
 * Every RCCL call of an engine -- ncclGroupStart/Send/Recv/End between a pack and an unpack kernel, the ncclAllReduce of the inner
 * products and checks -- is issued on ONE communication stream of the engine's own, ordered against the engine's stream by
 * events; an exchange runs beside the interior tiles of the pass (sweep, flux kernel) that needs it.  BiCGSTAB exchanges before
 * every operator launch.
 * world == 1 with n_peers == 0 attaches a STAND-ALONE rank: the row layout and launch structure of one rank of a larger partition
 * (n_core < n_owned: replayed layers; n_halo read-only rows) that never exchanges -- the rows outside the core keep what the caller
 * put there (measurement of a rank's compute side: tools/rank_step_profile.py).
 * Hosting: one process per GPU is the supported arrangement.  Several engines in ONE process, each driven by its own thread, work
 * as far as this library goes (every entry point switches its thread to the thread-local stream-capture mode, so one engine's
 * hipGraph capture does not refuse the other's calls) -- but the streams of one process share its copy-engine rings, which
 * execute in order: a communication layer that parks a copy behind a wait on a peer (the test stand-in does; RCCL's kernels do
 * not) can then block its process mate's copies. */
int32_t cwr_comm_unique_id(uint8_t id_out[128]);
int32_t cwr_attach_comm(cwr_engine* e, int32_t rank, int32_t world, const uint8_t unique_id[128], int32_t n_core,
                        int32_t exchange_every, int32_t n_peers, const int32_t* peers, const int32_t* send_ptr,
                        const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells);
/* Diagnostics of the communication path.  count > 0: a grouped ncclSend / ncclRecv of `count` doubles from this rank to
 * itself on the engine's communication stream, bracketed by the two events of the overlapped exchange, compared bit for
 * bit -- the RCCL point-to-point signatures and the stream / event plumbing, executable with ONE rank.  count = 0: only
 * the statistics.  overlapped_exchanges (may be NULL): halo exchanges that ran beside the interior tiles of a pass so far. */
int32_t cwr_comm_selftest(cwr_engine* e, int32_t count, int64_t* overlapped_exchanges);

/* ------------------------------------------------------------------ output side
 * Mass balance on the device (replaces the host post-processing of postproc_util.py:21-166, which needs the whole
 * (T, ncell) state and (T, nedge) flux history in RAM).
 * cwr_set_boundary_lines: the faces of every boundary-condition line (postproc_util.py:84-90: the 'Face Index' rows
 *   of boundary_data grouped by 'BC Line ID'), CSR over line_ptr[n_lines + 1]; clears the ledger.
 * A cwr_step taken with CWR_STEP_MASS_BALANCE adds, for every line and constituent, that step's
 *   sum over the line's faces of total_mass_flux (transport.py:414-429), its part <= 0 (inflow) and its part >= 0
 *   (outflow) to the ledger -- postproc_util.py:99-139; NaN propagates as in the reference.  The per-face flux arrays
 *   are not materialised for this.  Partitioned engines add the faces whose face1 they own; the host adds the ranks.
 * cwr_get_mass_balance: ledger[(line * 3 + q) * K + k], q = 0 total, 1 inflow part, 2 outflow part.
 * cwr_domain_mass: out[k] = sum over this engine's own real cells of volume[t_level, c] * state[c, k], out[K] = sum of
 *   volume[t_level, c]  (postproc_util.py:36-57, with the CURRENT state standing for level t_level). */
int32_t cwr_set_boundary_lines(cwr_engine* e, int32_t n_lines, const int32_t* line_ptr, const int32_t* line_faces);
int32_t cwr_reset_mass_balance(cwr_engine* e);
int32_t cwr_get_mass_balance(cwr_engine* e, double* ledger);
int32_t cwr_domain_mass(cwr_engine* e, int32_t t_level, double* out);

/* Streamed output (replaces the RAM-resident (T, ncell) / (T, nedge) float64 arrays of constituents.py:28-48 and the
 * write-everything-at-the-end of io/outputs.py): a ring of pinned host slots filled by asynchronous copies on a second
 * HIP stream while the next steps compute.
 * cwr_output_open: n_slots >= 1 slots of n_out * K doubles (+ 3 * n_edges * K with with_flux != 0).  row_order
 *   (n_out engine cell ids, or NULL for 0..n_out-1) selects and orders the state rows that are written.
 * cwr_output_push: snapshot the current state (and the flux arrays of the last step taken with CWR_STEP_MASS_FLUX)
 *   CONSTITUENT-MAJOR -- slot[k * n_out + i] = state[row_order[i], k], the (1, nface) chunk of constituent k's
 *   (time, nface) array -- and start its copy to the host; returns the slot.  Blocks only while that slot is still held.
 * cwr_output_wait: block until the slot's copy has landed; returns host pointers valid until cwr_output_release.
 *   flux (if any) = three consecutive (K, n_edges) blocks: advection, diffusion, total.
 * cwr_output_wait and cwr_output_release may be called from a second (writer) thread. */
int32_t cwr_output_open(cwr_engine* e, int32_t n_slots, int32_t with_flux, int32_t n_out, const int32_t* row_order);
int32_t cwr_output_push(cwr_engine* e, int32_t* slot);
/* The same snapshot copied straight into the CALLER's arrays instead of the ring slot: state_dst (K, n_out) and flux_dst
 * (3, K, n_edges; required when the ring was opened with_flux) -- e.g. row t+1 of a (T, K, ncell) history block, so that no host
 * copy is left between the device and mesh[name][t+1] (transport.py:252-273).  The copies are asynchronous when the
 * destination is page-locked (cwr_host_register: hipHostRegister with the engine's HIP runtime) and staged by HIP otherwise;
 * cwr_output_wait(slot) returns the two destinations once they are complete.
 * Round 5: a snapshot of up to 8 MB (CWR_OUTPUT_DIRECT_MB; 4 until round 6) whose destinations are page-locked is written in place by the snapshot
 * kernels through the destinations' device aliases (no staging buffer, no copy command: at the reference's own mesh sizes the
 * copies' submission cost more than the bytes); the destinations must then stay registered until cwr_output_wait has returned
 * for the slot -- as before -- and cwr_output_close drains the engine's stream as well as the ring's. */
int32_t cwr_output_push_into(cwr_engine* e, double* state_dst, double* flux_dst, int32_t* slot);
int32_t cwr_host_register(void* ptr, int64_t bytes);
int32_t cwr_host_unregister(void* ptr);
int32_t cwr_output_wait(cwr_engine* e, int32_t slot, const double** state, const double** flux);
int32_t cwr_output_release(cwr_engine* e, int32_t slot);
int32_t cwr_output_close(cwr_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* CWR_TRANSPORT_H */
