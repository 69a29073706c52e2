// cwr_engine_state.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): RCCL entry points (dlopen), process-exit guard, and the engine object.
#pragma once
namespace {

// (thread_local: two threads creating engines each keep their own message -- "no global state")
thread_local std::string g_create_error;
// compute units of the device the process last created an engine on (0: none yet); cwr_chain_min_rows reads it
std::atomic<int> g_n_cu{0};

// ---- RCCL, resolved lazily with dlopen so that a single-GPU engine has no RCCL dependency at all ----
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    // CWR_RCCL_LIB: explicit library path (the tests point it at a shared-memory stand-in so that several ranks
    // can share ONE GPU, which RCCL itself refuses)
    if (const char* over = getenv("CWR_RCCL_LIB")) lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { if (lib) break; lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!lib) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define CWR_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
    if (!field) { err = std::string("librccl lacks ") + name; return false; }
    CWR_SYM(GetUniqueId, "ncclGetUniqueId") CWR_SYM(CommInitRank, "ncclCommInitRank")
    CWR_SYM(CommDestroy, "ncclCommDestroy") CWR_SYM(Send, "ncclSend") CWR_SYM(Recv, "ncclRecv")
    CWR_SYM(AllReduce, "ncclAllReduce") CWR_SYM(GroupStart, "ncclGroupStart")
    CWR_SYM(GroupEnd, "ncclGroupEnd") CWR_SYM(GetErrorString, "ncclGetErrorString")
#undef CWR_SYM
    return true;
  }
};
Rccl g_rccl;

// ---- process exit ----
// The HIP runtime tears itself down from exit handlers of its own.  A host that still calls into this library after that -- the
// finalizer of a garbage-collected wrapper object, a static destructor of the embedding program -- would reach hip* on a dead
// runtime.  The first successful cwr_create registers ONE exit handler; exit handlers run in reverse order of registration and
// the runtime registered its own during the hip calls before that point, so this one runs BEFORE the runtime goes down.  It only
// raises a flag: from then on the three releasing entry points (cwr_destroy, cwr_output_close, cwr_host_unregister) return
// without touching HIP -- the process is about to give everything back anyway.  Nothing is destroyed here: a stream that waits
// for a dead peer must not keep the process from exiting.
std::atomic<bool> g_down{false};
std::atomic<bool> g_exit_hooked{false};
void on_process_exit() { g_down.store(true); }

constexpr int NCCL_FLOAT64 = 8;   // ncclDataType_t::ncclFloat64
constexpr int NCCL_SUM = 0;       // ncclRedOp_t::ncclSum

}  // namespace

struct cwr_engine {
  int dev = 0;
  hipStream_t stream = nullptr;
  int n_owned = 0, n_halo = 0, n_real = 0, n_cells = 0, n_ghost = 0, E = 0, K = 0;
  int Ku = 0;                   // the caller's constituents; K >= Ku is the engine's internal row width (pad_constituents: zero columns behind Ku)
  int n_core = 0;               // rows this rank owns (<= n_owned = rows it computes); inner products, results
  int exch_every = 1;           // Jacobi sweeps between two halo exchanges (= halo depth)
  int VW = 1, G = 1, R = 1;
  int max_degree = 0;            // most faces on one row
  int nnz = 0, U = 1, ntiles = 0, apply_grid = 0, last_apply_grid = 0, stage_cap = 0, cu_cap = 8;
  double* d_partial = nullptr;   // [max grid][4][K] per-block inner-product partials
  size_t apply_lds = 0;
  // static topology
  int32_t *d_f1 = nullptr, *d_f2 = nullptr, *d_ptr = nullptr, *d_ent_edge = nullptr, *d_ent_nb = nullptr;
  int32_t* d_face_orig = nullptr;          // internal face index -> reference face id (k_faces_in / k_faces_out)
  int32_t* d_face_pos = nullptr;           // reference face id -> internal face index
  uint8_t* d_sq_fast = nullptr;             // 1 where k_sq_numeric may take a row through its branch-free path
  uint8_t* d_row_ghost = nullptr;          // 1 where a computed row has a boundary (ghost) face
  std::vector<int32_t> bad_level;          // per time level: the zero-coefficient precondition is violated (k_check_ghost_levels)
  std::vector<int32_t> h_face_pos;         // reference face id -> internal face index
  // flow field: all T levels resident in HBM (W == T), or -- windowed (round 5: cwr_flow_window_open / _load) -- a ring of W < T
  // levels, level t in slot t % W, filled on a stream of its own beside the steps
  int T = 0, T_bc = 0;
  int W = 0;                               // levels the device arrays hold
  int flow_cap = 0;                        // ... and are allocated for
  bool windowed = false;
  float *d_adv = nullptr, *d_vel = nullptr, *d_vol = nullptr;
  double* d_dif = nullptr;
  size_t slot(int t) const { return windowed ? (size_t)(t % W) : (size_t)t; }
  float* adv_l(int t) const { return d_adv + slot(t) * (size_t)E; }
  double* dif_l(int t) const { return d_dif + slot(t) * (size_t)E; }
  float* vel_l(int t) const { return d_vel + slot(t) * (size_t)E; }
  float* vol_l(int t) const { return d_vol + slot(t) * (size_t)n_cells; }
  std::vector<int32_t> slot_level;         // windowed: the level every slot holds (-1: none)
  hipStream_t flow_stream = nullptr;       // windowed: upload, derivation, norms of the incoming levels
  std::vector<hipEvent_t> ev_level;        // [W] recorded on flow_stream when the slot's level is complete
  hipEvent_t ev_evict = nullptr;           // engine stream -> flow stream: every user of the level being replaced is done
  float *d_in_f = nullptr, *d_flow_l = nullptr;   // staging of ONE level: reference face order in, face flows in internal order
  double* d_dist = nullptr;                // face_to_face_dist in internal face order (kept by windowed engines)
  unsigned long long* d_jn = nullptr;      // [T] ||J||_inf bit patterns per step (windowed)
  double* d_lvlx = nullptr;                // partitioned + windowed: [W][world][3] a slot's level scalars laid out for their all-reduce (k_pack_level)
  std::vector<hipEvent_t> ev_lvl_local;    // [W] recorded on flow_stream when a rank's own scalars of the slot's level are packed
  int32_t* d_bad = nullptr;                // [T] zero-coefficient flags per level (windowed)
  double* d_lvl_view = nullptr;            // the device's address of h_lvl
  double* h_lvl = nullptr;                 // page-locked [T][2]: {||J||_inf of step t, flag of level t}: where flow_stream leaves them
  // loads asked for and not yet enqueued: cwr_flow_window_load only notes them; cwr_step enqueues them BEHIND the step's batch, while
  // the host would otherwise spin on the convergence check -- the ~0.15 ms of host calls a level costs (three copies from page-locked
  // memory, kernels, events) then overlap the step's passes instead of standing between two steps (profiles/r05_window.txt)
  struct PendingLoad { int t0, n; const float *ff, *ev, *vol; };
  std::vector<PendingLoad> pending_loads;
  // (round 6) boundary values of levels that travel with the flow-field ring (cwr_boundary_window_load): noted like the loads above,
  // copied on the flow stream into their rows of d_bc (all T_bc levels stay allocated: n_ghost x K doubles each), one event behind them
  struct PendingBc { int t0, n; const double* v; };
  std::vector<PendingBc> pending_bc;
  hipEvent_t ev_bc = nullptr;              // recorded on flow_stream behind the boundary rows of a flush
  bool bc_event_pending = false;           // ... and not yet waited for by the engine's stream (the next cwr_step does)
  double* d_bc_stage = nullptr;            // padded constituents (K > Ku): the caller's rows land here first
  size_t bc_stage_cap = 0;
  bool defer_loads = false;                // cwr_step in progress has decided to enqueue them behind its batch
  std::vector<char> lvl_final;             // windowed: jnorm / err_factor / bad_level of index t are final on the host
  // the Neumann vectors of refine_error_factors: ONE column (k_neumann), two of them, and the maxima of every sweep
  double *d_wa = nullptr, *d_wb = nullptr;
  unsigned long long* d_wmax = nullptr;
  std::vector<double> dt;
  double D = 0.0;
  double* d_bc = nullptr;
  // per-step operator
  FaceRec* d_rec = nullptr;
  double* d_diag = nullptr;
  int prepared_t = -1;
  // vectors: c is the full state [owned | halo | ghost] x K and doubles as the solver's x
  double *d_c = nullptr, *d_r = nullptr, *d_r0 = nullptr, *d_p = nullptr, *d_v = nullptr, *d_s = nullptr,
         *d_t = nullptr, *d_b = nullptr;
  double* d_chk = nullptr;       // [4][K] convergence-check scalars of the Jacobi path: ||x'-x||^2, ||bhat||^2 (sums) and the
                                 // element-wise measures max(|x'-x| - ew_rel |x'|), max |x'| (k_apply MODE 4)
  // (round 5) the check scalars of a single engine reach the host without a copy and without draining the stream: k_reduce_partials
  // stores them into this page-locked buffer and publishes a sequence number behind them (ReduceNote); the host spins on it
  double* h_note = nullptr;        // [5 K] doubles + the sequence word (hipHostMalloc, mapped)
  double* h_notex = nullptr;       // partitioned engines: the all-reduced check block [(2 + 2 world) K + 1] (hipHostMalloc, mapped)
  double* d_notex_view = nullptr;
  double* d_note_view = nullptr;   // the device's address of h_note
  unsigned long long* h_note_seq = nullptr;
  unsigned long long* d_note_state = nullptr;   // device: [0] the sequence counter, [1] (as unsigned int) the arrival counter
  unsigned long long note_expected = 0;         // notifications enqueued so far
  int fixed_sweeps = 0;            // CWR_TEST_FIXED_SWEEPS=N (measurement hook, tools/rank_step_profile.py): every step runs ONE batch of N sweeps
                                   // and takes its result whatever the check says -- the launch sequence of a converging step of that length, for a rank
                                   // stepped alone, whose halo rows nobody refreshes (its own iteration cannot converge: see the tool)
  bool use_note = true;            // CWR_NO_NOTE=1: the download of round 4 (A/B)
  bool fused_begin = true;         // k_begin_step: operator set-up + right-hand side + kept rows + ghost write-back in one launch (CWR_NO_FUSED_BEGIN=1: round 4's three)
  double* d_chkx = nullptr;      // partitioned engines: [rr | bb | world x (m1 | m2)] -- the one all-reduce of a check (gather_check)
  double* d_keep = nullptr;      // x_t (computed rows, written by k_rhs) and the ghost rows as the step found them: a failed
                                 // step restores the state from here
  // element-wise stopping rule on top of the norm criterion: |x'_i - x_i| <= ew_rel |x'_i| + ew_abs max|x'| for every cell
  // and constituent, with (ew_rel, ew_abs) = s (1e6 tol, tol) and s = 0.3 (1-rho)/rho from the measured contraction
  // (Jacobi's a-posteriori bound |e| <= rho/(1-rho) |x'-x|), i.e. forward error <= 1e-6 |x| + 1e-12 max|x| at tol = 1e-12
  bool ew_enabled = true;
  double ew_rel = 0.0, ew_abs = 0.0;
  bool ew_split = true;            // round 6: only the absolute part of the rule is floored at s = 1e-3 (CWR_EW_SPLIT=0: both, as until round 5)
  double ew_rel_floor = 1.0e-13;   // ... and the relative part at this size (CWR_EW_REL_FLOOR)
  std::vector<double> jnorm;     // per level t: ||J||_inf of step t's Jacobi iteration matrix (k_jnorm, when the flow field is loaded)
  // per level t: F_t with ||x* - x'||_inf <= F_t ||x' - x||_inf for a Jacobi sweep x -> x' of step t: what the element-wise rule
  // is scaled by.  ||J||_inf / (1 - ||J||_inf) where that is finite, and the row-wise bound of
  // refine_error_factors where that is smaller (near-dry rows, rows next to dry cells: see there)
  std::vector<double> err_factor;
  int neumann_sweeps = 128;      // sweeps refine_error_factors may spend per level (CWR_BOUND_SWEEPS; 0 = norm bound only)
  int neumann_sweeps_max = 2048; // ... on a level that has NO bound yet (CWR_BOUND_SWEEPS_MAX)
  bool neu_warm = true;          // the sweeps of a level start from the previous level's vector (CWR_BOUND_WARM=0: from 1, as until round 5)
  bool neu_holds_vector = false; // d_wa / d_wb hold a usable iterate of an earlier level
  bool neu_in_b = false;         // ... in d_wb
  int info_flags = 0;            // CWR_INFO_* bits of the step in progress
  bool ptr_exported = false;     // cwr_state_device_ptr handed the state out: the caller may rewrite it at any time
  // real-cell entries of input_array (levels >= 1): applied to the solved level before the mass fluxes
  std::map<int, std::pair<int, int>> in_levels;   // level -> (first entry, count)
  std::vector<char> in_any;                       // partitioned engines: level has real-cell inputs on SOME rank (sync_input_levels)
  int32_t* d_in_rows = nullptr;
  double* d_in_vals = nullptr;
  bool tail_done = false;        // the step's tail (step_tail) was enqueued speculatively and the check then passed
  int spec_t = -1, spec_flags = 0; // >= 0: solve_jacobi may enqueue step_tail(spec_t, spec_flags) before its check download
  bool halo_fresh = false;       // the halo rows of the state hold their owners' current values (set by the end-of-step
                                 // exchange of a CWR_STEP_MASS_FLUX step, cleared by anything that may change the state)
  double* d_react = nullptr;     // K x K reaction matrix of cwr_react_linear
  // ---- output side (8f-4)
  int n_lines = 0;
  int32_t *d_line_ptr = nullptr, *d_line_faces = nullptr;
  double *d_ledger = nullptr, *d_mass_out = nullptr;
  struct OutSlot { double* h = nullptr; hipEvent_t done = nullptr; std::atomic<bool> busy{false}; double *dst_state = nullptr, *dst_flux = nullptr; };
  std::vector<OutSlot> out_slots;
  hipStream_t out_stream = nullptr;
  hipEvent_t out_snap_ready = nullptr, out_copy_done = nullptr;
  double* d_snap = nullptr;      // device snapshot the copy stream reads while the next steps compute
  int32_t* d_out_order = nullptr;
  int out_n = 0, out_next = 0;
  bool out_flux = false, out_copy_pending = false;
  size_t out_direct_limit = 8u << 20;   // snapshots up to this many bytes are written in place into page-locked destinations (CWR_OUTPUT_DIRECT_MB; 4 until round 6:
                                        // 10 k cells x 12 with fluxes = 6.7 MB: facade 0.53 -> 0.51 ms per update(); 16 MB: no further gain)
  long out_direct_pushes = 0, out_copy_pushes = 0;   // (CWR_OUTPUT_DEBUG=1: printed by cwr_output_close)
  size_t out_state_cnt = 0, out_slot_cnt = 0;
  double* d_scal = nullptr;      // acc[3][ACC_N][K] | rho[3][K] | bb[K]
  int32_t* d_counters = nullptr; // 8 ints
  double *d_fadv = nullptr, *d_fdif = nullptr;      // (the total flux is their sum, formed by the readers: k_mass_flux)
  bool flux_valid = false;
  // communicator
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
  bool force_coll = false;      // CWR_FORCE_COLLECTIVES=1: issue the all-reduces even with one rank (test hook)
  std::vector<int> peers, send_ptr, recv_ptr;
  int32_t *d_send_cells = nullptr, *d_recv_cells = nullptr;
  double *d_sendbuf = nullptr, *d_recvbuf = nullptr;
  int n_send = 0, n_recv = 0;
  // overlap of a halo exchange with the interior tiles of the pass that needs it: the exchange runs on its own
  // stream between two events; `inner` tiles read core rows only, `outer` tiles read (or are) rows an exchange refreshes
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_halo = nullptr, ev_red_in = nullptr, ev_red_out = nullptr;
  bool one_comm_stream = true;                  // every RCCL call on comm_stream (see exchange_halo)
  bool overlap = true;
  bool test_poison_halo = false;                // CWR_TEST_POISON_HALO=1 (tests): NaN every halo row of both vectors in front of an overlapped exchange
  int overlap_reserve = 8 * N_XCD;              // block slots an overlapped interior launch leaves to the communication kernels
  int n_tile_inner = 0, n_tile_outer = 0;
  int32_t *d_tile_inner = nullptr, *d_tile_outer = nullptr;
  // the same split for the row tiles of the plain sweep (k_apply): the CLOSING sweep of a partitioned step runs its core tiles beside
  // the exchange that refreshes the halo rows and its cut tiles (and the replayed layers) behind it (round 4)
  int n_apply_inner = 0, n_apply_outer = 0;
  int32_t *d_apply_inner = nullptr, *d_apply_outer = nullptr;
  // ... and for the faces of the mass-flux kernel: the exchange at the end of a step (fresh halo rows for the fluxes of the cut faces
  // and for the next step's right-hand side) runs beside the faces between core cells
  int n_face_inner = 0, n_face_outer = 0;
  int32_t *d_face_inner = nullptr, *d_face_outer = nullptr;
  std::vector<int32_t> h_f1, h_f2;         // host copies of the face tables in the internal face order
  std::map<int, hipGraphExec_t> stretch_exec;   // exchange-free runs of passes of a partitioned engine, by (first parity, length)
  int64_t n_overlapped = 0;                     // exchanges that ran beside interior tiles (diagnostic, cwr_comm_stats)
  int step_exchanges = 0, step_overlapped = 0, step_checks = 0;   // of the step in progress (cwr_step_info)
  // measurement
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  bool profiling = false;
  int64_t prof_launches = 0;
  double prof_us = 0.0;
  // (round 6) the communication side of a profiled step (CWR_STEP_PROFILE on a partitioned engine): timing-event pairs around every
  // exchange (from "the packed rows are there and the communication stream is free" to "unpacked") and every all-reduce, on the stream
  // they run on; the host's wall time inside the check (cwr_comm_profile_read: what bench.py --gpus N puts into its line per rank)
  std::vector<hipEvent_t> cev;
  std::vector<char> cev_kind;              // per pair: 0 exchange (nothing beside it), 1 exchange beside compute, 2 all-reduce
  size_t cev_used = 0;
  double cprof_us[3] = {0.0, 0.0, 0.0};
  int64_t cprof_n[3] = {0, 0, 0};
  double cprof_check_wait_us = 0.0;
  int64_t cprof_checks = 0;
  // jacobi_limit: predicted sweeps beyond which a step is handed to BiCGSTAB.  Effectively off by default: measured on
  // 160x40 ... 1000x1000-cell meshes from CFL 2.5 to the steady-state limit (dt = 20 000 s), the block-asynchronous passes
  // need 60-900 sweep equivalents and beat BiCGSTAB (120-2000 iterations of ~4.7 sweeps' traffic each) by 10-20 x
  // (scratch/stiff_probe.py); BiCGSTAB stays as the fallback for a stalled or exhausted (max_iter) sweep phase.
  int last_iters = 0, last_sweeps = 0, jacobi_limit = 1 << 30;
  double last_rate = 0.0;       // contraction per sweep measured in the previous step (first-check prediction)
  // a batch of fused sweeps captured once as a hipGraph (kernel arguments never change between steps: only the
  // contents of the buffers do), replayed to keep small meshes from being host-launch-bound
  static constexpr int GRAPH_SWEEPS = 8;
  hipGraph_t sweep_graph = nullptr;
  hipGraphExec_t sweep_exec = nullptr;
  bool graph_tried = false, use_graphs = true;
  // squared operator J^2 (two Jacobi sweeps per launch; single GPU, K >= sq_min_k)
  std::vector<int32_t> h_ptr, h_nb;      // host copies of the adjacency for the symbolic J^2
  bool use_sq = true, sq_pattern = false, sq_failed = false;
  int sq_min_k = 1, nnz2 = 0, n_sq = 0, stage_cap2 = 0, apply_grid2 = 0;
  size_t apply_lds2 = 0;
  int32_t *d_ptr2 = nullptr, *d_col2 = nullptr, *d_row2 = nullptr, *d_pair_ptr = nullptr;
  uint8_t* d_slots = nullptr;
  bool sq_rowwise = false;
  size_t sqn_lds = 0;              // dynamic LDS of k_sq_numeric: the entries of the fullest 128-row block
  int sq_max_row = 0;              // longest J^2 row
  // tiled J^2 pass: per tile the distinct x rows it touches, and per J^2 entry the row's index in that list
  bool use_tcl = true, tcl_ready = false;
  int tcl_cfg = -1, tcl_vw = 0;   // tcl_vw: constituents per lane in the tiled pass (4 = wide rows, else VW)
  int local_reps = 2;              // J^2 applications per tile and pass (1 = exact Jacobi; > 1 = block-asynchronous)
  bool tcl_ell = false;            // (round 6 A/B, CWR_TCL_ELL=1) wave-sliced entry layout of the tiled pass where it applies.  Built and measured: the pass 2 % faster
                                   // at K = 16, the step not (numeric J^2 stores through an index, +10 % padded entries); K = 1: 64 rows per wave pad past the
                                   // kernel's 256-row configuration -- slower.  Off by default: profiles/r06_ell_ab.txt
  bool tcl_use_ell = false;        // ... in use by this engine's tiling
  int64_t tcl_entries = 0;         // entries of the tiled pass's weight / position arrays (CSR: nnz2; sliced: with padding)
  int32_t *d_eptr = nullptr, *d_ell_pos = nullptr;   // sliced layout: per-tile entry offsets; CSR entry -> sliced index (the numeric kernels store through it)
  int tcl_power = 2;               // 2: the passes apply J^2 (c2 = bhat + J bhat); 1 (CWR_TCL_POWER=1, round 6 A/B): the SAME kernels over J's own pattern --
                                   // a pass is one Jacobi sweep per tile-local application, the constant is bhat, no numeric J^2 and no c2 sweep per step
  double* c2() const { return tcl_power == 1 ? d_b : d_t; }
  bool reps_auto = true;           // chained passes: chosen per step from ||J||_inf (CWR_LOCAL_REPS fixes it)
  int reps_base = 2;               // the engine's default for ping-pong passes
  int n_tcl = 0, tcl_TR = 0, tcl_ntiles = 0, tcl_max_cols = 0, tcl_stage_cap = 0, tcl_grid = 0;
  size_t tcl_lds = 0, tcl_total_cols = 0;
  int32_t *d_tcl_ptr = nullptr, *d_tcl_cols = nullptr;
  int32_t *d_trow = nullptr, *d_vptr = nullptr;   // rows and virtual items (chunks 1.. of long rows) of every tile
  int tcl_nvmax = 0;                              // virtual items a tile may hold (LDS for their partial sums)
  int tcl_seg = 1 << 20;                          // J^2 entries per work item of the tiled pass (rows are summed in chunks of it)
  int32_t* d_meta = nullptr;                      // per tile: its rows' ptr2 entries, then the codes of its virtual items
  uint16_t* d_loc2 = nullptr;    // local (in-tile) column of every J^2 entry: 16 bits (a tile holds < 65 536 x rows)
  double* d_w2 = nullptr;
  // chained in-place passes (single GPU): a schedule [sched_depth][tcl_grid] of tile ids (-1 = end of a block's list); every
  // block walks chains of tiles linked along the flow of the level the schedule was built for
  int32_t* d_sched = nullptr;
  int sched_depth = 0, sched_cap = 0;
  // column reuse along a block's list (see k_sq_tiled, REUSE mode): per-schedule copy of the tiles' column lists
  bool chain_reuse = true;                 // CWR_CHAIN_REUSE=0: fetch every column, two interleaved streams per block (A/B)
  double chain_min_tiles = 1.75;           // tiles per block of the persistent grid from which schedules are built.  3 until the lane boundaries of
                                           // the numbering were smoothed (ordering.lane_order); since, lanes + chains over lists of two tiles beat the Hilbert curve +
                                           // ping-pong passes from ~1.5 tiles per block: 1.1-1.3: 0.60-0.62 vs 0.55-0.56 ms per step, 1.5: 0.57 vs 0.59, 1.8: 0.575 vs
                                           // 0.689 (119 k cells x 16: one rank of 8 of the 1 M-cell mesh; CFL 25: 2.49 vs 3.51), 2.3: 0.68 vs 0.81 (profiles/r04_x)
  int det_default_k = 8;                   // engines with up to this many constituents take the deterministic passes by default: they cost 1-3.5 % there
                                           // (K = 12: 19 %, K = 16: 14 %; profiles/r04_t_*); CWR_DET_DEFAULT_K=0: in place at every K
  bool det_walk = true;                    // deterministic steps walk the chain lists too (ping-pong between the vectors); CWR_DET_WALK=0: tile order
  int step_chained = 0;                    // the passes of the step in progress: 1 chained in place, 2 chained between two vectors (cwr_step_info.chained)
  bool deterministic = false;              // CWR_STEP_DETERMINISTIC of the step in progress: ping-pong passes
  int32_t* d_scols = nullptr;
  // partitioned engines: the interior and the cut tiles chained SEPARATELY, so that an exchange runs beside the interior lists
  // (one list position per tile in either: one shared copy of the column lists serves both)
  struct SubSched { int32_t* d = nullptr; int depth = 0, cap = 0, grid = 0; };
  SubSched sched_in, sched_out;
  int32_t* d_scols_io = nullptr;
  std::vector<int32_t> h_tile_inner, h_tile_outer;
  std::vector<int32_t> h_tcl_ptr, h_tcl_cols;             // host copies of the tiles' column lists
  std::vector<int32_t> h_trow;                            // ... and of their row ranges
  bool tiles_cut = false;                                 // some windows were cut into smaller tiles (build_tiling's limits): tile != row / TR
  std::vector<int32_t> sched_nxt;          // chain successor of every tile in the installed schedule (unchanged -> no rebuild)
  int own_cap = 0;                         // rows of the LDS staging area for a tile's results (tile rows when reuse is on)
  bool use_chains = true;
  bool shape_agreed = false, any_tiled = false;   // partitioned engines: see agree_on_pass_shape
  bool sched_user = false;                 // installed by cwr_set_tile_schedule: never rebuilt by the engine
  int sched_level = -1, sched_refresh = 64; // level the schedule was built for; rebuilt when the step is this many levels away
  int cur_t = 0;                           // level of the step in progress
  // static link structure of the tiles (built with the tiling): directed links (src tile -> dst tile) with their face entries
  std::vector<int32_t> h_edge;             // host copy of ent_edge (face index << 1 | side per adjacency entry)
  std::vector<int32_t> link_src, link_dst;
  int n_links = 0;
  int32_t *d_link_ptr = nullptr, *d_link_ent = nullptr;
  float* d_link_flux = nullptr;
  int64_t n_sched_builds = 0;
  std::map<int, hipGraphExec_t> batch_exec;   // whole-batch graphs by number of passes (see solve_jacobi)
  int batch_last = -1;
  hipGraph_t tcl_graph = nullptr;
  hipGraphExec_t tcl_exec = nullptr;
  bool tcl_graph_tried = false;
  FaceRec* d_rec2 = nullptr;
  double* d_w = nullptr;
  hipGraph_t sq_graph = nullptr;
  hipGraphExec_t sq_exec = nullptr;
  bool sq_graph_tried = false;
  int dominant_mode = 4;
  // Sweeps added to the previous step's need when the first batch of a step is sized (CWR_SWEEP_MARGIN).  The need drifts by a
  // sweep or two from step to step with the boundary series; a first batch that falls one sweep short costs a host round
  // trip, one more pass and another closing sweep (~0.25 ms at K = 16), a sweep of margin 0.05-0.09 ms.  Measured over 32
  // steps of the bench workload (profiles/r02_v_batch_shape.txt): margin 0: 6 steps with a second batch, 3.322 ms per step;
  // 1: none, 3.266; 2: none, 3.325.
  int sweep_margin = 1;
  int ew_batch_div = 16;             // CWR_EW_BATCH_DIV: a batch behind a norm-satisfied check is 1 / this of the sweeps so far (at least 8)
  bool two_closing = false;      // CWR_TWO_CLOSING=1: round 1's batch shape on one GPU too (even passes + two closing sweeps; A/B)
  bool use_small = true;         // one-workgroup-per-constituent LDS-resident solve for meshes that fit one CU
  double* d_info = nullptr;      // [K][5] results of k_small_jacobi
  int32_t* d_small_rows = nullptr;   // [rpt][1024] the row at position p of k_small_jacobi's internal order (-1: none)
  int32_t* d_small_recs = nullptr;   // [8][rpt][1024] record index of the q-th real neighbour of that row (-1: none)
  uint32_t* d_small_offs = nullptr;  // [4][rpt][1024] byte offsets of neighbours 2 qq / 2 qq + 1 in the LDS column (16 bits each)
  int small_rpt = 0;                 // rows per thread of the plan (0: not built)
  bool small_planned = false;
  int small_P = 1, small_D = 0, small_S = 0, small_R = 0;   // parts per constituent, halo layers, padded send / receive list lengths
  int small_parts = 0;               // CWR_SMALL_PARTS: parts per constituent (0: the fewest that fit)
  int small_depth = 12;              // CWR_SMALL_DEPTH: halo layers = sweeps between two exchanges of a plan of several parts
                                     // (profiles/r05_mid_mesh.txt: an exchange costs ~4 us, a sweep ~1.1: 10 k x 12 0.54 / 0.44 / 0.41 ms per step at 4 / 8 / 12)
  int small_max_parts = 12;          // CWR_SMALL_MAX_PARTS
  int small_spin_ms = 500;           // CWR_SMALL_SPIN_MS: bound of a part's wait for the others
  int n_cu = 256;                    // compute units of the device (cwr_create)
  int small_wg_cap = 128;            // workgroups one launch of the several-parts solver may have: half the CUs, 128 at most (one workgroup per CU)
  bool small_resident_checked = false;   // the occupancy query of the several-parts kernel has been made (solve_small)
  bool small_fell_back = false;      // a part's wait ran out once: the engine left the one-launch solver for good (CWR_INFO_SMALL_FALLBACK on every step since)
  int small_last_sweeps = 0;         // sweeps of the last step through k_small_jacobi (0: none, or it did not converge)
  double small_last_tol2 = -1.0;     // ... and the squared tolerance it ran with
  bool small_first_check = true;     // CWR_SMALL_FIRST_CHECK=0: convergence checks from the first sweeps on
  int small_fences = 1;              // CWR_SMALL_FENCES=0: the parts' hand-off without the agent-scope release / acquire pair (sc1 accesses only)
  int small_max_cells = 24576;       // CWR_SMALL_MAX_CELLS: meshes up to this size may take the one-launch solver with several parts
                                     // (24 k cells x 1: 0.47 against 0.56 ms with the multi-launch passes, 32-40 k: level with them; K x parts <= 128
                                     //  workgroups, so wide state vectors on the larger meshes stay with the passes; 0 = up to 4 096 cells only)
  int32_t *d_small_send_pos = nullptr, *d_small_send_cnt = nullptr, *d_small_recv_src = nullptr, *d_small_recv_pos = nullptr, *d_small_recv_cnt = nullptr;
  double *d_small_pub = nullptr, *d_small_red = nullptr;

  int nt_stream = 0;            // nt loads for the streamed operands (records, bhat/c2/r0): pays for wide rows only
  std::string err;

  double* acc(int slot) const { return d_scal + (size_t)slot * ACC_N * K; }
  double* rho(int slot) const { return d_scal + (size_t)3 * ACC_N * K + (size_t)slot * K; }
  double* bb() const { return d_scal + (size_t)3 * ACC_N * K + (size_t)3 * K; }
  size_t scal_count() const { return (size_t)3 * ACC_N * K + 3 * K + K; }
  // allocated / cleared size of d_scal (scalars + 8 counters + the precondition flag), a multiple of 256 bytes: ONE fill kernel per memset
  size_t scal_alloc() const { return (scal_count() + 5 + (size_t)K + 1 + 31) & ~(size_t)31; }
  // (behind the flag: the K arrival counters and the abort word of k_small_jacobi's parts -- zeroed by the step's one memset)
  unsigned long long* small_arrive() const { return reinterpret_cast<unsigned long long*>(d_scal + scal_count() + 5); }
  double* bad_flag() const { return d_scal + scal_count() + 4; }   // 1.0 when k_rhs met the zero-coefficient precondition (behind the 8 counters)
  bool ghost_bad_any = false;    // partitioned engines: some rank met it (all-reduced with the check scalars)
};
