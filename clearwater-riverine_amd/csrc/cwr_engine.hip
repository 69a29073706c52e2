// cwr_engine.hip -- host side of the transport engine: device memory, per-step orchestration, the
// batched BiCGSTAB driver, the optional RCCL halo exchange, and the C ABI of include/cwr_transport.h.
// Built for gfx950 only:  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -shared -fPIC
#include "../../include/cwr_transport.h"
#include "cwr_kernels.hpp"
#include "cwr_host_builders.hpp"

#include <dlfcn.h>
#include <atomic>
#include <map>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace cwr;

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
  size_t out_direct_limit = 4u << 20;   // snapshots up to this many bytes are written in place into page-locked destinations (CWR_OUTPUT_DIRECT_MB)
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

namespace {

// Every ABI entry that touches the device goes through here: the calling thread's stream-capture interaction mode becomes
// ThreadLocal (once per thread), then the device is selected.  A thread's mode defaults to Global, in which the HIP runtime refuses
// its "potentially unsafe" calls (allocations, synchronous copies, stream-memory operations) while ANY stream of the process is
// being captured -- two engines driven by two threads of one process (SURVEY 8b allows "one process (or thread) per GPU") then fail
// with "operation not permitted when stream is capturing" as soon as one of them captures a batch of passes into a hipGraph
// (gpurun_out/r04f_eight.log).  In ThreadLocal mode only the thread's OWN captures count, and those never enclose such a call.
hipError_t enter_device(int dev) {
  static thread_local bool mode_set = false;
  if (!mode_set) {
    hipStreamCaptureMode m = hipStreamCaptureModeThreadLocal;
    (void)hipThreadExchangeStreamCaptureMode(&m);
    mode_set = true;
  }
  return hipSetDevice(dev);
}

int fail(cwr_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg; else g_create_error = msg;
  return code;
}

#define HIP_TRY(e, call)                                                                       \
  do {                                                                                         \
    hipError_t _st = (call);                                                                   \
    if (_st != hipSuccess)                                                                     \
      return fail((e), CWR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_st));       \
  } while (0)

#define TRY_(call) do { int _rc = (call); if (_rc != CWR_OK) return _rc; } while (0)

#define NCCL_TRY(e, call)                                                                      \
  do {                                                                                         \
    int _st = (call);                                                                          \
    if (_st != 0)                                                                              \
      return fail((e), CWR_ERR_RCCL, std::string(#call) + ": " + g_rccl.GetErrorString(_st)); \
  } while (0)

template <typename T> int dev_alloc(cwr_engine* e, T** p, size_t count) {
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
  return CWR_OK;
}
// device temporary of a call: freed on every way out
template <typename T> struct DevTmp {
  T* p = nullptr;
  ~DevTmp() { if (p) hipFree(p); }
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
};
template <typename T> int upload(cwr_engine* e, T* dst, const T* src, size_t count) {
  if (count == 0) return CWR_OK;
  HIP_TRY(e, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return CWR_OK;
}
template <typename T> int download(cwr_engine* e, T* dst, const T* src, size_t count) {
  if (count == 0) return CWR_OK;
  HIP_TRY(e, hipMemcpyAsync(dst, src, count * sizeof(T), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return CWR_OK;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// rows of the caller's Ku values <-> the engine's rows of K >= Ku values (zero columns behind Ku; see pad_constituents)
int upload_cols(cwr_engine* e, double* dst, const double* src, size_t rows) {
  if (e->K == e->Ku) return upload(e, dst, src, rows * (size_t)e->K);
  if (rows == 0) return CWR_OK;
  DevTmp<double> tmp;
  TRY_(dev_alloc(e, &tmp.p, rows * (size_t)e->Ku));
  TRY_(upload(e, tmp.p, src, rows * (size_t)e->Ku));
  const int64_t total = (int64_t)rows * e->K;
  k_pad_cols<<<(int)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, BLOCK), 256 * 16)), BLOCK, 0, e->stream>>>(total, e->Ku, e->K, tmp.p, dst);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return CWR_OK;
}
int download_cols(cwr_engine* e, double* dst, const double* src, size_t rows) {
  if (e->K == e->Ku) return download(e, dst, src, rows * (size_t)e->K);
  if (rows == 0) return CWR_OK;
  DevTmp<double> tmp;
  TRY_(dev_alloc(e, &tmp.p, rows * (size_t)e->Ku));
  const int64_t total = (int64_t)rows * e->Ku;
  k_strip_cols<<<(int)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, BLOCK), 256 * 16)), BLOCK, 0, e->stream>>>(total, e->Ku, e->K, src, tmp.p);
  HIP_TRY(e, hipGetLastError());
  return download(e, dst, tmp.p, rows * (size_t)e->Ku);
}

// Blocks of `fn` (BLOCK threads, `lds` bytes of dynamic LDS) that are resident on a CU at once -- the size of a PERSISTENT grid,
// whose blocks walk a static share of the work: a block that is not resident from the start runs its share after the others
// are done.  The occupancy query counts 5 blocks of 32 704 B (tiled pass, K = 1) into the 160 KB of LDS and the hardware
// places 4: 1 280 blocks took 44.6 us per pass, 1 024 take 35.7 (profiles/r02_r_grid_sweep.txt).  So 2 KB of the LDS are left
// out of the count.
int resident_blocks(const void* fn, size_t lds) {
  int pc = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, BLOCK, lds) != hipSuccess || pc < 1) pc = 1;
  if (lds > 0) pc = std::min(pc, std::max(1, (int)((160 * 1024 - 2048) / ((lds + 511) / 512 * 512))));
  return pc;
}

// ---- launches ----------------------------------------------------------------------------------------
int prep_step(cwr_engine* e, int t) {
  if (e->prepared_t == t) return CWR_OK;
  k_prep_step<<<cdiv(e->n_owned, BLOCK), BLOCK, 0, e->stream>>>(
      e->n_owned, e->d_ptr, e->d_ent_edge, e->d_ent_nb, e->adv_l(t),
      e->dif_l(t), e->vol_l(t + 1), e->dt[t], e->d_rec, e->d_diag, e->d_w);
  HIP_TRY(e, hipGetLastError());
  e->prepared_t = t;
  return CWR_OK;
}

int reduce_partials(cwr_engine* e, int nslots, int ND, double* o0, double* o1 = nullptr, double* o2 = nullptr, double* o3 = nullptr,
                    int max_from = 1 << 20, bool notify = false) {
  ReduceOuts outs; outs.p[0] = o0; outs.p[1] = o1; outs.p[2] = o2; outs.p[3] = o3;
  ReduceNote note{nullptr, nullptr, nullptr, nullptr};
  if (notify && e->h_note && ND == 4 && o0 && o1 && o2 && o3)
    note = ReduceNote{e->d_note_view, reinterpret_cast<unsigned long long*>(e->d_note_view + 5 * (size_t)e->K),
                      reinterpret_cast<unsigned int*>(e->d_note_state + 1), e->d_note_state};
  k_reduce_partials<<<ND, RBLOCK, 0, e->stream>>>(nslots, ND, e->K, e->d_partial, outs, max_from, note);
  HIP_TRY(e, hipGetLastError());
  return CWR_OK;
}

// tile_list / n_list (optional): only these row tiles (of e->R * e->U rows, counted from row 0); slot0: first partials slot
template <int MODE>
int launch_apply(cwr_engine* e, const double* xin, double* yout, const double* r0, const double* bhat,
                 double* r0_out, double* p_out, int rows = -1, int row0 = 0, const int32_t* tile_list = nullptr, int n_list = 0, int slot0 = 0) {
  if (rows < 0) rows = e->n_owned;
  const int ntiles = tile_list ? n_list : cdiv(rows - row0, e->R * e->U);
  if (ntiles <= 0) { e->last_apply_grid = 0; return CWR_OK; }
  const bool sq = (MODE == 5);
  const int max_grid = sq ? e->apply_grid2 : e->apply_grid;
  const int grid = std::max(N_XCD, std::min(max_grid, cdiv(ntiles, N_XCD) * N_XCD));
  const int32_t* ptr = sq ? e->d_ptr2 : e->d_ptr;
  const FaceRec* rec = sq ? e->d_rec2 : e->d_rec;
  const int cap = sq ? e->stage_cap2 : e->stage_cap;
  const size_t lds = sq ? e->apply_lds2 : e->apply_lds;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->profiling && MODE == e->dominant_mode && e->ev_used + 2 <= e->ev.size()) {
    e0 = e->ev[e->ev_used++]; e1 = e->ev[e->ev_used++];
    HIP_TRY(e, hipEventRecord(e0, e->stream));
  }
  if (e->VW == 2)
    k_apply<2, MODE><<<grid, BLOCK, lds, e->stream>>>(row0, rows, e->n_core, e->K, e->G, e->U, ntiles, cap, e->nt_stream,
        ptr, rec, e->d_diag, xin, yout, r0, bhat, r0_out, p_out, e->d_partial, e->d_chk + 4 * (size_t)e->K, e->tcl_seg, tile_list, slot0);
  else
    k_apply<1, MODE><<<grid, BLOCK, lds, e->stream>>>(row0, rows, e->n_core, e->K, e->G, e->U, ntiles, cap, e->nt_stream,
        ptr, rec, e->d_diag, xin, yout, r0, bhat, r0_out, p_out, e->d_partial, e->d_chk + 4 * (size_t)e->K, e->tcl_seg, tile_list, slot0);
  e->last_apply_grid = grid;
  HIP_TRY(e, hipGetLastError());
  if (e1) HIP_TRY(e, hipEventRecord(e1, e->stream));
  return CWR_OK;
}

int vec_grid(const cwr_engine* e) {
  // memory-bound streaming kernels: cap the grid and grid-stride the rest
  return std::max(1, std::min(cdiv(e->n_core, e->R), 256 * 4));
}

// vec2 (optional): a second vector whose halo rows receive the same values -- the ping-pong partner of a J^2 pass, whose
// outermost (never computed) layers would otherwise keep the values of an exchange several passes back
int exchange_begin(cwr_engine* e, const double* vec);
int exchange_finish(cwr_engine* e, double* vec, double* vec2, bool beside = true);
int exchange_halo(cwr_engine* e, double* vec, double* vec2 = nullptr) {
  if (!e->comm || e->peers.empty()) return CWR_OK;
  // (round 4) ONE stream per communicator: every RCCL call of an engine -- the exchanges that run beside compute, the plain ones,
  // the all-reduces -- is issued on the communication stream, ordered against the engine's stream by events.  Round 3 issued the
  // plain exchanges and the all-reduces on the engine's stream and the overlapped ones on the communication stream: serialised by
  // the same events, but a communicator fed from two streams is exactly what RCCL documents as "serialise it yourself", and the
  // one-GPU box cannot show that the events are enough for the real library.  CWR_COMM_TWO_STREAMS=1: round 3's form.
  if (e->one_comm_stream && e->comm_stream) { const int rc = exchange_begin(e, vec); return rc != CWR_OK ? rc : exchange_finish(e, vec, vec2, false); }
  ++e->step_exchanges;
  const int64_t total = (int64_t)e->n_send * e->K;
  if (total > 0) {
    k_pack_rows<<<cdiv(total, BLOCK), BLOCK, 0, e->stream>>>(total, e->K, e->d_send_cells, vec, e->d_sendbuf);
    HIP_TRY(e, hipGetLastError());
  }
  NCCL_TRY(e, g_rccl.GroupStart());
  for (size_t i = 0; i < e->peers.size(); ++i) {
    const size_t ns = (size_t)(e->send_ptr[i + 1] - e->send_ptr[i]) * e->K;
    const size_t nr = (size_t)(e->recv_ptr[i + 1] - e->recv_ptr[i]) * e->K;
    if (ns) NCCL_TRY(e, g_rccl.Send(e->d_sendbuf + (size_t)e->send_ptr[i] * e->K, ns, NCCL_FLOAT64, e->peers[i], e->comm, e->stream));
    if (nr) NCCL_TRY(e, g_rccl.Recv(e->d_recvbuf + (size_t)e->recv_ptr[i] * e->K, nr, NCCL_FLOAT64, e->peers[i], e->comm, e->stream));
  }
  NCCL_TRY(e, g_rccl.GroupEnd());
  const int64_t rtotal = (int64_t)e->n_recv * e->K;
  if (rtotal > 0) {
    k_unpack_rows<<<cdiv(rtotal, BLOCK), BLOCK, 0, e->stream>>>(rtotal, e->K, e->d_recv_cells, e->d_recvbuf, vec, vec2);
    HIP_TRY(e, hipGetLastError());
  }
  return CWR_OK;
}

// The same exchange in two halves, for overlap: exchange_begin packs on the engine's stream and marks the spot; whatever the
// caller enqueues next on that stream (the interior tiles) runs beside exchange_finish, which sends / receives / unpacks
// on the communication stream and makes the engine's stream wait for the unpacked rows.
int exchange_begin(cwr_engine* e, const double* vec) {
  const int64_t total = (int64_t)e->n_send * e->K;
  if (total > 0) {
    k_pack_rows<<<cdiv(total, BLOCK), BLOCK, 0, e->stream>>>(total, e->K, e->d_send_cells, vec, e->d_sendbuf);
    HIP_TRY(e, hipGetLastError());
  }
  HIP_TRY(e, hipEventRecord(e->ev_packed, e->stream));
  return CWR_OK;
}
int exchange_finish(cwr_engine* e, double* vec, double* vec2, bool beside) {
  HIP_TRY(e, hipStreamWaitEvent(e->comm_stream, e->ev_packed, 0));
  NCCL_TRY(e, g_rccl.GroupStart());
  for (size_t i = 0; i < e->peers.size(); ++i) {
    const size_t ns = (size_t)(e->send_ptr[i + 1] - e->send_ptr[i]) * e->K;
    const size_t nr = (size_t)(e->recv_ptr[i + 1] - e->recv_ptr[i]) * e->K;
    if (ns) NCCL_TRY(e, g_rccl.Send(e->d_sendbuf + (size_t)e->send_ptr[i] * e->K, ns, NCCL_FLOAT64, e->peers[i], e->comm, e->comm_stream));
    if (nr) NCCL_TRY(e, g_rccl.Recv(e->d_recvbuf + (size_t)e->recv_ptr[i] * e->K, nr, NCCL_FLOAT64, e->peers[i], e->comm, e->comm_stream));
  }
  NCCL_TRY(e, g_rccl.GroupEnd());
  const int64_t rtotal = (int64_t)e->n_recv * e->K;
  if (rtotal > 0) {
    k_unpack_rows<<<cdiv(rtotal, BLOCK), BLOCK, 0, e->comm_stream>>>(rtotal, e->K, e->d_recv_cells, e->d_recvbuf, vec, vec2);
    HIP_TRY(e, hipGetLastError());
  }
  HIP_TRY(e, hipEventRecord(e->ev_halo, e->comm_stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_halo, 0));
  ++e->step_exchanges;
  if (beside) { ++e->n_overlapped; ++e->step_overlapped; }        // (the caller put work on the engine's stream between the two halves)
  return CWR_OK;
}

int allreduce(cwr_engine* e, double* p, size_t count) {
  if (!e->comm || (e->world == 1 && !e->force_coll)) return CWR_OK;
  if (e->one_comm_stream && e->comm_stream && e->ev_red_in) {      // on the communication stream, behind the producer, in front of the consumer
    HIP_TRY(e, hipEventRecord(e->ev_red_in, e->stream));
    HIP_TRY(e, hipStreamWaitEvent(e->comm_stream, e->ev_red_in, 0));
    NCCL_TRY(e, g_rccl.AllReduce(p, p, count, NCCL_FLOAT64, NCCL_SUM, e->comm, e->comm_stream));
    HIP_TRY(e, hipEventRecord(e->ev_red_out, e->comm_stream));
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_red_out, 0));
    return CWR_OK;
  }
  NCCL_TRY(e, g_rccl.AllReduce(p, p, count, NCCL_FLOAT64, NCCL_SUM, e->comm, e->stream));
  return CWR_OK;
}

#define TRY(call) do { int _rc = (call); if (_rc != CWR_OK) return _rc; } while (0)

// the closing check of a batch of sweeps: fold the per-block partials of the last MODE 4 launch into d_chk
int reduce_check(cwr_engine* e, bool notify = false) {
  const int K = e->K;
  return reduce_partials(e, e->last_apply_grid, 4, e->d_chk, e->d_chk + K, e->d_chk + 2 * K, e->d_chk + 3 * K, 2, notify);
}

// single engines (and a stand-alone rank): may this check be read from the notification buffer instead of a download?
bool check_by_note(const cwr_engine* e) { return e->use_note && e->h_note && (!e->comm || (e->world == 1 && !e->force_coll)); }

// Wait for the `note_expected`-th notification of k_reduce_partials and take the check scalars from the host buffer.  The host
// spins on a word of page-locked memory: no copy is enqueued and the stream is not drained -- whatever was enqueued BEHIND the
// reduction (the step's speculative tail) runs on while the host already decides and enqueues the next step.
int wait_check_note(cwr_engine* e, double* h, size_t count = 0, const double* from = nullptr) {
  const unsigned long long want = e->note_expected;
  for (unsigned long long spin = 1;; ++spin) {
    if (__atomic_load_n(e->h_note_seq, __ATOMIC_ACQUIRE) >= want) break;
    __builtin_ia32_pause();
    if ((spin & 0xFFFFu) == 0) {
      // a fault on the stream would otherwise leave the host spinning: ask the runtime every 65 536 spins
      const hipError_t st = hipStreamQuery(e->stream);
      (void)hipGetLastError();         // (hipErrorNotReady is the normal answer here, and HIP remembers it as the thread's last error: a
                                       //  library that checks hipGetLastError() afterwards -- RCCL's initialisation does -- would trip over it)
      if (st != hipSuccess && st != hipErrorNotReady) return fail(e, CWR_ERR_HIP, std::string("convergence check: ") + hipGetErrorString(st));
      if (st == hipSuccess && __atomic_load_n(e->h_note_seq, __ATOMIC_ACQUIRE) < want) {
        // (the stream is idle and the word has not moved: settle once more, then give up loudly)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (__atomic_load_n(e->h_note_seq, __ATOMIC_ACQUIRE) < want) return fail(e, CWR_ERR_HIP, "convergence check: the stream drained without the notification");
      }
    }
  }
  std::memcpy(h, from ? from : e->h_note, (count ? count : 4 * (size_t)e->K) * sizeof(double));
  return CWR_OK;
}

// The check scalars of every rank, on the host: h = [rr | bb | m1 | m2] with the sums added and the maxima taken over the
// ranks.  ONE all-reduce (sum) carries both: every rank adds its two maxima in its own slot of a (world x 2K) block that is
// zero elsewhere, and the host takes the maximum over the slots -- a second (max) collective would cost another 20-40 us of
// latency per check.  Single GPU: a plain download.
int gather_check(cwr_engine* e, double* h, bool noted = false) {
  const size_t K = (size_t)e->K;
  ++e->step_checks;
  if (noted) return wait_check_note(e, h);                      // (the reduction of this check was launched with a notification)
  if (!e->comm || (e->world == 1 && !e->force_coll)) return download(e, h, e->d_chk, 4 * K);
  // (+ one word: the zero-coefficient precondition flag of k_rhs, so that every rank learns of a violation on ANY rank with the
  // check it downloads anyway -- the step used to end with a second, blocking download of the rank's own counters, which also
  // waited for the speculative tail behind the check)
  const size_t W = (size_t)e->world, n = 2 * K + W * 2 * K + 1;
  // (round 5) one launch lays the block out (a memset and three copies before), and the all-reduced block reaches the host through
  // page-locked memory and a sequence word, as a single engine's check does: no copy, the stream not drained -- the stand-alone
  // rank budgets of profiles/r05_rank_budget.txt were measured that way, so a rank of a real run has to do the same
  k_pack_check<<<1, 256, 0, e->stream>>>((int)n, (int)K, e->rank, e->d_chk, e->bad_flag(), e->d_chkx);
  HIP_TRY(e, hipGetLastError());
  TRY(allreduce(e, e->d_chkx, n));
  std::vector<double> all(n);
  if (e->use_note && e->h_notex && e->h_note_seq) {
    k_note_out<<<1, 256, 0, e->stream>>>((int)n, e->d_chkx, e->d_notex_view, reinterpret_cast<unsigned long long*>(e->d_note_view + 5 * K), e->d_note_state);
    HIP_TRY(e, hipGetLastError());
    ++e->note_expected;
    TRY(wait_check_note(e, all.data(), n, e->h_notex));
  } else {
    TRY(download(e, all.data(), e->d_chkx, n));
  }
  e->ghost_bad_any = all[n - 1] > 0.0;
  for (size_t k = 0; k < 2 * K; ++k) h[k] = all[k];
  for (size_t k = 0; k < 2 * K; ++k) {
    double m = -INFINITY;
    for (size_t r = 0; r < W; ++r) m = std::max(m, all[2 * K + r * 2 * K + k]);
    h[2 * K + k] = m;
  }
  return CWR_OK;
}

// element-wise verdict from the downloaded check scalars h = [rr | bb | m1 | m2]; ratio: by how much |x'-x| still has to fall
bool elementwise_ok(const cwr_engine* e, const double* h, double* ratio) {
  const int K = e->K;
  double worst = 0.0;
  if (e->ew_enabled)
    for (int k = 0; k < K; ++k) {
      const double m1 = h[2 * K + k], m2 = h[3 * K + k];
      if (m1 > 0.0) worst = std::max(worst, m2 > 0.0 ? m1 / (e->ew_abs * m2) : (double)INFINITY);
    }
  if (ratio) *ratio = worst;
  return !(worst > 1.0);
}

int launch_rhs(cwr_engine* e, int t, const double* x, double* b, bool scale, double* keep = nullptr) {
  const int grid = std::max(1, std::min(cdiv(e->n_owned, e->R), 256 * 16));
  const float* vol_t = e->vol_l(t);
  const float* vel_n = e->vel_l(t + 1);
  const float* adv_n = e->adv_l(t + 1);
  const double* dif_n = e->dif_l(t + 1);
  const double* bc_n = e->d_bc + (size_t)(t + 1) * e->n_ghost * e->K;
  const int used = (e->D != 0.0) ? 1 : 0;
#define CWR_RHS(VWv, SC) k_rhs<VWv, SC><<<grid, BLOCK, 0, e->stream>>>(e->n_owned, e->K, e->G, e->d_ptr, e->d_ent_edge, \
    e->d_ent_nb, vol_t, e->dt[t], vel_n, adv_n, dif_n, used, bc_n, x, e->d_diag, e->d_row_ghost, b, e->d_counters, keep, e->n_owned, e->n_cells - e->n_owned, \
    e->d_chk + 4 * (size_t)e->K, e->ew_rel, e->bad_flag())
  if (e->VW == 2) { if (scale) CWR_RHS(2, true); else CWR_RHS(2, false); }
  else            { if (scale) CWR_RHS(1, true); else CWR_RHS(1, false); }
#undef CWR_RHS
  HIP_TRY(e, hipGetLastError());
  return CWR_OK;
}

// the opening of step t in one launch (k_begin_step): what prep_step + launch_rhs(scale, keep) + the ghost write-back of step_tail did
int launch_begin_step(cwr_engine* e, int t) {
  const int used = (e->D != 0.0) ? 1 : 0;
#define CWR_BEGIN(VWv) k_begin_step<VWv><<<cdiv(e->n_owned, BLOCK), BLOCK, 0, e->stream>>>(e->n_owned, e->n_real, e->n_cells, e->K, e->G, e->d_ptr, \
    e->d_ent_edge, e->d_ent_nb, e->adv_l(t), e->dif_l(t), e->vol_l(t + 1), e->dt[t], e->d_rec, \
    e->d_diag, e->d_w, e->vol_l(t), e->vel_l(t + 1), e->adv_l(t + 1), e->dif_l(t + 1), \
    used, e->d_bc + (size_t)(t + 1) * e->n_ghost * e->K, e->d_c, e->d_row_ghost, e->d_b, e->d_counters, e->d_keep, e->d_chk + 4 * (size_t)e->K, e->ew_rel, \
    e->bad_flag())
  if (e->VW == 2) CWR_BEGIN(2); else CWR_BEGIN(1);
#undef CWR_BEGIN
  HIP_TRY(e, hipGetLastError());
  e->prepared_t = t;
  return CWR_OK;
}

int one_iteration(cwr_engine* e, int it, double tol2) {
  const int K = e->K;
  const int slot = it % 3, prev = (it + 2) % 3, next = (it + 1) % 3;
  double* acc_cur = e->acc(slot);
  const double* rho_ptr = (it == 0) ? e->acc(2) + ACC_RR * K : e->rho(slot);
  const double* rr_prev = e->acc(prev) + ACC_RR * K;
  const int vg = vec_grid(e);
  TRY(exchange_halo(e, e->d_p));
  TRY(launch_apply<1>(e, e->d_p, e->d_v, e->d_r0, nullptr, nullptr, nullptr, e->n_core));
  TRY(reduce_partials(e, e->last_apply_grid, 1, acc_cur + ACC_R0V * K));
  TRY(allreduce(e, acc_cur + ACC_R0V * K, K));
  if (e->VW == 2) k_vec_s<2><<<vg, BLOCK, 0, e->stream>>>(e->n_core, K, e->G, e->d_r, e->d_v, e->d_s, rho_ptr, acc_cur, rr_prev, e->bb(), tol2);
  else            k_vec_s<1><<<vg, BLOCK, 0, e->stream>>>(e->n_core, K, e->G, e->d_r, e->d_v, e->d_s, rho_ptr, acc_cur, rr_prev, e->bb(), tol2);
  HIP_TRY(e, hipGetLastError());
  TRY(exchange_halo(e, e->d_s));
  TRY(launch_apply<2>(e, e->d_s, e->d_t, e->d_r0, nullptr, nullptr, nullptr, e->n_core));
  TRY(reduce_partials(e, e->last_apply_grid, 4, acc_cur + ACC_TS * K, acc_cur + ACC_TT * K, acc_cur + ACC_R0T * K, acc_cur + ACC_R0S * K));
  TRY(allreduce(e, acc_cur + ACC_TS * K, 4 * (size_t)K));
  if (e->VW == 2) k_vec_x<2><<<vg, BLOCK, 0, e->stream>>>(e->n_core, K, e->G, e->d_c, e->d_r, e->d_p, e->d_s, e->d_t, e->d_v, rho_ptr, e->rho(next), acc_cur, e->d_partial, rr_prev, e->bb(), tol2, e->d_counters);
  else            k_vec_x<1><<<vg, BLOCK, 0, e->stream>>>(e->n_core, K, e->G, e->d_c, e->d_r, e->d_p, e->d_s, e->d_t, e->d_v, rho_ptr, e->rho(next), acc_cur, e->d_partial, rr_prev, e->bb(), tol2, e->d_counters);
  HIP_TRY(e, hipGetLastError());
  TRY(reduce_partials(e, vg, 1, acc_cur + ACC_RR * K));
  TRY(allreduce(e, acc_cur + ACC_RR * K, K));
  return CWR_OK;
}

int flush_window_loads(cwr_engine* e);
int window_load_now(cwr_engine* e, int t0, int n_levels, const float* face_flow, const float* edge_velocity, const float* volume);
int check_level(cwr_engine* e, int t, bool need_next) {
  if (e->T <= 0) return fail(e, CWR_ERR_STATE, "no flow field loaded (cwr_load_flow_field / cwr_load_coefficients)");
  if (t < 0 || t + (need_next ? 1 : 0) >= e->T)
    return fail(e, CWR_ERR_STATE, "time level " + std::to_string(t) + " out of range for " + std::to_string(e->T) + " levels");
  if (e->windowed && (!e->pending_loads.empty() || !e->pending_bc.empty()) && !e->defer_loads) TRY(flush_window_loads(e));
  if (e->windowed)
    for (int q = t; q <= t + (need_next ? 1 : 0); ++q) {
      if (e->slot_level[e->slot(q)] != q)
        return fail(e, CWR_ERR_STATE, "time level " + std::to_string(q) + " is not in the flow-field window (cwr_flow_window_load: slot " +
                    std::to_string(e->slot(q)) + " holds level " + std::to_string(e->slot_level[e->slot(q)]) + ")");
      HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_level[e->slot(q)], 0));   // (whatever the caller enqueues next reads the arrived level)
    }
  return CWR_OK;
}

// Windowed engines: before step t runs, its two levels must have arrived (the engine's stream waits for the flow stream's events)
// and the host needs what the flow stream left for it -- ||J||_inf of step t, the zero-coefficient flag of level t + 1 -- and the
// row-wise error factor where the norm form is not good enough (refine_level: synchronously here, the price of a level whose
// worst row says nothing; uniform fields never pay it).  The same numbers a resident engine holds after cwr_load_flow_field.
int refine_level(cwr_engine* e, int t);
int finalize_level(cwr_engine* e, int t) {
  if (!e->windowed || e->lvl_final[(size_t)t]) return CWR_OK;      // (check_level has made the engine's stream wait for both levels)
  HIP_TRY(e, hipEventSynchronize(e->ev_level[e->slot(t + 1)]));    // (the later of the two: the flow stream works in load order)
  HIP_TRY(e, hipEventSynchronize(e->ev_level[e->slot(t)]));
  const double rho = e->h_lvl[2 * (size_t)t];
  e->jnorm[(size_t)t] = rho;
  e->bad_level[(size_t)t + 1] = e->h_lvl[2 * ((size_t)t + 1) + 1] != 0.0 ? 1 : 0;
  e->err_factor[(size_t)t] = (rho >= 0.0 && rho < 1.0) ? rho / (1.0 - rho) : INFINITY;
  if (e->neumann_sweeps > 0) TRY(refine_level(e, t));
  e->lvl_final[(size_t)t] = 1;
  return CWR_OK;
}

void collect_profile(cwr_engine* e) {
  for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]) == hipSuccess) { e->prof_us += 1000.0 * ms; e->prof_launches += 1; }
  }
  e->ev_used = 0;
}

int alloc_flow(cwr_engine* e, int T) {
  if (e->flow_cap != T) {
    hipFree(e->d_adv); hipFree(e->d_dif); hipFree(e->d_vel); hipFree(e->d_vol);
    e->d_adv = nullptr; e->d_dif = nullptr; e->d_vel = nullptr; e->d_vol = nullptr; e->T = 0; e->flow_cap = 0;
    TRY(dev_alloc(e, &e->d_adv, (size_t)T * e->E));
    TRY(dev_alloc(e, &e->d_dif, (size_t)T * e->E));
    TRY(dev_alloc(e, &e->d_vel, (size_t)T * e->E));
    TRY(dev_alloc(e, &e->d_vol, (size_t)T * e->n_cells));
    e->flow_cap = T;
  }
  e->T = T; e->W = T; e->windowed = false;
  e->prepared_t = -1;
  e->pending_loads.clear(); e->pending_bc.clear();   // (noted for another field: stale pointers, levels of another T / W -- ADVICE r05)
  return CWR_OK;
}

// flags of the reference's zero-coefficient ValueError for every loaded level (see k_check_ghost_levels)
int check_ghost_levels(cwr_engine* e) {
  const int T = e->T;
  e->bad_level.assign((size_t)T, 0);
  if (T <= 0 || e->E <= 0) return CWR_OK;
  DevTmp<int32_t> t_flags;
  TRY(dev_alloc(e, &t_flags.p, (size_t)T));
  int32_t* d_flags = t_flags.p;
  int rc = CWR_OK;
  if (hipMemsetAsync(d_flags, 0, (size_t)T * sizeof(int32_t), e->stream) != hipSuccess) rc = fail(e, CWR_ERR_HIP, "hipMemsetAsync failed");
  if (rc == CWR_OK) {
    const int64_t total = (int64_t)T * e->E;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, BLOCK), 256 * 16));
    k_check_ghost_levels<<<grid, BLOCK, 0, e->stream>>>(total, e->E, e->n_owned, e->n_real, e->d_f1, e->d_f2, e->d_vel, e->d_adv,
                                                       e->d_dif, e->D != 0.0 ? 1 : 0, d_flags);
    if (hipGetLastError() != hipSuccess) rc = fail(e, CWR_ERR_HIP, "k_check_ghost_levels failed");
  }
  if (rc == CWR_OK) rc = download(e, e->bad_level.data(), d_flags, (size_t)T);
  return rc;
}

int sync_jnorms(cwr_engine* e);
int refine_error_factors(cwr_engine* e);
void norm_error_factors(cwr_engine* e) {
  e->err_factor.assign(e->jnorm.size(), INFINITY);
  for (size_t t = 0; t < e->jnorm.size(); ++t) {
    const double rho = e->jnorm[t];
    if (rho >= 0.0 && rho < 1.0) e->err_factor[t] = rho / (1.0 - rho);
  }
}
// ||J||_inf of every step the loaded flow field allows (see k_jnorm); jnorm[T-1] = 0 (no step starts at the last level)
int compute_jnorms(cwr_engine* e) {
  const int T = e->T;
  e->jnorm.assign((size_t)std::max(T, 0), 0.0);
  if (T < 2) return CWR_OK;
  DevTmp<unsigned long long> t_jn; DevTmp<double> t_dt;
  TRY(dev_alloc(e, &t_jn.p, (size_t)T));
  TRY(dev_alloc(e, &t_dt.p, (size_t)T));
  HIP_TRY(e, hipMemsetAsync(t_jn.p, 0, (size_t)T * sizeof(unsigned long long), e->stream));
  TRY(upload(e, t_dt.p, e->dt.data(), (size_t)T));
  for (int t0 = 0; t0 < T - 1; t0 += 32768) {                   // (gridDim.y <= 65535)
    const int nt = std::min(32768, T - 1 - t0);
    k_jnorm<<<dim3((unsigned)std::max(1, std::min(cdiv(e->n_owned, BLOCK), 1024)), (unsigned)nt), BLOCK, 0, e->stream>>>(
        e->n_owned, e->E, e->n_cells, e->d_ptr, e->d_ent_edge, e->d_ent_nb, e->d_adv + (size_t)t0 * e->E, e->d_dif + (size_t)t0 * e->E,
        e->d_vol + (size_t)(t0 + 1) * e->n_cells, t_dt.p + t0, t_jn.p + t0, 0.0);
    HIP_TRY(e, hipGetLastError());
  }
  static_assert(sizeof(unsigned long long) == sizeof(double), "bit patterns");
  TRY(download(e, reinterpret_cast<unsigned long long*>(e->jnorm.data()), t_jn.p, (size_t)T));
  TRY(sync_jnorms(e));
  return refine_error_factors(e);
}

// Partitioned engines: every rank's norms become the maximum over the ranks (the element-wise rule of the GLOBAL matrix, as a
// single engine would apply it).  One sum all-reduce: every rank adds its values in its own slot of a (world x T) block that is
// zero elsewhere, and the host takes the maximum over the slots (cf. gather_check).  Collective: every rank calls it at the
// same point -- when the flow field is loaded with a communicator attached, or when the communicator is attached to an
// engine that already holds a flow field.
int sync_jnorms(cwr_engine* e) {
  norm_error_factors(e);
  if (!e->comm || e->world <= 1 || e->T <= 0 || e->jnorm.size() != (size_t)e->T) return CWR_OK;
  const size_t T = (size_t)e->T, W = (size_t)e->world;
  DevTmp<double> buf;
  TRY(dev_alloc(e, &buf.p, W * T));
  HIP_TRY(e, hipMemsetAsync(buf.p, 0, W * T * sizeof(double), e->stream));
  HIP_TRY(e, hipMemcpyAsync(buf.p + (size_t)e->rank * T, e->jnorm.data(), T * sizeof(double), hipMemcpyHostToDevice, e->stream));
  TRY(allreduce(e, buf.p, W * T));
  std::vector<double> all(W * T);
  TRY(download(e, all.data(), buf.p, W * T));
  for (size_t t = 0; t < T; ++t) {
    double m = 0.0;
    for (size_t r = 0; r < W; ++r) { const double v = all[r * T + t]; m = (v != v) ? INFINITY : std::max(m, v); }
    e->jnorm[t] = m;
  }
  norm_error_factors(e);
  return CWR_OK;
}

// The max-norm a-posteriori bound of a Jacobi sweep, row by row.  For x' = J x + bhat and the solution x* = J x* + bhat:
//     (I - J)(x* - x') = J (x' - x)   =>   |x* - x'| <= (I - J)^-1 J |x' - x| <= ((I - J)^-1 1 - 1) ||x' - x||_inf = (w - 1) ||x' - x||_inf
// with w = (I - J)^-1 1 >= 1 (J >= 0, rho(J) < 1: A is a column-diagonally-dominant M-matrix whatever the flow field does to its
// ROWS).  ||J||_inf / (1 - ||J||_inf) is the crude form of max(w) - 1: it is set by the single worst row -- a nearly dry cell with
// through-flow (row sum 1 - V_t / (dt sum_in): local CFL in the thousands at a wetting front), or the neighbour of a dry cell
// whose zeroed faces leave it an unbalanced budget (row sum > 1: no norm bound at all) -- although such a row simply follows its
// neighbours.  w is bounded rigorously from the Neumann series: w_m = sum_{k<=m} J^k 1 (m sweeps of w <- 1 + J w from 1),
// r_m = w_{m+1} - w_m = J^{m+1} 1 >= 0, and w - w_{m+1} = (I - J)^-1 J r_m <= ||r_m||_inf (w - 1), so
//     max(w) - 1 <= max(w_{m+1} - 1) / (1 - ||r_m||_inf)          once ||r_m||_inf < 1.
// Evaluated per loaded level with a matrix-free one-column sweep of its own (k_neumann: J's entries formed on the fly from the
// level's coefficients; round 4 ran the K-wide solver sweep on K identical columns).
// Partitioned engines (round 5): the same sweeps over the rank's computed rows (core + replayed layers) with one halo exchange per
// `exch_every` sweeps -- the deep halo serves the Neumann vector exactly as it serves the solver's sweeps -- and ONE all-reduce per
// check that carries every rank's (||r_m||_inf, max w) in a slot of its own: every rank ends with the factor of the GLOBAL matrix, the
// one a single engine would hold, and takes the same stop decisions.  COLLECTIVE then: called where the flow field is loaded with a
// communicator attached, or where the communicator is attached to an engine that holds a flow field (cwr_attach_comm).
constexpr int NEU_FIRST = 12, NEU_NEXT = 8;      // sweeps before the first / every later host decision

int neumann_cap(const cwr_engine* e) { return std::max(e->neumann_sweeps, e->neumann_sweeps_max); }   // sweeps a level without any bound may take
int neumann_buffers(cwr_engine* e) {
  if (e->d_wa) return CWR_OK;
  TRY(dev_alloc(e, &e->d_wa, (size_t)e->n_real));
  TRY(dev_alloc(e, &e->d_wb, (size_t)e->n_real));
  TRY(dev_alloc(e, &e->d_wmax, (size_t)2 * (neumann_cap(e) + NEU_FIRST + NEU_NEXT)));
  return CWR_OK;
}

// one-column halo exchange of the Neumann vector (partitioned engines; the solver's send / receive buffers serve: nothing else
// runs while a flow field is being loaded or a communicator attached)
int exchange_halo_1col(cwr_engine* e, double* vec, double* vec2) {
  if (!e->comm || e->peers.empty()) return CWR_OK;
  hipStream_t cs = (e->one_comm_stream && e->comm_stream) ? e->comm_stream : e->stream;
  if (e->n_send > 0) {
    k_pack_rows<<<cdiv(e->n_send, BLOCK), BLOCK, 0, e->stream>>>((int64_t)e->n_send, 1, e->d_send_cells, vec, e->d_sendbuf);
    HIP_TRY(e, hipGetLastError());
  }
  if (cs != e->stream) { HIP_TRY(e, hipEventRecord(e->ev_packed, e->stream)); HIP_TRY(e, hipStreamWaitEvent(cs, e->ev_packed, 0)); }
  NCCL_TRY(e, g_rccl.GroupStart());
  for (size_t i = 0; i < e->peers.size(); ++i) {
    const size_t ns = (size_t)(e->send_ptr[i + 1] - e->send_ptr[i]), nr = (size_t)(e->recv_ptr[i + 1] - e->recv_ptr[i]);
    if (ns) NCCL_TRY(e, g_rccl.Send(e->d_sendbuf + (size_t)e->send_ptr[i], ns, NCCL_FLOAT64, e->peers[i], e->comm, cs));
    if (nr) NCCL_TRY(e, g_rccl.Recv(e->d_recvbuf + (size_t)e->recv_ptr[i], nr, NCCL_FLOAT64, e->peers[i], e->comm, cs));
  }
  NCCL_TRY(e, g_rccl.GroupEnd());
  if (e->n_recv > 0) {
    k_unpack_rows<<<cdiv(e->n_recv, BLOCK), BLOCK, 0, cs>>>((int64_t)e->n_recv, 1, e->d_recv_cells, e->d_recvbuf, vec, vec2);
    HIP_TRY(e, hipGetLastError());
  }
  if (cs != e->stream) { HIP_TRY(e, hipEventRecord(e->ev_halo, cs)); HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_halo, 0)); }
  return CWR_OK;
}

// (||w_{m+1} - w_m||_inf, max w_{m+1}) of sweep `q` of the level in progress, over every rank
int bound_check(cwr_engine* e, int q, double* r, double* wmax) {
  double h[2];
  static_assert(sizeof(unsigned long long) == sizeof(double), "bit patterns");
  if (!e->comm || (e->world == 1 && !e->force_coll)) {
    TRY(download(e, reinterpret_cast<unsigned long long*>(h), e->d_wmax + 2 * (size_t)q, 2));
    *r = h[0]; *wmax = h[1];
    return CWR_OK;
  }
  const size_t W = (size_t)e->world, n = 2 * W;                 // (d_chkx holds (2 + 2 W) K + 1 doubles)
  HIP_TRY(e, hipMemsetAsync(e->d_chkx, 0, n * sizeof(double), e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_chkx + 2 * (size_t)e->rank, e->d_wmax + 2 * (size_t)q, 2 * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  TRY(allreduce(e, e->d_chkx, n));
  std::vector<double> all(n);
  TRY(download(e, all.data(), e->d_chkx, n));
  double rr = 0.0, ww = 0.0;
  for (size_t k = 0; k < W; ++k) {                              // (no bound on any rank is no bound for anybody)
    const double a = all[2 * k], b = all[2 * k + 1];
    rr = (a != a || rr != rr) ? NAN : std::max(rr, a);
    ww = (b != b || ww != ww) ? NAN : std::max(ww, b);
  }
  *r = rr; *wmax = ww;
  return CWR_OK;
}

// the row-wise factor of ONE step (level t, whose coefficients and V of level t + 1 must be on the device), synchronously
int refine_level(cwr_engine* e, int t) {
  if (e->err_factor[(size_t)t] <= 4.0) return CWR_OK;
  // (the scale s = 0.3 / F of the element-wise rule is held within [1e-3, 0.1]: a factor below 3 changes nothing, and one below 4
  // (||J||_inf <= 0.8: s >= 0.075) costs at most one sweep of ~35 against the ideal -- less than the twelve Neumann sweeps per level
  // that finding out would take (round 5: 3 -> 4; the bench field, 0.775 -> 3.44, is uniformly stiff and gained nothing from its
  // sweeps: a windowed run would have paid them at every step).  The sweeps themselves stop as soon as the bound is below 3.
  // Partitioned: err_factor comes from the all-reduced norms, so every rank skips the same levels)
  TRY(neumann_buffers(e));
  const bool part = e->comm && (e->world > 1 || e->force_coll);
  const int nr = e->n_real;
  const int cap = neumann_cap(e) + NEU_FIRST + NEU_NEXT;
  // (round 6) WARM START: the sweeps of a level begin from the vector the previous level's sweeps ended with instead of from 1.  The
  // iteration w <- 1 + J w converges to w = (I - J)^-1 1 from ANY start, and the bound holds for any iterate: with e_m = w - w_m,
  // r_m = w_{m+1} - w_m = (I - J) e_m gives |e_m| <= (I - J)^-1 |r_m| <= ||r_m||_inf w, e_{m+1} = J e_m, so
  //     w - 1 <= (w_{m+1} - 1) + ||r_m||_inf (w - 1)   =>   max(w) - 1 <= max(w_{m+1} - 1) / (1 - ||r_m||_inf)    (||r_m||_inf = max |r_m| < 1)
  // -- the same formula, r no longer of one sign (k_neumann folds |r|).  A flow field changes little from level to level: after the
  // first level a batch of twelve sweeps decides most levels, where the series from 1 needs one sweep per cell of the domain's length
  // (river band at dt = 14 400 s: none within 128 sweeps -- F = inf, CWR_INFO_ELEMENTWISE_CLAMPED on every step, profiles/r06_matrix_probe.txt).
  // A level WITHOUT any bound so far may therefore take up to CWR_BOUND_SWEEPS_MAX sweeps (2 048); its successors start from its vector.
  const bool warm = e->neu_warm && e->neu_holds_vector;
  if (!warm) {
    k_fill<<<std::max(1, std::min(cdiv(nr, BLOCK), 2048)), BLOCK, 0, e->stream>>>((int64_t)nr, 1.0, e->d_wa, e->d_wb);
    HIP_TRY(e, hipGetLastError());
  }
  HIP_TRY(e, hipMemsetAsync(e->d_wmax, 0, (size_t)2 * cap * sizeof(unsigned long long), e->stream));
  double* x = e->neu_in_b && warm ? e->d_wb : e->d_wa; double* y = x == e->d_wa ? e->d_wb : e->d_wa;
  double best = e->err_factor[(size_t)t];
  // (w_0 = 1 on every row, halo rows included: exact everywhere.  A warm start's halo rows hold what the previous level's last
  // exchange left: refreshed in front of the first sweep)
  int since_exchange = warm ? e->exch_every : 0;
  int q = 0;
  for (int done = 0; done < (std::isfinite(best) ? e->neumann_sweeps : neumann_cap(e));) {
    const int batch = done == 0 ? NEU_FIRST : NEU_NEXT;                      // (one host round trip decides most levels: see the stop rules below)
    for (int i = 0; i < batch; ++i, ++q) {
      if (part && since_exchange >= e->exch_every) { TRY(exchange_halo_1col(e, x, y)); since_exchange = 0; }
      k_neumann<<<std::max(1, std::min(cdiv(e->n_owned, BLOCK), 1024)), BLOCK, 0, e->stream>>>(
          e->n_owned, e->n_core, e->d_ptr, e->d_ent_edge, e->d_ent_nb, e->adv_l(t), e->dif_l(t), e->vol_l(t + 1), e->dt[(size_t)t], x, y,
          i == batch - 1 ? e->d_wmax + 2 * (size_t)q : nullptr);            // (only the last sweep of a batch is looked at)
      HIP_TRY(e, hipGetLastError());
      std::swap(x, y); ++since_exchange;
    }
    done += batch;
    double r = 0.0, wmax = 0.0;
    TRY(bound_check(e, q - 1, &r, &wmax));
    // r = ||w_{m+1} - w_m||_inf, wmax = max(w_{m+1}) (over the core rows of every rank)
    if (!std::isfinite(r) || !std::isfinite(wmax)) { e->neu_holds_vector = false; break; }   // NaN in the field: no bound from here (and no start for the next level)
    e->neu_holds_vector = true; e->neu_in_b = (x == e->d_wb);
    if (r < 1.0) best = std::min(best, (wmax - 1.0) / (1.0 - r));
    if (r <= 0.1 || best <= 3.0) break;                                      // within 11 % of max(w) - 1, or below what matters
    // a field whose rows are uniformly stiff gains nothing over its norm bound and would take the most sweeps to say so: where the
    // norm form is usable (s not clamped) and ||J^12 1|| is still above 0.3 (bulk row sums >= 0.9), stop (Ohio-sized band at CFL 18, 912 levels: 0.4 -> 0.1 s)
    if (!warm && done >= NEU_FIRST && r > 0.3 && e->err_factor[(size_t)t] < 300.0) break;
  }
  e->err_factor[(size_t)t] = best;
  return CWR_OK;
}

int refine_error_factors(cwr_engine* e) {
  const int T = e->T;
  if (T < 2 || e->neumann_sweeps <= 0 || e->err_factor.size() != (size_t)T || e->windowed) return CWR_OK;   // (windowed: per level, at the step)
  // an engine with halo rows and no communicator (yet): its halo rows would stay at w = 1 -- no bound of the global matrix;
  // cwr_attach_comm calls again
  if ((!e->comm || e->peers.empty()) && e->n_halo != 0) return CWR_OK;    // (a stand-alone rank likewise: its halo rows are frozen)
  for (int t = 0; t + 1 < T; ++t) TRY(refine_level(e, t));
  e->step_exchanges = e->step_overlapped = 0;
  return CWR_OK;
}

// Partitioned engines: a level at which ANY rank has real-cell inputs is taken non-speculatively by EVERY rank -- the step's
// tail holds a collective exchange, so all ranks must take the same path.  One sum all-reduce of a 0/1 vector over the levels;
// collective (every rank calls cwr_load_real_inputs, also with zero entries; or attaches its communicator afterwards).
int sync_input_levels(cwr_engine* e) {
  e->in_any.clear();
  if (!e->comm || e->world <= 1 || e->T <= 0) return CWR_OK;
  const size_t T = (size_t)e->T + 1;
  std::vector<double> flags(T, 0.0);
  for (const auto& kv : e->in_levels) if (kv.first >= 0 && (size_t)kv.first < T) flags[(size_t)kv.first] = 1.0;
  DevTmp<double> buf;
  TRY(dev_alloc(e, &buf.p, T));
  TRY(upload(e, buf.p, flags.data(), T));
  TRY(allreduce(e, buf.p, T));
  TRY(download(e, flags.data(), buf.p, T));
  e->in_any.assign(T, 0);
  for (size_t t = 0; t < T; ++t) e->in_any[t] = flags[t] > 0.0 ? 1 : 0;
  return CWR_OK;
}

// Constituent columns the engine carries internally for a caller's K (see cwr_create: zero columns where that is faster)
int pad_constituents(int K);

// Rows (lane-group slots) of a tile of the tiled pass for K constituents -- also what cwr_tile_rows tells a host wrapper that
// wants to arrange its cell numbering in tiles (ordering.balance_windows).
int tile_rows_for(int K, bool* four_wide) {
  int tr_target = 64;
  if (const char* v = getenv("CWR_TCL_ROWS")) tr_target = std::max(1, atoi(v));
  // wide rows: four constituents per lane halve the lanes that re-read every (weight, index) pair from LDS
  bool want4 = (K % 4 == 0) && K >= 8;
  if (const char* v = getenv("CWR_TCL_VW")) want4 = atoi(v) == 4 && (K % 4 == 0);
  const int VW = (K % 2 == 0) ? 2 : 1;
  const int Rt = want4 ? BLOCK / (K / 4) : BLOCK / (K / VW);     // rows one pass of the compute mapping covers
  int tr = tr_target;
  while (tr > Rt && (tr % Rt) != 0) --tr;
  tr = std::max(tr, Rt);
  if (want4) {
    // four-wide mapping: one row per lane group.  Two rows per lane group (84-102-row tiles at K = 20-24, 64-row tiles at
    // K = 32; configurations 7 and 8) were measured SLOWER on the merged 1 M-cell mesh: 191 / 222 / 271 us per pass against
    // 148 / 169 / 202 us at K = 20 / 24 / 32 (profiles/r02_b_per_K.txt) -- the extra prefetch registers cost a block per CU
    int ut = 1;
    if (const char* v = getenv("CWR_TCL_UT")) ut = std::max(1, std::min(2, atoi(v)));
    tr = Rt * ut;
  }
  if (four_wide) *four_wide = want4;
  return std::max(1, std::min(tr, BLOCK));
}

// (round 5) Constituent counts off the kernels' wide mappings fall off a cliff: odd K runs one constituent per lane, K = 10 two per
// lane on 5-lane groups -- 1 M cells, ms per step (profiles/r04_zc_per_K_final.txt): K = 3: 1.316 vs 4: 1.199; 5: 1.834 vs 6: 1.629 and
// 8: 1.757; 10: 2.285 vs 12: 2.146.  The reference's cost is linear in K (transport.py:231), so the engine carries such a K as the next
// count that runs well: zero columns behind the caller's (zero state, zero boundary values: they solve to zero, pass every check at
// once and are stripped at every read-out).  Table measured once (profiles/r05_per_K.txt); CWR_K_PAD=0: the caller's K as it is.
int pad_constituents(int K) {
  if (const char* v = getenv("CWR_K_PAD")) if (atoi(v) == 0) return K;
  if (K <= 2 || K > 252) return K;
  if (K <= 8) return (K & 1) ? K + 1 : K;                       // 3 -> 4, 5 -> 6, 7 -> 8
  if (K == 18) return K;                                         // (measured: 2.98 ms per step as it is, 3.14 as 20)
  return (K + 3) & ~3;                                           // 9, 10, 11 -> 12; 13, 14, 15 -> 16; 17, 19 -> 20; 21, 22, 23 -> 24; ...
}

#define CWR_TCL_K(VWv, Q) k_sq_tiled<VWv, TCL_CFG[Q].wrn, TCL_CFG[Q].ut, TCL_CFG[Q].xr>
#define CWR_TCL_PICK(KM)                                                                                                                 \
  if (vw == 4) return cfg == 3 ? reinterpret_cast<const void*>(&KM(4, 3)) : cfg == 4 ? reinterpret_cast<const void*>(&KM(4, 4))             \
                    : cfg == 5 ? reinterpret_cast<const void*>(&KM(4, 5)) : cfg == 6 ? reinterpret_cast<const void*>(&KM(4, 6))             \
                    : cfg == 7 ? reinterpret_cast<const void*>(&KM(4, 7)) : reinterpret_cast<const void*>(&KM(4, 8));                       \
  if (vw == 2) return cfg == 0 ? reinterpret_cast<const void*>(&KM(2, 0)) : cfg == 1 ? reinterpret_cast<const void*>(&KM(2, 1))             \
                    : cfg == 9 ? reinterpret_cast<const void*>(&KM(2, 9)) : reinterpret_cast<const void*>(&KM(2, 2));                       \
  return cfg == 0 ? reinterpret_cast<const void*>(&KM(1, 0)) : cfg == 1 ? reinterpret_cast<const void*>(&KM(1, 1))                          \
       : cfg == 9 ? reinterpret_cast<const void*>(&KM(1, 9)) : reinterpret_cast<const void*>(&KM(1, 2));
const void* tcl_kernel(int vw, int cfg) { CWR_TCL_PICK(CWR_TCL_K) }
#undef CWR_TCL_PICK

// Symbolic J^2 (once): row c of J^2 has the columns reachable in two face steps.  Numeric values per step on
// the device (k_sq_numeric; k_build_sq for very long rows), then c2 = bhat + J bhat with one plain sweep of bhat.
int ensure_sq_pattern(cwr_engine* e) {
  if (e->sq_pattern || e->sq_failed) return CWR_OK;
  // symbolic J^2 on the host (cwr_host_builders.hpp: also what the CPU sanitizer build exercises)
  host::SqPattern sqp;
  if (e->tcl_power == 1) { if (!host::symbolic_j(e->n_owned, e->n_core, e->h_ptr, e->h_nb, sqp)) { e->sq_failed = true; return CWR_OK; } }
  else
  if (!host::symbolic_sq(e->n_owned, e->n_core, e->max_degree, e->h_ptr, e->h_nb, sqp)) { e->sq_failed = true; return CWR_OK; }   // halo too shallow: plain sweeps only
  const int n = sqp.n_sq;
  e->n_sq = n;
  const std::vector<int32_t>& ptr2 = sqp.ptr2; const std::vector<int32_t>& col2 = sqp.col2; const std::vector<int32_t>& pair_ptr = sqp.pair_ptr;
  const std::vector<uint8_t>& slots = sqp.slots;
  bool rowwise = sqp.rowwise;
  e->sq_max_row = std::max(e->sq_max_row, sqp.max_row);
  e->nnz2 = (int)col2.size();
  const int TR = e->R * e->U;
  TRY(dev_alloc(e, &e->d_sq_fast, (size_t)n));
  TRY(upload(e, e->d_sq_fast, sqp.fast.data(), (size_t)n));
  int cap = 1;
  for (int b = 0; b * TR < n; ++b) cap = std::max(cap, ptr2[std::min((b + 1) * TR, n)] - ptr2[b * TR]);
  if (cap > 8192) { e->sq_failed = true; return CWR_OK; }      // would not fit LDS staging: stay with plain sweeps
  e->stage_cap2 = cap;
  e->apply_lds2 = ((size_t)cap * sizeof(FaceRec) + (size_t)red_doubles(e->G, e->VW) * sizeof(double) + (size_t)(TR + 1) * sizeof(int32_t) + 15) & ~(size_t)15;
  const void* fn = (e->VW == 2) ? reinterpret_cast<const void*>(&k_apply<2, 5>) : reinterpret_cast<const void*>(&k_apply<1, 5>);
  if (e->apply_lds2 > 48 * 1024) HIP_TRY(e, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->apply_lds2));
  int per_cu = 1, n_cu = 256;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, e->dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
  per_cu = resident_blocks(fn, e->apply_lds2);
  per_cu = std::min(per_cu, e->cu_cap);
  e->apply_grid2 = std::max(N_XCD, std::min(cdiv(cdiv(n, TR), N_XCD) * N_XCD, (n_cu * per_cu / N_XCD) * N_XCD));
  if (e->apply_grid2 > std::max(e->apply_grid, 256 * 8)) e->apply_grid2 = std::max(e->apply_grid, 256 * 8);   // partials buffer size
  TRY(dev_alloc(e, &e->d_ptr2, (size_t)n + 1));
  TRY(dev_alloc(e, &e->d_col2, (size_t)e->nnz2));
  TRY(dev_alloc(e, &e->d_row2, (size_t)e->nnz2));
  TRY(dev_alloc(e, &e->d_rec2, (size_t)e->nnz2));
  TRY(upload(e, e->d_ptr2, ptr2.data(), (size_t)n + 1));
  TRY(upload(e, e->d_col2, col2.data(), (size_t)e->nnz2));
  {
    int most = 1;
    for (int b = 0; b * SQN_THREADS < n; ++b) most = std::max(most, ptr2[std::min((b + 1) * SQN_THREADS, n)] - ptr2[b * SQN_THREADS]);
    e->sqn_lds = (size_t)most * sizeof(double);
    if (e->sqn_lds > 64 * 1024) rowwise = false;
    else if (e->sqn_lds > 48 * 1024)
      for (const void* fn : {reinterpret_cast<const void*>(&k_sq_numeric<4>), reinterpret_cast<const void*>(&k_sq_numeric<6>), reinterpret_cast<const void*>(&k_sq_numeric<8>)})
        HIP_TRY(e, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->sqn_lds));
  }
  if (rowwise && slots.size() < 2000000000u) {
    TRY(dev_alloc(e, &e->d_pair_ptr, (size_t)n + 1));
    TRY(dev_alloc(e, &e->d_slots, slots.size() + SQN_PAD));
    TRY(upload(e, e->d_pair_ptr, pair_ptr.data(), (size_t)n + 1));
    TRY(upload(e, e->d_slots, slots.data(), slots.size()));
    e->sq_rowwise = true;
  }
  {
    std::vector<int32_t> row2((size_t)e->nnz2);
    for (int c = 0; c < n; ++c) for (int q = ptr2[c]; q < ptr2[c + 1]; ++q) row2[q] = c;
    TRY(upload(e, e->d_row2, row2.data(), (size_t)e->nnz2));
  }
  // ---- tiled variant: distinct x rows per tile (own rows first) and local indices; only where a tile fits LDS
  // (partitioned engines too: the lists simply reach into the halo rows of x)
  if (e->use_tcl) {
    // partitioned engines: first try to tile every J^2 row (partition.py numbers the replayed layers 1..s-2 along the
    // cell curve, so their tiles are as compact as core tiles); if a halo tile does not fit, tile the core rows only and
    // run the replayed layers through the un-tiled pass
    for (int attempt = 0; attempt < 2 && !e->tcl_ready; ++attempt) {
    const int n_t = (e->comm && attempt == 1) ? e->n_core : n;
    if (attempt == 1 && (!e->comm || e->n_core == n)) break;
    bool want4 = false;
    int tr = tile_rows_for(e->K, &want4);
    const int R4 = want4 ? BLOCK / (e->K / 4) : 0;
    // a tile that holds too many entries or distinct rows for every compiled configuration (dense adjacency: many 5-8-face
    // cells; narrow rows use 256-row tiles) is retried at half the rows -- part of the lanes then idle in the compute phase,
    // which still beats the un-tiled exact pass by far
    for (int shrink = 0; shrink < 3 && !e->tcl_ready; ++shrink, tr = std::max(16, tr / 2)) {
    // tiles of work items (see k_sq_tiled; a -DCWR_WORK_ITEMS=1 build with CWR_TCL_SPLIT=1 only): a row of more than TCL_SEG
    // entries occupies one lane group per chunk, so a tile takes rows while rows + extra chunks fit the tr lane-group slots
    // of a pass.  Measured on the merged 1 M-cell mesh, us per pass split / not: K = 2: 46.1 / 44.6, 3: 83 / 73, 4: 70 / 64,
    // 6: 76 / 69, 16: 119 / 110 (profiles/r02_f_split_sweep.txt); K = 1: 38.3 / 37.1 with the tile-balanced numbering the
    // unsplit tiles allow (profiles/r02_r_k1_ab.txt): off by default everywhere.
    bool split = false;
    if (const char* v = getenv("CWR_TCL_SPLIT")) split = atoi(v) != 0;
    split = split && CWR_WORK_ITEMS && e->VW == 1 && !want4;     // (only the one-constituent-per-lane kernels carry the item logic)
    const int seg = split ? TCL_SEG : (1 << 20);
    const int nvmax = split ? TCL_NVMAX : 0;
    host::Tiling tl;
    if (!host::build_tiling(n_t, tr, seg, nvmax, e->K, e->n_real, ptr2, col2, tl)) continue;
    {
      // (round 5) a FEW windows heavier than the cheapest kernel configuration allows (a rank's window in which the replayed strips of
      // two neighbours meet) are cut into smaller tiles instead of deciding the configuration of all: see build_tiling
      const int q_first = want4 ? 3 : TCL_NARROW[0];
      const int col_lim = want4 ? TCL_CFG[q_first].xr * (BLOCK / (e->K / 2)) : TCL_CFG[q_first].xr * e->R;
      const int ent_lim = TCL_CFG[q_first].wrn * BLOCK;
      if (!split && !getenv("CWR_NO_TILE_CUT") && (tl.max_cols > col_lim || tl.cap2 > ent_lim)) {
        host::Tiling cut; int heavy = 0;
        if (host::build_tiling(n_t, tr, seg, nvmax, e->K, e->n_real, ptr2, col2, cut, col_lim, ent_lim, &heavy) &&
            heavy > 0 && heavy * 50 <= tl.ntiles() && cut.max_cols <= col_lim && cut.cap2 <= ent_lim) {
          if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] tiled J^2: %d of %d windows cut into smaller tiles (they held up to %d distinct x rows / %d entries; limits %d / %d)\n",
                                             heavy, tl.ntiles(), tl.max_cols, tl.cap2, col_lim, ent_lim);
          tl = std::move(cut);
          e->tiles_cut = true;
        }
      }
    }
    const std::vector<int32_t>&trow = tl.trow, &vptr = tl.vptr, &tptr = tl.tptr, &tcols = tl.tcols;
    const std::vector<uint16_t>&vtab = tl.vtab, &loc2 = tl.loc2;
    const int nt = tl.ntiles(), max_cols = tl.max_cols, cap2 = tl.cap2;
    // (chained passes: + a staging area for a tile's results, which the next tile of the block's list carries over -- only where
    // the lists will be long enough to chain, so that engines below that size keep their LDS footprint and resident blocks)
    auto lds_for = [&](int own) {
      return ((size_t)(max_cols + nvmax + own) * e->K * sizeof(double) + (size_t)cap2 * (sizeof(double) + sizeof(uint16_t)) +
              (size_t)(tr + 1 + nvmax) * sizeof(int32_t) + 15) & ~(size_t)15; };
    int own_cap = 0;
    size_t lds = lds_for(0);
    // the kernel's compile-time prefetch depths bound what a tile may hold; otherwise the plain J^2 pass stays
    e->tcl_cfg = -1;
    int q0 = 0;
    if (const char* v = getenv("CWR_TCL_CFG")) q0 = std::max(0, std::min(3, atoi(v)));       // (index into TCL_NARROW)
    e->tcl_vw = e->VW;
    // (fetch mapping: K/2 lanes per row; compute mapping: K/4 lanes per row)
    int q4 = 3;
    if (const char* v = getenv("CWR_TCL_CFG")) q4 = std::max(3, std::min(8, atoi(v)));
    for (int q = q4; q < 9 && want4 && e->tcl_cfg < 0; ++q)
      if (max_cols <= TCL_CFG[q].xr * (BLOCK / (e->K / 2)) && cap2 <= TCL_CFG[q].wrn * BLOCK && tr <= TCL_CFG[q].ut * R4) { e->tcl_cfg = q; e->tcl_vw = 4; }
    for (int qi = q0; qi < 4 && e->tcl_cfg < 0; ++qi) {
      const int q = TCL_NARROW[qi];
      if (max_cols <= TCL_CFG[q].xr * e->R && cap2 <= TCL_CFG[q].wrn * BLOCK && tr <= TCL_CFG[q].ut * e->R) e->tcl_cfg = q;
    }
    if (lds <= 64 * 1024 && e->tcl_cfg >= 0 && tr <= BLOCK && (int64_t)max_cols * e->K <= 65535) {
      const void* fn6 = tcl_kernel(e->tcl_vw, e->tcl_cfg);
      auto grid_for = [&](size_t l) {
        int pcq = std::min(resident_blocks(fn6, l), 8);
        if (const char* v = getenv("CWR_TCL_BLOCKS_PER_CU")) pcq = std::max(1, std::min(pcq, atoi(v)));
        int g = std::max(N_XCD, std::min(cdiv(nt, N_XCD) * N_XCD, (n_cu * pcq / N_XCD) * N_XCD));
        if (const char* v = getenv("CWR_TCL_GRID")) g = std::max(N_XCD, std::min(g, atoi(v) / N_XCD * N_XCD));
        return g; };
      if (e->use_chains && e->chain_reuse && !split && lds_for(tr) <= 64 * 1024 && nt >= e->chain_min_tiles * grid_for(lds_for(tr))) {
        own_cap = tr; lds = lds_for(tr);
      }
      e->n_tcl = n_t;
      e->tcl_seg = seg; e->tcl_nvmax = nvmax;
      e->tcl_TR = tr; e->tcl_ntiles = nt; e->tcl_max_cols = max_cols; e->tcl_stage_cap = cap2; e->tcl_lds = lds;
      e->tcl_total_cols = tcols.size();
      e->own_cap = own_cap;
      if (own_cap > 0) { e->h_tcl_ptr = tptr; e->h_tcl_cols = tcols; }
      e->h_trow = trow;
      e->tcl_grid = grid_for(lds);
      TRY(dev_alloc(e, &e->d_tcl_ptr, (size_t)nt + 1));
      TRY(dev_alloc(e, &e->d_trow, (size_t)nt + 1));
      TRY(dev_alloc(e, &e->d_vptr, (size_t)nt + 1));
      {
        const std::vector<int32_t> meta = host::tile_meta(n_t, ptr2, tl);
        TRY(dev_alloc(e, &e->d_meta, meta.size()));
        TRY(upload(e, e->d_meta, meta.data(), meta.size()));
      }
      TRY(upload(e, e->d_trow, trow.data(), (size_t)nt + 1));
      TRY(upload(e, e->d_vptr, vptr.data(), (size_t)nt + 1));
      TRY(dev_alloc(e, &e->d_tcl_cols, tcols.size()));
      TRY(dev_alloc(e, &e->d_loc2, (size_t)e->nnz2));
      TRY(dev_alloc(e, &e->d_w2, (size_t)e->nnz2));
      TRY(upload(e, e->d_tcl_ptr, tptr.data(), (size_t)nt + 1));
      TRY(upload(e, e->d_tcl_cols, tcols.data(), tcols.size()));
      TRY(upload(e, e->d_loc2, loc2.data(), (size_t)e->nnz2));
      if (e->comm) {
        // interior tiles: every row they hold and every x row they read is a core row -- no exchange touches them
        std::vector<int32_t> inner, outer;
        host::split_interior(e->n_core, tl, inner, outer);
        e->n_tile_inner = (int)inner.size(); e->n_tile_outer = (int)outer.size();
        e->h_tile_inner = inner; e->h_tile_outer = outer;
        TRY(dev_alloc(e, &e->d_tile_inner, inner.size()));
        TRY(dev_alloc(e, &e->d_tile_outer, outer.size()));
        TRY(upload(e, e->d_tile_inner, inner.data(), inner.size()));
        TRY(upload(e, e->d_tile_outer, outer.data(), outer.size()));
        if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] rank %d: %d interior tiles overlap the exchange, %d cut tiles wait for it\n", e->rank, e->n_tile_inner, e->n_tile_outer);
      }
      e->tcl_ready = true;
      if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] tiled J^2: %d tiles of <= %d items (%.1f rows + %.2f extra chunks of long rows each), cfg %d, %.2f distinct x rows per row, max %d per tile, lds=%zu, grid=%d\n",
                                         nt, tr, (double)n_t / nt, (double)vtab.size() / nt, e->tcl_cfg, (double)tcols.size() / n_t, max_cols, lds, e->tcl_grid);
    } else if (getenv("CWR_VERBOSE")) {
      fprintf(stderr, "[cwr] tiled J^2 not used over %d rows: tile=%d rows, max %d distinct x rows per tile (limit %d), %d entries per tile (limit %d), lds=%zu\n",
              n_t, tr, max_cols, TCL_CFG[2].xr * e->R, cap2, TCL_CFG[9].wrn * BLOCK, lds);
    }
    }
    }
  }
  e->sq_pattern = true;
  if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] J^2: nnz2=%d (%.1f per row) stage_cap2=%d lds2=%zu grid2=%d\n", e->nnz2, (double)e->nnz2 / n, e->stage_cap2, e->apply_lds2, e->apply_grid2);
  return CWR_OK;
}

// ---- chained passes: the tile schedule along the flow --------------------------------------------------------------------
// The tiled pass is a persistent grid whose blocks each walk a list of tiles.  Walked in the default order and ping-ponging
// between two vectors it is a block-Jacobi iteration between tiles: information crosses a tile per pass.  Here the tiles are
// linked into CHAINS along the flow of one time level -- tile t -> the neighbour tile u that takes most of t's outflow, kept
// when u's largest inflow comes from t -- and every block walks chains IN PLACE, so that a tile reads what its upstream
// neighbour of the same chain has just written: block Gauss-Seidel along the flow, with no block ever waiting for another
// (the order only decides how fresh the values a tile reads are: a schedule built for another flow direction costs passes,
// never correctness; Chazan-Miranker: converges whenever rho(|J|) < 1).  tests/models/chain_gs_probe.py is the numpy model,
// clearwater-riverine_amd/schedule.py the numpy specification of this builder (compared in tests/test_gpu_chains.py).
// Measured on the 1 M-cell bench mesh x 16: 57 -> 43-47 sweep equivalents per step at CFL 2.5, 502 -> 181 at CFL 25 with four
// tile-local applications (profiles/r03_c_chained_passes.txt).
int build_tile_links(cwr_engine* e) {
  if (e->n_links > 0 || !e->tcl_ready || e->tcl_seg < (1 << 20)) return CWR_OK;     // (fixed-size tiles only: tile = row / TR)
  host::TileLinks lk;
  host::build_links(e->n_tcl, e->tcl_TR, e->tcl_ntiles, e->h_ptr, e->h_nb, e->h_edge, lk, e->tiles_cut ? &e->h_trow : nullptr);
  e->link_src = lk.src; e->link_dst = lk.dst;
  const std::vector<int32_t>&lptr = lk.lptr, &lent = lk.lent;
  e->n_links = (int)e->link_src.size();
  if (e->n_links == 0) return CWR_OK;
  TRY(dev_alloc(e, &e->d_link_ptr, lptr.size()));
  TRY(dev_alloc(e, &e->d_link_ent, lent.size()));
  TRY(dev_alloc(e, &e->d_link_flux, (size_t)e->n_links));
  TRY(upload(e, e->d_link_ptr, lptr.data(), lptr.size()));
  TRY(upload(e, e->d_link_ent, lent.data(), lent.size()));
  return CWR_OK;
}

using host::chains_to_schedule;          // chains -> schedule [depth][grid], -1 padded (cwr_host_builders.hpp; schedule.py: the same construction)

int install_schedule(cwr_engine* e, const std::vector<int32_t>& sched, int depth) {
  const size_t cnt = sched.size();
  if ((int)cnt > e->sched_cap || depth != e->sched_depth) {
    // (buffer pointer and depth are captured kernel arguments of the batch graphs)
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (auto& kv : e->batch_exec) if (kv.second) hipGraphExecDestroy(kv.second);
    e->batch_exec.clear(); e->batch_last = -1;
    for (auto& kv : e->stretch_exec) if (kv.second) hipGraphExecDestroy(kv.second);
    e->stretch_exec.clear();
  }
  if ((int)cnt > e->sched_cap) {
    hipFree(e->d_sched); e->d_sched = nullptr; e->sched_cap = 0;
    TRY(dev_alloc(e, &e->d_sched, cnt + 1024));
    e->sched_cap = (int)(cnt + 1024);
  }
  TRY(upload(e, e->d_sched, sched.data(), cnt));
  e->sched_depth = depth;
  return CWR_OK;
}

// Per-schedule column lists: a column the PREVIOUS tile of the same list holds in LDS is coded -2 - (its position there)
// (cwr_host_builders.hpp).  scols starts as a copy of the tiles' column lists; only the tiles of `sched` are rewritten.
void reuse_codes(const cwr_engine* e, const std::vector<int32_t>& sched, int grid, int depth, std::vector<int32_t>& scols) {
  host::reuse_codes(e->n_real, e->h_tcl_ptr, e->h_tcl_cols, sched, grid, depth, scols);
}

int install_sub_schedule(cwr_engine* e, cwr_engine::SubSched& ss, const std::vector<int32_t>& sched, int depth, int grid) {
  const size_t cnt = sched.size();
  if ((int)cnt > ss.cap) {
    if (ss.d) hipFree(ss.d);
    ss.d = nullptr; ss.cap = 0;
    TRY(dev_alloc(e, &ss.d, cnt + 1024));
    ss.cap = (int)(cnt + 1024);
  }
  if (cnt > 0) TRY(upload(e, ss.d, sched.data(), cnt));
  ss.depth = depth; ss.grid = grid;
  return CWR_OK;
}

int build_chain_schedule(cwr_engine* e, int t) {
  const auto w0 = std::chrono::steady_clock::now();
  TRY(build_tile_links(e));
  if (e->n_links == 0) { e->sched_level = t; return CWR_OK; }                // a single tile, or variable tiles: nothing to chain
  const int nt = e->tcl_ntiles, L = e->n_links;
  k_link_flux<<<cdiv(L, BLOCK), BLOCK, 0, e->stream>>>(L, e->d_link_ptr, e->d_link_ent, e->adv_l(t), e->d_link_flux);
  HIP_TRY(e, hipGetLastError());
  std::vector<float> flux((size_t)L);
  TRY(download(e, flux.data(), e->d_link_flux, (size_t)L));
  // tile t -> nxt[t]: the destination of its largest outflow, kept when that tile's largest inflow comes from t
  std::vector<int32_t> nxt;
  {
    host::TileLinks lk;                                                      // (only src / dst are read)
    lk.src = e->link_src; lk.dst = e->link_dst;
    host::chains_from_flux(nt, lk, flux, nxt);
  }
  e->sched_level = t;
  ++e->n_sched_builds;
  if (e->sched_depth > 0 && !e->sched_user && nxt == e->sched_nxt) {                     // the same chains: the lists stand
    if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] chained passes: level %d keeps the chains of the installed schedule (%.2f ms)\n", t,
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    return CWR_OK;
  }
  const bool reuse = e->own_cap > 0 && !e->h_tcl_ptr.empty();
  std::vector<int32_t> sched; int depth = 0;
  chains_to_schedule(nt, e->tcl_grid, reuse ? 1 : 2, nxt, sched, depth);
  if (reuse) {
    std::vector<int32_t> scols(e->h_tcl_cols);
    reuse_codes(e, sched, e->tcl_grid, depth, scols);
    if (!e->d_scols) TRY(dev_alloc(e, &e->d_scols, scols.size()));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    TRY(upload(e, e->d_scols, scols.data(), scols.size()));
  }
  TRY(install_schedule(e, sched, depth));
  if (e->comm && e->overlap && e->comm_stream && !e->h_tile_inner.empty() && reuse) {
    // the same chains cut at the boundary between interior and cut tiles: two schedules for the pass an exchange runs beside
    // (the interior lists leave a few block slots to RCCL's copy kernels, like the interior launch of the ping-pong passes)
    int gi = e->tcl_grid;
    if (gi > 4 * e->overlap_reserve) gi = std::max(N_XCD, (gi - e->overlap_reserve) / N_XCD * N_XCD);
    gi = std::max(N_XCD, std::min(gi, cdiv((int)e->h_tile_inner.size(), N_XCD) * N_XCD));
    const int go = std::max(N_XCD, std::min(e->tcl_grid, cdiv((int)e->h_tile_outer.size(), N_XCD) * N_XCD));
    std::vector<int32_t> s_in, s_out; int d_in = 0, d_out = 0;
    chains_to_schedule(nt, gi, 1, nxt, s_in, d_in, &e->h_tile_inner);
    chains_to_schedule(nt, go, 1, nxt, s_out, d_out, &e->h_tile_outer);
    std::vector<int32_t> scols(e->h_tcl_cols);
    reuse_codes(e, s_in, gi, d_in, scols);
    reuse_codes(e, s_out, go, d_out, scols);
    if (!e->d_scols_io) TRY(dev_alloc(e, &e->d_scols_io, scols.size()));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->comm_stream) HIP_TRY(e, hipStreamSynchronize(e->comm_stream));
    TRY(upload(e, e->d_scols_io, scols.data(), scols.size()));
    TRY(install_sub_schedule(e, e->sched_in, s_in, d_in, gi));
    TRY(install_sub_schedule(e, e->sched_out, s_out, d_out, go));
  }
  e->sched_nxt = nxt;
  if (getenv("CWR_VERBOSE")) {
    int linked = 0; for (int a = 0; a < nt; ++a) linked += nxt[(size_t)a] >= 0;
    fprintf(stderr, "[cwr] chained passes: schedule for level %d: %d of %d tiles have a chain successor, %d lists x %d slots (%.2f ms)\n", t, linked, nt, e->tcl_grid, depth,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
  }
  return CWR_OK;
}

// Partitioned engines, once, at the first Jacobi solve (collective): the exchanges of a batch must be the same on every rank, and
// they follow from the batch shape -- chained ranks always close with one plain sweep, ping-pong ranks choose by the sweep count;
// tiled ranks refresh their halos before the closing sweep.  So: every rank chains or none does (ranks of very different size,
// or a middle rank whose two halos lift it over the three-tiles-per-block threshold, would otherwise differ), and the closing
// exchange is forced everywhere as soon as one rank runs tiled passes.
int agree_on_pass_shape(cwr_engine* e, bool tiled) {
  const bool can_chain = tiled && e->use_chains && !e->two_closing &&
                         (e->sched_user ? e->sched_depth > 0 : e->tcl_ntiles >= e->chain_min_tiles * e->tcl_grid);
  double h[2] = {can_chain ? 0.0 : 1.0, tiled ? 1.0 : 0.0};
  DevTmp<double> buf;
  TRY(dev_alloc(e, &buf.p, 2));
  TRY(upload(e, buf.p, h, 2));
  TRY(allreduce(e, buf.p, 2));
  TRY(download(e, h, buf.p, 2));
  if (h[0] > 0.0 && e->use_chains) {
    e->use_chains = false;
    if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] rank %d: %d rank(s) cannot chain their tiles: ping-pong passes on all ranks\n", e->rank, (int)h[0]);
  }
  e->any_tiled = h[1] > 0.0;
  e->shape_agreed = true;
  return CWR_OK;
}

// numeric J^2 and c2 (into d_t) for the step whose operator is prepared; active = false -> plain sweeps only
int prepare_sq(cwr_engine* e, bool& active) {
  active = false;
  if (!e->use_sq || e->sq_failed || e->K < e->sq_min_k) return CWR_OK;
  TRY(ensure_sq_pattern(e));
  if (!e->sq_pattern) return CWR_OK;
  // (the entry weights w were written by k_prep_step)
  const bool need_rec2 = !e->tcl_ready || e->n_sq > e->n_tcl;      // the un-tiled pass reads FaceRec-format rows
  if (e->tcl_power == 1) {
    // (A/B, round 6) J's own entries on the merged pattern (faces between the same two cells summed in face order); the passes' constant is bhat
    k_j_numeric<<<cdiv(e->n_sq, BLOCK), BLOCK, 0, e->stream>>>(e->n_sq, e->d_ptr, e->d_ent_nb, e->d_w, e->d_ptr2, e->d_col2, need_rec2 ? e->d_rec2 : nullptr,
                                                          e->tcl_ready ? e->d_w2 : nullptr);
    HIP_TRY(e, hipGetLastError());
    active = true;
    return CWR_OK;
  }
#define CWR_SQN(DEGv) k_sq_numeric<DEGv><<<cdiv(e->n_sq, SQN_THREADS), SQN_THREADS, e->sqn_lds, e->stream>>>(e->n_sq, e->d_ptr, e->d_ent_nb, \
        e->d_w, e->d_ptr2, e->d_col2, e->d_pair_ptr, e->d_slots, e->d_sq_fast, need_rec2 ? e->d_rec2 : nullptr, e->tcl_ready ? e->d_w2 : nullptr)
  if (e->sq_rowwise) { if (e->max_degree <= 4) CWR_SQN(4); else if (e->max_degree <= 6) CWR_SQN(6); else CWR_SQN(8); }
#undef CWR_SQN
  else
    k_build_sq<<<cdiv(e->nnz2, BLOCK), BLOCK, 0, e->stream>>>(e->nnz2, e->d_ptr, e->d_ent_nb, e->d_w, e->d_row2, e->d_col2, e->d_rec2, e->tcl_ready ? e->d_w2 : nullptr);
  HIP_TRY(e, hipGetLastError());
  const int keep = e->dominant_mode; e->dominant_mode = -1;                    // this set-up launch is not a profiled sweep
  const int rc = launch_apply<4>(e, e->d_b, e->d_t, nullptr, e->d_b, nullptr, nullptr);   // c2 = bhat + J bhat
  e->dominant_mode = keep;
  if (rc != CWR_OK) return rc;
  active = true;
  return CWR_OK;
}

// tile_list (device, optional): the launch covers only these `n_list` tiles (interior / cut tiles of a partitioned engine)
// chained = true: every block walks its own list of the schedule e->d_sched (all tiles; single GPU); xin == yout then makes
// the pass an in-place (block Gauss-Seidel along the chains) relaxation
// sub (optional, with chained): walk this schedule (the interior or the cut tiles of a partitioned engine) instead of the full one
int launch_sq_tiled(cwr_engine* e, const double* xin, double* yout, const int32_t* tile_list = nullptr, int n_list = 0, bool tail = true,
                    bool chained = false, const cwr_engine::SubSched* sub = nullptr) {
  const int ntiles = tile_list ? n_list : e->tcl_ntiles;
  if (sub && sub->depth <= 0) {                        // (no such tiles on this rank)
    if (tail && e->n_sq > e->n_tcl) TRY(launch_apply<5>(e, xin, yout, nullptr, e->c2(), nullptr, nullptr, e->n_sq, e->n_tcl));
    return CWR_OK;
  }
  if (ntiles <= 0) return CWR_OK;
  int grid = std::max(N_XCD, std::min(e->tcl_grid, cdiv(ntiles, N_XCD) * N_XCD));
  // an interior launch that runs beside an exchange leaves a few block slots free: the grid is persistent (every resident
  // slot taken until the launch ends), so RCCL's copy kernels could otherwise only start when it is over
  if (tile_list && !tail && grid > 4 * e->overlap_reserve) grid -= e->overlap_reserve;
  int depth = 0;
  const int32_t* scols = nullptr;
  if (chained) { tile_list = e->d_sched; depth = e->sched_depth; grid = e->tcl_grid; scols = e->d_scols; }
  if (chained && sub) { tile_list = sub->d; depth = sub->depth; grid = sub->grid; scols = e->d_scols_io; }
  const int inplace = (xin == yout) ? 1 : 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->profiling && e->dominant_mode == 6 && e->ev_used + 2 <= e->ev.size()) {
    e0 = e->ev[e->ev_used++]; e1 = e->ev[e->ev_used++];
    HIP_TRY(e, hipEventRecord(e0, e->stream));
  }
#define CWR_TILED(VWv, Q) CWR_TCL_K(VWv, Q)<<<grid, BLOCK, e->tcl_lds, e->stream>>>(e->K, e->K / VWv, e->tcl_TR, ntiles, tile_list, depth, inplace,    \
      e->d_trow, e->d_ptr2, e->d_loc2, e->d_w2, e->d_tcl_ptr, e->d_tcl_cols, e->d_vptr, e->d_meta, e->tcl_max_cols, e->tcl_stage_cap,    \
      e->local_reps, e->tcl_seg, e->tcl_nvmax, xin, e->c2(), yout, scols, e->own_cap)
  if (e->tcl_vw == 4) { if (e->tcl_cfg == 3) CWR_TILED(4, 3); else if (e->tcl_cfg == 4) CWR_TILED(4, 4); else if (e->tcl_cfg == 5) CWR_TILED(4, 5);
                        else if (e->tcl_cfg == 6) CWR_TILED(4, 6); else if (e->tcl_cfg == 7) CWR_TILED(4, 7); else CWR_TILED(4, 8); }
  else if (e->VW == 2) { if (e->tcl_cfg == 0) CWR_TILED(2, 0); else if (e->tcl_cfg == 1) CWR_TILED(2, 1); else if (e->tcl_cfg == 9) CWR_TILED(2, 9); else CWR_TILED(2, 2); }
  else            { if (e->tcl_cfg == 0) CWR_TILED(1, 0); else if (e->tcl_cfg == 1) CWR_TILED(1, 1); else if (e->tcl_cfg == 9) CWR_TILED(1, 9); else CWR_TILED(1, 2); }
#undef CWR_TILED
  HIP_TRY(e, hipGetLastError());
  if (e1) HIP_TRY(e, hipEventRecord(e1, e->stream));
  if (tail && e->n_sq > e->n_tcl)                     // replayed halo layers (partitioned engines): un-tiled J^2 rows
    TRY(launch_apply<5>(e, xin, yout, nullptr, e->c2(), nullptr, nullptr, e->n_sq, e->n_tcl));
  return CWR_OK;
}

// What follows the solve of step t: ghost write-back (transport.py:258-264) and, on request, the per-face mass fluxes
// (transport.py:406-429).  On one GPU the Jacobi path enqueues it SPECULATIVELY right behind the batch whose convergence
// check is about to be downloaded: when the check passes (the steady state: one check per step) the GPU went straight on
// instead of idling through the host round trip; when it fails, more sweeps follow and the tail simply runs again
// (it only writes ghost rows, which no sweep reads, and the flux arrays).
int step_tail(cwr_engine* e, int t, int flags) {
  const int K = e->K;
  const int64_t gk = (int64_t)e->n_ghost * K;
  {
    auto it = e->in_levels.find(t + 1);             // transport.py:258-264 on real cells (never speculative: see cwr_step)
    if (it != e->in_levels.end() && it->second.second > 0) {
      const int64_t total = (int64_t)it->second.second * K;
      k_apply_inputs<<<cdiv(total, BLOCK), BLOCK, 0, e->stream>>>(total, K, e->d_in_rows + it->second.first,
                                                                  e->d_in_vals + (size_t)it->second.first * K, e->d_c);
      HIP_TRY(e, hipGetLastError());
    }
  }
  if (gk > 0 && !e->fused_begin) {                    // (k_begin_step has written them)
    k_ghost_writeback<<<cdiv(gk, BLOCK), BLOCK, 0, e->stream>>>(gk, e->d_bc + (size_t)(t + 1) * gk, e->d_c + (size_t)e->n_real * K);
    HIP_TRY(e, hipGetLastError());
  }
  if (flags & CWR_STEP_MASS_FLUX) {
    if (!e->d_fadv) {
      const size_t cnt = (size_t)e->E * K;
      TRY(dev_alloc(e, &e->d_fadv, cnt)); TRY(dev_alloc(e, &e->d_fdif, cnt));
    }
    const float* adv_t = e->adv_l(t);
    const double* dif_t = e->dif_l(t);
    auto flux = [&](const int32_t* list, int n_list) -> int {
      const int nf = list ? n_list : e->E;
      if (nf <= 0) return CWR_OK;
      const int grid = std::max(1, std::min(cdiv(nf, e->R), 256 * 8));
      if (e->VW == 2) k_mass_flux<2><<<grid, BLOCK, 0, e->stream>>>(e->E, e->n_core, K, e->G, e->d_f1, e->d_f2, adv_t, dif_t, e->dt[t], e->d_c, e->d_fadv, e->d_fdif, list, n_list);
      else            k_mass_flux<1><<<grid, BLOCK, 0, e->stream>>>(e->E, e->n_core, K, e->G, e->d_f1, e->d_f2, adv_t, dif_t, e->dt[t], e->d_c, e->d_fadv, e->d_fdif, list, n_list);
      HIP_TRY(e, hipGetLastError());
      return CWR_OK;
    };
    // (round 4) partitioned engines: the exchange that closes the step runs beside the faces between core cells
    const bool split = e->comm && e->overlap && e->comm_stream && e->n_face_inner > 0 && !e->peers.empty() && !getenv("CWR_NO_TAIL_OVERLAP");
    if (split) {
      if (e->test_poison_halo && e->n_real > e->n_core)
        HIP_TRY(e, hipMemsetAsync(e->d_c + (size_t)e->n_core * K, 0xFF, (size_t)(e->n_real - e->n_core) * K * sizeof(double), e->stream));
      TRY(exchange_begin(e, e->d_c));
      TRY(flux(e->d_face_inner, e->n_face_inner));
      TRY(exchange_finish(e, e->d_c, nullptr));
      TRY(flux(e->d_face_outer, e->n_face_outer));
    } else {
      TRY(exchange_halo(e, e->d_c));
      TRY(flux(nullptr, 0));
    }
    e->flux_valid = true;
    e->halo_fresh = true;
  }
  return CWR_OK;
}

struct SolveStats {
  int iterations = 0, sweeps = 0, restarts = 0, launches = 0, status = CWR_OK, sweep_kernel = 0;
  double max_rel = 0.0;
};

// Fully fused Jacobi sweeps x <- x + (bhat - D^-1 A x): one operator launch (and, partitioned, one halo
// exchange) per sweep, no inner products between checks.  ||x' - x|| of a sweep is the scaled residual of
// its input, so the check after a batch is exact.  The measured contraction predicts the sweeps still
// needed; when that exceeds what BiCGSTAB would cost (stiff steps: large CFL), or the residual grows,
// the caller switches to BiCGSTAB from the current iterate.
int solve_jacobi(cwr_engine* e, double tol2, int max_iter, bool forced, SolveStats& st, bool& need_bicg) {
  const int K = e->K;
  need_bicg = false;
  std::vector<double> h(4 * (size_t)K);
  // the reduced check scalars (||x'-x||^2, ||bhat||^2 | element-wise maxima) land side by side: one download per check
  double prev_worst = -1.0;
  int prev_sweeps = 0;
  int since_exchange = 0;                     // the caller exchanged the state's halo just before the right-hand side
  // sweeps the next batch should add (prediction, unrounded; the margin only where batches come in steps of one or two
  // sweeps: the even-passes shape rounds up to 2 (mod 4) and has its slack built in)
  const int margin = e->two_closing ? 0 : e->sweep_margin;
  int want = (e->last_sweeps > 0) ? std::max(2, e->last_sweeps + margin) : 8;
  if (e->fixed_sweeps > 0) want = e->fixed_sweeps;
  int batch = 0;
  const int sweep_limit = forced ? max_iter : std::min(max_iter, e->jacobi_limit);
  bool sq = false;
  TRY(prepare_sq(e, sq));
  const bool tiled = sq && e->tcl_ready;
  if (e->comm && !e->shape_agreed) TRY(agree_on_pass_shape(e, tiled));
  if (tiled && e->use_chains && !e->two_closing && !e->sched_user && e->tcl_ntiles >= e->chain_min_tiles * e->tcl_grid && (!e->deterministic || e->det_walk) &&
      (e->sched_level < 0 || std::abs(e->cur_t - e->sched_level) >= e->sched_refresh))
    // (worth it from a few tiles per block up: CWR_CHAIN_MIN_TILES, default 3)
    TRY(build_chain_schedule(e, e->cur_t));
  if (e->comm && sq && e->n_real > e->n_core)
    // the ping-pong partner starts with this step's halo values too (its never-computed outer layers would otherwise
    // still hold the previous step's): block-asynchronous passes spread what those layers hold four rows per pass
    HIP_TRY(e, hipMemcpyAsync(e->d_p + (size_t)e->n_core * K, e->d_c + (size_t)e->n_core * K,
                              (size_t)(e->n_real - e->n_core) * K * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  e->dominant_mode = tiled ? 6 : (sq ? 5 : 4);
  st.sweep_kernel = e->dominant_mode;
  const bool noted = check_by_note(e);
  for (;;) {
    want = std::min(want, std::max(2, sweep_limit - st.sweeps));       // max_iter bounds the first batch too
    batch = std::max(2, std::min((want + 1) & ~1, 4096));               // even: the result lands in the state vector
    int launches = batch;
    int todo = batch;
    bool batch_graph = false;
    if (sq) {
      // One GPU: batch = 2*doubles + 1 -- `doubles` J^2 passes, then ONE plain sweep whose ||x'-x|| is the exact scaled
      // residual of its input (the convergence criterion) and whose output is the answer.  The launches ping-pong between
      // the state vector and its partner and must END in the state vector: with an odd number of launches the first pass
      // reads x_t from the copy k_rhs keeps for a failed step (d_keep: every computed row) and writes the state vector,
      // which shifts the parity -- first batch of a step only; later batches take one pass more instead.  (Round 1 used an
      // even number of passes and two closing sweeps: 4-sweep granularity and a plain sweep, 85 us at K = 16, more.)
      // That shape is still the cheaper one when the sweeps wanted are 2 (mod 4): N passes + 2 sweeps against N + 1 passes +
      // 1 sweep, and a plain sweep costs less than a pass.  Partitioned engines always use it: their halo layers and
      // exchanges are counted in pairs of sweeps.
      // Chained passes (one GPU, a tile schedule along the flow is installed): the passes relax IN PLACE in the partner vector
      // -- the first pass of a step reads x_t from the state vector and writes the partner, every later pass reads and writes
      // the partner -- and the closing sweep carries the result into the state vector: any number of passes, one closing sweep.
      // (A later batch of the same step goes on in the partner; the closing sweep's own progress is not used.)
      // Partitioned engines (round 3): the same, between the halo exchanges -- which then run on the engine's stream in front of
      // the pass that needs them, or -- where a rank has interior tiles -- beside the lists of its interior tiles, which are
      // chained separately from the cut tiles for that pass (build_chain_schedule).  Every rank chains or none does
      // (agree_on_pass_shape): the batch shape, and with it the exchanges of a batch, must be the same on all ranks.
      const bool chained = tiled && e->use_chains && e->sched_depth > 0 && !e->two_closing && !e->deterministic;
      // Deterministic steps of a single engine WALK the same lists, ping-ponging between the two vectors: a tile takes its
      // predecessor's rows from LDS (fresh: block Gauss-Seidel along a list, which is where the flow carries the information) and
      // every other row from the pass's input vector, which no block writes -- nothing depends on timing.
      // (partitioned engines too, when the ranks agreed to chain: the same lists, cut into interior and cut tiles for the passes with an exchange)
      const bool walk = chained || (tiled && e->use_chains && e->sched_depth > 0 && !e->two_closing && e->deterministic && e->det_walk && e->d_scols &&
                                    (!e->comm || !e->sched_user));
      const bool first_batch = st.sweeps == 0;
      e->step_chained = chained ? 1 : (walk ? 2 : 0);
      if (e->reps_auto) {
        // Tile-local applications per visit.  A chain carries information from tile to tile only as far as the applications
        // carry it across a tile, and the stiffer the step the more of its sweeps are transport along the flow.  Measured on the
        // 1 M-cell mesh x 16 with column reuse (profiles/r03_c_chained_passes.txt, E): CFL 2.5 (||J||_inf 0.78): x2 2.77 ms per
        // step, x3 3.01; CFL 25 (0.973): x2 15.3, x4 10.8, x6 12.0; CFL 62 (0.989): x4 22.1, x6 19.0; CFL 225 (0.9969): x4 39.2,
        // x8 34.9 (ping-pong x2: 3.69 / 27.4 / 55.7 / 140.8).  ||J||_inf of the step is known from the flow field (k_jnorm).
        const double rho = ((size_t)e->cur_t < e->jnorm.size()) ? e->jnorm[(size_t)e->cur_t] : 0.0;
        // Ping-pong passes (engines below the chain threshold, deterministic steps; round 4): information crosses one tile per pass
        // whatever the applications, but a stiff step relaxes its tiles' interiors too slowly with two -- same box, ms per step at
        // x2 / x3 / x4 / x6 (profiles/r04_f_small_engines.txt): 10 k cells x 12 at CFL 18: 0.75 / 0.67 / 0.66 / 0.69; 8 k x 1: 0.59 /
        // 0.50 / 0.48 / 0.45; 119 k x 16 at CFL 25: 3.83 / 3.42 / 3.52 / 3.73; x 1: 1.38 / 1.21 / 1.17 / 1.21 -- while at CFL 2.5
        // (||J||_inf 0.78) two stay the cheapest (119 k x 16: 0.70 / 0.71 / 0.76).
        const int pp = (rho < 0.9 || getenv("CWR_NO_PP_REPS")) ? e->reps_base : std::max(e->reps_base, e->K <= 2 ? 6 : 4);   // (CWR_NO_PP_REPS=1: round 3's fixed count, A/B)
        // (round 4, after the numbering changed -- smoothed lane boundaries, 3-cell tiles -- the stiff steps want FEWER applications than
        // round 3 measured: 1 M x 16, ms per step at x2 / x3 / x4 / x6 / x8 (profiles/r04_zb): CFL 12 (||J||_inf 0.95): 6.49 / 5.71 / 6.21; CFL 25
        // (0.973): 10.66 / 8.67 / 9.19 / 11.73; CFL 62 (0.989): - / 15.43 / 15.08 / 18.84; CFL 225 (0.9969): - / - / 25.9 / 35.1 / 43.0)
        e->local_reps = !walk ? pp : (rho < 0.9 ? 2 : (rho < 0.98 ? 3 : 4));
      }
      // (round 3: partitioned engines take the one-closing shape too -- k_rhs keeps the read-only halo rows of x_t beside the
      // computed rows, so a first pass may start from the copy there as well: one plain sweep and one exchange fewer per step)
      const bool one_closing = chained || (!e->two_closing && want % 4 != 2);
      int doubles;
      if (one_closing) {
        doubles = std::max(1, std::min(want / 2, 2047));
        if (!chained && !(doubles & 1) && st.sweeps > 0) ++doubles;     // (a later batch cannot start from the copy: odd, ends in the state vector)
        batch = 2 * doubles + 1;
      } else {
        // batch = 2*doubles + 2 with an even number of J^2 passes, then two plain sweeps
        batch += (6 - batch % 4) % 4;                                   // round up to 2 (mod 4)
        doubles = (batch - 2) / 2;
      }
      const bool from_keep = !chained && one_closing && !(doubles & 1);  // doubles + 1 launches, odd: start from the copy
      auto srcb = [&](int i) -> double* {
        if (chained) return (i == 0 && first_batch) ? e->d_c : e->d_p;
        if (!from_keep) return (i & 1) ? e->d_p : e->d_c;
        return i == 0 ? e->d_keep : ((i & 1) ? e->d_c : e->d_p); };
      auto dstb = [&](int i) -> double* {
        if (chained) return i < doubles ? e->d_p : e->d_c;
        if (!from_keep) return (i & 1) ? e->d_c : e->d_p;
        return (i & 1) ? e->d_p : e->d_c; };
      const int passes = doubles;
      // passes [i, i + cnt) of the batch, none of which needs an exchange
      auto launch_passes = [&](int i, int cnt) -> int {
        for (int q = 0; q < cnt; ++q) {
          if (tiled) TRY(launch_sq_tiled(e, srcb(i + q), dstb(i + q), nullptr, 0, true, walk));
          else TRY(launch_apply<5>(e, srcb(i + q), dstb(i + q), nullptr, e->c2(), nullptr, nullptr, e->n_sq));
        }
        return CWR_OK;
      };
      launches = doubles + (one_closing ? 1 : 2);
      todo = 0;
      // steady state (the same batch shape as the previous check): the WHOLE batch -- passes, closing sweeps, reduction --
      // is one hipGraph, captured the second time a shape is seen (the kernel arguments of a batch never change)
      if (!e->comm && !e->profiling && e->use_graphs) {
        const int shape = 2 * doubles + (one_closing ? 1 : 0) + (chained ? (first_batch ? (1 << 20) : (1 << 21)) : 0) + (walk && !chained ? (1 << 22) : 0) + (e->local_reps << 24);
        auto it = e->batch_exec.find(shape);
        if (it == e->batch_exec.end() && e->batch_last == shape && e->batch_exec.size() < 12) {
          hipGraphExec_t ex = nullptr;
          if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = launch_passes(0, doubles);
            if (rc == CWR_OK) rc = launch_apply<4>(e, srcb(doubles), dstb(doubles), nullptr, e->d_b, nullptr, nullptr);
            if (rc == CWR_OK && !one_closing) rc = launch_apply<4>(e, srcb(doubles + 1), dstb(doubles + 1), nullptr, e->d_b, nullptr, nullptr);
            if (rc == CWR_OK) rc = reduce_check(e, noted);
            hipGraph_t g = nullptr;
            const hipError_t ec = hipStreamEndCapture(e->stream, &g);
            if (!(rc == CWR_OK && ec == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess)) { ex = nullptr; (void)hipGetLastError(); }
            if (g) hipGraphDestroy(g);
          }
          it = e->batch_exec.emplace(shape, ex).first;        // (nullptr: capture failed, do not try this shape again)
        }
        e->batch_last = shape;
        if (it != e->batch_exec.end() && it->second) {
          HIP_TRY(e, hipGraphLaunch(it->second, e->stream));
          batch_graph = true;
        }
      }
      if (!batch_graph) {
      if (!e->comm && !e->profiling && e->use_graphs && !from_keep && !walk) {
        hipGraphExec_t& exec = tiled ? e->tcl_exec : e->sq_exec;
        hipGraph_t& graph = tiled ? e->tcl_graph : e->sq_graph;
        bool& tried = tiled ? e->tcl_graph_tried : e->sq_graph_tried;
        if (!tried) {
          tried = true;
          if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int rc = CWR_OK;
            for (int i = 0; i < cwr_engine::GRAPH_SWEEPS && rc == CWR_OK; ++i) {
              double* src = (i & 1) ? e->d_p : e->d_c;
              double* dst = (i & 1) ? e->d_c : e->d_p;
              rc = tiled ? launch_sq_tiled(e, src, dst) : launch_apply<5>(e, src, dst, nullptr, e->c2(), nullptr, nullptr, e->n_sq);
            }
            hipGraph_t g = nullptr;
            const hipError_t ec = hipStreamEndCapture(e->stream, &g);
            if (rc == CWR_OK && ec == hipSuccess && g && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess) graph = g;
            else { if (g) hipGraphDestroy(g); exec = nullptr; (void)hipGetLastError(); }
          }
        }
        while (exec && doubles >= cwr_engine::GRAPH_SWEEPS) { HIP_TRY(e, hipGraphLaunch(exec, e->stream)); doubles -= cwr_engine::GRAPH_SWEEPS; }
      }
      // a J^2 pass uses up two halo layers of validity, a plain sweep one
      const bool can_overlap = e->comm && tiled && e->overlap && e->comm_stream && e->n_tile_inner > 0 && !e->peers.empty() &&
                               (!walk || (e->sched_in.depth > 0 && e->d_scols_io && !e->sched_user));
      for (int i = 0; i < doubles;) {
        double* src = srcb(i);
        double* dst = dstb(i);
        if (e->comm && from_keep && i == 0 && since_exchange + 2 <= e->exch_every) {
          // the pass that starts from the kept copy of x_t (it shifts the ping-pong parity): on its own, outside the stretch graphs
          TRY(launch_passes(i, 1));
          since_exchange += 2; ++i;
          continue;
        }
        if (since_exchange + 2 > e->exch_every) {
          if (can_overlap) {
            // pack the cut rows, start the interior tiles (they read core rows only), exchange beside them on the
            // communication stream, then the tiles that read or are refreshed rows, and the un-tiled tail
            if (e->test_poison_halo && e->n_real > e->n_core) {
              // test hook: every row an exchange refreshes is NaN in BOTH vectors before the pack (and so before ev_packed, which
              // the unpack on the communication stream waits for).  The result is unchanged only if the interior tiles read
              // no such row and the cut tiles really wait for the unpacked values (ev_halo)
              const size_t off = (size_t)e->n_core * K, cnt = (size_t)(e->n_real - e->n_core) * K * sizeof(double);
              HIP_TRY(e, hipMemsetAsync(src + off, 0xFF, cnt, e->stream));
              HIP_TRY(e, hipMemsetAsync(dst + off, 0xFF, cnt, e->stream));
            }
            TRY(exchange_begin(e, src));
            if (walk) {
              // (in place, or from one vector into the other: deterministic steps) along the interior lists (they read and write core rows only; the rows just packed may be among them:
              // the pack precedes this launch on the stream), then along the lists of the cut tiles behind the unpack
              TRY(launch_sq_tiled(e, src, dst, nullptr, 0, false, true, &e->sched_in));
              TRY(exchange_finish(e, src, dst != src ? dst : nullptr));
              TRY(launch_sq_tiled(e, src, dst, nullptr, 0, true, true, &e->sched_out));
              since_exchange = 2; ++i;
              continue;
            }
            TRY(launch_sq_tiled(e, src, dst, e->d_tile_inner, e->n_tile_inner, false));
            TRY(exchange_finish(e, src, dst));
            TRY(launch_sq_tiled(e, src, dst, e->d_tile_outer, e->n_tile_outer, true));
            if (e->n_tile_outer == 0 && e->n_sq > e->n_tcl) TRY(launch_apply<5>(e, src, dst, nullptr, e->c2(), nullptr, nullptr, e->n_sq, e->n_tcl));
            since_exchange = 2; ++i;
            continue;
          }
          TRY(exchange_halo(e, src, dst)); since_exchange = 0;
        }
        // exchange-free stretch: as many passes as the halo depth still covers, replayed as one hipGraph per (parity, length)
        int run = std::min(doubles - i, std::max(1, (e->exch_every - since_exchange) / 2));
        if (!e->comm) run = doubles - i;
        if (e->comm && tiled && run >= 3 && e->use_graphs && !e->profiling) {
          const int key = (src == e->d_c ? 0 : (src == e->d_p ? 1 : 2)) * 4096 + run + (walk ? (chained ? (1 << 16) : (1 << 17)) + (e->local_reps << 20) : 0);   // (which vector the stretch starts from)
          auto it = e->stretch_exec.find(key);
          if (it == e->stretch_exec.end() && e->stretch_exec.size() < 32) {
            hipGraphExec_t ex = nullptr;
            if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
              const int rc = launch_passes(i, run);
              hipGraph_t g = nullptr;
              const hipError_t ec = hipStreamEndCapture(e->stream, &g);
              if (!(rc == CWR_OK && ec == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess)) { ex = nullptr; (void)hipGetLastError(); }
              if (g) hipGraphDestroy(g);
            }
            it = e->stretch_exec.emplace(key, ex).first;
          }
          if (it != e->stretch_exec.end() && it->second) {
            HIP_TRY(e, hipGraphLaunch(it->second, e->stream));
            since_exchange += 2 * run; i += run;
            continue;
          }
        }
        TRY(launch_passes(i, run));
        since_exchange += 2 * run; i += run;
      }
      // block-asynchronous passes leave the replayed halo layers only approximately equal to their owners' rows: refresh
      // them so that the two plain sweeps below are exact on the core and the check is the true residual
      if (one_closing) {
        bool split = false;
        if (e->comm) {
          if (e->any_tiled && e->local_reps > 1 && passes > 0) since_exchange = e->exch_every;
          if (since_exchange + 1 > e->exch_every) {
            // (round 4) the exchange in front of the closing sweep runs BESIDE the sweep's core tiles: pack, core tiles on the engine's
            // stream, send / receive / unpack on the communication stream, then the cut tiles and the replayed layers behind ev_halo.
            // The two launches leave their partials side by side; one reduction folds both.
            split = e->overlap && e->comm_stream && e->n_apply_inner > 0 && !e->peers.empty() && !getenv("CWR_NO_CLOSING_OVERLAP");
            if (split) {
              double* src = srcb(passes);
              if (e->test_poison_halo && e->n_real > e->n_core)
                HIP_TRY(e, hipMemsetAsync(src + (size_t)e->n_core * K, 0xFF, (size_t)(e->n_real - e->n_core) * K * sizeof(double), e->stream));
              TRY(exchange_begin(e, src));
              TRY(launch_apply<4>(e, src, dstb(passes), nullptr, e->d_b, nullptr, nullptr, -1, 0, e->d_apply_inner, e->n_apply_inner, 0));
              const int g_in = e->last_apply_grid;
              TRY(exchange_finish(e, src, nullptr));
              TRY(launch_apply<4>(e, src, dstb(passes), nullptr, e->d_b, nullptr, nullptr, -1, 0, e->d_apply_outer, e->n_apply_outer, g_in));
              e->last_apply_grid += g_in;
            } else TRY(exchange_halo(e, srcb(passes)));
            since_exchange = 0;
          }
          ++since_exchange;
        }
        if (!split) TRY(launch_apply<4>(e, srcb(passes), dstb(passes), nullptr, e->d_b, nullptr, nullptr));
      } else {
      if (e->comm && e->any_tiled && e->local_reps > 1 && passes > 0) since_exchange = e->exch_every;
      if (since_exchange + 1 > e->exch_every) { TRY(exchange_halo(e, e->d_c)); since_exchange = 0; }
      TRY(launch_apply<4>(e, e->d_c, e->d_p, nullptr, e->d_b, nullptr, nullptr));
      ++since_exchange;
      if (since_exchange + 1 > e->exch_every) { TRY(exchange_halo(e, e->d_p)); since_exchange = 0; }
      TRY(launch_apply<4>(e, e->d_p, e->d_c, nullptr, e->d_b, nullptr, nullptr));
      ++since_exchange;
      }
      }
    } else if (!e->comm && !e->profiling && e->use_graphs) {
      if (!e->graph_tried) {                               // capture GRAPH_SWEEPS sweeps once
        e->graph_tried = true;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          int rc = CWR_OK;
          for (int i = 0; i < cwr_engine::GRAPH_SWEEPS && rc == CWR_OK; ++i)
            rc = launch_apply<4>(e, (i & 1) ? e->d_p : e->d_c, (i & 1) ? e->d_c : e->d_p, nullptr, e->d_b, nullptr, nullptr);
          hipGraph_t g = nullptr;
          const hipError_t ec = hipStreamEndCapture(e->stream, &g);
          if (rc == CWR_OK && ec == hipSuccess && g && hipGraphInstantiate(&e->sweep_exec, g, nullptr, nullptr, 0) == hipSuccess) {
            e->sweep_graph = g;
          } else {
            if (g) hipGraphDestroy(g);
            e->sweep_exec = nullptr;
            (void)hipGetLastError();
          }
        }
      }
      while (e->sweep_exec && todo >= cwr_engine::GRAPH_SWEEPS) {
        HIP_TRY(e, hipGraphLaunch(e->sweep_exec, e->stream));
        todo -= cwr_engine::GRAPH_SWEEPS;
      }
    }
    for (int i = 0; i < todo; ++i) {                        // remainder (even), partitioned or profiled runs
      double* src = (i & 1) ? e->d_p : e->d_c;
      double* dst = (i & 1) ? e->d_c : e->d_p;
      if (since_exchange >= e->exch_every) { TRY(exchange_halo(e, src)); since_exchange = 0; }
      TRY(launch_apply<4>(e, src, dst, nullptr, e->d_b, nullptr, nullptr));
      ++since_exchange;
    }
    st.sweeps += batch; st.launches += launches;
    if (!batch_graph) TRY(reduce_check(e, noted));
    if (noted) ++e->note_expected;                                        // (one notifying reduction per batch, replayed graph or not)
    bool speculated = false;
    // (partitioned engines too: every rank takes the same decisions from the all-reduced check, so a speculative tail --
    // whose exchange is a collective -- is entered and, if the check fails, repeated by all ranks alike)
    if (e->spec_t >= 0) { TRY(step_tail(e, e->spec_t, e->spec_flags)); speculated = true; }
    // (windowed flow field: the levels asked for since the last step go to the flow stream now -- the batch is on its way and the
    // host has nothing to do but wait for the check)
    if (e->defer_loads) { e->defer_loads = false; TRY(flush_window_loads(e)); }
    TRY(gather_check(e, h.data(), noted));
    // (a rank met the zero-coefficient precondition: its right-hand side is NaN-poisoned, every rank leaves here with the same code)
    if (e->comm && e->ghost_bad_any) { st.status = CWR_ERR_GHOST_COEFF; return CWR_ERR_GHOST_COEFF; }
    bool ok = true;
    double worst = 0.0;                                                   // max over columns of rr / (tol^2 bb)
    st.max_rel = 0.0;
    for (int k = 0; k < K; ++k) {
      const double rr = h[k], bb = h[K + k];
      if (!std::isfinite(rr) || !std::isfinite(bb)) { st.status = CWR_ERR_NONFINITE; return CWR_ERR_NONFINITE; }
      st.max_rel = std::max(st.max_rel, bb > 0.0 ? std::sqrt(rr / bb) : (rr > 0.0 ? (double)INFINITY : 0.0));
      if (rr > tol2 * bb) ok = false;
      worst = std::max(worst, bb > 0.0 ? rr / (tol2 * bb) : (rr > 0.0 ? (double)INFINITY : 0.0));
    }
    if (e->fixed_sweeps > 0) { e->last_sweeps = 0; e->tail_done = speculated; return CWR_OK; }   // (measurement hook: see fixed_sweeps)
    // element-wise rule: every |x'_i - x_i| within ew_rel |x'_i| + ew_abs max|x'| (plume fronts far below the peak are
    // invisible to the 2-norm).  Folded into `worst` (a squared ratio) so that the sweep prediction serves both rules.
    double ew_ratio = 0.0;
    const bool norm_ok = ok;                                              // (the 2-norm criterion alone)
    const double worst_norm = worst;                                      // (... and its measure: what contracts geometrically)
    if (!elementwise_ok(e, h.data(), &ew_ratio)) ok = false;
    worst = std::max(worst, ew_ratio * ew_ratio);
    if (ok) {
      // remember the sweeps this step really needed (the margin below the tolerance, converted with the measured
      // contraction), so that the next step's first batch neither overshoots nor needs a second check
      int extra = 0;
      if (worst > 0.0 && worst < 1.0 && e->last_rate > 0.0 && e->last_rate < 1.0)
        extra = (int)std::floor(0.5 * std::log(1.0 / worst) / -std::log(e->last_rate));
      e->last_sweeps = std::max(2, st.sweeps - extra);
      // (six digits below the tolerance the residual has most likely reached its rounding floor, where the margin says nothing about
      // the sweeps that were too many: come down by a quarter at least)
      if (worst < 1.0e-12) e->last_sweeps = std::max(2, std::min(e->last_sweeps, st.sweeps * 3 / 4));
      e->tail_done = speculated;
      return CWR_OK;
    }
    if (st.sweeps >= sweep_limit) {                                       // max_iter bounds the sweeps and the BiCGSTAB iterations each
      if (forced) { st.status = CWR_ERR_NOT_CONVERGED; return CWR_ERR_NOT_CONVERGED; }
      need_bicg = true; e->last_sweeps = 0; return CWR_OK;
    }
    // contraction per sweep from the last two checks (worst is a squared, normalised residual).  The RATE -- and the verdict
    // "stalled" -- come from the 2-norm measure alone, and only from checks at which that criterion is still open: the element-wise
    // measure max(|dx| - ew_rel |x'|) is no geometric sequence (it may rise between two checks a few sweeps apart), and since the
    // batches behind a norm-satisfied check are short, reading it as a rate sent converging steps to BiCGSTAB (117 k cells at CFL 72
    // with dry cells: 1 892 iterations, 54 ms).
    int predicted = 16;
    if (prev_worst <= 0.0 && e->last_rate > 0.0 && e->last_rate < 1.0 && std::isfinite(worst))
      predicted = (int)std::ceil(0.5 * std::log(worst) / -std::log(e->last_rate)) + 1;
    if (!norm_ok && prev_worst > 0.0 && std::isfinite(worst_norm)) {
      double rate = std::pow(worst_norm / prev_worst, 0.5 / (st.sweeps - prev_sweeps));
      // ||J||_inf of the step bounds the asymptotic contraction of a sweep from above (and the passes contract faster than a sweep):
      // a measured rate above it is two checks at the rounding floor, not slow convergence.  Unclamped, such a rate (0.9999...)
      // sized the next batch at the sweep limit and the over-converged steps after it came down by ~240 sweeps a step only
      // (profiles/r05_mid_mesh.txt: 18 k cells x 4 / 8 / 16 at CFL 18 through the passes: 163, 832, 593, 353, 684 ... 2002 sweeps)
      const double rho_t = ((size_t)e->cur_t < e->jnorm.size()) ? e->jnorm[(size_t)e->cur_t] : 0.0;
      if (rho_t > 0.0 && rho_t < 1.0 && rate > rho_t && rate < 1.0) rate = rho_t;
      if (rate > 0.0 && rate < 1.0) e->last_rate = rate;
      if (!(rate < 1.0)) {                                               // stalled or diverging
        if (forced) { predicted = 64; } else { need_bicg = true; e->last_sweeps = 0; return CWR_OK; }
      } else {
        predicted = (int)std::ceil(0.5 * std::log(worst) / -std::log(rate)) + 1;
        if (!forced && st.sweeps + predicted > e->jacobi_limit) { need_bicg = true; e->last_sweeps = 0; return CWR_OK; }
      }
    }
    // Bounds of the next batch.  (1) Only the element-wise rule is open: its measure max(|dx| - ew_rel |x'|) does not fall
    // geometrically -- it drops through zero within a few sweeps of the norm criterion -- so log(worst) over-predicts by hundreds of
    // sweeps (18 k cells x 4 at CFL 18: 193 -> 403 sweeps every third step): a short batch, and another check if need be.
    // (2) In general no batch more than doubles what the step has taken: a wrong rate costs a check, not a step.
    if (norm_ok) predicted = std::min(predicted, std::max(8, st.sweeps / e->ew_batch_div));
    predicted = std::min(predicted, std::max(32, st.sweeps));
    if (!norm_ok) { prev_worst = worst_norm; prev_sweeps = st.sweeps; }
    else prev_worst = -1.0;                                               // (no rate across a norm-satisfied check)
    want = predicted;
  }
}

// The tables of k_small_jacobi (see there and host::build_small_plan): built once per engine, uploaded, with the exchange buffers of
// a plan of several parts.  use_small goes false when no plan exists (a row with more than 8 real neighbours, a mesh too large).
int ensure_small_plan(cwr_engine* e) {
  if (e->small_planned) return CWR_OK;
  e->small_planned = true;
  host::SmallPlan pl;
  // the deepest halo that fits: 12 layers on a band, fewer on a wide patch (whose breadth-first levels are long: the halo rows of
  // 12 of them no longer fit beside a part's own) -- an exchange every 8, 6, 4 ... sweeps then (profiles/r05_mid_mesh.txt)
  bool planned = false;
  for (int depth : {e->small_depth, 8, 6, 4, 3, 2}) {
    if (depth > e->small_depth) continue;
    // (K x parts <= 128 workgroups: a wide state vector gets fewer, larger parts -- 4 rows per thread where 3 would need too many)
    const int max_parts = std::max(1, std::min(e->small_max_parts, e->small_wg_cap / std::max(1, e->K)));
    if (host::build_small_plan(e->n_owned, e->h_ptr, e->h_nb, SMALL_THREADS, 4, e->small_parts, depth, max_parts, pl)) { planned = true; break; }
    if (e->n_owned <= 4 * SMALL_THREADS) break;  // (one workgroup: the depth plays no part)
  }
  if (!planned) {
    e->use_small = false;
    return CWR_OK;
  }
  auto up32 = [&](int32_t** d, const void* h, size_t count) -> int {
    if (count == 0) count = 1;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(d), count * sizeof(int32_t)));
    if (h) HIP_TRY(e, hipMemcpy(*d, h, count * sizeof(int32_t), hipMemcpyHostToDevice));
    return CWR_OK;
  };
  TRY(up32(&e->d_small_rows, pl.rows.data(), pl.rows.size()));
  TRY(up32(&e->d_small_recs, pl.recs.data(), pl.recs.size()));
  TRY(up32(reinterpret_cast<int32_t**>(&e->d_small_offs), pl.offs.data(), pl.offs.size()));
  e->small_rpt = pl.rpt; e->small_P = pl.P; e->small_D = pl.depth; e->small_S = pl.S; e->small_R = pl.R;
  if (pl.P > 1) {
    // (recv_src travels as part * 2 S + slot: the two publication buffers of a part lie side by side)
    std::vector<int32_t> src(pl.recv_src);
    for (auto& v : src) v = (v / pl.S) * 2 * pl.S + v % pl.S;
    TRY(up32(&e->d_small_send_pos, pl.send_pos.data(), pl.send_pos.size()));
    TRY(up32(&e->d_small_send_cnt, pl.send_cnt.data(), pl.send_cnt.size()));
    TRY(up32(&e->d_small_recv_src, src.data(), src.size()));
    TRY(up32(&e->d_small_recv_pos, pl.recv_pos.data(), pl.recv_pos.size()));
    TRY(up32(&e->d_small_recv_cnt, pl.recv_cnt.data(), pl.recv_cnt.size()));
    const size_t K = (size_t)e->K;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_small_pub), K * pl.P * 2 * pl.S * sizeof(double)));
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_small_red), K * pl.P * 2 * 4 * sizeof(double)));
    // (ON THE ENGINE'S STREAM: a hipMemset goes to the null stream, which this non-blocking stream does not wait for -- under load it
    // ran after the first launch had begun and zeroed values between a part's store and its neighbour's load: the one failure of
    // test_the_parts_exchange_correctly_while_another_engine_loads_the_chip, 2 runs in 14, found by that test)
    HIP_TRY(e, hipMemsetAsync(e->d_small_pub, 0, K * pl.P * 2 * pl.S * sizeof(double), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_small_red, 0, K * pl.P * 2 * 4 * sizeof(double), e->stream));
  }
  return CWR_OK;
}

// Meshes that fit one CU's LDS: the whole Jacobi solve of every constituent in ONE launch (k_small_jacobi).
// handled = false: not applicable; need_bicg = true: the sweeps did not converge within the limit.
int solve_small(cwr_engine* e, double tol2, int max_iter, bool forced, SolveStats& st, bool& handled, bool& need_bicg) {
  handled = false; need_bicg = false;
  // only where the rows' weights fit registers (<= 4 rows per thread, <= 8 real neighbours per row): then a sweep touches LDS
  // only.  Streaming the records from L2 instead was measured SLOWER than the multi-launch path (4.3 vs 1.5 ms at 8-10 k
  // cells), so larger meshes do not come here.
  // (round 5: meshes of up to ~20 000 cells -- BASELINE configs 1 / 2 at "~10 k cells" -- come here too: several workgroups per
  // constituent, each with halo layers around its rows, exchanging every few sweeps: k_small_jacobi<RPT, true>)
  if (e->comm || !e->use_small || e->n_halo != 0 || e->n_owned > std::max(SMALL_THREADS * 4, e->small_max_cells)) return CWR_OK;
  const int K = e->K, n = e->n_owned;
  TRY(ensure_small_plan(e));
  if (!e->use_small) return CWR_OK;              // no plan (a row with more than 8 real neighbours, too many parts): the multi-launch path
  const int rpt = e->small_rpt, P = e->small_P;
  if (!e->d_info) {                               // [K][5] results + the parts' sticky abort word (see the end of k_small_jacobi)
    TRY(dev_alloc(e, &e->d_info, (size_t)5 * K + 1));
    HIP_TRY(e, hipMemsetAsync(e->d_info, 0, ((size_t)5 * K + 1) * sizeof(double), e->stream));
  }
  size_t lds = (2 * (size_t)rpt * SMALL_THREADS + 64) * sizeof(double);  // two columns + the scratch of block_reduce3 (3 x 16 wave results)
  if (P > 1) lds = std::max(lds + ((size_t)e->small_S + 2 * (size_t)e->small_R) * sizeof(int32_t),   // + the part's exchange lists
                            (size_t)84 * 1024);                          // more than half a CU's LDS: one workgroup per CU (the hand-off's measured form)
  SmallCoop co{};
  if (P > 1) {
    if ((long long)K * P > e->small_wg_cap) return CWR_OK;   // a part that is not resident would be waited for: one workgroup per CU, half the chip at most
    // (the arrival counters and the abort word lie in the scalar block cwr_step zeroed at its start: no memset of their own)
    co = SmallCoop{P, e->small_D, e->small_S, e->small_R, e->d_small_send_pos, e->d_small_send_cnt, e->d_small_recv_src, e->d_small_recv_pos,
                   e->d_small_recv_cnt, e->d_small_pub, e->d_small_red, e->small_arrive(),
                   (long long)e->small_spin_ms * 100000ll, e->small_fences};
  }
  const int limit = forced ? max_iter : std::min(max_iter, e->jacobi_limit);
  // (round 5) the five numbers per constituent reach the host through the notification buffer of the sweeps' check: no download,
  // and -- what counts at 0.2-0.35 ms per step -- no copy's round trip behind the one launch
  const bool noted = check_by_note(e);
  ReduceNote note{nullptr, nullptr, nullptr, nullptr};
  if (noted) note = ReduceNote{e->d_note_view, reinterpret_cast<unsigned long long*>(e->d_note_view + 5 * (size_t)K),
                               reinterpret_cast<unsigned int*>(e->d_note_state + 1), e->d_note_state};
  // one workgroup per constituent: no convergence check (two barriers and a reduction each, every fourth sweep) before three
  // quarters of the sweeps the last step took -- the step before is the best guess there is, and a step that needs fewer only
  // sweeps on to that point (CWR_SMALL_FIRST_CHECK=0: check from the start)
  // (several parts: the exchanges before that point carry the halo rows only -- no block reduction, no partial norms)
  // (only while the tolerance stays what it was: the last step's count says nothing about a looser one)
  const int first_check = (e->small_first_check && !forced && tol2 == e->small_last_tol2) ? (e->small_last_sweeps * 3 / 4) / 4 * 4 : 0;
  e->small_last_tol2 = tol2;
#define CWR_SMALL(RPTv, COOPv) do {                                                                                   \
    static bool attr_done = false;                                                                                    \
    if (!attr_done) { HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small_jacobi<RPTv, COOPv>),     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done = true; }        \
    if (COOPv && !e->small_resident_checked) {                                                                        \
      /* (round 6) the parts wait for each other inside ONE ordinary launch: all K x P workgroups must be resident together.  Asked  \
         of the runtime for THIS kernel, block size and LDS request on THIS device instead of assumed from gfx950's constants */     \
      int pc = 0;                                                                                                     \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, reinterpret_cast<const void*>(&k_small_jacobi<RPTv, COOPv>), SMALL_THREADS, lds) != hipSuccess) { pc = 0; (void)hipGetLastError(); } \
      e->small_resident_checked = true;                                                                               \
      if ((long long)pc * e->n_cu < (long long)K * P) {                                                               \
        if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] one-launch solver: %d x %d workgroups, %d resident at once on %d CUs: not taken\n", K, P, pc * e->n_cu, e->n_cu); \
        e->use_small = false; return CWR_OK;                                                                          \
      }                                                                                                               \
    }                                                                                                                 \
    k_small_jacobi<RPTv, COOPv><<<K * P, SMALL_THREADS, lds, e->stream>>>(n, K, e->d_small_rows, e->d_small_recs, e->d_small_offs, e->d_rec, \
        e->d_diag, e->d_b, e->d_c, tol2, e->ew_enabled ? e->ew_rel : 1.0, e->ew_enabled ? e->ew_abs : 1.0, limit, 4, e->d_info, note, co, first_check); } while (0)
  if (P == 1) {
    if (rpt == 1) CWR_SMALL(1, false);
    else if (rpt == 2) CWR_SMALL(2, false);
    else if (rpt == 3) CWR_SMALL(3, false);
    else CWR_SMALL(4, false);
  } else {
    if (rpt == 1) CWR_SMALL(1, true);
    else if (rpt == 2) CWR_SMALL(2, true);
    else if (rpt == 3) CWR_SMALL(3, true);
    else CWR_SMALL(4, true);
  }
#undef CWR_SMALL
  HIP_TRY(e, hipGetLastError());
  std::vector<double> h((size_t)5 * K + 1, 0.0);
  if (noted) { ++e->note_expected; TRY(wait_check_note(e, h.data(), (size_t)5 * K)); if (P > 1) h[(size_t)5 * K] = e->h_note[(size_t)5 * K + 1]; }
  else TRY(download(e, h.data(), e->d_info, (size_t)5 * K + (P > 1 ? 1 : 0)));
  bool gave_up = h[(size_t)5 * K] != 0.0;          // SOME part gave up (any part says so: the sticky word behind the numbers)
  for (int k = 0; k < K; ++k) if (h[5 * (size_t)k] < 0.0) gave_up = true;
  if (gave_up) {
    // a part was waited for longer than the bound (never seen; a CU shortage would do it).  Parts that had passed their last exchange
    // before the abort was raised may have written their rows: the state goes back to the kept copy of x_t (k_begin_step wrote it),
    // the multi-launch path takes the step from the same start, and this engine stays with it -- said in every step's flags from here on
    HIP_TRY(e, hipMemcpyAsync(e->d_c, e->d_keep, (size_t)e->n_owned * K * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] the one-launch solver's parts did not all arrive within %d ms; this engine uses the multi-launch passes from here on\n", e->small_spin_ms);
    e->use_small = false;
    e->small_fell_back = true;
    e->info_flags |= CWR_INFO_SMALL_FALLBACK;
    return CWR_OK;
  }
  handled = true;
  st.launches += 1;
  st.sweep_kernel = 7;
  bool ok = true;
  int sweeps = 0;
  st.max_rel = 0.0;
  for (int k = 0; k < K; ++k) {
    const double rr = h[5 * k + 1], bb = h[5 * k + 2];
    sweeps = std::max(sweeps, (int)h[5 * k]);
    if (!std::isfinite(rr) || !std::isfinite(bb)) { st.status = CWR_ERR_NONFINITE; st.sweeps += sweeps; return CWR_ERR_NONFINITE; }
    st.max_rel = std::max(st.max_rel, bb > 0.0 ? std::sqrt(rr / bb) : (rr > 0.0 ? (double)INFINITY : 0.0));
    if (rr > tol2 * bb) ok = false;
    if (e->ew_enabled && h[5 * k + 3] > e->ew_abs * h[5 * k + 4]) ok = false;      // element-wise rule (see k_apply MODE 4)
  }
  st.sweeps += sweeps;
  e->small_last_sweeps = ok ? sweeps : 0;
  if (!ok) {
    // sweeps exhausted: the reference's direct solve has no such outcome, so unless the caller forced the sweeps
    // BiCGSTAB continues from the current iterate (as solve_jacobi does)
    if (forced) { st.status = CWR_ERR_NOT_CONVERGED; return CWR_ERR_NOT_CONVERGED; }
    need_bicg = true;
  }
  return CWR_OK;
}

int solve_bicgstab(cwr_engine* e, double tol2, int max_iter, SolveStats& st) {
  const int K = e->K;
  std::vector<double> h_scal(e->scal_count());
  int32_t h_cnt[8];
  int total_it = 0, restarts = 0, launches = 0, status = CWR_OK;
  double max_rel = st.max_rel;
  bool converged = false;
  e->dominant_mode = 1;                       // profile the first-product launches of BiCGSTAB steps
  HIP_TRY(e, hipMemsetAsync(e->d_scal, 0, e->scal_count() * sizeof(double), e->stream));
  int round = 0;
  // ew_try: after the norm criterion is met the element-wise rule is verified with two plain Jacobi sweeps (whose
  // ||x'-x|| measures are the ones solve_jacobi uses); if it fails, BiCGSTAB restarts from there with tol / 10
  for (int ew_try = 0;; ++ew_try) {
  converged = false;
  for (int round0 = round; !converged; ++round) {
    // (re)start: true residual of the current x; r0 = p = r
    if (round > 0) HIP_TRY(e, hipMemsetAsync(e->d_scal, 0, (size_t)3 * ACC_N * K * sizeof(double) + (size_t)3 * K * sizeof(double), e->stream));
    TRY(exchange_halo(e, e->d_c));
    TRY(launch_apply<3>(e, e->d_c, e->d_r, nullptr, e->d_b, e->d_r0, e->d_p, e->n_core));
    TRY(reduce_partials(e, e->last_apply_grid, 2, e->acc(2) + ACC_RR * K, round == 0 ? e->bb() : nullptr));
    ++launches;
    TRY(allreduce(e, e->acc(2) + ACC_RR * K, K));
    if (round == 0) TRY(allreduce(e, e->bb(), K));
    if (round > 0) {
      TRY(download(e, h_scal.data(), e->d_scal, e->scal_count()));
      const double* rr = h_scal.data() + (size_t)2 * ACC_N * K + ACC_RR * K;
      const double* bbh = h_scal.data() + (size_t)3 * ACC_N * K + 3 * K;
      bool ok = true, loose = true;
      max_rel = 0.0;
      for (int k = 0; k < K; ++k) {
        if (!std::isfinite(rr[k])) { status = CWR_ERR_NONFINITE; break; }
        const double rel = (bbh[k] > 0.0) ? std::sqrt(rr[k] / bbh[k]) : (rr[k] > 0.0 ? INFINITY : 0.0);
        max_rel = std::max(max_rel, rel);
        if (rr[k] > tol2 * bbh[k]) ok = false;
        if (rr[k] > 1.0e4 * tol2 * bbh[k]) loose = false;
      }
      if (status != CWR_OK) break;
      if (ok) { converged = true; break; }
      if (total_it >= max_iter || round - round0 > 6) {
        // stagnation within 100 x tol after 6 verified restarts: the attainable accuracy of this system in float64.
        // Accepted, but never silently: CWR_INFO_LOOSE_RESIDUAL is set in cwr_step_info.flags (the facade warns)
        if (loose && round - round0 > 6) { converged = true; e->info_flags |= CWR_INFO_LOOSE_RESIDUAL; break; }
        status = CWR_ERR_NOT_CONVERGED; break;
      }
      ++restarts;
    }
    // iterate until the recurrence residual says converged, a breakdown is flagged, or max_iter
    int it = 0;
    int batch = (round == 0) ? std::max(2, e->last_iters) : 2;
    bool inner_done = false;
    while (!inner_done) {
      batch = std::min(batch, std::max(1, max_iter - total_it));
      for (int b = 0; b < batch; ++b) { TRY(one_iteration(e, it, tol2)); ++it; ++total_it; launches += 2; }
      TRY(download(e, h_scal.data(), e->d_scal, e->scal_count()));
      TRY(download(e, h_cnt, e->d_counters, (size_t)8));
      const double* rr = h_scal.data() + (size_t)((it - 1) % 3) * ACC_N * K + ACC_RR * K;
      const double* bbh = h_scal.data() + (size_t)3 * ACC_N * K + 3 * K;
      bool ok = true;
      for (int k = 0; k < K; ++k) {
        if (!std::isfinite(rr[k])) { status = CWR_ERR_NONFINITE; }
        if (rr[k] > tol2 * bbh[k]) ok = false;
      }
      if (h_cnt[2]) status = CWR_ERR_GHOST_COEFF;
      if (h_cnt[3] && status == CWR_OK) status = CWR_ERR_NONFINITE;
      if (status != CWR_OK) break;
      if (ok || h_cnt[1] || total_it >= max_iter) inner_done = true;
      if (h_cnt[1]) HIP_TRY(e, hipMemsetAsync(e->d_counters + 1, 0, sizeof(int32_t), e->stream));
      batch = 2;
    }
    if (status != CWR_OK) break;
  }
  if (status != CWR_OK || !e->ew_enabled) break;
  {
    std::vector<double> h(4 * (size_t)K);
    const int keep = e->dominant_mode; e->dominant_mode = -1;
    int rc = exchange_halo(e, e->d_c);
    if (rc == CWR_OK) rc = launch_apply<4>(e, e->d_c, e->d_p, nullptr, e->d_b, nullptr, nullptr, e->n_core);
    if (rc == CWR_OK) rc = exchange_halo(e, e->d_p);
    if (rc == CWR_OK) rc = launch_apply<4>(e, e->d_p, e->d_c, nullptr, e->d_b, nullptr, nullptr, e->n_core);
    e->dominant_mode = keep;
    if (rc != CWR_OK) return rc;
    launches += 2; st.sweeps += 2;
    TRY(reduce_check(e));
    TRY(gather_check(e, h.data()));
    bool finite = true;
    for (int k = 0; k < K; ++k) if (!std::isfinite(h[k])) finite = false;
    if (!finite) { status = CWR_ERR_NONFINITE; break; }
    if (elementwise_ok(e, h.data(), nullptr)) break;
    if (ew_try >= 3 || total_it >= max_iter) { e->info_flags |= CWR_INFO_ELEMENTWISE_MISSED; break; }
    tol2 *= 1.0e-2;
  }
  }
  st.iterations += total_it; st.restarts += restarts; st.launches += launches; st.max_rel = max_rel; st.status = status;
  if (status == CWR_OK) e->last_iters = std::max(1, total_it - 1);
  return status;
}

}  // namespace

// ====================================================================================================
extern "C" {

int32_t cwr_abi_version(void) { return 7; }

int32_t cwr_tile_rows(int32_t n_constituents) {
  if (n_constituents < 1 || n_constituents > 256) return 0;
  // (a work-item build with CWR_TCL_SPLIT=1 splits long rows itself and its tiles hold a variable number of rows, so a fixed
  // window would straddle tiles -- measured 38 -> 52 us per pass with sorted 256-row windows: no arrangement wanted then)
  if (CWR_WORK_ITEMS && n_constituents == 1 && getenv("CWR_TCL_SPLIT") && atoi(getenv("CWR_TCL_SPLIT")) != 0) return 0;
  return tile_rows_for(pad_constituents(n_constituents), nullptr);
}

// From how many rows an engine with K constituents chains its tiles (the rule of ensure_sq_pattern, evaluated for the usual four
// resident blocks per CU): what a host wrapper that chooses the cell numbering BEFORE it creates the engine asks, so that numbering
// (lanes along the flow for chains, the Hilbert curve for ping-pong passes) and engine follow ONE threshold, CWR_CHAIN_MIN_TILES
// included (the wrapper used to carry its own copy of the constant and a hard-coded grid).
int32_t cwr_chain_min_rows(int32_t n_constituents) {
  if (n_constituents < 1 || n_constituents > 256) return 0;
  double min_tiles = 1.75;
  if (const char* v = getenv("CWR_CHAIN_MIN_TILES")) min_tiles = std::max(1.0, atof(v));
  if (const char* v = getenv("CWR_NO_CHAINS")) if (atoi(v) != 0) return INT32_MAX;
  // (no HIP call here: the question is asked before an engine exists, also by processes that must not open the GPU -- a test runner
  // counting its processes on the card, bench.py's launcher.  gfx950 / MI355X: 256 CUs, what cwr_create finds on the device)
  // (a partitioned or smaller device has fewer: the count cwr_create found, once an engine exists in this process, or CWR_N_CU)
  int n_cu = g_n_cu.load() > 0 ? g_n_cu.load() : 256;
  if (const char* v = getenv("CWR_N_CU")) n_cu = std::max(N_XCD, atoi(v));
  int per_cu = 4;
  if (const char* v = getenv("CWR_TCL_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(8, atoi(v)));
  int grid = (n_cu * per_cu / N_XCD) * N_XCD;
  if (const char* v = getenv("CWR_TCL_GRID")) grid = std::max(N_XCD, std::min(grid, atoi(v) / N_XCD * N_XCD));
  const double rows = std::ceil(min_tiles * grid) * (double)tile_rows_for(pad_constituents(n_constituents), nullptr);
  return rows >= 2147483647.0 ? INT32_MAX : (int32_t)rows;
}

const char* cwr_last_error(const cwr_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int32_t cwr_create(int32_t n_owned, int32_t n_halo, int32_t n_cells, int32_t n_edges, int32_t K_user,
                   const int32_t* face1, const int32_t* face2, int32_t device, cwr_engine** out) {
  if (!out) return fail(nullptr, CWR_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (n_owned <= 0 || n_halo < 0 || n_edges < 0 || K_user <= 0 || K_user > 256 || !face1 || !face2 ||
      n_cells < n_owned + n_halo)
    return fail(nullptr, CWR_ERR_BAD_ARG, "cwr_create: bad sizes or NULL topology");
  const int K = pad_constituents(K_user);       // the engine's internal row width (zero columns behind the caller's: see there)
  const int n_real = n_owned + n_halo;
  if ((double)n_cells * K * 8.0 >= 4294967296.0)
    return fail(nullptr, CWR_ERR_BAD_ARG, "cwr_create: n_cells * K * 8 bytes must stay below 4 GiB per engine (32-bit row offsets); partition the mesh");
  std::vector<int32_t> cnt((size_t)n_owned + 1, 0);
  for (int e = 0; e < n_edges; ++e) {
    const int P = face1[e], N = face2[e];
    if (P < 0 || P >= n_real || N < 0 || N >= n_cells)
      return fail(nullptr, CWR_ERR_BAD_ARG, "cwr_create: face " + std::to_string(e) +
                  " has face1 outside the real cells or face2 outside the mesh (face1 must be a real cell, io/hdf.py:268)");
    if (P < n_owned) cnt[P + 1]++;
    if (N < n_owned) cnt[N + 1]++;
  }
  for (int c = 0; c < n_owned; ++c) cnt[c + 1] += cnt[c];
  const int nnz = cnt[n_owned];
  // internal face order: ascending smaller cell id (stable), so per-face data of neighbouring cells is contiguous
  std::vector<int32_t> face_orig((size_t)n_edges), face_pos((size_t)n_edges);
  for (int e = 0; e < n_edges; ++e) face_orig[(size_t)e] = e;
  // (Round 1 kept the reference's order for rows shorter than a 64-byte sector: k_mass_flux then wrote one output row per
  // face in reference order, a scatter -- 23 -> 58 us at K = 1.  Since the flux arrays are written in the INTERNAL order and
  // read out through the face map, the sorted order pays at every K: K = 1 1.025 -> 1.003 ms per step, K = 2 1.174 -> 1.153,
  // K = 4 1.571 -> 1.555; CWR_FACE_ORDER_MIN_K=8 restores the old threshold.)
  int face_order_min_k = 1;
  if (const char* v = getenv("CWR_FACE_ORDER_MIN_K")) face_order_min_k = atoi(v);
  if (!getenv("CWR_NO_FACE_ORDER") && K >= face_order_min_k)
    std::stable_sort(face_orig.begin(), face_orig.end(), [&](int32_t a, int32_t b) {
      const int ka = (face2[a] < n_real) ? std::min(face1[a], face2[a]) : face1[a];
      const int kb = (face2[b] < n_real) ? std::min(face1[b], face2[b]) : face1[b];
      return ka < kb;
    });
  for (int p = 0; p < n_edges; ++p) face_pos[(size_t)face_orig[(size_t)p]] = p;
  std::vector<int32_t> f1p((size_t)std::max(n_edges, 1)), f2p((size_t)std::max(n_edges, 1));
  for (int p = 0; p < n_edges; ++p) { f1p[(size_t)p] = face1[face_orig[(size_t)p]]; f2p[(size_t)p] = face2[face_orig[(size_t)p]]; }
  std::vector<int32_t> ent_edge((size_t)std::max(nnz, 1)), ent_nb((size_t)std::max(nnz, 1)), fill(cnt.begin(), cnt.end() - 1);
  for (int e = 0; e < n_edges; ++e) {           // ascending REFERENCE face id inside every cell (last-write-wins order)
    const int P = face1[e], N = face2[e];
    const int pe = face_pos[(size_t)e];
    if (P < n_owned) { const int j = fill[P]++; ent_edge[j] = (pe << 1); ent_nb[j] = (N < n_real) ? N : -1 - (N - n_real); }
    if (N < n_owned) { const int j = fill[N]++; ent_edge[j] = (pe << 1) | 1; ent_nb[j] = P; }
  }

  cwr_engine* eng = new cwr_engine();
  eng->dev = device;
  eng->h_ptr = cnt;
  for (int c = 0; c < n_owned; ++c) eng->max_degree = std::max(eng->max_degree, cnt[c + 1] - cnt[c]);
  eng->h_nb.assign(ent_nb.begin(), ent_nb.begin() + nnz);
  eng->h_edge.assign(ent_edge.begin(), ent_edge.begin() + nnz);
  eng->n_core = n_owned;
  eng->n_owned = n_owned; eng->n_halo = n_halo; eng->n_real = n_real; eng->n_cells = n_cells;
  eng->n_ghost = n_cells - n_real; eng->E = n_edges; eng->K = K; eng->Ku = K_user; eng->nnz = nnz;
  eng->VW = (K % 2 == 0) ? 2 : 1;
  eng->G = K / eng->VW;
  eng->R = BLOCK / eng->G;
  int tile_rows = 128, cu_cap = 5;                               // measured: 4-8 blocks/CU within 3 %, 5 best (profiles/)                               // tunables (measured defaults; env overrides for sweeps)
  if (const char* v = getenv("CWR_TILE_ROWS")) tile_rows = std::max(1, atoi(v));
  if (const char* v = getenv("CWR_BLOCKS_PER_CU")) cu_cap = std::max(1, atoi(v));
  if (const char* v = getenv("CWR_JACOBI_LIMIT")) eng->jacobi_limit = std::max(2, atoi(v));
  if (const char* v = getenv("CWR_NO_GRAPHS")) eng->use_graphs = atoi(v) == 0;
  if (const char* v = getenv("CWR_NO_SQ")) eng->use_sq = atoi(v) == 0;
  if (const char* v = getenv("CWR_NO_SMALL")) eng->use_small = atoi(v) == 0;
  if (const char* v = getenv("CWR_TWO_CLOSING")) eng->two_closing = atoi(v) != 0;
  if (const char* v = getenv("CWR_EW_BATCH_DIV")) eng->ew_batch_div = std::max(1, atoi(v));
  if (const char* v = getenv("CWR_SWEEP_MARGIN")) eng->sweep_margin = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_NO_TCL")) eng->use_tcl = atoi(v) == 0;
  if (const char* v = getenv("CWR_NO_CHAINS")) eng->use_chains = atoi(v) == 0;
  if (const char* v = getenv("CWR_DET_WALK")) eng->det_walk = atoi(v) != 0;
  if (const char* v = getenv("CWR_DET_DEFAULT_K")) eng->det_default_k = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_CHAIN_REFRESH")) eng->sched_refresh = std::max(1, atoi(v));
  if (const char* v = getenv("CWR_CHAIN_REUSE")) eng->chain_reuse = atoi(v) != 0;
  if (const char* v = getenv("CWR_CHAIN_MIN_TILES")) eng->chain_min_tiles = std::max(1.0, atof(v));
  if (const char* v = getenv("CWR_BOUND_SWEEPS")) eng->neumann_sweeps = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_BOUND_SWEEPS_MAX")) eng->neumann_sweeps_max = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_BOUND_WARM")) eng->neu_warm = atoi(v) != 0;
  // tile-local J^2 applications per pass: each costs LDS time only (measured 15-25 us at K = 16, 4 us at K = 1 on 1 M cells)
  // and cuts the passes from 46 to 28 (x2) / 24 (x3); narrow rows gain from the third application, wide rows do not
  // two everywhere (round 1 ran three at K <= 4).  Same box, ms per step at 2 / 3 / 4 applications (profiles/r02_w_local_reps.txt):
  // K = 1 1.007 / 1.011 / 1.096, K = 2 1.150 / 1.146 / 1.260, K = 4 1.476 / 1.555 / 1.745, K = 8 2.162 / 2.265 / 2.598, K = 16 3.64-3.67 / 3.786 / 4.234
  // (small meshes at narrow K keep three: their passes are a single round of tiles, bound by its latency, and an application
  // more is nearly free -- 8 000 cells, K = 1, CFL 18: 110 sweeps and 0.53 ms per step with three, 226 and 0.57 with two)
  eng->local_reps = (K <= 4 && n_owned < 100000) ? 3 : 2;
  eng->reps_base = eng->local_reps;
  if (const char* v = getenv("CWR_LOCAL_REPS")) { eng->local_reps = std::max(1, std::min(16, atoi(v))); eng->reps_auto = false; }
  eng->nt_stream = (K >= 8) ? 1 : 0;
  if (const char* v = getenv("CWR_NT_STREAM")) eng->nt_stream = atoi(v) != 0;
  if (const char* v = getenv("CWR_SQ_MIN_K")) eng->sq_min_k = std::max(1, atoi(v));
  if (const char* v = getenv("CWR_TCL_POWER")) eng->tcl_power = atoi(v) == 1 ? 1 : 2;
  eng->U = std::max(1, std::min(4, tile_rows / eng->R));
  int TR = 0;
  for (;;) {                                                       // the records of one tile must fit the LDS staging area
    TR = eng->R * eng->U;
    eng->ntiles = cdiv(n_owned, TR);
    int cap = 0;
    for (int b = 0; b < eng->ntiles; ++b) {
      const int c0 = b * TR, c1 = std::min(c0 + TR, n_owned);
      cap = std::max(cap, cnt[c1] - cnt[c0]);
    }
    eng->stage_cap = std::max(cap, 1);
    eng->apply_lds = (size_t)eng->stage_cap * sizeof(FaceRec) + (size_t)red_doubles(eng->G, eng->VW) * sizeof(double) +
                     (size_t)(TR + 1) * sizeof(int32_t);
    eng->apply_lds = (eng->apply_lds + 15) & ~(size_t)15;
    if (eng->apply_lds <= 64 * 1024 || eng->U == 1) break;
    eng->U /= 2;
  }
  if (eng->apply_lds > 160 * 1024) {
    delete eng;
    return fail(nullptr, CWR_ERR_BAD_ARG, "cwr_create: a block of cells has too many faces for the LDS staging area");
  }
  eng->cu_cap = cu_cap;

#define CREATE_TRY(call) do { int _rc = (call); if (_rc != CWR_OK) { g_create_error = eng->err; cwr_destroy(eng); return _rc; } } while (0)
#define CREATE_HIP(call) do { hipError_t _st = (call); if (_st != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(_st); cwr_destroy(eng); return CWR_ERR_HIP; } } while (0)
  CREATE_HIP(enter_device(device));
  CREATE_HIP(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
  if (eng->apply_lds > 48 * 1024) {
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
    CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apply<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->apply_lds));
  }
  {
    // persistent grid = what is really co-resident: blocks/CU from the occupancy query (LDS, registers, waves),
    // times the CU count, rounded down to a multiple of 8 (one share per XCD); a block that had to wait for a
    // free CU slot would run its whole tile range as a tail
    int per_cu = 1, n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    eng->n_cu = n_cu;
    eng->small_wg_cap = std::max(1, std::min(128, n_cu / 2));
    g_n_cu.store(n_cu);                           // (what cwr_chain_min_rows answers with from now on: it makes no HIP call itself)
    const void* fn = (eng->VW == 2) ? reinterpret_cast<const void*>(&k_apply<2, 2>) : reinterpret_cast<const void*>(&k_apply<1, 2>);
    per_cu = resident_blocks(fn, eng->apply_lds);
    per_cu = std::min(per_cu, eng->cu_cap);
    eng->apply_grid = std::max(N_XCD, std::min(cdiv(eng->ntiles, N_XCD) * N_XCD, (n_cu * per_cu / N_XCD) * N_XCD));
  }
  const size_t nK = (size_t)n_real * K;
  CREATE_TRY(dev_alloc(eng, &eng->d_f1, (size_t)n_edges));
  CREATE_TRY(dev_alloc(eng, &eng->d_f2, (size_t)n_edges));
  CREATE_TRY(dev_alloc(eng, &eng->d_ptr, (size_t)n_owned + 1));
  CREATE_TRY(dev_alloc(eng, &eng->d_ent_edge, (size_t)nnz));
  CREATE_TRY(dev_alloc(eng, &eng->d_ent_nb, (size_t)nnz + SQN_PAD));
  CREATE_TRY(dev_alloc(eng, &eng->d_rec, (size_t)nnz));
  CREATE_TRY(dev_alloc(eng, &eng->d_diag, (size_t)n_owned));
  CREATE_TRY(dev_alloc(eng, &eng->d_w, (size_t)nnz + SQN_PAD));
  CREATE_TRY(dev_alloc(eng, &eng->d_chk, 4 * (size_t)K + 2));      // (+ ew_rel, read by k_apply MODE 4)
  CREATE_HIP(hipMemsetAsync(eng->d_chk, 0, (4 * (size_t)K + 2) * sizeof(double), eng->stream));   // (stream-ordered: see ensure_small_plan)
  CREATE_TRY(dev_alloc(eng, &eng->d_keep, (size_t)n_cells * K));
  if (const char* v = getenv("CWR_TEST_FIXED_SWEEPS")) eng->fixed_sweeps = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_NO_NOTE")) eng->use_note = atoi(v) == 0;
  if (const char* v = getenv("CWR_SMALL_PARTS")) eng->small_parts = std::max(0, std::min(16, atoi(v)));
  if (const char* v = getenv("CWR_SMALL_DEPTH")) eng->small_depth = std::max(1, std::min(16, atoi(v)));
  if (const char* v = getenv("CWR_SMALL_MAX_PARTS")) eng->small_max_parts = std::max(1, std::min(16, atoi(v)));
  if (const char* v = getenv("CWR_SMALL_MAX_CELLS")) eng->small_max_cells = std::max(0, atoi(v));
  if (const char* v = getenv("CWR_SMALL_FIRST_CHECK")) eng->small_first_check = atoi(v) != 0;
  if (const char* v = getenv("CWR_SMALL_FENCES")) eng->small_fences = atoi(v) != 0;
  if (const char* v = getenv("CWR_SMALL_SPIN_MS")) eng->small_spin_ms = std::max(0, atoi(v));   // (0: a test's way to the abort path -- any part that has to wait at all gives up)
  if (const char* v = getenv("CWR_OUTPUT_DIRECT_MB")) eng->out_direct_limit = (size_t)std::max(0, atoi(v)) << 20;   // 0: always the copy engine
  if (const char* v = getenv("CWR_NO_FUSED_BEGIN")) eng->fused_begin = atoi(v) == 0;
  if (eng->use_note) {
    // (a runtime that cannot map host memory leaves h_note null: the checks are downloaded as before)
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, (5 * (size_t)K + 2) * sizeof(double), hipHostMallocMapped) == hipSuccess && hp &&
        hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
      std::memset(hp, 0, (5 * (size_t)K + 2) * sizeof(double));
      eng->h_note = static_cast<double*>(hp);                      // (4 K check rows of the sweeps, or the 5 K numbers of k_small_jacobi)
      eng->h_note_seq = reinterpret_cast<unsigned long long*>(eng->h_note + 5 * (size_t)K);
      eng->d_note_view = static_cast<double*>(dp);                 // (the same address on this platform; asked for, not assumed)
      CREATE_TRY(dev_alloc(eng, &eng->d_note_state, 2));
      CREATE_HIP(hipMemsetAsync(eng->d_note_state, 0, 2 * sizeof(unsigned long long), eng->stream));
    } else (void)hipGetLastError();
  }
  if (const char* v = getenv("CWR_NO_ELEMENTWISE")) eng->ew_enabled = atoi(v) == 0;
  if (const char* v = getenv("CWR_EW_SPLIT")) eng->ew_split = atoi(v) != 0;
  if (const char* v = getenv("CWR_EW_REL_FLOOR")) eng->ew_rel_floor = std::max(1.0e-15, atof(v));
  CREATE_TRY(dev_alloc(eng, &eng->d_c, (size_t)n_cells * K));
  CREATE_TRY(dev_alloc(eng, &eng->d_r, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_r0, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_p, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_v, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_s, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_t, nK));
  CREATE_TRY(dev_alloc(eng, &eng->d_b, nK));
  // (the 8 step counters live behind the solver scalars: one memset clears both at the start of a step)
  CREATE_TRY(dev_alloc(eng, &eng->d_scal, eng->scal_alloc()));      // (+ 8 int32 counters + the precondition flag as a double)
  eng->d_counters = reinterpret_cast<int32_t*>(eng->d_scal + eng->scal_count());
  CREATE_TRY(dev_alloc(eng, &eng->d_partial, (size_t)2 * std::max(eng->apply_grid, 256 * 8) * 4 * K));   // (x 2: a sweep in two launches, see n_apply_inner)
  if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] K=%d VW=%d G=%d U=%d tiles=%d stage_cap=%d lds=%zu grid=%d\n", K, eng->VW, eng->G, eng->U, eng->ntiles, eng->stage_cap, eng->apply_lds, eng->apply_grid);
  CREATE_TRY(upload(eng, eng->d_f1, f1p.data(), (size_t)n_edges));
  CREATE_TRY(upload(eng, eng->d_f2, f2p.data(), (size_t)n_edges));
  CREATE_TRY(dev_alloc(eng, &eng->d_face_orig, (size_t)std::max(n_edges, 1)));
  CREATE_TRY(upload(eng, eng->d_face_orig, face_orig.data(), (size_t)n_edges));
  eng->h_face_pos = face_pos;
  eng->h_f1.assign(f1p.begin(), f1p.begin() + n_edges); eng->h_f2.assign(f2p.begin(), f2p.begin() + n_edges);
  CREATE_TRY(dev_alloc(eng, &eng->d_face_pos, (size_t)std::max(n_edges, 1)));
  CREATE_TRY(upload(eng, eng->d_face_pos, face_pos.data(), (size_t)n_edges));
  {
    std::vector<uint8_t> row_ghost((size_t)n_owned, 0);
    for (int c = 0; c < n_owned; ++c)
      for (int j = cnt[c]; j < cnt[c + 1]; ++j) if (ent_nb[(size_t)j] < 0) row_ghost[(size_t)c] = 1;
    CREATE_TRY(dev_alloc(eng, &eng->d_row_ghost, (size_t)n_owned));
    CREATE_TRY(upload(eng, eng->d_row_ghost, row_ghost.data(), (size_t)n_owned));
  }
  CREATE_TRY(upload(eng, eng->d_ptr, cnt.data(), (size_t)n_owned + 1));
  CREATE_TRY(upload(eng, eng->d_ent_edge, ent_edge.data(), (size_t)nnz));
  CREATE_TRY(upload(eng, eng->d_ent_nb, ent_nb.data(), (size_t)nnz));
  CREATE_HIP(hipMemsetAsync(eng->d_c, 0, (size_t)n_cells * K * sizeof(double), eng->stream));
  for (double* v : {eng->d_r, eng->d_r0, eng->d_p, eng->d_v, eng->d_s, eng->d_t, eng->d_b})
    CREATE_HIP(hipMemsetAsync(v, 0, nK * sizeof(double), eng->stream));
  CREATE_HIP(hipMemsetAsync(eng->d_scal, 0, eng->scal_alloc() * sizeof(double), eng->stream));
  CREATE_HIP(hipStreamSynchronize(eng->stream));
#undef CREATE_TRY
#undef CREATE_HIP
  if (!g_exit_hooked.exchange(true)) std::atexit(on_process_exit);
  *out = eng;
  return CWR_OK;
}

void cwr_destroy(cwr_engine* e) {
  if (!e || g_down.load()) return;                   // (after the library's exit handler: see g_down)
  hipSetDevice(e->dev);
  if (e->stream) hipStreamSynchronize(e->stream);
  cwr_output_close(e);
  if (e->comm_stream) hipStreamSynchronize(e->comm_stream);
  if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
  for (auto& kv : e->stretch_exec) if (kv.second) hipGraphExecDestroy(kv.second);
  if (e->ev_packed) hipEventDestroy(e->ev_packed);
  if (e->ev_halo) hipEventDestroy(e->ev_halo);
  if (e->ev_red_in) hipEventDestroy(e->ev_red_in);
  if (e->ev_red_out) hipEventDestroy(e->ev_red_out);
  if (e->comm_stream) hipStreamDestroy(e->comm_stream);
  if (e->sweep_exec) hipGraphExecDestroy(e->sweep_exec);
  if (e->sweep_graph) hipGraphDestroy(e->sweep_graph);
  if (e->sq_exec) hipGraphExecDestroy(e->sq_exec);
  for (auto& kv : e->batch_exec) if (kv.second) hipGraphExecDestroy(kv.second);
  if (e->tcl_exec) hipGraphExecDestroy(e->tcl_exec);
  if (e->tcl_graph) hipGraphDestroy(e->tcl_graph);
  if (e->sq_graph) hipGraphDestroy(e->sq_graph);
  for (hipEvent_t ev : e->ev) hipEventDestroy(ev);
  void* ptrs[] = {e->d_f1, e->d_f2, e->d_ptr, e->d_ent_edge, e->d_ent_nb, e->d_adv, e->d_vel, e->d_vol, e->d_dif,
                  e->d_bc, e->d_rec, e->d_diag, e->d_c, e->d_r, e->d_r0, e->d_p, e->d_v, e->d_s, e->d_t, e->d_b,
                  e->d_scal, e->d_partial, e->d_fadv, e->d_fdif, e->d_send_cells, e->d_sendbuf, e->d_recv_cells, e->d_recvbuf, e->d_ptr2, e->d_col2, e->d_row2, e->d_rec2, e->d_w, e->d_react, e->d_info, e->d_tcl_ptr, e->d_tcl_cols, e->d_loc2, e->d_w2, e->d_pair_ptr, e->d_slots, e->d_line_ptr, e->d_line_faces, e->d_ledger, e->d_mass_out, e->d_chk, e->d_face_orig, e->d_row_ghost, e->d_keep, e->d_in_rows, e->d_in_vals, e->d_face_pos, e->d_trow, e->d_vptr, e->d_meta, e->d_tile_inner, e->d_tile_outer, e->d_apply_inner, e->d_apply_outer, e->d_face_inner, e->d_face_outer, e->d_chkx, e->d_sq_fast, e->d_sched, e->d_link_ptr, e->d_link_ent, e->d_link_flux, e->d_scols, e->d_scols_io, e->sched_in.d, e->sched_out.d, e->d_small_rows, e->d_small_recs, e->d_small_offs, e->d_small_send_pos, e->d_small_send_cnt, e->d_small_recv_src, e->d_small_recv_pos, e->d_small_recv_cnt, e->d_small_pub, e->d_small_red};
  for (void* p : ptrs) if (p) hipFree(p);
  for (void* p : {(void*)e->d_in_f, (void*)e->d_flow_l, (void*)e->d_dist, (void*)e->d_jn, (void*)e->d_bad, (void*)e->d_wa, (void*)e->d_wb, (void*)e->d_wmax, (void*)e->d_bc_stage})
    if (p) hipFree(p);
  if (e->h_lvl) hipHostFree(e->h_lvl);
  if (e->flow_stream) { hipStreamSynchronize(e->flow_stream); hipStreamDestroy(e->flow_stream); }
  for (hipEvent_t ev : e->ev_level) if (ev) hipEventDestroy(ev);
  for (hipEvent_t ev : e->ev_lvl_local) if (ev) hipEventDestroy(ev);
  if (e->d_lvlx) hipFree(e->d_lvlx);
  if (e->ev_evict) hipEventDestroy(e->ev_evict);
  if (e->ev_bc) hipEventDestroy(e->ev_bc);
  if (e->d_note_state) hipFree(e->d_note_state);
  if (e->h_note) hipHostFree(e->h_note);
  if (e->h_notex) hipHostFree(e->h_notex);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
}

int32_t cwr_load_flow_field(cwr_engine* e, int32_t T, const float* face_flow, const float* edge_velocity,
                            const float* volume, const double* dt, const double* dist, double D) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (T < 2 || !face_flow || !edge_velocity || !volume || !dt || !dist)
    return fail(e, CWR_ERR_BAD_ARG, "cwr_load_flow_field: need >= 2 time levels and non-NULL arrays");
  HIP_TRY(e, enter_device(e->dev));
  TRY(alloc_flow(e, T));
  const size_t TE = (size_t)T * e->E;
  // the host arrays arrive in the reference's face order: upload to temporaries, gather into the internal face order
  DevTmp<float> t_flow, t_tmpf; DevTmp<double> t_dist, t_tmpd;
  TRY(dev_alloc(e, &t_flow.p, TE));
  TRY(dev_alloc(e, &t_tmpf.p, TE));
  TRY(dev_alloc(e, &t_dist.p, (size_t)e->E));
  TRY(dev_alloc(e, &t_tmpd.p, (size_t)e->E));
  float *d_flow = t_flow.p, *d_tmpf = t_tmpf.p; double *d_dist = t_dist.p, *d_tmpd = t_tmpd.p;
  const int gridTE = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)TE, BLOCK), 256 * 16));
  int rc = upload(e, d_tmpf, face_flow, TE);
  if (rc == CWR_OK && TE > 0) k_faces_in<float><<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, e->d_face_orig, d_tmpf, d_flow);
  if (rc == CWR_OK) rc = upload(e, d_tmpf, edge_velocity, TE);         // (upload synchronises: the gather above is done)
  if (rc == CWR_OK && TE > 0) k_faces_in<float><<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, e->d_face_orig, d_tmpf, e->d_vel);
  if (rc == CWR_OK) rc = upload(e, e->d_vol, volume, (size_t)T * e->n_cells);
  if (rc == CWR_OK) rc = upload(e, d_tmpd, dist, (size_t)e->E);
  if (rc == CWR_OK && e->E > 0) k_faces_in<double><<<cdiv(e->E, BLOCK), BLOCK, 0, e->stream>>>((int64_t)e->E, e->E, e->d_face_orig, d_tmpd, d_dist);
  if (rc == CWR_OK && TE > 0) {
    k_derive_coeff<<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, d_flow, e->d_vel, d_dist, (float)D, e->d_adv, e->d_dif);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
      rc = fail(e, CWR_ERR_HIP, "k_derive_coeff failed");
  }
  if (rc != CWR_OK) { e->T = 0; return rc; }
  e->dt.assign(dt, dt + T);
  e->D = D;
  TRY(check_ghost_levels(e));
  return compute_jnorms(e);
}

// ---- windowed flow-field residency ("time-series streaming") ------------------------------------------------------
// cwr_load_flow_field keeps all T levels in HBM: adv f32 + dif f64 + vel f32 per face and vol f32 per cell, ~37 MB per level at 1 M
// cells -- ~7 000 levels are the ceiling there, and the reference's own fixture has 10 801 stamps (tests/data/simple_test_cases/
// plan01_10x5), its reader windows a file by datetime_range (io/hdf.py:149-191) and utilities.py:513-541 derives per level.  Here
// the device holds a RING of W levels; cwr_flow_window_load uploads further levels on a stream of its own, derives their
// coefficients, the zero-coefficient flag and ||J||_inf there, beside the steps, and cwr_step(t) runs for any t whose levels t and
// t + 1 are in the ring.
int32_t cwr_flow_window_open(cwr_engine* e, int32_t T, int32_t W, const double* dt, const double* dist, double D) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (T < 2 || W < 2 || !dt || !dist) return fail(e, CWR_ERR_BAD_ARG, "cwr_flow_window_open: need >= 2 time levels, a window of >= 2 levels and non-NULL arrays");
  // (round 6) partitioned engines too: every rank holds a ring of ITS slices of W levels; what a single engine leaves for the host per level
  // (zero-coefficient flag, ||J||_inf) is all-reduced on the communication stream at the load's point -- see window_load_now.  Collective in
  // effect: every rank opens and loads the same levels at the same steps.
  if (e->comm && !(e->one_comm_stream && e->comm_stream))
    return fail(e, CWR_ERR_STATE, "cwr_flow_window_open: a partitioned engine needs its communication stream for windowed residency (not with CWR_COMM_TWO_STREAMS=1)");
  HIP_TRY(e, enter_device(e->dev));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  W = std::min(W, T);
  e->pending_loads.clear(); e->pending_bc.clear();   // (loads noted for a previous window: ADVICE r05)
  TRY(alloc_flow(e, W));                             // (W levels of the four arrays)
  e->T = T; e->W = W; e->windowed = W < T;
  e->dt.assign(dt, dt + T);
  e->D = D;
  e->slot_level.assign((size_t)W, -1);
  e->jnorm.assign((size_t)T, NAN); e->jnorm[(size_t)T - 1] = 0.0;
  e->err_factor.assign((size_t)T, INFINITY);
  e->bad_level.assign((size_t)T, 0);
  e->lvl_final.assign((size_t)T, 0);
  e->windowed = true;                                // (also with W == T: the levels still arrive one load at a time)
  for (void* p : {(void*)e->d_in_f, (void*)e->d_flow_l, (void*)e->d_dist, (void*)e->d_jn, (void*)e->d_bad}) if (p) hipFree(p);
  e->d_in_f = e->d_flow_l = nullptr; e->d_dist = nullptr; e->d_jn = nullptr; e->d_bad = nullptr;
  if (e->h_lvl) { hipHostFree(e->h_lvl); e->h_lvl = nullptr; }
  TRY(dev_alloc(e, &e->d_in_f, (size_t)std::max(e->E, e->n_cells)));
  TRY(dev_alloc(e, &e->d_flow_l, (size_t)e->E));
  TRY(dev_alloc(e, &e->d_dist, (size_t)e->E));
  TRY(dev_alloc(e, &e->d_jn, (size_t)T));
  TRY(dev_alloc(e, &e->d_bad, (size_t)T));
  HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_lvl), (size_t)2 * T * sizeof(double), hipHostMallocMapped));
  std::memset(e->h_lvl, 0, (size_t)2 * T * sizeof(double));
  { void* dp = nullptr; HIP_TRY(e, hipHostGetDevicePointer(&dp, e->h_lvl, 0)); e->d_lvl_view = static_cast<double*>(dp); }
  HIP_TRY(e, hipMemsetAsync(e->d_jn, 0, (size_t)T * sizeof(unsigned long long), e->stream));
  HIP_TRY(e, hipMemsetAsync(e->d_bad, 0, (size_t)T * sizeof(int32_t), e->stream));
  {
    DevTmp<double> tmp;                              // face_to_face_dist: reference face order -> internal
    TRY(dev_alloc(e, &tmp.p, (size_t)e->E));
    TRY(upload(e, tmp.p, dist, (size_t)e->E));
    if (e->E > 0) k_faces_in<double><<<cdiv(e->E, BLOCK), BLOCK, 0, e->stream>>>((int64_t)e->E, e->E, e->d_face_orig, tmp.p, e->d_dist);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  if (!e->flow_stream) HIP_TRY(e, hipStreamCreateWithFlags(&e->flow_stream, hipStreamNonBlocking));
  if (!e->ev_evict) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_evict, hipEventDisableTiming));
  for (hipEvent_t ev : e->ev_level) hipEventDestroy(ev);
  e->ev_level.assign((size_t)W, nullptr);
  for (auto& ev : e->ev_level) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (hipEvent_t ev : e->ev_lvl_local) hipEventDestroy(ev);
  e->ev_lvl_local.assign((size_t)W, nullptr);
  for (auto& ev : e->ev_lvl_local) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (e->d_lvlx) { hipFree(e->d_lvlx); e->d_lvlx = nullptr; }
  e->sched_level = -1;
  return CWR_OK;
}

int32_t cwr_flow_window_load(cwr_engine* e, int32_t t0, int32_t n_levels, const float* face_flow, const float* edge_velocity, const float* volume) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (!e->windowed || !e->flow_stream) return fail(e, CWR_ERR_STATE, "cwr_flow_window_load: cwr_flow_window_open first");
  if (t0 < 0 || n_levels < 1 || t0 + n_levels > e->T || n_levels > e->W || !face_flow || !edge_velocity || !volume)
    return fail(e, CWR_ERR_BAD_ARG, "cwr_flow_window_load: levels outside the run, more levels than the window holds, or NULL arrays");
  // noted, not enqueued: the next cwr_step (or any call that needs a level: check_level) sends it to the flow stream -- behind its batch
  // of passes where the step itself does not need the levels (see pending_loads)
  if (getenv("CWR_WINDOW_EAGER")) return window_load_now(e, t0, n_levels, face_flow, edge_velocity, volume);     // (A/B: enqueue at the call)
  e->pending_loads.push_back(cwr_engine::PendingLoad{t0, n_levels, face_flow, edge_velocity, volume});
  return CWR_OK;
}
}  // extern "C" (closed for the two helpers below)

namespace {
int flush_window_loads(cwr_engine* e) {
  if (!e->pending_bc.empty()) {
    // the boundary rows first: the flow levels enqueued behind them record the events a step waits for, and ev_bc says it outright
    HIP_TRY(e, enter_device(e->dev));
    std::vector<cwr_engine::PendingBc> bcs;
    bcs.swap(e->pending_bc);
    if (!e->ev_bc) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_bc, hipEventDisableTiming));
    for (const auto& pb : bcs) {
      const size_t rows = (size_t)pb.n * e->n_ghost;
      double* dst = e->d_bc + (size_t)pb.t0 * e->n_ghost * e->K;
      if (e->K == e->Ku) {
        HIP_TRY(e, hipMemcpyAsync(dst, pb.v, rows * e->K * sizeof(double), hipMemcpyHostToDevice, e->flow_stream));
      } else {
        if (e->bc_stage_cap < rows * e->Ku) {
          HIP_TRY(e, hipStreamSynchronize(e->flow_stream));
          hipFree(e->d_bc_stage); e->d_bc_stage = nullptr; e->bc_stage_cap = 0;
          TRY(dev_alloc(e, &e->d_bc_stage, rows * e->Ku));
          e->bc_stage_cap = rows * e->Ku;
        }
        HIP_TRY(e, hipMemcpyAsync(e->d_bc_stage, pb.v, rows * e->Ku * sizeof(double), hipMemcpyHostToDevice, e->flow_stream));
        const int64_t total = (int64_t)rows * e->K;
        k_pad_cols<<<(int)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, BLOCK), 256 * 16)), BLOCK, 0, e->flow_stream>>>(total, e->Ku, e->K, e->d_bc_stage, dst);
        HIP_TRY(e, hipGetLastError());
      }
    }
    HIP_TRY(e, hipEventRecord(e->ev_bc, e->flow_stream));
    e->bc_event_pending = true;
  }
  std::vector<cwr_engine::PendingLoad> todo;
  todo.swap(e->pending_loads);
  for (const auto& pl : todo) TRY(window_load_now(e, pl.t0, pl.n, pl.ff, pl.ev, pl.vol));
  return CWR_OK;
}

int window_load_now(cwr_engine* e, int t0, int n_levels, const float* face_flow, const float* edge_velocity, const float* volume) {
  HIP_TRY(e, enter_device(e->dev));
  const size_t E = (size_t)e->E, nc = (size_t)e->n_cells;
  hipStream_t fs = e->flow_stream;
  e->prepared_t = -1;
  // CWR_WINDOW_DEBUG (measurement only, tools/r05_window_debug3.sh): 1 = bookkeeping and events only (no copy, no kernel: the slots keep
  // stale levels), 2 = the copies without the kernels -- where a windowed step's extra time goes
  static const int dbg = getenv("CWR_WINDOW_DEBUG") ? atoi(getenv("CWR_WINDOW_DEBUG")) : 0;
  // the slots about to be overwritten may still be read by what the engine's stream holds (a step's closing flux kernel reads the
  // coefficients of its level): the flow stream waits for everything enqueued there so far
  HIP_TRY(e, hipEventRecord(e->ev_evict, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(fs, e->ev_evict, 0));
  const int gE = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)E, BLOCK), 256 * 16));
  for (int i = 0; i < n_levels; ++i) {
    const int L = t0 + i;
    const size_t sl = e->slot(L);
    const int old = e->slot_level[sl];
    e->slot_level[sl] = L;
    if (old >= 0 && old != L) {                      // what was derived WITH the level that leaves must be derived again if it ever returns
      e->lvl_final[(size_t)old] = 0;
      if (old > 0) e->lvl_final[(size_t)old - 1] = 0;
    }
    e->lvl_final[(size_t)L] = 0;
    if (L > 0) e->lvl_final[(size_t)L - 1] = 0;
    // face flows and velocities arrive in the reference's face order: two staging arrays, then ONE kernel gathers them into the
    // internal order, derives the coefficients and sets the level's zero-coefficient flag (k_level_in); the volumes go straight
    // into their slot
    if (dbg == 1 && old >= 0) { e->h_lvl[2 * (size_t)L] = e->h_lvl[2 * (size_t)old]; e->h_lvl[2 * (size_t)L + 1] = 0.0; if (L > 0 && old > 0) e->h_lvl[2 * (size_t)L - 2] = e->h_lvl[2 * (size_t)old - 2];
                                HIP_TRY(e, hipEventRecord(e->ev_level[sl], fs)); continue; }
    HIP_TRY(e, hipMemsetAsync(e->d_bad + L, 0, sizeof(int32_t), fs));
    HIP_TRY(e, hipMemcpyAsync(e->d_in_f, face_flow + (size_t)i * E, E * sizeof(float), hipMemcpyHostToDevice, fs));
    HIP_TRY(e, hipMemcpyAsync(e->d_flow_l, edge_velocity + (size_t)i * E, E * sizeof(float), hipMemcpyHostToDevice, fs));
    if (E > 0 && !(dbg == 2 && old >= 0)) k_level_in<<<gE, BLOCK, 0, fs>>>(e->E, e->n_owned, e->n_real, e->d_face_orig, e->d_f1, e->d_f2, e->d_in_f, e->d_flow_l, e->d_dist, (float)e->D,
                                               e->D != 0.0 ? 1 : 0, e->vel_l(L), e->adv_l(L), e->dif_l(L), e->d_bad + L);
    HIP_TRY(e, hipMemcpyAsync(e->vol_l(L), volume + (size_t)i * nc, nc * sizeof(float), hipMemcpyHostToDevice, fs));
    HIP_TRY(e, hipGetLastError());
    // ||J||_inf of the steps this level completes: step L - 1 (its coefficients, this level's volumes) and step L (when level L + 1 is
    // already here: levels loaded out of order); then ONE tiny kernel leaves the level's flag and those norms in page-locked memory
    int st_done[2] = {-1, -1};
    for (int q = 0; q < 2; ++q) {
      const int st = L - 1 + q;
      if (st < 0 || st + 1 >= e->T) continue;
      if (e->slot_level[e->slot(st)] != st || e->slot_level[e->slot(st + 1)] != st + 1) continue;
      if (dbg == 2 && old >= 0) { e->h_lvl[2 * (size_t)st] = 0.7836; continue; }
      HIP_TRY(e, hipMemsetAsync(e->d_jn + st, 0, sizeof(unsigned long long), fs));
      k_jnorm<<<dim3((unsigned)std::max(1, std::min(cdiv(e->n_owned, BLOCK), 1024)), 1u), BLOCK, 0, fs>>>(e->n_owned, e->E, e->n_cells, e->d_ptr, e->d_ent_edge,
          e->d_ent_nb, e->adv_l(st), e->dif_l(st), e->vol_l(st + 1), nullptr, e->d_jn + st, e->dt[(size_t)st]);
      st_done[q] = st;
    }
    if (e->comm && (e->world > 1 || e->force_coll)) {
      // partitioned: this rank's scalars into its slot of the slot's block (flow stream), ONE sum all-reduce of world x 3 doubles on the
      // communication stream behind it, the fold over the ranks into the page-locked words, and the slot's event -- the one a step waits
      // for -- recorded THERE: the level is complete when every rank's share of it is.  Every rank reaches this point with the same
      // level at the same place in its sequence of communication calls (loads are noted and flushed by rules that depend on t alone).
      if (!(e->one_comm_stream && e->comm_stream)) return fail(e, CWR_ERR_STATE, "windowed flow field on a partitioned engine: needs the communication stream (not with CWR_COMM_TWO_STREAMS=1)");
      const size_t blk = (size_t)3 * e->world;
      if (!e->d_lvlx) TRY(dev_alloc(e, &e->d_lvlx, (size_t)e->W * blk));
      hipStream_t cs = e->comm_stream;
      k_pack_level<<<1, 64, 0, fs>>>(e->world, e->rank, e->d_bad + L, st_done[0] >= 0 ? e->d_jn + st_done[0] : nullptr,
                                    st_done[1] >= 0 ? e->d_jn + st_done[1] : nullptr, e->d_lvlx + sl * blk);
      HIP_TRY(e, hipGetLastError());
      HIP_TRY(e, hipEventRecord(e->ev_lvl_local[sl], fs));
      HIP_TRY(e, hipStreamWaitEvent(cs, e->ev_lvl_local[sl], 0));
      NCCL_TRY(e, g_rccl.AllReduce(e->d_lvlx + sl * blk, e->d_lvlx + sl * blk, blk, NCCL_FLOAT64, NCCL_SUM, e->comm, cs));
      k_note_level_ranks<<<1, 64, 0, cs>>>(e->world, e->d_lvlx + sl * blk, e->d_lvl_view + 2 * (size_t)L + 1,
                                          st_done[0] >= 0 ? e->d_lvl_view + 2 * (size_t)st_done[0] : nullptr,
                                          st_done[1] >= 0 ? e->d_lvl_view + 2 * (size_t)st_done[1] : nullptr);
      HIP_TRY(e, hipGetLastError());
      HIP_TRY(e, hipEventRecord(e->ev_level[sl], cs));
      continue;
    }
    k_note_level<<<1, 1, 0, fs>>>(e->d_bad + L, e->d_lvl_view + 2 * (size_t)L + 1,
                                 st_done[0] >= 0 ? e->d_jn + st_done[0] : nullptr, st_done[0] >= 0 ? e->d_lvl_view + 2 * (size_t)st_done[0] : nullptr,
                                 st_done[1] >= 0 ? e->d_jn + st_done[1] : nullptr, st_done[1] >= 0 ? e->d_lvl_view + 2 * (size_t)st_done[1] : nullptr);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipEventRecord(e->ev_level[sl], fs));
  }
  return CWR_OK;
}
}  // namespace

extern "C" {

int32_t cwr_load_coefficients(cwr_engine* e, int32_t T, const float* adv, const double* dif, const float* vel,
                              const float* volume, const double* dt, double D) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (T < 2 || !adv || !dif || !vel || !volume || !dt)
    return fail(e, CWR_ERR_BAD_ARG, "cwr_load_coefficients: need >= 2 time levels and non-NULL arrays");
  HIP_TRY(e, enter_device(e->dev));
  TRY(alloc_flow(e, T));
  const size_t TE = (size_t)T * e->E;
  DevTmp<float> t_tmpf; DevTmp<double> t_tmpd;                     // reference face order -> internal face order
  TRY(dev_alloc(e, &t_tmpf.p, TE));
  TRY(dev_alloc(e, &t_tmpd.p, TE));
  float* d_tmpf = t_tmpf.p; double* d_tmpd = t_tmpd.p;
  const int gridTE = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)TE, BLOCK), 256 * 16));
  int rc = upload(e, d_tmpf, adv, TE);
  if (rc == CWR_OK) k_faces_in<float><<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, e->d_face_orig, d_tmpf, e->d_adv);
  if (rc == CWR_OK) rc = upload(e, d_tmpf, vel, TE);
  if (rc == CWR_OK) k_faces_in<float><<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, e->d_face_orig, d_tmpf, e->d_vel);
  if (rc == CWR_OK) rc = upload(e, d_tmpd, dif, TE);
  if (rc == CWR_OK) k_faces_in<double><<<gridTE, BLOCK, 0, e->stream>>>((int64_t)TE, e->E, e->d_face_orig, d_tmpd, e->d_dif);
  if (rc == CWR_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)) rc = fail(e, CWR_ERR_HIP, "k_faces_in failed");
  if (rc != CWR_OK) { e->T = 0; return rc; }
  TRY(upload(e, e->d_vol, volume, (size_t)T * e->n_cells));
  e->dt.assign(dt, dt + T);
  e->D = D;
  TRY(check_ghost_levels(e));
  return compute_jnorms(e);
}

int32_t cwr_get_coefficients(cwr_engine* e, int32_t t, float* adv, double* dif) {
  if (!e) return CWR_ERR_BAD_ARG;
  TRY(check_level(e, t, false));
  HIP_TRY(e, enter_device(e->dev));
  DevTmp<float> t_tmpf; DevTmp<double> t_tmpd;                     // internal face order -> reference face order
  float*& d_tmpf = t_tmpf.p; double*& d_tmpd = t_tmpd.p;
  int rc = CWR_OK;
  if (adv) {
    rc = dev_alloc(e, &d_tmpf, (size_t)std::max(e->E, 1));
    if (rc == CWR_OK) { k_faces_out<float><<<cdiv(std::max(e->E, 1), BLOCK), BLOCK, 0, e->stream>>>(e->E, e->d_face_orig, e->adv_l(t), d_tmpf);
                        rc = download(e, adv, d_tmpf, (size_t)e->E); }
  }
  if (rc == CWR_OK && dif) {
    rc = dev_alloc(e, &d_tmpd, (size_t)std::max(e->E, 1));
    if (rc == CWR_OK) { k_faces_out<double><<<cdiv(std::max(e->E, 1), BLOCK), BLOCK, 0, e->stream>>>(e->E, e->d_face_orig, e->dif_l(t), d_tmpd);
                        rc = download(e, dif, d_tmpd, (size_t)e->E); }
  }
  return rc;
}

int32_t cwr_load_boundary(cwr_engine* e, int32_t T, const double* ghost_conc) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (T < 1) return fail(e, CWR_ERR_BAD_ARG, "cwr_load_boundary: bad arguments");
  HIP_TRY(e, enter_device(e->dev));
  const size_t cnt = (size_t)T * e->n_ghost * e->K;
  e->pending_bc.clear();
  if (e->T_bc != T) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    hipFree(e->d_bc); e->d_bc = nullptr; e->T_bc = 0;
    TRY(dev_alloc(e, &e->d_bc, cnt));
    e->T_bc = T;
  }
  if (!ghost_conc) {
    // (round 6) NULL: T levels of zeros ("no boundary value") -- the levels then arrive a few at a time (cwr_boundary_window_load,
    // cwr_set_boundary_level): a run that streams its flow field level by level never holds all T levels of boundary values on the host
    if (cnt > 0) HIP_TRY(e, hipMemsetAsync(e->d_bc, 0, cnt * sizeof(double), e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return CWR_OK;
  }
  TRY(upload_cols(e, e->d_bc, ghost_conc, (size_t)T * e->n_ghost));
  return CWR_OK;
}

// Boundary values of levels t0 .. t0 + n_levels - 1 ((n_levels, n_ghost, K) doubles; the reference's input_array[t, ghost cells],
// constituents.py:153-164) into their rows of the array cwr_load_boundary allocated.  On a windowed engine (cwr_flow_window_open) the
// call only NOTES the pointer, like cwr_flow_window_load: the copy runs on the engine's flow stream beside the steps, enqueued by the
// next cwr_step in front of the flow levels noted with it, and the step that reads level t + 1 waits for it on the device.  The host
// array stays untouched until a cwr_step that reads the levels, or cwr_synchronize, has returned.  Other engines: a blocking upload.
int32_t cwr_boundary_window_load(cwr_engine* e, int32_t t0, int32_t n_levels, const double* ghost_conc) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (e->T_bc <= 0) return fail(e, CWR_ERR_STATE, "cwr_boundary_window_load: cwr_load_boundary first (it allocates the levels; NULL values: zeros)");
  if (t0 < 0 || n_levels < 1 || t0 + n_levels > e->T_bc || (!ghost_conc && e->n_ghost > 0))
    return fail(e, CWR_ERR_BAD_ARG, "cwr_boundary_window_load: levels outside the loaded boundary array, or NULL values");
  if (e->n_ghost == 0) return CWR_OK;
  if (e->windowed && e->flow_stream && !getenv("CWR_WINDOW_EAGER")) {
    e->pending_bc.push_back(cwr_engine::PendingBc{t0, n_levels, ghost_conc});
    return CWR_OK;
  }
  HIP_TRY(e, enter_device(e->dev));
  TRY(upload_cols(e, e->d_bc + (size_t)t0 * e->n_ghost * e->K, ghost_conc, (size_t)n_levels * e->n_ghost));
  return CWR_OK;
}

int32_t cwr_set_boundary_level(cwr_engine* e, int32_t t, const double* level) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (t < 0 || t >= e->T_bc) return fail(e, CWR_ERR_STATE, "cwr_set_boundary_level: level outside the loaded boundary array");
  HIP_TRY(e, enter_device(e->dev));
  TRY(upload_cols(e, e->d_bc + (size_t)t * e->n_ghost * e->K, level, (size_t)e->n_ghost));
  return CWR_OK;
}

int32_t cwr_load_real_inputs(cwr_engine* e, int32_t n_entries, const int32_t* level, const int32_t* row, const double* values) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (n_entries < 0 || (n_entries > 0 && (!level || !row || !values))) return fail(e, CWR_ERR_BAD_ARG, "cwr_load_real_inputs: bad arguments");
  for (int i = 0; i < n_entries; ++i) {
    if (row[i] < 0 || row[i] >= e->n_core) return fail(e, CWR_ERR_BAD_ARG, "cwr_load_real_inputs: row outside this engine's own real cells");
    if (level[i] < 1) return fail(e, CWR_ERR_BAD_ARG, "cwr_load_real_inputs: level must be >= 1 (level 0 is the initial state: cwr_set_state)");
    if (i > 0 && level[i] < level[i - 1]) return fail(e, CWR_ERR_BAD_ARG, "cwr_load_real_inputs: entries must be sorted by level");
  }
  HIP_TRY(e, enter_device(e->dev));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  hipFree(e->d_in_rows); hipFree(e->d_in_vals);
  e->d_in_rows = nullptr; e->d_in_vals = nullptr; e->in_levels.clear();
  if (n_entries == 0) return sync_input_levels(e);
  TRY(dev_alloc(e, &e->d_in_rows, (size_t)n_entries));
  TRY(dev_alloc(e, &e->d_in_vals, (size_t)n_entries * e->K));
  TRY(upload(e, e->d_in_rows, row, (size_t)n_entries));
  TRY(upload_cols(e, e->d_in_vals, values, (size_t)n_entries));
  for (int i = 0; i < n_entries; ++i) {
    auto it = e->in_levels.find(level[i]);
    if (it == e->in_levels.end()) e->in_levels[level[i]] = std::make_pair(i, 1); else it->second.second += 1;
  }
  return sync_input_levels(e);
}

int32_t cwr_set_state(cwr_engine* e, const double* conc_owned) {
  if (!e || !conc_owned) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_set_state: NULL") : CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  e->halo_fresh = false;
  TRY(upload_cols(e, e->d_c, conc_owned, (size_t)e->n_core));
  return CWR_OK;
}

int32_t cwr_react_linear(cwr_engine* e, const double* M) {
  if (!e || !M) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_react_linear: NULL") : CWR_ERR_BAD_ARG;
  if (e->K > BLOCK) return fail(e, CWR_ERR_BAD_ARG, "cwr_react_linear: K too large");
  HIP_TRY(e, enter_device(e->dev));
  const int K = e->K;
  e->halo_fresh = false;
  if (!e->d_react) TRY(dev_alloc(e, &e->d_react, (size_t)K * K));
  {
    std::vector<double> Mp((size_t)K * K, 0.0);                  // (the caller's Ku x Ku block; padded columns stay zero)
    for (int i = 0; i < e->Ku; ++i) for (int j = 0; j < e->Ku; ++j) Mp[(size_t)i * K + j] = M[(size_t)i * e->Ku + j];
    TRY(upload(e, e->d_react, Mp.data(), (size_t)K * K));
  }
  const int rows_pb = BLOCK / K;
  const size_t lds = ((size_t)rows_pb * K + (size_t)K * K) * sizeof(double);
  const int grid = std::max(1, std::min(cdiv(e->n_core, rows_pb), 256 * 8));
  k_react_linear<<<grid, BLOCK, lds, e->stream>>>(e->n_core, K, e->d_react, e->d_c);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return CWR_OK;
}

int32_t cwr_state_device_ptr(cwr_engine* e, void** state, void** stream) {
  if (!e || !state) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_state_device_ptr: NULL") : CWR_ERR_BAD_ARG;
  *state = e->d_c;
  e->halo_fresh = false;                              // the caller may rewrite the state,
  e->ptr_exported = true;                             // now and between any two later steps (the pointer never changes)
  if (stream) *stream = e->stream;
  return CWR_OK;
}

int32_t cwr_state_row_stride(const cwr_engine* e) { return e ? e->K : 0; }

int32_t cwr_get_state(cwr_engine* e, double* conc_all) {
  if (!e || !conc_all) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_get_state: NULL") : CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  TRY(download_cols(e, conc_all, e->d_c, (size_t)e->n_cells));
  return CWR_OK;
}

int32_t cwr_apply(cwr_engine* e, int32_t t, const double* x, double* y) {
  if (!e || !x || !y) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_apply: NULL") : CWR_ERR_BAD_ARG;
  TRY(check_level(e, t, true));
  HIP_TRY(e, enter_device(e->dev));
  TRY(upload_cols(e, e->d_p, x, (size_t)e->n_real));
  TRY(prep_step(e, t));
  TRY(launch_apply<0>(e, e->d_p, e->d_v, nullptr, nullptr, nullptr, nullptr));
  TRY(download_cols(e, y, e->d_v, (size_t)e->n_owned));
  return CWR_OK;
}

int32_t cwr_rhs(cwr_engine* e, int32_t t, const double* x_t, double* b) {
  if (!e || !x_t || !b) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_rhs: NULL") : CWR_ERR_BAD_ARG;
  TRY(check_level(e, t, true));
  if (e->T_bc < t + 2) return fail(e, CWR_ERR_STATE, "cwr_rhs: boundary values of level t+1 not loaded");
  HIP_TRY(e, enter_device(e->dev));
  TRY(upload_cols(e, e->d_s, x_t, (size_t)e->n_owned));
  HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 8 * sizeof(int32_t), e->stream));
  TRY(launch_rhs(e, t, e->d_s, e->d_t, false));
  int32_t cnt[8];
  TRY(download(e, cnt, e->d_counters, (size_t)8));
  if (cnt[2]) return fail(e, CWR_ERR_GHOST_COEFF, "active ghost face with a zero advection/diffusion coefficient at level t+1 "
                          "(the reference raises a shape-mismatch ValueError, linalg.py:349-351)");
  TRY(download_cols(e, b, e->d_t, (size_t)e->n_owned));
  return CWR_OK;
}

int32_t cwr_step(cwr_engine* e, int32_t t, double tol, int32_t max_iter, int32_t flags, cwr_step_info* info) {
  if (!e) return CWR_ERR_BAD_ARG;
  const auto w0 = std::chrono::steady_clock::now();
  cwr_step_info local; std::memset(&local, 0, sizeof(local));
  if (info) *info = local;
  e->defer_loads = false;
  if (e->windowed && (!e->pending_loads.empty() || !e->pending_bc.empty())) {
    // loads this step needs -- or that would replace a level it reads -- are enqueued now; all others behind the step's batch
    bool now = false;
    for (const auto& pl : e->pending_loads)
      for (int L = pl.t0; L < pl.t0 + pl.n; ++L)
        if (L == t || L == t + 1 || e->slot(L) == e->slot(t) || e->slot(L) == e->slot(t + 1)) now = true;
    for (const auto& pb : e->pending_bc) if (pb.t0 <= t + 1 && t + 1 < pb.t0 + pb.n) now = true;   // (the boundary values this step reads)
    e->defer_loads = !now;
  }
  struct DeferGuard { cwr_engine* e; ~DeferGuard() { if (e->defer_loads || !e->pending_loads.empty() || !e->pending_bc.empty()) { e->defer_loads = false; (void)flush_window_loads(e); } } } defer_guard{e};
  TRY(check_level(e, t, true));
  if (e->T_bc < t + 2) return fail(e, CWR_ERR_STATE, "cwr_step: boundary values of level t+1 not loaded (cwr_load_boundary)");
  if (!(tol > 0.0) || max_iter < 1) return fail(e, CWR_ERR_BAD_ARG, "cwr_step: tol must be > 0 and max_iter >= 1");
  HIP_TRY(e, enter_device(e->dev));
  if (e->windowed && !e->defer_loads && !e->pending_bc.empty()) TRY(flush_window_loads(e));   // (boundary rows noted without flow levels)
  if (e->bc_event_pending) {                           // boundary rows copied on the flow stream since the last step: this step's kernels behind them
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_bc, 0));
    e->bc_event_pending = false;
  }
  TRY(finalize_level(e, t));
  const int K = e->K;
  const double tol2 = tol * tol;
  e->profiling = (flags & CWR_STEP_PROFILE) != 0;
  if (e->profiling && e->ev.empty()) {
    e->ev.resize(1024);
    for (auto& ev : e->ev) HIP_TRY(e, hipEventCreate(&ev));
  }
  e->ev_used = 0;
  e->flux_valid = false;
  e->tail_done = false;
  e->info_flags = e->small_fell_back ? CWR_INFO_SMALL_FALLBACK : 0;
  e->cur_t = t;
  e->deterministic = (flags & CWR_STEP_DETERMINISTIC) != 0 || (e->det_walk && e->K <= e->det_default_k);   // (the same on every rank: K and the environment are)
  e->step_chained = 0;
  e->step_exchanges = e->step_overlapped = e->step_checks = 0;
  {
    // element-wise rule: targets (1e6 tol, tol) = (1e-6, 1e-12) at the default tolerance, scaled by s = 0.3 (1 - rho) / rho with
    // rho = ||J||_inf of THIS step's iteration matrix (exact, from the flow field: k_jnorm) -- Jacobi's a-posteriori bound
    // ||x* - x'||_inf <= rho / (1 - rho) ||x' - x||_inf then keeps the forward error within 0.3 (1e6 tol + tol) max|x| in the
    // max norm, rigorously.  (Round 2 used the measured 2-norm contraction of an earlier check, which is not a bound.)
    // s is kept within [1e-3, 0.1]: below 1e-3 (rho > 0.9967, CFL of several hundred) |x' - x| would have to fall under the
    // rounding of a sweep; the step then runs at s = 1e-3 and says so: CWR_INFO_ELEMENTWISE_CLAMPED.
    // (round 4: the factor rho / (1 - rho) is replaced by the row-wise bound F_t of refine_error_factors where that is smaller --
    // meshes with dry or nearly dry cells, whose worst row sum says nothing about the error of a sweep)
    // (round 6: the two parts of the rule are floored SEPARATELY.  What rounding limits is |x' - x| against the cell's own size: a
    // sweep sums at most nine non-negative terms (J >= 0, b >= 0 for concentrations), so a converged sweep repeats itself to a few
    // 1e-16 |x'_i| -- the RELATIVE part may follow s = 0.3 / F down to ew_rel = 1e-13 (s = 1e-7 at tol = 1e-12: F = 3e6), seven decades
    // below the 1e-9 it was held at.  Only the ABSOLUTE part, s tol max|x'|, is at rounding size already at s = 1e-3 (1e-15 max|x'|)
    // and stays floored there.  The step's bound is then F (ew_rel + ew_abs) max|x'| = (0.3 * 1e6 tol + 1e-3 F tol) max|x'|, and
    // CWR_INFO_ELEMENTWISE_CLAMPED is raised only when THAT exceeds the target (1e6 tol + tol) max|x'| -- F > 7e8 -- or F is no
    // bound at all.  Before: every river-band mesh with a dry cell at dt = 3600 s (F = 300 ... 5000) ran clamped, VERDICT r05 weak 2;
    // CWR_EW_SPLIT=0 restores that rule, A/B)
    double F = ((size_t)t < e->err_factor.size()) ? e->err_factor[(size_t)t] : INFINITY;
    if (!(F >= 0.0)) F = INFINITY;
    const double R = std::min(1.0e-2, 1.0e6 * tol), A = tol;
    const double s_raw = (F > 0.0) ? 0.3 / F : 0.1;
    const double sc = std::min(0.1, std::max(1.0e-3, s_raw));
    bool clamped = s_raw < 1.0e-3;
    e->ew_rel = sc * R;
    e->ew_abs = sc * A;
    if (e->ew_split && clamped && std::isfinite(F)) {
      e->ew_rel = std::max(s_raw * R, std::min(1.0e-3 * R, e->ew_rel_floor));
      clamped = F * (e->ew_rel + e->ew_abs) > R + A;
    }
    if (e->ew_enabled && clamped) e->info_flags |= CWR_INFO_ELEMENTWISE_CLAMPED;
  }

  // one GPU: the zero-coefficient precondition of level t+1 is known from the flow field (check_ghost_levels): stop before
  // anything touches the state, without a device round trip.  Partitioned runs keep going instead -- the violating
  // rank's right-hand side is NaN-poisoned by k_rhs, so every rank leaves the solve together (no rank is left in a collective)
  if (!e->comm && (size_t)(t + 1) < e->bad_level.size() && e->bad_level[(size_t)t + 1])
    return fail(e, CWR_ERR_GHOST_COEFF, "active ghost face with a zero advection/diffusion coefficient at level t+1 "
                "(the reference raises a shape-mismatch ValueError, linalg.py:349-351)");
  if (!e->fused_begin) TRY(prep_step(e, t));
  HIP_TRY(e, hipMemsetAsync(e->d_scal, 0, e->scal_alloc() * sizeof(double), e->stream));   // (+ the counters and the precondition flag behind them)
  // the inner halo layers need x_t for their right-hand sides; the exchange that closed the previous step (for its face
  // fluxes) already delivered it unless the state was touched in between.  Every rank makes the same calls, so every
  // rank takes the same branch.
  // (a caller holding the state pointer may have rewritten the state since: then the exchange is never skipped)
  if (!e->halo_fresh || e->ptr_exported) TRY(exchange_halo(e, e->d_c));
  e->halo_fresh = false;
  // keep x_t and the ghost rows (k_rhs writes both aside): a failed solve restores them
  // (round 5: operator set-up, right-hand side, kept rows and the ghost rows' values of level t+1 in one launch)
  if (e->fused_begin) TRY(launch_begin_step(e, t));
  else TRY(launch_rhs(e, t, e->d_c, e->d_b, true, e->d_keep));
  // the tail writes real rows then: it must not run speculatively (partitioned: on any rank -- the tail is collective)
  const bool has_inputs = e->in_levels.count(t + 1) != 0 || ((size_t)(t + 1) < e->in_any.size() && e->in_any[(size_t)t + 1]);
  SolveStats st;
  int rc_solve = CWR_OK;
  const bool force_bicg = (flags & CWR_STEP_FORCE_BICGSTAB) != 0;
  const bool force_jac = (flags & CWR_STEP_FORCE_JACOBI) != 0;
  bool need_bicg = force_bicg;
  if (!force_bicg) {
    bool handled = false;
    rc_solve = solve_small(e, tol2, max_iter, force_jac, st, handled, need_bicg);
    if (rc_solve != CWR_OK && st.status == CWR_OK) return rc_solve;
    if (!handled) {
      e->spec_t = (!e->profiling && !has_inputs) ? t : -1; e->spec_flags = flags;
      rc_solve = solve_jacobi(e, tol2, max_iter, force_jac, st, need_bicg);
      e->spec_t = -1;
      if (rc_solve != CWR_OK && st.status == CWR_OK) return rc_solve;     // HIP / RCCL failure
    }
  }
  if (need_bicg && st.status == CWR_OK) {
    rc_solve = solve_bicgstab(e, tol2, max_iter, st);
    if (rc_solve != CWR_OK && st.status == CWR_OK) return rc_solve;
  }
  const int status = st.status;
  const int total_it = st.iterations + st.sweeps;
  const double max_rel = st.max_rel;
  if (e->comm && (st.status != CWR_OK || st.iterations > 0 || force_bicg)) {
    // (one GPU: checked before the step began.  Partitioned: the sweeps learn of it with their check, on every rank; this
    // download of the rank's own counters is left for the paths without that check -- BiCGSTAB, failed steps)
    int32_t h_cnt[8];
    TRY(download(e, h_cnt, e->d_counters, (size_t)8));
    if (h_cnt[2] || e->ghost_bad_any) st.status = CWR_ERR_GHOST_COEFF;     // takes precedence over the NaN it caused
  }
  e->ghost_bad_any = false;
  if (e->profiling) { hipStreamSynchronize(e->stream); collect_profile(e); }
  e->profiling = false;
  local.iterations = st.iterations; local.sweeps = st.sweeps; local.restarts = st.restarts; local.operator_launches = st.launches;
  local.max_rel_residual = max_rel; local.status = st.status;
  local.solver = (st.iterations == 0 && !force_bicg) ? 0 : (st.sweeps == 0 ? 1 : 2);
  local.sweep_kernel = st.sweep_kernel;
  local.flags = e->info_flags;
  local.exchanges = e->step_exchanges; local.overlapped = e->step_overlapped; local.checks = e->step_checks;
  local.local_reps = (st.sweep_kernel == 6) ? e->local_reps : 0;
  local.chained = (st.sweep_kernel == 6) ? e->step_chained : 0;
  if (st.status != CWR_OK) {
    e->flux_valid = false; e->halo_fresh = false; e->tail_done = false;   // (a speculative tail may have run)
    // the solver iterated in place: put x_t and the ghost rows back, so that the state is what the step found and the
    // caller may retry (other tolerance, other solver) or read it
    hipMemcpyAsync(e->d_c, e->d_keep, (size_t)e->n_owned * K * sizeof(double), hipMemcpyDeviceToDevice, e->stream);
    if (e->n_ghost > 0)
      hipMemcpyAsync(e->d_c + (size_t)e->n_real * K, e->d_keep + (size_t)e->n_real * K, (size_t)e->n_ghost * K * sizeof(double),
                     hipMemcpyDeviceToDevice, e->stream);
    e->last_sweeps = 0;
    if (info) *info = local;
    switch (st.status) {
      case CWR_ERR_GHOST_COEFF: return fail(e, st.status, "active ghost face with a zero advection/diffusion coefficient at level t+1 "
                                                        "(the reference raises a shape-mismatch ValueError, linalg.py:349-351)");
      case CWR_ERR_NONFINITE: return fail(e, st.status, "non-finite value met in the implicit solve (NaN/Inf in state, flow field or boundary values)");
      default: {
        char buf[256];
        snprintf(buf, sizeof(buf), "implicit solve did not reach tol = %.3e in %d Jacobi sweeps + %d BiCGSTAB iterations "
                 "(max relative residual %.3e)", tol, st.sweeps, st.iterations, max_rel);
        return fail(e, st.status, buf);
      }
    }
  }
  (void)status; (void)total_it;

  if (!e->tail_done) TRY(step_tail(e, t, flags));
  e->tail_done = false;
  if (flags & CWR_STEP_MASS_BALANCE) {
    if (e->n_lines <= 0) return fail(e, CWR_ERR_STATE, "cwr_step: CWR_STEP_MASS_BALANCE without cwr_set_boundary_lines");
    k_line_mass<<<e->n_lines, BLOCK, 0, e->stream>>>(K, e->n_core, e->d_line_ptr, e->d_line_faces, e->d_f1, e->d_f2,
        e->adv_l(t), e->dif_l(t), e->dt[t], e->d_c, e->d_ledger);
    HIP_TRY(e, hipGetLastError());
  }
  // no synchronisation here: convergence is known, and the tail kernels are ordered on the engine's stream before
  // everything a later call does (read-outs synchronise themselves), so the host can already enqueue the next step
  local.exchanges = e->step_exchanges; local.overlapped = e->step_overlapped; local.checks = e->step_checks;   // (incl. the tail's exchange)
  local.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  if (info) *info = local;
  return CWR_OK;
}

int32_t cwr_get_mass_flux(cwr_engine* e, double* adv, double* dif, double* tot) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (!e->flux_valid) return fail(e, CWR_ERR_STATE, "cwr_get_mass_flux: the last step was not taken with CWR_STEP_MASS_FLUX");
  HIP_TRY(e, enter_device(e->dev));
  const size_t cnt = (size_t)e->E * e->Ku;
  if (cnt == 0) return CWR_OK;
  DevTmp<double> tmp;                                   // internal face order -> reference face order, on the device
  TRY(dev_alloc(e, &tmp.p, cnt));
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)cnt, BLOCK), 256 * 16));
  double* outs[3] = {adv, dif, tot};
  const double* srcs[3] = {e->d_fadv, e->d_fdif, e->d_fadv};
  for (int q = 0; q < 3; ++q) {
    if (!outs[q]) continue;
    k_face_rows_out<<<grid, BLOCK, 0, e->stream>>>((int64_t)cnt, e->Ku, e->K, e->d_face_orig, srcs[q], q == 2 ? e->d_fdif : nullptr, tmp.p);
    HIP_TRY(e, hipGetLastError());
    TRY(download(e, outs[q], tmp.p, cnt));
  }
  return CWR_OK;
}

int32_t cwr_get_jacobi_norms(cwr_engine* e, int32_t n_times, double* norms) {
  if (!e || !norms) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_get_jacobi_norms: NULL") : CWR_ERR_BAD_ARG;
  if (n_times != e->T || e->jnorm.size() != (size_t)e->T) return fail(e, CWR_ERR_STATE, "cwr_get_jacobi_norms: n_times must be the number of loaded levels");
  std::copy(e->jnorm.begin(), e->jnorm.end(), norms);
  return CWR_OK;
}

int32_t cwr_set_jacobi_norms(cwr_engine* e, int32_t n_times, const double* norms) {
  if (!e || !norms) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_set_jacobi_norms: NULL") : CWR_ERR_BAD_ARG;
  if (n_times != e->T || e->T <= 0) return fail(e, CWR_ERR_STATE, "cwr_set_jacobi_norms: n_times must be the number of loaded levels");
  e->jnorm.assign(norms, norms + n_times);
  norm_error_factors(e);                             // (the caller's norms are the whole truth then: no row-wise refinement)
  return CWR_OK;
}

int32_t cwr_get_error_factors(cwr_engine* e, int32_t n_times, double* factors) {
  if (!e || !factors) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_get_error_factors: NULL") : CWR_ERR_BAD_ARG;
  if (n_times != e->T || e->err_factor.size() != (size_t)e->T) return fail(e, CWR_ERR_STATE, "cwr_get_error_factors: n_times must be the number of loaded levels");
  std::copy(e->err_factor.begin(), e->err_factor.end(), factors);
  return CWR_OK;
}

// Tiling of the dominant sweep kernel: out = {tiled pass ready, tiles, blocks of its persistent grid, rows per tile}
int32_t cwr_tiling_info(cwr_engine* e, int32_t out[4]) {
  if (!e || !out) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_tiling_info: NULL") : CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  if (e->use_sq && !e->sq_failed && e->K >= e->sq_min_k) TRY(ensure_sq_pattern(e));
  out[0] = e->tcl_ready ? 1 : 0; out[1] = e->tcl_ntiles; out[2] = e->tcl_grid; out[3] = e->tcl_TR;
  return CWR_OK;
}

// Install a tile schedule for the chained in-place passes: sched[it * n_lists + b] = it-th tile of block b, -1 = end of its list
// (n_lists must be the grid of the tiled pass, every tile must appear exactly once).  depth = 0 removes it.
int32_t cwr_set_tile_schedule(cwr_engine* e, int32_t n_lists, int32_t depth, const int32_t* sched) {
  if (!e) return CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (depth <= 0) { e->sched_depth = 0; e->sched_user = false; e->sched_level = -1; return CWR_OK; }
  if (!sched || !e->tcl_ready || n_lists != e->tcl_grid) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_tile_schedule: n_lists must equal the grid of the tiled pass (cwr_tiling_info)");
  if (const char* why = host::validate_schedule(e->tcl_ntiles, n_lists, depth, sched))
    return fail(e, CWR_ERR_BAD_ARG, std::string("cwr_set_tile_schedule: ") + why);
  TRY(install_schedule(e, std::vector<int32_t>(sched, sched + (size_t)n_lists * depth), depth));
  e->sched_user = true;
  // (the column lists of the engine's own schedule do not fit another one: plain lists, every column fetched)
  if (e->d_scols && !e->h_tcl_cols.empty()) TRY(upload(e, e->d_scols, e->h_tcl_cols.data(), e->h_tcl_cols.size()));
  e->sched_nxt.clear();
  return CWR_OK;
}

// The installed schedule (built by the engine at the first tiled step of a level range, or set by the caller): out is
// [depth][n_lists]; info = {depth, n_lists, level it was built for (-1: none / the caller's), schedules built so far}
int32_t cwr_get_tile_schedule(cwr_engine* e, int32_t info[4], int32_t* out, int64_t out_cap) {
  if (!e || !info) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_get_tile_schedule: NULL") : CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  info[0] = e->sched_depth; info[1] = e->sched_depth > 0 ? e->tcl_grid : 0; info[2] = e->sched_user ? -1 : e->sched_level; info[3] = (int32_t)e->n_sched_builds;
  const size_t cnt = (size_t)e->sched_depth * e->tcl_grid;
  if (out && cnt > 0) {
    if ((int64_t)cnt > out_cap) return fail(e, CWR_ERR_BAD_ARG, "cwr_get_tile_schedule: buffer too small");
    TRY(download(e, out, e->d_sched, cnt));
  }
  return CWR_OK;
}

int32_t cwr_time_apply(cwr_engine* e, int32_t t, int32_t variant, int32_t reps, double* avg_us) {
  if (!e || !avg_us || reps < 1) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_time_apply: bad arguments") : CWR_ERR_BAD_ARG;
  TRY(check_level(e, t, true));
  HIP_TRY(e, enter_device(e->dev));
  TRY(prep_step(e, t));
  const size_t nK = (size_t)e->n_real * e->K;
  // operands: the current state and its image, so the numbers are those of a real step
  HIP_TRY(e, hipMemcpyAsync(e->d_p, e->d_c, nK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_s, e->d_c, nK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->d_r0, e->d_c, nK * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
  hipEvent_t e0, e1;
  HIP_TRY(e, hipEventCreate(&e0)); HIP_TRY(e, hipEventCreate(&e1));
  const bool was = e->profiling; e->profiling = false;
  int rc = CWR_OK;
  auto body = [&](int i) -> int {
    if (variant == 0) {                                        // the last step's dominant sweep kernel
      if (e->dominant_mode == 6) return launch_sq_tiled(e, (i & 1) ? e->d_s : e->d_p, e->d_v);
      if (e->dominant_mode == 5)
        return (i & 1) ? launch_apply<5>(e, e->d_s, e->d_v, nullptr, e->d_t, nullptr, nullptr, e->n_sq)
                       : launch_apply<5>(e, e->d_p, e->d_v, nullptr, e->d_t, nullptr, nullptr, e->n_sq);
      return (i & 1) ? launch_apply<4>(e, e->d_s, e->d_v, nullptr, e->d_b, nullptr, nullptr)
                     : launch_apply<4>(e, e->d_p, e->d_v, nullptr, e->d_b, nullptr, nullptr);
    }
    if (variant == 2) {                                        // BiCGSTAB's first product
      return (i & 1) ? launch_apply<1>(e, e->d_s, e->d_t, e->d_r0, nullptr, nullptr, nullptr)
                     : launch_apply<1>(e, e->d_p, e->d_v, e->d_r0, nullptr, nullptr, nullptr);
    }
    const double* xin = (i & 1) ? e->d_s : e->d_p;
    double* yo = (i & 1) ? e->d_t : e->d_v;
    const int g1 = cdiv(e->n_owned, e->R), g2 = cdiv(e->E, e->R);
    const float* adv_t = e->adv_l(t);
    const double* dif_t = e->dif_l(t);
    if (e->VW == 2) {
      k_scatter_diag<2><<<g1, BLOCK, 0, e->stream>>>(e->n_owned, e->K, e->G, e->d_diag, xin, yo);
      k_scatter_faces<2><<<g2, BLOCK, 0, e->stream>>>(e->E, e->n_owned, e->n_real, e->K, e->G, e->d_f1, e->d_f2, adv_t, dif_t, xin, yo);
    } else {
      k_scatter_diag<1><<<g1, BLOCK, 0, e->stream>>>(e->n_owned, e->K, e->G, e->d_diag, xin, yo);
      k_scatter_faces<1><<<g2, BLOCK, 0, e->stream>>>(e->E, e->n_owned, e->n_real, e->K, e->G, e->d_f1, e->d_f2, adv_t, dif_t, xin, yo);
    }
    return hipGetLastError() == hipSuccess ? CWR_OK : fail(e, CWR_ERR_HIP, "scatter variant launch failed");
  };
  for (int i = 0; i < 2 && rc == CWR_OK; ++i) rc = body(i);          // warm-up
  if (rc == CWR_OK) {
    hipEventRecord(e0, e->stream);
    for (int i = 0; i < reps && rc == CWR_OK; ++i) rc = body(i);
    hipEventRecord(e1, e->stream);
    if (hipEventSynchronize(e1) != hipSuccess) rc = fail(e, CWR_ERR_HIP, "cwr_time_apply: event synchronize failed");
    float ms = 0.f;
    if (rc == CWR_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *avg_us = 1000.0 * ms / reps;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  e->profiling = was;
  // the timing loop used the solver's work vectors and accumulators: leave them clean
  hipMemsetAsync(e->d_scal, 0, e->scal_count() * sizeof(double), e->stream);
  hipStreamSynchronize(e->stream);
  return rc;
}

int32_t cwr_profile_read(cwr_engine* e, int64_t* launches, double* total_us) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (launches) *launches = e->prof_launches;
  if (total_us) *total_us = e->prof_us;
  e->prof_launches = 0; e->prof_us = 0.0;
  return CWR_OK;
}

// ------------------------------------------------------------------ output side (8f-4)
int32_t cwr_set_boundary_lines(cwr_engine* e, int32_t n_lines, const int32_t* line_ptr, const int32_t* line_faces) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (n_lines < 1 || !line_ptr) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: need >= 1 line and line_ptr");
  if (e->K > BLOCK) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: K too large");
  if (line_ptr[0] != 0) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: line_ptr[0] must be 0");
  for (int l = 0; l < n_lines; ++l)
    if (line_ptr[l + 1] < line_ptr[l]) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: line_ptr must not decrease");
  const int nf = line_ptr[n_lines];
  if (nf > 0 && !line_faces) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: line_faces is NULL");
  for (int i = 0; i < nf; ++i)
    if (line_faces[i] < 0 || line_faces[i] >= e->E) return fail(e, CWR_ERR_BAD_ARG, "cwr_set_boundary_lines: face id out of range");
  HIP_TRY(e, enter_device(e->dev));
  hipFree(e->d_line_ptr); hipFree(e->d_line_faces); hipFree(e->d_ledger);
  e->d_line_ptr = nullptr; e->d_line_faces = nullptr; e->d_ledger = nullptr; e->n_lines = 0;
  TRY(dev_alloc(e, &e->d_line_ptr, (size_t)n_lines + 1));
  TRY(dev_alloc(e, &e->d_line_faces, (size_t)std::max(nf, 1)));
  TRY(dev_alloc(e, &e->d_ledger, (size_t)n_lines * 3 * e->K));
  TRY(upload(e, e->d_line_ptr, line_ptr, (size_t)n_lines + 1));
  if (nf > 0) {
    std::vector<int32_t> internal((size_t)nf);
    for (int i = 0; i < nf; ++i) internal[(size_t)i] = e->h_face_pos[(size_t)line_faces[i]];
    TRY(upload(e, e->d_line_faces, internal.data(), (size_t)nf));
  }
  e->n_lines = n_lines;
  return cwr_reset_mass_balance(e);
}

int32_t cwr_reset_mass_balance(cwr_engine* e) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (e->n_lines <= 0) return fail(e, CWR_ERR_STATE, "cwr_reset_mass_balance: no boundary lines set");
  HIP_TRY(e, enter_device(e->dev));
  HIP_TRY(e, hipMemsetAsync(e->d_ledger, 0, (size_t)e->n_lines * 3 * e->K * sizeof(double), e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return CWR_OK;
}

int32_t cwr_get_mass_balance(cwr_engine* e, double* ledger) {
  if (!e || !ledger) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_get_mass_balance: NULL") : CWR_ERR_BAD_ARG;
  if (e->n_lines <= 0) return fail(e, CWR_ERR_STATE, "cwr_get_mass_balance: no boundary lines set");
  HIP_TRY(e, enter_device(e->dev));
  TRY(download_cols(e, ledger, e->d_ledger, (size_t)e->n_lines * 3));
  return CWR_OK;
}

int32_t cwr_domain_mass(cwr_engine* e, int32_t t_level, double* out) {
  if (!e || !out) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_domain_mass: NULL") : CWR_ERR_BAD_ARG;
  TRY(check_level(e, t_level, false));
  if (e->K > BLOCK) return fail(e, CWR_ERR_BAD_ARG, "cwr_domain_mass: K too large");
  HIP_TRY(e, enter_device(e->dev));
  const int K = e->K, per = BLOCK / K;
  const int grid = std::max(1, std::min(cdiv(e->n_core, per), 512));
  if (!e->d_mass_out) TRY(dev_alloc(e, &e->d_mass_out, (size_t)513 * (K + 1)));
  k_domain_mass<<<grid, BLOCK, 0, e->stream>>>(e->n_core, K, e->vol_l(t_level), e->d_c, e->d_mass_out);
  k_fold_partials<<<1, BLOCK, 0, e->stream>>>(grid, K + 1, e->d_mass_out, e->d_mass_out + (size_t)512 * (K + 1));
  HIP_TRY(e, hipGetLastError());
  std::vector<double> h((size_t)K + 1);
  TRY(download(e, h.data(), e->d_mass_out + (size_t)512 * (K + 1), (size_t)K + 1));
  for (int k = 0; k < e->Ku; ++k) out[k] = h[(size_t)k];
  out[e->Ku] = h[(size_t)K];                          // (the volume sum sits behind the engine's K columns)
  return CWR_OK;
}

int32_t cwr_output_close(cwr_engine* e) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (!e->out_stream || g_down.load()) return CWR_OK;
  hipSetDevice(e->dev);
  hipStreamSynchronize(e->stream);               // (snapshots written in place run on the engine's stream)
  hipStreamSynchronize(e->out_stream);
  if (getenv("CWR_OUTPUT_DEBUG")) fprintf(stderr, "cwr_output_close: %ld snapshots written in place, %ld through the copy engine\n", e->out_direct_pushes, e->out_copy_pushes);
  for (auto& sl : e->out_slots) { if (sl.h) hipHostFree(sl.h); if (sl.done) hipEventDestroy(sl.done); }
  e->out_slots.clear();
  if (e->out_snap_ready) hipEventDestroy(e->out_snap_ready);
  if (e->out_copy_done) hipEventDestroy(e->out_copy_done);
  e->out_snap_ready = e->out_copy_done = nullptr;
  hipFree(e->d_snap); hipFree(e->d_out_order);
  e->d_snap = nullptr; e->d_out_order = nullptr;
  hipStreamDestroy(e->out_stream);
  e->out_stream = nullptr;
  e->out_copy_pending = false;
  return CWR_OK;
}

int32_t cwr_output_open(cwr_engine* e, int32_t n_slots, int32_t with_flux, int32_t n_out, const int32_t* row_order) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (e->out_stream) return fail(e, CWR_ERR_STATE, "cwr_output_open: already open");
  if (n_slots < 1 || n_slots > 64 || n_out < 1 || n_out > e->n_cells) return fail(e, CWR_ERR_BAD_ARG, "cwr_output_open: bad n_slots / n_out");
  if (row_order)
    for (int i = 0; i < n_out; ++i)
      if (row_order[i] < 0 || row_order[i] >= e->n_cells) return fail(e, CWR_ERR_BAD_ARG, "cwr_output_open: row_order entry out of range");
  HIP_TRY(e, enter_device(e->dev));
  e->out_n = n_out; e->out_flux = with_flux != 0; e->out_next = 0;
  e->out_state_cnt = (size_t)n_out * e->Ku;
  e->out_slot_cnt = e->out_state_cnt + (e->out_flux ? (size_t)3 * e->E * e->Ku : 0);
  HIP_TRY(e, hipStreamCreateWithFlags(&e->out_stream, hipStreamNonBlocking));
  int rc = CWR_OK;
  if (hipEventCreateWithFlags(&e->out_snap_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->out_copy_done, hipEventDisableTiming) != hipSuccess) rc = fail(e, CWR_ERR_HIP, "cwr_output_open: event creation failed");
  if (rc == CWR_OK) rc = dev_alloc(e, &e->d_snap, e->out_slot_cnt);
  if (rc == CWR_OK && row_order) { rc = dev_alloc(e, &e->d_out_order, (size_t)n_out); if (rc == CWR_OK) rc = upload(e, e->d_out_order, row_order, (size_t)n_out); }
  if (rc == CWR_OK) {
    e->out_slots = std::vector<cwr_engine::OutSlot>((size_t)n_slots);
    for (auto& sl : e->out_slots) {
      if (hipHostMalloc(reinterpret_cast<void**>(&sl.h), e->out_slot_cnt * sizeof(double), hipHostMallocDefault) != hipSuccess ||
          hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) { rc = fail(e, CWR_ERR_HIP, "cwr_output_open: pinned host allocation failed"); break; }
    }
  }
  if (rc != CWR_OK) { cwr_output_close(e); return rc; }
  const size_t lds = (size_t)e->Ku * (SNAP_ROWS + 1) * sizeof(double);
  if (lds > 48 * 1024) HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_snapshot_t), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return CWR_OK;
}

namespace {
// snapshot the state (and fluxes) constituent-major and start the copy to the host: into the ring slot, or -- state_dst given --
// straight into the caller's own (page-locked) arrays
int output_push_impl(cwr_engine* e, int32_t* slot, double* state_dst, double* flux_dst) {
  if (!e->out_stream) return fail(e, CWR_ERR_STATE, "cwr_output_push: cwr_output_open first");
  if (e->out_flux && !e->flux_valid) return fail(e, CWR_ERR_STATE, "cwr_output_push: the last step was not taken with CWR_STEP_MASS_FLUX");
  HIP_TRY(e, enter_device(e->dev));
  const int s = e->out_next;
  cwr_engine::OutSlot& sl = e->out_slots[(size_t)s];
  for (int waited = 0; sl.busy.load(std::memory_order_acquire); ++waited) {           // the consumer still holds this slot
    if (waited > 120000) return fail(e, CWR_ERR_STATE, "cwr_output_push: output ring full for 120 s (slot never released)");
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  // Small snapshots into page-locked destinations (the facade's history blocks at the reference's own mesh sizes) are written
  // IN PLACE by the snapshot kernels through the destinations' device aliases: no staging buffer, no copy commands, no second
  // stream -- at 2 943 cells x 12 the copies' submission and completion cost more than the 2 MB they moved
  // (profiles/r05_small_mesh.txt).  Larger ones keep the copy engine (a kernel writing across PCIe holds CUs for the duration).
  double* dst_dev = nullptr; double* dstf_dev = nullptr;
  bool direct = false;
  if (state_dst && e->out_direct_limit > 0 && e->out_slot_cnt * sizeof(double) <= e->out_direct_limit) {
    void* p = nullptr; void* pf = nullptr;
    if (hipHostGetDevicePointer(&p, state_dst, 0) == hipSuccess && p &&
        (!(e->out_flux && flux_dst) || (hipHostGetDevicePointer(&pf, flux_dst, 0) == hipSuccess && pf))) {
      direct = true; dst_dev = static_cast<double*>(p); dstf_dev = static_cast<double*>(pf);
    } else {
      (void)hipGetLastError();                     // pageable destination: the copy path below
    }
  }
  // the device snapshot is rewritten only after the previous copy out of it has finished
  if (!direct && e->out_copy_pending) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->out_copy_done, 0));
  const size_t lds = (size_t)e->Ku * (SNAP_ROWS + 1) * sizeof(double);
  const int grid = std::max(1, std::min(cdiv(e->out_n, SNAP_ROWS), 256 * 8));
  k_snapshot_t<<<grid, BLOCK, lds, e->stream>>>(e->out_n, e->Ku, e->K, e->d_out_order, e->d_c, nullptr, direct ? dst_dev : e->d_snap, nullptr, nullptr);
  if (e->out_flux && (!direct || dstf_dev)) {
    const int gridf = std::max(1, std::min(cdiv(e->E, SNAP_ROWS), 256 * 8));
    const size_t EK = (size_t)e->E * e->Ku;
    double* fo = direct ? dstf_dev : e->d_snap + e->out_state_cnt;
    // (output index = the reference's face id; its row sits at the face's internal position)
    k_snapshot_t<<<gridf, BLOCK, lds, e->stream>>>(e->E, e->Ku, e->K, e->d_face_pos, e->d_fadv, e->d_fdif, fo, fo + EK, fo + 2 * EK);
  }
  HIP_TRY(e, hipGetLastError());
  ++(direct ? e->out_direct_pushes : e->out_copy_pushes);
  if (direct) {
    sl.dst_state = state_dst; sl.dst_flux = flux_dst;
    HIP_TRY(e, hipEventRecord(sl.done, e->stream));
    sl.busy.store(true, std::memory_order_release);
    e->out_next = (s + 1) % (int)e->out_slots.size();
    *slot = s;
    return CWR_OK;
  }
  HIP_TRY(e, hipEventRecord(e->out_snap_ready, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->out_stream, e->out_snap_ready, 0));
  if (state_dst) {
    HIP_TRY(e, hipMemcpyAsync(state_dst, e->d_snap, e->out_state_cnt * sizeof(double), hipMemcpyDeviceToHost, e->out_stream));
    if (e->out_flux && flux_dst)
      HIP_TRY(e, hipMemcpyAsync(flux_dst, e->d_snap + e->out_state_cnt, (e->out_slot_cnt - e->out_state_cnt) * sizeof(double), hipMemcpyDeviceToHost, e->out_stream));
  } else {
    HIP_TRY(e, hipMemcpyAsync(sl.h, e->d_snap, e->out_slot_cnt * sizeof(double), hipMemcpyDeviceToHost, e->out_stream));
  }
  sl.dst_state = state_dst; sl.dst_flux = flux_dst;
  HIP_TRY(e, hipEventRecord(sl.done, e->out_stream));
  HIP_TRY(e, hipEventRecord(e->out_copy_done, e->out_stream));
  e->out_copy_pending = true;
  sl.busy.store(true, std::memory_order_release);
  e->out_next = (s + 1) % (int)e->out_slots.size();
  *slot = s;
  return CWR_OK;
}
}  // namespace

int32_t cwr_output_push(cwr_engine* e, int32_t* slot) {
  if (!e || !slot) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_output_push: NULL") : CWR_ERR_BAD_ARG;
  return output_push_impl(e, slot, nullptr, nullptr);
}

int32_t cwr_output_push_into(cwr_engine* e, double* state_dst, double* flux_dst, int32_t* slot) {
  if (!e || !slot || !state_dst) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_output_push_into: NULL") : CWR_ERR_BAD_ARG;
  if (e->out_flux && !flux_dst) return fail(e, CWR_ERR_BAD_ARG, "cwr_output_push_into: the ring was opened with fluxes: flux_dst needed");
  return output_push_impl(e, slot, state_dst, flux_dst);
}

int32_t cwr_host_register(void* ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return CWR_ERR_BAD_ARG;
  if (hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return CWR_ERR_HIP; }
  return CWR_OK;
}
int32_t cwr_host_unregister(void* ptr) {
  if (!ptr) return CWR_ERR_BAD_ARG;
  if (g_down.load()) return CWR_OK;
  if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return CWR_ERR_HIP; }
  return CWR_OK;
}

int32_t cwr_output_wait(cwr_engine* e, int32_t slot, const double** state, const double** flux) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (!e->out_stream || slot < 0 || slot >= (int)e->out_slots.size()) return fail(e, CWR_ERR_BAD_ARG, "cwr_output_wait: bad slot");
  cwr_engine::OutSlot& sl = e->out_slots[(size_t)slot];
  if (!sl.busy.load(std::memory_order_acquire)) return fail(e, CWR_ERR_STATE, "cwr_output_wait: slot holds no snapshot");
  // (no hipSetDevice: events carry their device; this may run on a consumer thread)
  // (a short spin first: a snapshot of the reference's own mesh sizes lands within tens of microseconds, less than a blocking
  // wait's wake-up)
  bool landed = false;
  for (int spin = 0; spin < 4000 && !landed; ++spin) {
    const hipError_t q = hipEventQuery(sl.done);
    if (q == hipSuccess) landed = true;
    else if (q != hipErrorNotReady) return fail(e, CWR_ERR_HIP, "cwr_output_wait: event query failed");
  }
  (void)hipGetLastError();                       // (hipErrorNotReady of the queries is no error)
  if (!landed && hipEventSynchronize(sl.done) != hipSuccess) return fail(e, CWR_ERR_HIP, "cwr_output_wait: event synchronize failed");
  if (state) *state = sl.dst_state ? sl.dst_state : sl.h;
  if (flux) *flux = !e->out_flux ? nullptr : (sl.dst_state ? sl.dst_flux : sl.h + e->out_state_cnt);
  return CWR_OK;
}

int32_t cwr_output_release(cwr_engine* e, int32_t slot) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (!e->out_stream || slot < 0 || slot >= (int)e->out_slots.size()) return fail(e, CWR_ERR_BAD_ARG, "cwr_output_release: bad slot");
  e->out_slots[(size_t)slot].busy.store(false, std::memory_order_release);
  return CWR_OK;
}

int32_t cwr_synchronize(cwr_engine* e) {
  if (!e) return CWR_ERR_BAD_ARG;
  HIP_TRY(e, enter_device(e->dev));
  if (!e->pending_loads.empty() || !e->pending_bc.empty()) TRY(flush_window_loads(e));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->flow_stream) HIP_TRY(e, hipStreamSynchronize(e->flow_stream));   // (windowed flow field: every enqueued level has arrived)
  return CWR_OK;
}

int32_t cwr_apply_bytes(const cwr_engine* e, int64_t* bytes_read, int64_t* bytes_written) {
  if (!e) return CWR_ERR_BAD_ARG;
  // algorithmic bytes of one launch of the last step's dominant operator kernel: adjacency records (16 B each; of J^2
  // when the double sweep is active), CSR row pointers, diagonal (plain sweep only), the input vector (every real row
  // once), the bhat / c2 / r0 operand; one output row per computed row
  const int64_t K = e->Ku;                             // (the caller's constituents: padded columns are not counted as useful bytes)
  const bool sq = (e->dominant_mode == 5 || e->dominant_mode == 6);
  const int64_t entries = sq ? e->nnz2 : e->nnz;
  const int64_t rows = (e->dominant_mode == 6) ? e->n_tcl : (sq ? e->n_sq : e->n_owned);
  // tiled J^2 pass: 8-B weight + 2-B local index per entry, + the per-tile lists of distinct x rows
  const int64_t extra = (e->dominant_mode == 6) ? 4LL * (int64_t)e->tcl_total_cols - 6LL * e->nnz2 : 0LL;
  if (bytes_read) *bytes_read = 16LL * entries + extra + 4LL * (rows + 1) + (sq ? 0LL : 8LL * rows) +
                                8LL * K * e->n_real + 8LL * K * rows;
  if (bytes_written) *bytes_written = 8LL * K * rows;
  return CWR_OK;
}

int32_t cwr_comm_unique_id(uint8_t id_out[128]) {
  std::string err;
  if (!id_out) return CWR_ERR_BAD_ARG;
  if (!g_rccl.load(err)) return fail(nullptr, CWR_ERR_RCCL, err);
  NcclUniqueId id;
  const int st = g_rccl.GetUniqueId(&id);
  if (st != 0) return fail(nullptr, CWR_ERR_RCCL, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(st));
  std::memcpy(id_out, id.internal, 128);
  return CWR_OK;
}

int32_t cwr_attach_comm(cwr_engine* e, int32_t rank, int32_t world, const uint8_t unique_id[128], int32_t n_core,
                        int32_t exchange_every, int32_t n_peers, const int32_t* peers, const int32_t* send_ptr,
                        const int32_t* send_cells, const int32_t* recv_ptr, const int32_t* recv_cells) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (world < 1 || rank < 0 || rank >= world || !unique_id || n_peers < 0 || n_core < 1 || n_core > e->n_owned ||
      exchange_every < 1 || (n_peers > 0 && (!peers || !send_ptr || !recv_ptr)))
    return fail(e, CWR_ERR_BAD_ARG, "cwr_attach_comm: bad arguments");
  for (int i = 0; i < n_peers; ++i) {
    if (peers[i] < 0 || peers[i] >= world || peers[i] == rank || send_ptr[i + 1] < send_ptr[i] || recv_ptr[i + 1] < recv_ptr[i])
      return fail(e, CWR_ERR_BAD_ARG, "cwr_attach_comm: malformed peer lists");
  }
  const int n_send = n_peers ? send_ptr[n_peers] : 0;
  const int n_recv = n_peers ? recv_ptr[n_peers] : 0;
  // (world == 1 without peers: a STAND-ALONE rank -- the row layout of one rank of a larger partition (core, replayed layers, read-only
  // layer) with nobody to exchange with: the rows outside the core keep what the caller put there.  The launch structure of a rank's
  // step without its exchanges: tools/rank_step_profile.py)
  const bool standalone = world == 1 && n_peers == 0;
  if (n_recv != e->n_real - n_core && !standalone)
    return fail(e, CWR_ERR_BAD_ARG, "cwr_attach_comm: receive lists must cover every real row outside the core exactly once");
  for (int i = 0; i < n_send; ++i)
    if (!send_cells || send_cells[i] < 0 || send_cells[i] >= n_core)
      return fail(e, CWR_ERR_BAD_ARG, "cwr_attach_comm: send cell outside the core block");
  {
    std::vector<char> seen((size_t)e->n_real, 0);
    for (int i = 0; i < n_recv; ++i) {
      if (!recv_cells || recv_cells[i] < n_core || recv_cells[i] >= e->n_real || seen[recv_cells[i]])
        return fail(e, CWR_ERR_BAD_ARG, "cwr_attach_comm: receive cell outside the halo rows or listed twice");
      seen[recv_cells[i]] = 1;
    }
  }
  if (e->comm) return fail(e, CWR_ERR_STATE, "cwr_attach_comm: this engine has a communicator already (its buffers are sized for that world: ADVICE r05)");
  if (e->windowed)
    for (int lvl : e->slot_level)
      if (lvl >= 0) return fail(e, CWR_ERR_STATE, "cwr_attach_comm: levels were loaded into the flow-field window already -- attach the communicator first "
                                                  "(a level's norms are all-reduced where it is loaded)");
  std::string err;
  if (!g_rccl.load(err)) return fail(e, CWR_ERR_RCCL, err);
  HIP_TRY(e, enter_device(e->dev));
  (void)hipGetLastError();             // (no stale error of this thread may reach the communication library's own checks)
  NcclUniqueId id; std::memcpy(id.internal, unique_id, 128);
  NCCL_TRY(e, g_rccl.CommInitRank(&e->comm, world, id, rank));
  e->rank = rank; e->world = world;
  e->n_core = n_core; e->exch_every = exchange_every;
  if (const char* v = getenv("CWR_FORCE_COLLECTIVES")) e->force_coll = atoi(v) != 0;
  e->peers.assign(peers, peers + n_peers);
  e->send_ptr.assign(send_ptr, send_ptr + (n_peers ? n_peers + 1 : 0));
  e->recv_ptr.assign(recv_ptr, recv_ptr + (n_peers ? n_peers + 1 : 0));
  e->n_send = n_send; e->n_recv = n_recv;
  TRY(dev_alloc(e, &e->d_send_cells, (size_t)n_send));
  TRY(dev_alloc(e, &e->d_sendbuf, (size_t)n_send * e->K));
  TRY(dev_alloc(e, &e->d_recv_cells, (size_t)n_recv));
  TRY(dev_alloc(e, &e->d_recvbuf, (size_t)n_recv * e->K));
  TRY(upload(e, e->d_send_cells, send_cells, (size_t)n_send));
  TRY(upload(e, e->d_recv_cells, recv_cells, (size_t)n_recv));
  TRY(dev_alloc(e, &e->d_chkx, (size_t)(2 + 2 * world) * e->K + 1));
  if (e->h_note && !e->h_notex) {                // (the check block's way to the host without a copy: gather_check)
    void* hp = nullptr; void* dp = nullptr;
    const size_t bytes = ((size_t)(2 + 2 * world) * e->K + 1) * sizeof(double);
    if (hipHostMalloc(&hp, bytes, hipHostMallocMapped) == hipSuccess && hp && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
      std::memset(hp, 0, bytes);
      e->h_notex = static_cast<double*>(hp); e->d_notex_view = static_cast<double*>(dp);
    } else {
      if (hp) (void)hipHostFree(hp);
      (void)hipGetLastError();
    }
  }
  if (const char* v = getenv("CWR_NO_OVERLAP")) e->overlap = atoi(v) == 0;
  if (const char* v = getenv("CWR_TEST_POISON_HALO")) e->test_poison_halo = atoi(v) != 0;
  if (const char* v = getenv("CWR_OVERLAP_RESERVE")) e->overlap_reserve = std::max(0, atoi(v)) / N_XCD * N_XCD;
  {
    // row tiles of the plain sweep whose rows and neighbours are all core rows: no exchange touches what they read or write
    const int TR = e->R * e->U, nt = cdiv(e->n_owned, TR);
    std::vector<int32_t> inner, outer;
    for (int t = 0; t < nt; ++t) {
      const int c0 = t * TR, c1 = std::min(c0 + TR, e->n_owned);
      bool in = c1 <= n_core;
      for (int c = c0; c < c1 && in; ++c)
        for (int j = e->h_ptr[(size_t)c]; j < e->h_ptr[(size_t)c + 1] && in; ++j) in = e->h_nb[(size_t)j] < n_core;
      (in ? inner : outer).push_back(t);
    }
    e->n_apply_inner = (int)inner.size(); e->n_apply_outer = (int)outer.size();
    TRY(dev_alloc(e, &e->d_apply_inner, inner.size()));
    TRY(dev_alloc(e, &e->d_apply_outer, outer.size()));
    TRY(upload(e, e->d_apply_inner, inner.data(), inner.size()));
    TRY(upload(e, e->d_apply_outer, outer.data(), outer.size()));
    // faces whose flux reads no halo row: both cells core rows, or a core cell and a ghost (boundary) cell
    std::vector<int32_t> fin, fout;
    for (int f = 0; f < e->E; ++f) {
      const int P = e->h_f1[(size_t)f], N = e->h_f2[(size_t)f];
      const bool in = P < n_core && (N < n_core || N >= e->n_real);
      (in ? fin : fout).push_back(f);
    }
    e->n_face_inner = (int)fin.size(); e->n_face_outer = (int)fout.size();
    TRY(dev_alloc(e, &e->d_face_inner, fin.size()));
    TRY(dev_alloc(e, &e->d_face_outer, fout.size()));
    TRY(upload(e, e->d_face_inner, fin.data(), fin.size()));
    TRY(upload(e, e->d_face_outer, fout.data(), fout.size()));
  }
  HIP_TRY(e, hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_packed, hipEventDisableTiming));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_halo, hipEventDisableTiming));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_red_in, hipEventDisableTiming));
  HIP_TRY(e, hipEventCreateWithFlags(&e->ev_red_out, hipEventDisableTiming));
  if (const char* v = getenv("CWR_COMM_TWO_STREAMS")) e->one_comm_stream = atoi(v) == 0;
  if (!e->windowed) {                                            // (a windowed field: per level, where it is loaded and where its step runs)
    TRY(sync_jnorms(e));
    TRY(refine_error_factors(e));                                 // (collective: the row-wise bound of the global matrix, see there)
  }
  return sync_input_levels(e);
}

int32_t cwr_comm_selftest(cwr_engine* e, int32_t count, int64_t* overlapped_exchanges) {
  if (!e) return CWR_ERR_BAD_ARG;
  if (overlapped_exchanges) *overlapped_exchanges = e->n_overlapped;
  if (count <= 0) return CWR_OK;                               // (statistics only)
  if (!e->comm) return fail(e, CWR_ERR_STATE, "cwr_comm_selftest: no communicator attached");
  HIP_TRY(e, enter_device(e->dev));
  // a grouped ncclSend / ncclRecv of this rank to ITSELF on the communication stream, bracketed by the two events of
  // the overlapped exchange: the call signatures and the stream / event plumbing of exchange_begin / exchange_finish,
  // executable with a single rank (the one-GPU box cannot host two RCCL ranks)
  DevTmp<double> a, b;
  TRY(dev_alloc(e, &a.p, (size_t)count));
  TRY(dev_alloc(e, &b.p, (size_t)count));
  std::vector<double> h((size_t)count), back((size_t)count, -1.0);
  for (int i = 0; i < count; ++i) h[(size_t)i] = 1.5 * i - 7.0;
  TRY(upload(e, a.p, h.data(), (size_t)count));
  HIP_TRY(e, hipMemsetAsync(b.p, 0, (size_t)count * sizeof(double), e->stream));
  HIP_TRY(e, hipEventRecord(e->ev_packed, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->comm_stream, e->ev_packed, 0));
  NCCL_TRY(e, g_rccl.GroupStart());
  NCCL_TRY(e, g_rccl.Send(a.p, (size_t)count, NCCL_FLOAT64, e->rank, e->comm, e->comm_stream));
  NCCL_TRY(e, g_rccl.Recv(b.p, (size_t)count, NCCL_FLOAT64, e->rank, e->comm, e->comm_stream));
  NCCL_TRY(e, g_rccl.GroupEnd());
  HIP_TRY(e, hipEventRecord(e->ev_halo, e->comm_stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_halo, 0));
  TRY(download(e, back.data(), b.p, (size_t)count));
  for (int i = 0; i < count; ++i)
    if (back[(size_t)i] != h[(size_t)i]) return fail(e, CWR_ERR_RCCL, "cwr_comm_selftest: self send/recv returned different data");
  return CWR_OK;
}

}  // extern "C"
