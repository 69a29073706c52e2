// cwr_engine_abi_create.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): C ABI: version, tile / chain thresholds, create / destroy, flow field (all levels resident).
#pragma once
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
  if (const char* v = getenv("CWR_TCL_ELL")) eng->tcl_ell = atoi(v) != 0;
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
  for (hipEvent_t ev : e->cev) hipEventDestroy(ev);
  void* ptrs[] = {e->d_f1, e->d_f2, e->d_ptr, e->d_ent_edge, e->d_ent_nb, e->d_adv, e->d_vel, e->d_vol, e->d_dif,
                  e->d_bc, e->d_rec, e->d_diag, e->d_c, e->d_r, e->d_r0, e->d_p, e->d_v, e->d_s, e->d_t, e->d_b,
                  e->d_scal, e->d_partial, e->d_fadv, e->d_fdif, e->d_send_cells, e->d_sendbuf, e->d_recv_cells, e->d_recvbuf, e->d_ptr2, e->d_col2, e->d_row2, e->d_rec2, e->d_w, e->d_react, e->d_info, e->d_tcl_ptr, e->d_tcl_cols, e->d_loc2, e->d_w2, e->d_pair_ptr, e->d_slots, e->d_line_ptr, e->d_line_faces, e->d_ledger, e->d_mass_out, e->d_chk, e->d_face_orig, e->d_row_ghost, e->d_keep, e->d_in_rows, e->d_in_vals, e->d_face_pos, e->d_trow, e->d_vptr, e->d_meta, e->d_tile_inner, e->d_tile_outer, e->d_apply_inner, e->d_apply_outer, e->d_face_inner, e->d_face_outer, e->d_chkx, e->d_sq_fast, e->d_sched, e->d_link_ptr, e->d_link_ent, e->d_link_flux, e->d_scols, e->d_scols_io, e->sched_in.d, e->sched_out.d, e->d_small_rows, e->d_small_recs, e->d_small_offs, e->d_small_send_pos, e->d_small_send_cnt, e->d_small_recv_src, e->d_small_recv_pos, e->d_small_recv_cnt, e->d_small_pub, e->d_small_red, e->d_eptr, e->d_ell_pos};
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

}  // extern "C"
