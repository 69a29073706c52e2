// cwr_engine_tiling.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): constituent padding, tile shapes, symbolic J^2 and the tiling of the dominant pass, tile links / chains / schedules along the flow.
#pragma once
namespace {
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
#define CWR_TCL_K_ELL(VWv, Q) k_sq_tiled<VWv, TCL_CFG[Q].wrn, TCL_CFG[Q].ut, TCL_CFG[Q].xr, true>
const void* tcl_kernel_csr(int vw, int cfg) { CWR_TCL_PICK(CWR_TCL_K) }
const void* tcl_kernel_ell(int vw, int cfg) { CWR_TCL_PICK(CWR_TCL_K_ELL) }
const void* tcl_kernel(int vw, int cfg, bool ell = false) { return ell ? tcl_kernel_ell(vw, cfg) : tcl_kernel_csr(vw, cfg); }
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
    const int nt = tl.ntiles(), max_cols = tl.max_cols;
    int cap2 = tl.cap2;
    // (round 6) the wave-sliced entry layout (host::build_ell, k_sq_tiled<..., true>): where a row's lanes never straddle two waves (G a power
    // of two: K = 1, 2, 4, 8, 16, 32) and the tiles are plain rows.  A first build gives the padded entry count per tile, which the kernel
    // configuration is chosen by; the layout is then rebuilt with that configuration's slices per tile.
    host::EllLayout ell;
    bool use_ell = false;
    {
      const int Gt = want4 ? e->K / 4 : e->K / e->VW;
      if (e->tcl_ell && !split && Gt >= 1 && (Gt & (Gt - 1)) == 0 && 64 % Gt == 0) {
        const int rpw = 64 / Gt, Rt = BLOCK / Gt;
        if (tr <= 4 * Rt && host::build_ell(tl, ptr2, e->K, rpw, 4 * Rt, ell)) { use_ell = true; cap2 = ell.cap; }
      }
    }
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
      if (use_ell) {                                       // the layout again, with the slices per tile of the chosen configuration
        const int Gt = e->K / e->tcl_vw, rpw = 64 / Gt, Rt = BLOCK / Gt;
        if (!host::build_ell(tl, ptr2, e->K, rpw, TCL_CFG[e->tcl_cfg].ut * Rt, ell) || ell.cap != cap2) use_ell = false;
      }
      if (e->tcl_ell && !use_ell && getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] tiled J^2: the sliced entry layout does not apply here (K = %d); entries in CSR order\n", e->K);
      if (!use_ell) cap2 = tl.cap2;
      e->tcl_use_ell = use_ell;
      const void* fn6 = tcl_kernel(e->tcl_vw, e->tcl_cfg, use_ell);
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
      if (use_ell) {                                       // per tile: its slice offsets (what the row pointers are to the CSR form)
        TRY(dev_alloc(e, &e->d_meta, ell.sl.size()));
        TRY(upload(e, e->d_meta, ell.sl.data(), ell.sl.size()));
        TRY(dev_alloc(e, &e->d_eptr, ell.eptr.size()));
        TRY(upload(e, e->d_eptr, ell.eptr.data(), ell.eptr.size()));
        TRY(dev_alloc(e, &e->d_ell_pos, ell.pos.size()));
        TRY(upload(e, e->d_ell_pos, ell.pos.data(), ell.pos.size()));
        e->tcl_entries = (int64_t)ell.total();
      } else {
        const std::vector<int32_t> meta = host::tile_meta(n_t, ptr2, tl);
        TRY(dev_alloc(e, &e->d_meta, meta.size()));
        TRY(upload(e, e->d_meta, meta.data(), meta.size()));
        e->tcl_entries = (int64_t)e->nnz2;
      }
      TRY(upload(e, e->d_trow, trow.data(), (size_t)nt + 1));
      TRY(upload(e, e->d_vptr, vptr.data(), (size_t)nt + 1));
      TRY(dev_alloc(e, &e->d_tcl_cols, tcols.size()));
      TRY(dev_alloc(e, &e->d_loc2, (size_t)e->tcl_entries));
      TRY(dev_alloc(e, &e->d_w2, (size_t)e->tcl_entries));
      TRY(upload(e, e->d_tcl_ptr, tptr.data(), (size_t)nt + 1));
      TRY(upload(e, e->d_tcl_cols, tcols.data(), tcols.size()));
      if (use_ell) {
        TRY(upload(e, e->d_loc2, ell.loc.data(), ell.total()));
        HIP_TRY(e, hipMemsetAsync(e->d_w2, 0, ell.total() * sizeof(double), e->stream));     // (the padding entries: weight 0, never written again)
        if (getenv("CWR_VERBOSE")) fprintf(stderr, "[cwr] tiled J^2: wave-sliced entries: %lld for %d CSR entries (+%.1f %% padding), <= %d per tile (CSR: %d)\n",
                                           (long long)ell.total(), e->nnz2, 100.0 * ((double)ell.total() / std::max(1, ptr2[(size_t)n_t]) - 1.0), ell.cap, tl.cap2);
      } else TRY(upload(e, e->d_loc2, loc2.data(), (size_t)e->nnz2));
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
}  // namespace
