// cwr_engine_abi_step.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): C ABI: coefficients, boundary values, state, reaction hook, apply / rhs / STEP, read-outs, tiling and schedule access, timing.
#pragma once
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

// The communication side of the steps taken with CWR_STEP_PROFILE since the last call (partitioned engines; zeros otherwise):
// out[0..1] exchanges with nothing beside them: count, us on the communication stream (grouped send / receive + unpack, peers' lateness included)
// out[2..3] exchanges that ran beside compute: count, us            out[4..5] all-reduces: count, us
// out[6..7] convergence checks: count, us of host wall time inside them (from "batch enqueued" to "verdict known": GPU time of the batch included)
int32_t cwr_comm_profile_read(cwr_engine* e, double out[8]) {
  if (!e || !out) return e ? fail(e, CWR_ERR_BAD_ARG, "cwr_comm_profile_read: NULL") : CWR_ERR_BAD_ARG;
  if (e->cev_used > 0) {                              // (the exchange that closed the last profiled step was enqueued behind its collection)
    HIP_TRY(e, enter_device(e->dev));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->comm_stream) HIP_TRY(e, hipStreamSynchronize(e->comm_stream));
    collect_comm_profile(e);
  }
  for (int k = 0; k < 3; ++k) { out[2 * k] = (double)e->cprof_n[k]; out[2 * k + 1] = e->cprof_us[k]; e->cprof_n[k] = 0; e->cprof_us[k] = 0.0; }
  out[6] = (double)e->cprof_checks; out[7] = e->cprof_check_wait_us;
  e->cprof_checks = 0; e->cprof_check_wait_us = 0.0;
  return CWR_OK;
}

}  // extern "C"
