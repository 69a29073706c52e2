// cwr_engine_flow.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): the flow field on the device: levels and their window, zero-coefficient flags, ||J||_inf, the row-wise error factor.
#pragma once
namespace {
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

void collect_comm_profile(cwr_engine* e) {
  for (size_t i = 0; i + 1 < e->cev_used; i += 2) {
    float ms = 0.f;
    const int kind = e->cev_kind[i / 2];
    if (hipEventElapsedTime(&ms, e->cev[i], e->cev[i + 1]) == hipSuccess) { e->cprof_us[kind] += 1000.0 * ms; e->cprof_n[kind] += 1; }
    else (void)hipGetLastError();
  }
  e->cev_used = 0;
}
void collect_profile(cwr_engine* e) {
  for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]) == hipSuccess) { e->prof_us += 1000.0 * ms; e->prof_launches += 1; }
  }
  e->ev_used = 0;
  collect_comm_profile(e);
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

}  // namespace
