// cwr_engine_abi_window.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): C ABI: windowed flow-field residency (ring of levels) and the loads on the flow stream.
#pragma once
extern "C" {
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
    if (e->ev_evict) {                                // (rows being REPLACED -- a level loaded again -- may still be read by what the engine's stream holds)
      HIP_TRY(e, hipEventRecord(e->ev_evict, e->stream));
      HIP_TRY(e, hipStreamWaitEvent(e->flow_stream, e->ev_evict, 0));
    }
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

