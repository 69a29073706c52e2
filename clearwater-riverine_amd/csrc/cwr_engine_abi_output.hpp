// cwr_engine_abi_output.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): C ABI: output side (8f-4): boundary-line ledger, domain mass, pinned output ring, host registration; synchronize, byte counts.
#pragma once
extern "C" {
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

}  // extern "C"
