// cwr_engine_launch.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): error plumbing, uploads / downloads, kernel launches, halo exchange, all-reduce, convergence checks, BiCGSTAB iteration.
#pragma once
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
bool comm_prof_pair(cwr_engine* e, int kind, hipEvent_t* e0, hipEvent_t* e1);
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
  hipEvent_t p0, p1;
  if (comm_prof_pair(e, beside ? 1 : 0, &p0, &p1)) HIP_TRY(e, hipEventRecord(p0, e->comm_stream));
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
  if (p1) HIP_TRY(e, hipEventRecord(p1, e->comm_stream));
  HIP_TRY(e, hipEventRecord(e->ev_halo, e->comm_stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_halo, 0));
  ++e->step_exchanges;
  if (beside) { ++e->n_overlapped; ++e->step_overlapped; }        // (the caller put work on the engine's stream between the two halves)
  return CWR_OK;
}

// a pair of timing events for the communication profile of a profiled step (nullptr: not profiling, or the pool is used up)
bool comm_prof_pair(cwr_engine* e, int kind, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!e->profiling || !e->comm) return false;
  if (e->cev.empty()) {
    e->cev.resize(512); e->cev_kind.assign(256, 0);
    for (auto& ev : e->cev) if (hipEventCreate(&ev) != hipSuccess) { e->cev.clear(); (void)hipGetLastError(); return false; }
  }
  if (e->cev_used + 2 > e->cev.size()) return false;
  e->cev_kind[e->cev_used / 2] = (char)kind;
  *e0 = e->cev[e->cev_used++]; *e1 = e->cev[e->cev_used++];
  return true;
}

int allreduce(cwr_engine* e, double* p, size_t count) {
  if (!e->comm || (e->world == 1 && !e->force_coll)) return CWR_OK;
  if (e->one_comm_stream && e->comm_stream && e->ev_red_in) {      // on the communication stream, behind the producer, in front of the consumer
    HIP_TRY(e, hipEventRecord(e->ev_red_in, e->stream));
    HIP_TRY(e, hipStreamWaitEvent(e->comm_stream, e->ev_red_in, 0));
    hipEvent_t p0, p1;
    if (comm_prof_pair(e, 2, &p0, &p1)) HIP_TRY(e, hipEventRecord(p0, e->comm_stream));
    NCCL_TRY(e, g_rccl.AllReduce(p, p, count, NCCL_FLOAT64, NCCL_SUM, e->comm, e->comm_stream));
    if (p1) HIP_TRY(e, hipEventRecord(p1, e->comm_stream));
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
  struct WaitClock {                                            // (profiled steps: the host's wall time inside the check, whichever way it is read)
    cwr_engine* e; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~WaitClock() { if (e->profiling) { e->cprof_check_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); ++e->cprof_checks; } }
  } wait_clock{e};
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

}  // namespace
