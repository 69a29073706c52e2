// cwr_engine_abi_comm.hpp -- part of cwr_engine.hip (included there, in this order; not a stand-alone header): C ABI: RCCL communicator of a partitioned engine.
#pragma once
extern "C" {
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
