// mock_rccl.cpp -- TEST INFRASTRUCTURE ONLY.
// A stand-in for the handful of RCCL entry points the engine resolves with dlopen (csrc/cwr_engine.hip, struct Rccl),
// so that the partitioned solver loop can be run with several ranks on ONE GPU (real RCCL refuses two ranks on one
// device: "Duplicate GPU detected").  Ranks are processes; data moves device -> POSIX shared memory -> device.
//
// Two modes:
//  * STREAM-ASYNCHRONOUS (default, round 3): like RCCL, every call only ENQUEUES work on the caller's stream and returns.
//    The shared segment is page-locked and mapped into the device address space (hipHostRegister); a send is
//        hipStreamWaitValue64(mailbox free)  ->  hipMemcpyAsync(device -> mailbox)  ->  hipStreamWriteValue64(published)
//    a receive is
//        hipStreamWaitValue64(published)     ->  hipMemcpyAsync(mailbox -> device)  ->  hipStreamWriteValue64(consumed)
//    and an all-reduce is wait(all slots free) -> copy to the own slot -> write(published) -> wait(all published) -> a
//    one-block kernel that folds the slots in rank order -> write(done).  Nothing drains a stream and no host thread
//    touches the data, so a missing hipStreamWaitEvent / a race between the engine's two streams (exchange_begin /
//    exchange_finish, k_unpack_rows on the communication stream beside the interior tiles) shows up as wrong halo rows
//    instead of being hidden by a host-side hipStreamSynchronize.  A watchdog thread releases every wait after
//    CWR_MOCK_TIMEOUT_S (default 60) seconds without progress, so a dead peer cannot hang the GPU.
//  * HOST-SYNCHRONOUS (CWR_MOCK_ASYNC=0, or when the device cannot wait on memory values): round 1's form -- the stream is
//    drained before data is read and copies complete before returning.
// Semantics kept in both: stream ordering, grouped send/recv, in-place all-reduce summed in rank order (identical result on
// every rank).
// Build: hipcc -O2 --offload-arch=gfx950 -fPIC -shared tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmock_rccl.so -lrt
#include <cmath>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int MAX_RANKS = 8;
constexpr size_t MAILBOX_BYTES = 4u << 20;      // per (src, dst) pair
constexpr size_t REDUCE_BYTES = 1u << 16;       // per rank
struct Shared {
  std::atomic<int> arrived;                       // init barrier
  std::atomic<int> left;                          // ranks that destroyed their communicator
  std::atomic<uint64_t> send_seq[MAX_RANKS][MAX_RANKS];   // [src][dst] messages published
  std::atomic<uint64_t> recv_seq[MAX_RANKS][MAX_RANKS];   // [src][dst] messages consumed
  std::atomic<uint64_t> send_bytes[MAX_RANKS][MAX_RANKS];
  std::atomic<uint64_t> red_seq[MAX_RANKS];       // all-reduce rounds each rank has published
  std::atomic<uint64_t> red_done[MAX_RANKS];      // all-reduce rounds each rank has finished reading
  unsigned char reduce[MAX_RANKS][REDUCE_BYTES];
  unsigned char mailbox[MAX_RANKS][MAX_RANKS][MAILBOX_BYTES];
};
static_assert(sizeof(std::atomic<uint64_t>) == 8, "flags are waited on as plain 64-bit words");
struct Comm {
  int rank, world;
  Shared* sh;
  std::string name;
  uint64_t red_round = 0;
  // asynchronous mode
  bool async = false;
  Shared* dsh = nullptr;                          // device view of the registered segment
  uint64_t sent_n[MAX_RANKS] = {}, recv_n[MAX_RANKS] = {};   // messages ENQUEUED per peer (host-side sequence numbers)
  std::atomic<bool> failed{false}, stop{false};
  std::atomic<uint64_t> enqueued{0};              // operations enqueued so far (watchdog)
  std::thread watchdog;
  hipStream_t wd_stream = nullptr;                // the watchdog releases waits from the GPU side too
};
struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

double timeout_s() { const char* v = getenv("CWR_MOCK_TIMEOUT_S"); return v ? atof(v) : 60.0; }

template <typename F> bool spin_until(F cond, double limit = -1.0) {
  if (limit < 0) limit = timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  while (!cond()) {
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
  }
  return true;
}
size_t dtype_size(int dt) { return dt == 8 ? 8 : (dt == 7 ? 4 : (dt <= 1 ? 1 : 4)); }

// ---------------------------------------------------------------- host-synchronous mode
int run_ops_sync(std::vector<Op>& ops) {
  // publish every send first, then complete the receives: no ordering between peers can deadlock
  for (Op& o : ops) if (o.send) {
    Comm* c = o.comm;
    if (o.bytes > MAILBOX_BYTES) return 5;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    auto& sent = c->sh->send_seq[c->rank][o.peer];
    auto& taken = c->sh->recv_seq[c->rank][o.peer];
    if (!spin_until([&] { return taken.load() == sent.load(); })) return 6;   // previous message consumed
    if (hipMemcpy(c->sh->mailbox[c->rank][o.peer], o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    c->sh->send_bytes[c->rank][o.peer].store(o.bytes);
    sent.fetch_add(1);
  }
  for (Op& o : ops) if (!o.send) {
    Comm* c = o.comm;
    auto& sent = c->sh->send_seq[o.peer][c->rank];
    auto& taken = c->sh->recv_seq[o.peer][c->rank];
    if (!spin_until([&] { return sent.load() > taken.load(); })) return 6;
    if (c->sh->send_bytes[o.peer][c->rank].load() != o.bytes) return 4;      // count mismatch between the two sides
    if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    if (hipMemcpy(o.ptr, c->sh->mailbox[o.peer][c->rank], o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    taken.fetch_add(1);
  }
  return 0;
}

// ---------------------------------------------------------------- stream-asynchronous mode
// (device addresses of the flags / buffers of the registered segment: same offsets as on the host)
template <typename T> T* dev(Comm* c, T* host) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(c->dsh) + (reinterpret_cast<char*>(host) - reinterpret_cast<char*>(c->sh)));
}
#define MOCK_HIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return 1; } } while (0)

int wait_ge(Comm* c, hipStream_t s, std::atomic<uint64_t>* flag, uint64_t v) {
  MOCK_HIP(hipStreamWaitValue64(s, dev(c, flag), v, hipStreamWaitValueGte, 0xffffffffffffffffull));
  return 0;
}
int write_val(Comm* c, hipStream_t s, std::atomic<uint64_t>* flag, uint64_t v) {
  MOCK_HIP(hipStreamWriteValue64(s, dev(c, flag), v, 0));
  return 0;
}

int run_ops_async(std::vector<Op>& ops) {
  // every send first, then the receives (as in the synchronous mode); all of it only ENQUEUED, in stream order
  for (Op& o : ops) if (o.send) {
    Comm* c = o.comm;
    if (c->failed.load()) return 6;
    if (o.bytes > MAILBOX_BYTES) return 5;
    const uint64_t n = ++c->sent_n[o.peer];                                  // this is the n-th message to that peer
    if (wait_ge(c, o.stream, &c->sh->recv_seq[c->rank][o.peer], n - 1)) return 1;      // the previous one has been consumed
    MOCK_HIP(hipMemcpyAsync(dev(c, &c->sh->mailbox[c->rank][o.peer][0]), o.ptr, o.bytes, hipMemcpyDefault, o.stream));
    if (write_val(c, o.stream, &c->sh->send_bytes[c->rank][o.peer], o.bytes)) return 1;
    if (write_val(c, o.stream, &c->sh->send_seq[c->rank][o.peer], n)) return 1;
    c->enqueued.fetch_add(1);
  }
  for (Op& o : ops) if (!o.send) {
    Comm* c = o.comm;
    if (c->failed.load()) return 6;
    const uint64_t n = ++c->recv_n[o.peer];
    if (wait_ge(c, o.stream, &c->sh->send_seq[o.peer][c->rank], n)) return 1;          // the n-th message is there
    MOCK_HIP(hipMemcpyAsync(o.ptr, dev(c, &c->sh->mailbox[o.peer][c->rank][0]), o.bytes, hipMemcpyDefault, o.stream));
    if (write_val(c, o.stream, &c->sh->recv_seq[o.peer][c->rank], n)) return 1;
    c->enqueued.fetch_add(1);
  }
  return 0;
}

__global__ void k_mock_reduce(const unsigned char* slots, size_t slot_bytes, int world, size_t count, int op, double* out) {
  for (size_t i = threadIdx.x; i < count; i += blockDim.x) {
    double acc = (op == 2) ? -INFINITY : 0.0;
    for (int r = 0; r < world; ++r) {                                        // rank order: the same sum on every rank
      const double v = reinterpret_cast<const volatile double*>(slots + (size_t)r * slot_bytes)[i];
      acc = (op == 2) ? fmax(acc, v) : acc + v;
    }
    out[i] = acc;
  }
}

int allreduce_async(Comm* c, const void* sendbuf, void* recvbuf, size_t count, int op, hipStream_t s) {
  if (c->failed.load()) return 6;
  const uint64_t round = ++c->red_round;
  for (int r = 0; r < c->world; ++r)                                         // every rank has read the previous round
    if (wait_ge(c, s, &c->sh->red_done[r], round - 1)) return 1;
  MOCK_HIP(hipMemcpyAsync(dev(c, &c->sh->reduce[c->rank][0]), sendbuf, count * 8, hipMemcpyDefault, s));
  if (write_val(c, s, &c->sh->red_seq[c->rank], round)) return 1;
  for (int r = 0; r < c->world; ++r)
    if (wait_ge(c, s, &c->sh->red_seq[r], round)) return 1;
  k_mock_reduce<<<1, 256, 0, s>>>(dev(c, &c->sh->reduce[0][0]), REDUCE_BYTES, c->world, count, op, static_cast<double*>(recvbuf));
  MOCK_HIP(hipGetLastError());
  if (write_val(c, s, &c->sh->red_done[c->rank], round)) return 1;
  c->enqueued.fetch_add(1);
  return 0;
}

// Releases every wait when nothing this rank waits for has moved for the time limit although operations are outstanding
// (a peer died or took another path): the flags are set to "infinitely many messages", the streams drain with garbage, and
// every later call fails.  Without it a lost peer would leave the GPU's command processor polling forever.
void watchdog_main(Comm* c) {
  auto snapshot = [&] {
    uint64_t h = c->enqueued.load();
    for (int r = 0; r < c->world; ++r) {
      h = h * 1315423911u + c->sh->send_seq[r][c->rank].load() + 3 * c->sh->recv_seq[c->rank][r].load() +
          5 * c->sh->red_seq[r].load() + 7 * c->sh->red_done[r].load() + 11 * c->sh->send_seq[c->rank][r].load() +
          13 * c->sh->recv_seq[r][c->rank].load();
    }
    return h;
  };
  auto pending = [&] {
    for (int r = 0; r < c->world; ++r) {
      if (c->sh->send_seq[c->rank][r].load() < c->sent_n[r]) return true;      // (sent_n / recv_n are written by the API thread:
      if (c->sh->recv_seq[r][c->rank].load() < c->recv_n[r]) return true;      //  a stale read only delays the verdict)
    }
    return c->sh->red_done[c->rank].load() < c->red_round;
  };
  uint64_t last = snapshot();
  auto t_last = std::chrono::steady_clock::now();
  const double limit = timeout_s();
  while (!c->stop.load()) {
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    const uint64_t now = snapshot();
    if (now != last || !pending()) { last = now; t_last = std::chrono::steady_clock::now(); continue; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_last).count() < limit) continue;
    c->failed.store(true);
    std::fprintf(stderr, "[mock_rccl] rank %d: no progress for %.0f s with operations outstanding -- releasing every wait\n", c->rank, limit);
    {
      // what this rank still waits for (a deadlock report: which pair, which kind)
      std::string msg = "[mock_rccl] rank " + std::to_string(c->rank) + " state: all-reduce round " + std::to_string(c->red_round) + ", published/done by rank:";
      for (int r = 0; r < c->world; ++r) msg += " " + std::to_string(c->sh->red_seq[r].load()) + "/" + std::to_string(c->sh->red_done[r].load());
      for (int r = 0; r < c->world; ++r) {
        if (c->sent_n[r] == 0 && c->recv_n[r] == 0) continue;
        msg += "; peer " + std::to_string(r) + ": sent " + std::to_string(c->sh->send_seq[c->rank][r].load()) + "/" + std::to_string(c->sent_n[r]) +
               " (consumed " + std::to_string(c->sh->recv_seq[c->rank][r].load()) + "), received " + std::to_string(c->sh->recv_seq[r][c->rank].load()) + "/" +
               std::to_string(c->recv_n[r]) + " (published " + std::to_string(c->sh->send_seq[r][c->rank].load()) + ")";
      }
      std::fprintf(stderr, "%s\n", msg.c_str());
    }
    const uint64_t big = 0x7fffffffffffffffull;
    for (int r = 0; r < c->world; ++r) {
      std::atomic<uint64_t>* flags[4] = {&c->sh->send_seq[r][c->rank], &c->sh->recv_seq[c->rank][r], &c->sh->red_seq[r], &c->sh->red_done[r]};
      for (auto* f : flags) {
        f->store(big);
        if (c->wd_stream) (void)hipStreamWriteValue64(c->wd_stream, dev(c, f), big, 0);   // (and through the GPU's own path)
      }
    }
    return;
  }
}

int run_ops(std::vector<Op>& ops) {
  if (ops.empty()) return 0;
  return ops[0].comm->async ? run_ops_async(ops) : run_ops_sync(ops);
}
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, 128);
  std::snprintf(id->internal, 128, "/cwr_mock_rccl_%d_%ld", (int)getpid(),
                (long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}

// 1 = this communicator runs in the stream-asynchronous mode (tests assert it, so that a silent fall-back to the synchronous
// mode cannot pass for coverage of the stream / event plumbing)
int mockRcclIsAsync(void* comm) { return comm && static_cast<Comm*>(comm)->async ? 1 : 0; }
int mockRcclLastCommAsync();
static std::atomic<int> g_last_async{-1};
int mockRcclLastCommAsync() { return g_last_async.load(); }

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks > MAX_RANKS) return 4;
  const std::string name(id.internal);
  int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;
  if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  Comm* c = new Comm();
  c->rank = rank; c->world = nranks; c->sh = static_cast<Shared*>(p); c->name = name;
  const char* mode = getenv("CWR_MOCK_ASYNC");
  if (!mode || atoi(mode) != 0) {
    // page-lock the segment and map it into the device address space; the device must be able to wait on memory values
    int dev_id = 0, can_wait = 0;
    void* dptr = nullptr;
    if (hipGetDevice(&dev_id) == hipSuccess &&
        hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev_id) == hipSuccess && can_wait &&
        hipHostRegister(p, sizeof(Shared), hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) {
      if (hipHostGetDevicePointer(&dptr, p, 0) == hipSuccess && dptr) { c->async = true; c->dsh = static_cast<Shared*>(dptr); }
      else (void)hipHostUnregister(p);
    }
    if (c->async) {
      // self-test before anything depends on it: a stream waits for a value that only the HOST writes into the segment (what a
      // peer process does); if the wait does not see it within 2 s it is released through the GPU's own write path and the
      // communicator falls back to the synchronous mode
      hipStream_t ts = nullptr;
      bool ok = hipStreamCreateWithFlags(&ts, hipStreamNonBlocking) == hipSuccess &&
                hipStreamCreateWithFlags(&c->wd_stream, hipStreamNonBlocking) == hipSuccess;
      std::atomic<uint64_t>* flag = &c->sh->red_seq[rank];                   // (zero so far; reset below)
      if (ok) ok = hipStreamWaitValue64(ts, dev(c, flag), 7, hipStreamWaitValueGte, 0xffffffffffffffffull) == hipSuccess;
      if (ok) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        flag->store(7);
        ok = spin_until([&] { return hipStreamQuery(ts) == hipSuccess; }, 2.0);
        if (!ok) {
          (void)hipStreamWriteValue64(c->wd_stream, dev(c, flag), 7, 0);
          (void)hipStreamSynchronize(c->wd_stream);
          (void)spin_until([&] { return hipStreamQuery(ts) == hipSuccess; }, 5.0);
        }
        flag->store(0);
      }
      if (ts) (void)hipStreamDestroy(ts);
      if (!ok) { c->async = false; (void)hipHostUnregister(p); if (c->wd_stream) { (void)hipStreamDestroy(c->wd_stream); c->wd_stream = nullptr; } }
    }
    (void)hipGetLastError();
    if (!c->async && mode && atoi(mode) > 1) { munmap(p, sizeof(Shared)); delete c; return 3; }   // CWR_MOCK_ASYNC=2: asynchronous or fail
  }
  g_last_async.store(c->async ? 1 : 0);
  c->sh->arrived.fetch_add(1);
  if (!spin_until([&] { return c->sh->arrived.load() >= nranks; })) return 6;
  if (c->async) c->watchdog = std::thread(watchdog_main, c);
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->async) {
    (void)hipDeviceSynchronize();                                              // nothing of ours may still be enqueued
    c->stop.store(true);
    if (c->watchdog.joinable()) c->watchdog.join();
    if (c->wd_stream) (void)hipStreamDestroy(c->wd_stream);
    (void)hipHostUnregister(c->sh);
  }
  // the LAST rank to leave unlinks the segment (a peer may still be reading a mailbox when rank 0 is done)
  if (c->sh->left.fetch_add(1) + 1 >= c->world) shm_unlink(c->name.c_str());
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}

int ncclGroupStart() { ++g_group_depth; return 0; }
int ncclGroupEnd() {
  if (--g_group_depth > 0) return 0;
  const int rc = run_ops(g_ops);
  g_ops.clear();
  return rc;
}
int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) {
  g_ops.push_back(Op{true, const_cast<void*>(buf), count * dtype_size(dt), peer, static_cast<Comm*>(comm), s});
  if (g_group_depth == 0) { const int rc = run_ops(g_ops); g_ops.clear(); return rc; }
  return 0;
}
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) {
  g_ops.push_back(Op{false, buf, count * dtype_size(dt), peer, static_cast<Comm*>(comm), s});
  if (g_group_depth == 0) { const int rc = run_ops(g_ops); g_ops.clear(); return rc; }
  return 0;
}

int ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, int dt, int op, void* comm, hipStream_t s) {
  Comm* c = static_cast<Comm*>(comm);
  if (dt != 8 || (op != 0 && op != 2) || count * 8 > REDUCE_BYTES) return 4;   // float64 sum (0) and max (2) only
  if (c->async) return allreduce_async(c, sendbuf, recvbuf, count, op, s);
  if (hipStreamSynchronize(s) != hipSuccess) return 1;
  const uint64_t round = ++c->red_round;
  // wait until every rank has finished reading the previous round before overwriting our slot
  for (int r = 0; r < c->world; ++r)
    if (!spin_until([&] { return c->sh->red_done[r].load() >= round - 1; })) return 6;
  if (hipMemcpy(c->sh->reduce[c->rank], sendbuf, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  c->sh->red_seq[c->rank].store(round);
  for (int r = 0; r < c->world; ++r)
    if (!spin_until([&] { return c->sh->red_seq[r].load() >= round; })) return 6;
  std::vector<double> acc(count, op == 2 ? -INFINITY : 0.0);
  for (int r = 0; r < c->world; ++r) {
    const double* src = reinterpret_cast<const double*>(c->sh->reduce[r]);
    for (size_t i = 0; i < count; ++i) acc[i] = (op == 2) ? std::fmax(acc[i], src[i]) : acc[i] + src[i];
  }
  c->sh->red_done[c->rank].store(round);
  if (hipMemcpy(recvbuf, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "ok";
    case 1: return "mock: HIP error";
    case 2: return "mock: shared memory error";
    case 3: return "mock: the stream-asynchronous mode was demanded (CWR_MOCK_ASYNC=2) and is not available on this device";
    case 4: return "mock: invalid argument / count mismatch";
    case 5: return "mock: message larger than the mailbox";
    case 6: return "mock: timed out waiting for a peer";
    default: return "mock: error";
  }
}
}
