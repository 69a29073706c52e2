// mock_rccl.cpp -- TEST INFRASTRUCTURE ONLY.
// A stand-in for the handful of RCCL entry points the engine resolves with dlopen (csrc/cwr_engine.hip, struct Rccl),
// so that the partitioned solver loop can be run with several ranks on ONE GPU (real RCCL refuses two ranks on one
// device: "Duplicate GPU detected").  Ranks are processes; data moves device -> POSIX shared memory -> device.
// Semantics kept: stream ordering (the stream is drained before data is read, copies complete before returning),
// grouped send/recv, in-place all-reduce summed in rank order (identical result on every rank).
// Build: hipcc -O2 -fPIC -shared tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmock_rccl.so -lrt
#include <cmath>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
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
  std::atomic<uint64_t> send_seq[MAX_RANKS][MAX_RANKS];   // [src][dst] messages published
  std::atomic<uint64_t> recv_seq[MAX_RANKS][MAX_RANKS];   // [src][dst] messages consumed
  std::atomic<uint64_t> send_bytes[MAX_RANKS][MAX_RANKS];
  std::atomic<uint64_t> red_seq[MAX_RANKS];       // all-reduce rounds each rank has published
  std::atomic<uint64_t> red_done[MAX_RANKS];      // all-reduce rounds each rank has finished reading
  unsigned char reduce[MAX_RANKS][REDUCE_BYTES];
  unsigned char mailbox[MAX_RANKS][MAX_RANKS][MAILBOX_BYTES];
};
struct Comm {
  int rank, world;
  Shared* sh;
  std::string name;
  uint64_t red_round = 0;
};
struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

template <typename F> bool spin_until(F cond, double timeout_s = 60.0) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!cond()) {
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
  }
  return true;
}
size_t dtype_size(int dt) { return dt == 8 ? 8 : (dt == 7 ? 4 : (dt <= 1 ? 1 : 4)); }

int run_ops(std::vector<Op>& ops) {
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
}  // namespace

extern "C" {
struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, 128);
  std::snprintf(id->internal, 128, "/cwr_mock_rccl_%d_%ld", (int)getpid(),
                (long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks > MAX_RANKS) return 4;
  const std::string name(id.internal);
  int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;
  if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  Comm* c = new Comm{rank, nranks, static_cast<Shared*>(p), name};
  c->sh->arrived.fetch_add(1);
  if (!spin_until([&] { return c->sh->arrived.load() >= nranks; })) return 6;
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->rank == 0) shm_unlink(c->name.c_str());
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
    case 4: return "mock: invalid argument / count mismatch";
    case 5: return "mock: message larger than the mailbox";
    case 6: return "mock: timed out waiting for a peer";
    default: return "mock: error";
  }
}
}
