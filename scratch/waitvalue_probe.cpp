// probe: can a stream wait on / write to a 64-bit word of page-locked POSIX shared memory?  (for tests/mock_rccl)
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); printf("%-70s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); if (e_ != hipSuccess) return 1; } while (0)
int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 64) << 20;
  int can = -1;
  CK(hipSetDevice(0));
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("can_wait=%d\n", can); fflush(stdout);
  int fd = shm_open("/cwr_probe", O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, bytes) != 0) { printf("shm failed\n"); return 1; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd); shm_unlink("/cwr_probe");
  printf("mmap %p\n", p); fflush(stdout);
  CK(hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  void* d = nullptr;
  CK(hipHostGetDevicePointer(&d, p, 0));
  printf("device view %p\n", d); fflush(stdout);
  auto* flag = reinterpret_cast<std::atomic<uint64_t>*>(p);
  auto* dflag = reinterpret_cast<uint64_t*>(d);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  // 1. GPU write visible to host
  CK(hipStreamWriteValue64(s1, dflag + 1, 42, 0));
  CK(hipStreamSynchronize(s1));
  printf("host reads %llu (expect 42)\n", (unsigned long long)flag[1].load()); fflush(stdout);
  // 2. wait satisfied by a host store
  CK(hipStreamWaitValue64(s1, dflag, 7, hipStreamWaitValueGte, 0xffffffffffffffffull));
  std::this_thread::sleep_for(std::chrono::milliseconds(5));
  printf("query before store: %s\n", hipGetErrorString(hipStreamQuery(s1))); fflush(stdout);
  flag[0].store(7);
  auto t0 = std::chrono::steady_clock::now();
  while (hipStreamQuery(s1) != hipSuccess && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) usleep(100);
  printf("after host store: %s (%.3f ms)\n", hipGetErrorString(hipStreamQuery(s1)), 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); fflush(stdout);
  if (hipStreamQuery(s1) != hipSuccess) { CK(hipStreamWriteValue64(s2, dflag, 7, 0)); CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s1)); printf("released by a GPU write\n"); }
  // 3. copies to / from the mapped segment with hipMemcpyDefault
  double* dev; CK(hipMalloc(&dev, 1 << 20));
  CK(hipMemsetAsync(dev, 0x11, 1 << 20, s1));
  CK(hipMemcpyAsync(reinterpret_cast<char*>(d) + 4096, dev, 1 << 20, hipMemcpyDefault, s1));
  CK(hipMemcpyAsync(dev, reinterpret_cast<char*>(d) + 4096, 1 << 20, hipMemcpyDefault, s1));
  CK(hipStreamSynchronize(s1));
  printf("segment byte %02x\n", reinterpret_cast<unsigned char*>(p)[4096 + 5]); fflush(stdout);
  CK(hipHostUnregister(p));
  printf("probe ok\n");
  return 0;
}
