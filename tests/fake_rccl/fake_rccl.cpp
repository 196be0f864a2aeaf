// TEST INFRASTRUCTURE (tests/ only) — an all-gather double with RCCL's entry points for ranks that SHARE one GPU.
//
// RCCL refuses two ranks on one device ("duplicate GPU"), and the GPU pool this repository is tested on has one GPU per
// box, so no rank > 0 of the exchange in zk-apps_amd/csrc/comm.hip could ever run there.  This library implements the five
// entry points comm.hip resolves (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString)
// over a POSIX shared-memory segment: every rank copies its slot device -> host -> segment, the ranks meet at a barrier,
// every rank copies all slots segment -> device.  ZKMI_RCCL_LIB=<this file> makes libzkmi.so use it (and nothing else).
// It proves nothing about RCCL or xGMI; it lets the rank arithmetic of the product -- slices, plans, slot offsets,
// per-rank combination -- execute with 2, 4 and 8 ranks on hardware.  Synchronous on purpose (the collective has
// completed when the call returns), which is within what a caller of an asynchronous collective may assume.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <new>

namespace {
constexpr size_t SLOT_CAP = 1u << 20;  // bytes a rank may contribute per collective
constexpr size_t HEADER = 4096;
struct Header {
  std::atomic<uint32_t> count, gen, failed;
};
struct Comm {
  Header* h = nullptr;
  uint8_t* data = nullptr;
  size_t map_bytes = 0;
  int nranks = 0, rank = 0;
  char name[64];
};
struct UniqueId {
  char internal[128];
};
double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
// sense-reversing barrier over the segment; a rank that waits longer than 300 s gives up and poisons the communicator
bool barrier(Comm* c) {
  Header* h = c->h;
  const uint32_t g = h->gen.load();
  if (h->count.fetch_add(1) + 1 == (uint32_t)c->nranks) {
    h->count.store(0);
    h->gen.fetch_add(1);
    return h->failed.load() == 0;
  }
  const double t0 = now_s();
  while (h->gen.load() == g) {
    if (h->failed.load()) return false;
    if (now_s() - t0 > 300.0) {
      h->failed.store(1);
      return false;
    }
    struct timespec ts = {0, 20000};
    nanosleep(&ts, nullptr);
  }
  return h->failed.load() == 0;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  if (!id) return 4;
  memset(id->internal, 0, sizeof(id->internal));
  unsigned char rnd[12] = {0};
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd >= 0) {
    (void)!read(fd, rnd, sizeof(rnd));
    close(fd);
  }
  char* p = id->internal;
  p += sprintf(p, "/zkmi_fake_rccl_");
  for (unsigned char b : rnd) p += sprintf(p, "%02x", b);
  return 0;
}

int ncclCommInitRank(void** out, int nranks, UniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return 4;
  Comm* c = new (std::nothrow) Comm();
  if (!c) return 1;
  c->nranks = nranks;
  c->rank = rank;
  id.internal[63] = 0;
  strcpy(c->name, id.internal);
  c->map_bytes = HEADER + SLOT_CAP * (size_t)nranks;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return 2;
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return 2;
  }
  c->h = static_cast<Header*>(m);  // a fresh segment is zero-filled: count = gen = failed = 0
  c->data = static_cast<uint8_t*>(m) + HEADER;
  if (!barrier(c)) {  // the communicator exists once every rank has attached (RCCL's init is collective too)
    munmap(m, c->map_bytes);
    delete c;
    return 3;
  }
  *out = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 4;
  munmap(c->h, c->map_bytes);
  shm_unlink(c->name);  // the first rank to get here removes the name; the others' mappings stay valid until they unmap
  delete c;
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || dtype != 1 /* ncclUint8 */ || count > SLOT_CAP || (count && (!send || !recv))) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (count && hipMemcpy(c->data + SLOT_CAP * (size_t)c->rank, send, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (!barrier(c)) return 3;
  for (int k = 0; k < c->nranks && count; k++)
    if (hipMemcpy(static_cast<uint8_t*>(recv) + count * (size_t)k, c->data + SLOT_CAP * (size_t)k, count, hipMemcpyHostToDevice) != hipSuccess)
      return 1;
  if (!barrier(c)) return 3;  // nobody overwrites a slot before every rank has read it
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case 0: return "fake_rccl: success";
    case 1: return "fake_rccl: HIP call failed";
    case 2: return "fake_rccl: shared-memory segment unavailable";
    case 3: return "fake_rccl: a rank did not reach the barrier";
    default: return "fake_rccl: invalid argument";
  }
}

}  // extern "C"
