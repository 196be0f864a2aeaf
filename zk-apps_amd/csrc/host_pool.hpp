// zkmi — the host side of the prover: how many CPUs this process may use, one persistent pool of assembly threads, and a
// wait for device events that does not burn a core.
//
// Why: the GPU box grants a CONTAINER 16 CPUs' worth of time on a 256-thread host (cgroup cpu.max = "1600000 100000").
// Round 4 sized the assembly pool by std::thread::hardware_concurrency() (256, capped at 16), started the threads anew
// for every group of proofs and let the driving thread spin inside hipEventSynchronize: eight ranks of a node would have
// asked for 8 x 17 runnable threads on 16 CPUs -- the quota is then burnt in a fraction of each period and the whole
// cgroup is throttled.  Now:
//   * host_cpu_budget() = min(logical CPUs, affinity mask, cgroup quota) / (ranks of this node: LOCAL_WORLD_SIZE, as
//     torchrun exports it), at least 1; ZKMI_HOST_THREADS or zkmi_set_host_threads() override it;
//   * HostPool: budget - 1 workers created once per process, the calling thread works too; several callers (one driving
//     thread per device in zkmi_groth16_prove_batch_multi) share it;
//   * wait_event(): a short poll, then sleeping polls -- a rank's driving thread costs ~3 % of a CPU while it waits for the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "tune.hpp"

namespace zkmi {

// CPUs' worth of time the process may use: logical CPUs, cut down by the affinity mask and the cgroup quota (v2, then v1)
inline unsigned host_cpus_granted() {
  long n = (long)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const long a = CPU_COUNT(&set);
    if (a >= 1 && a < n) n = a;
  }
  auto quota = [](const char* path, const char* path_period) -> double {
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    char a[64] = {0}, b[64] = {0};
    const int got = fscanf(f, "%63s %63s", a, b);
    fclose(f);
    if (got < 1 || !strcmp(a, "max") || atof(a) <= 0) return 0;
    double period = got >= 2 ? atof(b) : 0;
    if (period <= 0 && path_period) {
      FILE* g = fopen(path_period, "r");
      if (g) {
        if (fscanf(g, "%63s", b) == 1) period = atof(b);
        fclose(g);
      }
    }
    return period > 0 ? atof(a) / period : 0;
  };
  double q = quota("/sys/fs/cgroup/cpu.max", nullptr);
  if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  if (q > 0) {
    const long c = (long)(q + 0.999);
    if (c >= 1 && c < n) n = c;
  }
  return (unsigned)n;
}

// processes of this job on this node (they share the grant): torchrun / torch.distributed.run export LOCAL_WORLD_SIZE
inline unsigned host_local_ranks() {
  const char* e = getenv("LOCAL_WORLD_SIZE");
  const int v = e ? atoi(e) : 0;
  return (unsigned)(v >= 1 && v <= 1024 ? v : 1);
}

constexpr unsigned HOST_THREADS_MAX = 64;
inline std::atomic<unsigned>& host_threads_override() {
  static std::atomic<unsigned> v{0};  // zkmi_set_host_threads
  return v;
}
// threads the assembly of a group of proofs may occupy (the calling thread included)
inline unsigned host_cpu_budget() {
  if (const unsigned o = host_threads_override().load()) return o;
  static const unsigned v = [] {
    const char* env = getenv("ZKMI_HOST_THREADS");
    const int forced = env ? atoi(env) : 0;
    if (forced >= 1 && forced <= 256) return (unsigned)(forced > (int)HOST_THREADS_MAX ? HOST_THREADS_MAX : forced);
    unsigned n = host_cpus_granted() / host_local_ranks();
    if (n < 1) n = 1;
    if (n > 16) n = 16;  // a group has at most 64 proofs of ~0.7 ms each: more threads than this only add wake-ups
    return n;
  }();
  return v;
}

class HostPool {
 public:
  static HostPool& instance() {
    static HostPool p;
    return p;
  }
  // fn(0) .. fn(n - 1), on at most `width` threads (the caller is one of them); returns when all have run
  void run(uint32_t n, unsigned width, const std::function<void(uint32_t)>& fn) {
    if (n == 0) return;
    if (width > n) width = n;
    if (width <= 1) {
      for (uint32_t i = 0; i < n; i++) fn(i);
      return;
    }
    Job job;
    job.fn = &fn;
    job.n = n;
    {
      std::lock_guard<std::mutex> g(mu_);
      grow(width - 1);
      // `width - 1` tickets: a worker that takes one works on this job until its indices run out
      for (unsigned k = 0; k + 1 < width; k++) queue_.push_back(&job);
    }
    cv_.notify_all();
    work(job);
    // tickets nobody took yet must not outlive the job
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto it = queue_.begin(); it != queue_.end();) it = (*it == &job) ? queue_.erase(it) : it + 1;
    }
    std::unique_lock<std::mutex> lk(job.mu);
    job.cv.wait(lk, [&] { return job.active == 0 && job.done.load() == n; });
  }
  unsigned workers() {
    std::lock_guard<std::mutex> g(mu_);
    return (unsigned)threads_.size();
  }

 private:
  struct Job {
    const std::function<void(uint32_t)>* fn = nullptr;
    uint32_t n = 0;
    std::atomic<uint32_t> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
    unsigned active = 0;  // workers inside work() for this job (guarded by mu)
  };
  static void work(Job& job) {
    for (;;) {
      const uint32_t i = job.next.fetch_add(1);
      if (i >= job.n) break;
      (*job.fn)(i);
      job.done.fetch_add(1);
    }
  }
  void grow(unsigned want) {  // mu_ held
    if (want > HOST_THREADS_MAX) want = HOST_THREADS_MAX;
    while (threads_.size() < want) threads_.emplace_back([this] { loop(); });
  }
  void loop() {
    for (;;) {
      Job* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_) return;
        job = queue_.front();
        queue_.pop_front();
        std::lock_guard<std::mutex> g(job->mu);  // (taken under mu_: the owner cannot retire the job between the two)
        job->active++;
      }
      work(*job);
      {
        // (notified under the lock: the owner cannot see active == 0, return and destroy the job while this thread still
        // holds a reference to its condition variable)
        std::lock_guard<std::mutex> g(job->mu);
        job->active--;
        job->cv.notify_all();
      }
    }
  }
  HostPool() = default;
  ~HostPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job*> queue_;
  std::vector<std::thread> threads_;
  bool stop_ = false;
};

// Wait for a device event without holding a CPU: poll for ~100 us, then sleep 50 us between polls (a rank's driving thread
// costs a few per cent of a CPU while the GPU works; hipEventSynchronize on an event without hipEventBlockingSync spins).
// spin = true keeps hipEventSynchronize: ONE proof by itself, whose latency is what counts and which waits four times.
// (A/B library: ZKMI_HOST_WAIT=0 spins everywhere.)
inline hipError_t wait_event(hipEvent_t ev, bool spin) {
  if (spin || ZK_TUNE("ZKMI_HOST_WAIT", 1) == 0) return hipEventSynchronize(ev);
  const auto t0 = std::chrono::steady_clock::now();
  bool polled_busy = false;
  hipError_t e;
  for (;;) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
    polled_busy = true;
    if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(100)) continue;
    struct timespec ts = {0, 50000};
    nanosleep(&ts, nullptr);
  }
  // hipErrorNotReady is a status, not a failure: it must not surface from the hipGetLastError() behind the next launch
  if (polled_busy && e == hipSuccess) (void)hipGetLastError();
  return e;
}

}  // namespace zkmi
