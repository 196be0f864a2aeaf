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
#include <pthread.h>
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
#include <new>
#include <thread>
#include <vector>
#include "tune.hpp"

namespace zkmi {

// Where the two numbers below came from (zkmi_host_info_string: "cpus=16 (cgroup2 /sys/fs/cgroup/cpu.max) ranks=8 (LOCAL_WORLD_SIZE)")
struct HostGrant {
  unsigned cpus = 1;
  char cpu_source[160] = "logical CPUs";
  unsigned ranks = 1;
  char rank_source[48] = "single process";
};

// CPUs' worth of time the process may use: logical CPUs, cut down by the affinity mask and by the SMALLEST CPU quota on the
// path from the process's own cgroup up to the root (cgroup v2: cpu.max; v1: cpu.cfs_quota_us / cpu.cfs_period_us).  The
// process's cgroup is read from /proc/self/cgroup: inside a cgroup namespace that path is "/" and the walk is the one file
// the namespace root shows; without a namespace (or under a nested / hybrid hierarchy) the quota sits further down and the
// namespace-root file alone would miss it.
inline HostGrant host_grant_probe() {
  HostGrant g;
  long n = (long)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const long a = CPU_COUNT(&set);
    if (a >= 1 && a < n) {
      n = a;
      snprintf(g.cpu_source, sizeof(g.cpu_source), "affinity mask");
    }
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
      FILE* h = fopen(path_period, "r");
      if (h) {
        if (fscanf(h, "%63s", b) == 1) period = atof(b);
        fclose(h);
      }
    }
    return period > 0 ? atof(a) / period : 0;
  };
  double best = 0;
  char best_src[160] = "";
  auto consider = [&](double q, const char* kind, const char* path) {
    if (q > 0 && (best <= 0 || q < best)) {
      best = q;
      snprintf(best_src, sizeof(best_src), "%s %s", kind, path);
    }
  };
  // every directory from `rel` (a cgroup path, "/a/b") up to the mount point `mnt`
  auto walk = [&](const char* mnt, const char* rel, bool v2) {
    char dir[512];
    snprintf(dir, sizeof(dir), "%s%s", mnt, rel);
    for (;;) {
      size_t len = strlen(dir);
      while (len > 1 && dir[len - 1] == '/') dir[--len] = 0;
      char f1[600], f2[600];
      if (v2) {
        snprintf(f1, sizeof(f1), "%s/cpu.max", dir);
        consider(quota(f1, nullptr), "cgroup2", f1);
      } else {
        snprintf(f1, sizeof(f1), "%s/cpu.cfs_quota_us", dir);
        snprintf(f2, sizeof(f2), "%s/cpu.cfs_period_us", dir);
        consider(quota(f1, f2), "cgroup1", f1);
      }
      if (strlen(dir) <= strlen(mnt)) break;
      char* slash = strrchr(dir, '/');
      if (!slash || slash == dir) break;
      *slash = 0;
      if (strlen(dir) < strlen(mnt)) break;
    }
  };
  bool walked = false;
  if (FILE* f = fopen("/proc/self/cgroup", "r")) {
    char line[600];
    while (fgets(line, sizeof(line), f)) {
      // "<id>:<controllers>:<path>"
      char* c1 = strchr(line, ':');
      char* c2 = c1 ? strchr(c1 + 1, ':') : nullptr;
      if (!c2) continue;
      *c2 = 0;
      char* path = c2 + 1;
      path[strcspn(path, "\n")] = 0;
      if (path[0] != '/' || strstr(path, "..")) continue;
      const char* ctl = c1 + 1;
      if (!*ctl) {  // v2 (unified): "0::/path"
        walk("/sys/fs/cgroup", path, true);
        walked = true;
      } else if (strstr(ctl, "cpu") && !strstr(ctl, "cpuset")) {  // v1: "4:cpu,cpuacct:/path"
        char mnt[256];
        snprintf(mnt, sizeof(mnt), "/sys/fs/cgroup/%s", ctl);
        walk(mnt, path, false);
        walk("/sys/fs/cgroup/cpu", path, false);
        walked = true;
      }
    }
    fclose(f);
  }
  if (!walked) {  // no /proc: the files the namespace root shows
    consider(quota("/sys/fs/cgroup/cpu.max", nullptr), "cgroup2", "/sys/fs/cgroup/cpu.max");
    consider(quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"), "cgroup1", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us");
  }
  if (best > 0) {
    const long c = (long)(best + 0.999);
    if (c >= 1 && c < n) {
      n = c;
      snprintf(g.cpu_source, sizeof(g.cpu_source), "%s", best_src);
    }
  }
  g.cpus = (unsigned)n;
  // processes of this job on this node (they share the grant): torchrun / torch.distributed.run export LOCAL_WORLD_SIZE,
  // Open MPI OMPI_COMM_WORLD_LOCAL_SIZE, Slurm SLURM_NTASKS_PER_NODE, MPICH / Intel MPI MPI_LOCALNRANKS
  for (const char* name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "MPI_LOCALNRANKS"}) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    if (v >= 1 && v <= 1024) {
      g.ranks = (unsigned)v;
      snprintf(g.rank_source, sizeof(g.rank_source), "%s", name);
      break;
    }
  }
  return g;
}
inline const HostGrant& host_grant() {
  static const HostGrant g = host_grant_probe();
  return g;
}
inline unsigned host_cpus_granted() { return host_grant().cpus; }
inline unsigned host_local_ranks() { return host_grant().ranks; }

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
  // fork(): the child has none of the workers (only the forking thread survives), but it has their std::thread objects --
  // ~HostPool would join() threads that do not exist and never return from exit().  The child forgets them (the objects
  // are leaked on purpose: destroying a joinable std::thread terminates) and starts from an empty pool; the locks are
  // re-made because a worker may have held them at the instant of the fork.  A fork in the MIDDLE of run() is the
  // caller's own problem only as far as its job goes: the owner drains its own indices.
  static void atfork_child() {
    HostPool& p = instance();
    new (&p.mu_) std::mutex();
    new (&p.cv_) std::condition_variable();
    (void)new std::vector<std::thread>(std::move(p.threads_));
    new (&p.threads_) std::vector<std::thread>();
    p.queue_.clear();
    p.stop_ = false;
  }
  HostPool() { (void)pthread_atfork(nullptr, nullptr, &HostPool::atfork_child); }
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
