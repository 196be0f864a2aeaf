// zkmi — tuning switches.
//
// The PRODUCT library (default build, libzkmi.so) has ONE schedule: every ZK_TUNE(...) below compiles to its default
// value and the library reads no tuning variable from the environment -- a stale or hostile environment cannot select a
// slower or less-tested path, and the kernel generations the switches used to select are not in its code objects.
// What the product library does read (the complete list): ZKMI_HOST_THREADS (host threads for proof
// assembly), ZKMI_BACKTRACE (crash handler), ZKMI_DEBUG (diagnostics on stderr), ZKMI_RCCL_LIB (comm.hip: the RCCL library
// file of the zkmi_comm_* exchange -- absolute path, owned by root or the effective user, not world-writable, else refused),
// and the launcher's LOCAL_WORLD_SIZE / OMPI_COMM_WORLD_LOCAL_SIZE / SLURM_NTASKS_PER_NODE / MPI_LOCALNRANKS (host_pool.hpp:
// ranks that share this node's CPUs).
//
// `make experiments` builds the A/B library libzkmi_exp.so with -DZKMI_EXPERIMENTS: there every switch is read from the
// environment (once) and the retired kernels are compiled in.  tests/test_gpu_sizes.py::test_ab_switches_do_not_change_any_result
// keeps every variant of that library byte-identical to the product; scripts/env_ab.sh measures them (ZKMI_LIB selects
// the library in the Python binding).
#pragma once
#include <stdlib.h>

namespace zkmi {
#ifdef ZKMI_EXPERIMENTS
inline int tune_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
#define ZK_TUNE(name, dflt) ([]() -> int { static const int v = ::zkmi::tune_env(name, dflt); return v; }())
#else
#define ZK_TUNE(name, dflt) (dflt)
#endif
// ZKMI_DEBUG (both libraries): 1 = diagnostics of failing launches on stderr, 2 = also host-side timestamps of single proofs
inline int debug_level() {
  static const int v = [] {
    const char* e = getenv("ZKMI_DEBUG");
    return (e && *e) ? (atoi(e) > 0 ? atoi(e) : 1) : 0;
  }();
  return v;
}
}  // namespace zkmi
