// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into or called by the
// product path (zk-apps_amd/); used by tests/ as a fast CPU checker and by
// bench.py's cpu_baseline leg ("port": in-repo C++ CPU restatement of the
// arkworks algorithm shapes, NOT arkworks itself).
//
// PARITY UNPINNED against the reference: /root/reference holds no prover, MSM
// or NTT.  This file restates the published algorithms named by
// BASELINE.json's north_star and is itself pinned to oracle/*.py (big-int
// definitions) by tests/test_oracle_cpp.py on the committed golden vectors:
//   * NTT  : ark_poly 0.4.2 Radix2EvaluationDomain in-place radix-2 (bit-reverse
//            + Cooley-Tukey), coset generator 7              [SURVEY row a6]
//   * MSM  : ark_ec 0.4.2 VariableBaseMSM::msm_bigint — unsigned c-bit windows,
//            c = 3 if n < 32 else floor(log2(n)*69/100)+2, one bucket set per
//            window, running-sum reduction; (window, point-chunk) tasks in parallel  [rows a8/a9]
//   * prove: ark_groth16 0.4 create_proof_with_assignment + LibsnarkReduction
//            witness map                                       [rows a7/a10]
// Deliberately different from the product: 64-bit limbs (unsigned __int128),
// Jacobian coordinates, unsigned digits, (window x point-chunk) threading.
//
// Build: g++ -O3 -march=native -std=c++17 -shared -fPIC -pthread
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

typedef unsigned __int128 u128;

template <int N, const uint64_t* MOD, uint64_t INV, const uint64_t* RR, const uint64_t* ONE_>
struct Fp {
  uint64_t l[N];
  static Fp zero() { Fp r; memset(r.l, 0, sizeof(r.l)); return r; }
  static Fp one() { Fp r; memcpy(r.l, ONE_, sizeof(r.l)); return r; }
  bool is_zero() const { uint64_t a = 0; for (int i = 0; i < N; i++) a |= l[i]; return a == 0; }
  bool operator==(const Fp& o) const { return memcmp(l, o.l, sizeof(l)) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }
  static bool geq_mod(const uint64_t* a) {
    for (int i = N - 1; i >= 0; i--) if (a[i] != MOD[i]) return a[i] > MOD[i];
    return true;
  }
  static void sub_mod(uint64_t* a) {
    u128 b = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)a[i] - MOD[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
  }
  Fp operator+(const Fp& o) const {
    Fp r; u128 c = 0;
    for (int i = 0; i < N; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r; u128 b = 0;
    for (int i = 0; i < N; i++) { u128 d = (u128)l[i] - o.l[i] - (uint64_t)b; r.l[i] = (uint64_t)d; b = (d >> 64) & 1; }
    if (b) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)r.l[i] + MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
  }
  Fp neg() const { return is_zero() ? *this : zero() - *this; }
  Fp dbl() const { return *this + *this; }
  Fp operator*(const Fp& o) const {
    // separated operand scanning: full product then Montgomery reduction
    uint64_t t[2 * N + 1];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < N; i++) {
      u128 c = 0;
      for (int j = 0; j < N; j++) { c += (u128)l[j] * o.l[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
      t[i + N] = (uint64_t)c;
    }
    for (int i = 0; i < N; i++) {
      uint64_t m = t[i] * INV;
      u128 c = 0;
      for (int j = 0; j < N; j++) { c += (u128)m * MOD[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + N; c && k <= 2 * N; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
    }
    Fp r; memcpy(r.l, t + N, sizeof(r.l));
    if (t[2 * N] || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  Fp sqr() const { return *this * *this; }
  Fp pow(const uint64_t* e, int n) const {
    Fp res = one();
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { res = res.sqr(); if ((e[i] >> b) & 1) res = res * *this; }
    return res;
  }
  Fp inv() const {
    uint64_t e[N]; memcpy(e, MOD, sizeof(e));
    // e = MOD - 2 (low limb of both moduli is >= 2)
    e[0] -= 2;
    return pow(e, N);
  }
  static Fp from_u64(uint64_t v) { Fp a = zero(); a.l[0] = v; return a.to_mont(); }
  Fp to_mont() const { Fp rr; memcpy(rr.l, RR, sizeof(rr.l)); return *this * rr; }
  Fp from_mont() const { Fp o = zero(); o.l[0] = 1; return *this * o; }
  static Fp from_bytes(const uint8_t* b) { Fp a; memcpy(a.l, b, sizeof(a.l)); return a.to_mont(); }
  void to_bytes(uint8_t* b) const { Fp c = from_mont(); memcpy(b, c.l, sizeof(c.l)); }
  bool lex_larger() const {  // canonical value > (m-1)/2
    Fp c = from_mont(); uint64_t d[N + 1]; uint64_t carry = 0;
    for (int i = 0; i < N; i++) { d[i] = (c.l[i] << 1) | carry; carry = c.l[i] >> 63; }
    if (carry) return true;
    return geq_mod(d);
  }
};

static const uint64_t FR_MOD[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
static const uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
static const uint64_t FQ_MOD[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t FQ_R2[6] = {0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull, 0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull};
static const uint64_t FQ_ONE[6] = {0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull};
typedef Fp<4, FR_MOD, 0xfffffffeffffffffull, FR_R2, FR_ONE> Fr;
typedef Fp<6, FQ_MOD, 0x89f3fffcfffcfffdull, FQ_R2, FQ_ONE> Fq;

struct Fq2 {
  Fq c0, c1;
  static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static Fq2 one() { return {Fq::one(), Fq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(const Fq2& o) const { return !(*this == o); }
  Fq2 operator+(const Fq2& o) const { return {c0 + o.c0, c1 + o.c1}; }
  Fq2 operator-(const Fq2& o) const { return {c0 - o.c0, c1 - o.c1}; }
  Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  Fq2 operator*(const Fq2& o) const {  // schoolbook (4 mul), unlike the product's Karatsuba
    return {c0 * o.c0 - c1 * o.c1, c0 * o.c1 + c1 * o.c0};
  }
  Fq2 sqr() const { return *this * *this; }
  Fq2 inv() const { Fq d = (c0.sqr() + c1.sqr()).inv(); return {c0 * d, (c1 * d).neg()}; }
  static Fq2 from_bytes(const uint8_t* b) { return {Fq::from_bytes(b), Fq::from_bytes(b + 48)}; }
  void to_bytes(uint8_t* b) const { c0.to_bytes(b); c1.to_bytes(b + 48); }
  bool lex_larger() const { return c1.is_zero() ? c0.lex_larger() : c1.lex_larger(); }
};

// Jacobian points (X, Y, Z), infinity <=> Z == 0
template <class F>
struct Jac {
  F x, y, z;
  static Jac inf() { return {F::one(), F::one(), F::zero()}; }
  bool is_inf() const { return z.is_zero(); }
  Jac neg() const { return {x, y.neg(), z}; }
  Jac dbl() const {
    if (is_inf() || y.is_zero()) return inf();
    F a = x.sqr(), b = y.sqr(), c = b.sqr();
    F d = ((x + b).sqr() - a - c).dbl();
    F e = a.dbl() + a, f = e.sqr();
    F x3 = f - d.dbl();
    F y3 = e * (d - x3) - c.dbl().dbl().dbl();
    F z3 = (y * z).dbl();
    return {x3, y3, z3};
  }
  Jac add(const Jac& o) const {
    if (is_inf()) return o;
    if (o.is_inf()) return *this;
    F z1z1 = z.sqr(), z2z2 = o.z.sqr();
    F u1 = x * z2z2, u2 = o.x * z1z1;
    F s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
    if (u1 == u2) return s1 == s2 ? dbl() : inf();
    F h = u2 - u1, r = s2 - s1;
    F hh = h.sqr(), hhh = h * hh, v = u1 * hh;
    F x3 = r.sqr() - hhh - v.dbl();
    F y3 = r * (v - x3) - s1 * hhh;
    F z3 = z * o.z * h;
    return {x3, y3, z3};
  }
  // mixed addition with an affine point (ax, ay), not infinity
  Jac add_affine(const F& ax, const F& ay) const {
    if (is_inf()) return {ax, ay, F::one()};
    F z1z1 = z.sqr();
    F u2 = ax * z1z1, s2 = ay * z * z1z1;
    if (x == u2) return y == s2 ? dbl() : inf();
    F h = u2 - x, r = s2 - y;
    F hh = h.sqr(), hhh = h * hh, v = x * hh;
    F x3 = r.sqr() - hhh - v.dbl();
    F y3 = r * (v - x3) - y * hhh;
    F z3 = z * h;
    return {x3, y3, z3};
  }
  void to_affine(F* ax, F* ay, bool* inf_) const {
    if (is_inf()) { *inf_ = true; *ax = F::zero(); *ay = F::zero(); return; }
    F zi = z.inv(), zi2 = zi.sqr();
    *ax = x * zi2; *ay = y * zi2 * zi; *inf_ = false;
  }
  Jac mul(const uint64_t* k, int n) const {
    Jac acc = inf();
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { acc = acc.dbl(); if ((k[i] >> b) & 1) acc = acc.add(*this); }
    return acc;
  }
};

template <class F> struct Wire;
template <> struct Wire<Fq> { enum { W = 96, H = 48 }; };
template <> struct Wire<Fq2> { enum { W = 192, H = 96 }; };

static bool all_zero(const uint8_t* b, size_t n) { for (size_t i = 0; i < n; i++) if (b[i]) return false; return true; }

template <class F>
static void read_affine(const uint8_t* b, F* x, F* y, bool* inf) {
  if (all_zero(b, Wire<F>::W)) { *inf = true; return; }
  *inf = false; *x = F::from_bytes(b); *y = F::from_bytes(b + Wire<F>::H);
}
template <class F>
static void write_affine(const Jac<F>& p, uint8_t* b) {
  F x, y; bool inf; p.to_affine(&x, &y, &inf);
  if (inf) { memset(b, 0, Wire<F>::W); return; }
  x.to_bytes(b); y.to_bytes(b + Wire<F>::H);
}

// Threads worth starting = CPUs this process may actually use: logical CPUs, cut down by the affinity mask and by the
// cgroup CPU quota (a container on a 256-thread host with cpu.max = "1600000 100000" gets 16 CPUs' worth of time; 256
// runnable threads then burn the quota in a fraction of each period and the whole group is throttled: measured 0.49 s
// against 0.23 s for a 2^18-term MSM).  bench.py reports this number as `cores`.
static int n_threads() {
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
  double q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                                  // cgroup v2
  if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");  // cgroup v1
  if (q > 0) {
    const long c = (long)(q + 0.999);
    if (c >= 1 && c < n) n = c;
  }
  return (int)n;
}

template <class Fn>
static void parallel_for(size_t n, int threads, Fn fn) {
  if (threads <= 1 || n < 2) { for (size_t i = 0; i < n; i++) fn(i); return; }
  std::vector<std::thread> th;
  size_t T = std::min<size_t>(threads, n);
  for (size_t t = 0; t < T; t++) th.emplace_back([=]() { for (size_t i = t; i < n; i += T) fn(i); });
  for (auto& x : th) x.join();
}

// ---------------------------------------------------------------------------
// MSM: arkworks msm_bigint shape
// ---------------------------------------------------------------------------
static int ark_window(size_t n) {
  if (n < 32) return 3;
  int lg = 63 - __builtin_clzll((unsigned long long)n);
  return lg * 69 / 100 + 2;
}

// Work split: arkworks' msm_bigint hands whole WINDOWS to rayon, which keeps at most ceil(255 / c) = 17 threads busy at
// n = 2^20 -- on a 256-thread host that understates what a CPU can do.  Here the points are cut into chunks as well: a
// task = (window, chunk) with its own bucket set, the chunk sums of a window are added afterwards.  The window width
// follows from the thread count (more tasks -> fewer points per bucket set -> narrower windows): with one thread per
// window it is arkworks' own rule.  The result is the same group element either way.
struct MsmSplit {
  int c, nwin, nchunk;
};
static MsmSplit msm_split(size_t n, int threads) {
  const int c_ark = ark_window(n);
  MsmSplit best = {c_ark, (255 + c_ark - 1) / c_ark, 1};
  if (threads <= 1 || n < 4096) return best;
  // time ~ rounds of tasks x (insertions of a task + 3 additions per bucket of its running-sum reduction); the search
  // covers arkworks' own width (one chunk) and every narrower width with up to 4 tasks per thread
  double best_cost = -1;
  for (int c = 6; c <= c_ark; c++) {
    const int nwin = (255 + c - 1) / c;
    const int max_chunk = (4 * threads + nwin - 1) / nwin;
    for (int nchunk = 1; nchunk <= max_chunk; nchunk++) {
      if (nchunk > 1 && n / (size_t)nchunk < (size_t(1) << c)) break;  // at least one point per bucket
      const size_t tasks = (size_t)nwin * nchunk;
      const double rounds = (double)((tasks + threads - 1) / threads);
      const double cost = rounds * ((double)n / nchunk + 3.0 * (double)(size_t(1) << c));
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best = {c, nwin, nchunk};
      }
    }
  }
  return best;
}

template <class F>
static Jac<F> msm(const uint8_t* scalars, const uint8_t* bases, size_t n, int threads) {
  const MsmSplit sp = msm_split(n, threads);
  const int c = sp.c, nwin = sp.nwin, nchunk = sp.nchunk;
  std::vector<F> bx(n), by(n);
  std::vector<uint8_t> binf(n);
  {
    const size_t par = n >= 4096 ? (size_t)threads : 1;
    parallel_for(par, (int)par, [&](size_t t) {
      for (size_t i = n * t / par; i < n * (t + 1) / par; i++) {
        bool inf;
        read_affine<F>(bases + (size_t)Wire<F>::W * i, &bx[i], &by[i], &inf);
        binf[i] = inf;
      }
    });
  }
  std::vector<Jac<F>> part((size_t)nwin * nchunk, Jac<F>::inf());
  parallel_for((size_t)nwin * nchunk, threads, [&](size_t task) {
    const size_t w = task / nchunk, q = task % nchunk;
    const size_t lo = n * q / nchunk, hi = n * (q + 1) / nchunk;
    std::vector<Jac<F>> buckets((size_t(1) << c) - 1, Jac<F>::inf());
    const int bit = (int)w * c;
    for (size_t i = lo; i < hi; i++) {
      if (binf[i]) continue;
      const uint64_t* k = reinterpret_cast<const uint64_t*>(scalars + 32 * i);
      uint64_t limbs[5] = {k[0], k[1], k[2], k[3], 0};
      int li = bit >> 6, sh = bit & 63;
      uint64_t v = limbs[li] >> sh;
      if (sh && li + 1 < 5) v |= limbs[li + 1] << (64 - sh);
      v &= (uint64_t(1) << c) - 1;
      if (v) buckets[v - 1] = buckets[v - 1].add_affine(bx[i], by[i]);
    }
    Jac<F> run = Jac<F>::inf(), acc = Jac<F>::inf();
    for (size_t b = buckets.size(); b-- > 0;) { run = run.add(buckets[b]); acc = acc.add(run); }
    part[task] = acc;
  });
  Jac<F> total = Jac<F>::inf();
  for (int w = nwin - 1; w >= 0; w--) {
    for (int i = 0; i < c; i++) total = total.dbl();
    for (int q = 0; q < nchunk; q++) total = total.add(part[(size_t)w * nchunk + q]);
  }
  return total;
}

// ---------------------------------------------------------------------------
// NTT
// ---------------------------------------------------------------------------
static Fr root_of_unity(int log_n) {
  static const uint64_t ROOT[4] = {0x3829971f439f0d2bull, 0xb63683508c2280b9ull, 0xd09b681922c813b4ull, 0x16a2a19edfe81f20ull};
  Fr w; memcpy(w.l, ROOT, 32); w = w.to_mont();
  for (int i = 32; i > log_n; i--) w = w.sqr();
  return w;
}

static void ntt_inplace(std::vector<Fr>& a, int log_n, bool inverse, int threads) {
  const size_t n = size_t(1) << log_n;
  const size_t par = n >= 8192 ? (size_t)threads : 1;
  parallel_for(par, (int)par, [&](size_t t) {  // bit-reversal: pairs (i, rev i) are disjoint
    for (size_t i = n * t / par; i < n * (t + 1) / par; i++) {
      size_t r = 0;
      for (int b = 0; b < log_n; b++) if (i >> b & 1) r |= size_t(1) << (log_n - 1 - b);
      if (i < r) std::swap(a[i], a[r]);
    }
  });
  Fr wn = root_of_unity(log_n);
  if (inverse) wn = wn.inv();
  std::vector<Fr> tw(n / 2 ? n / 2 : 1);
  parallel_for(par, (int)par, [&](size_t t) {
    const size_t lo = (n / 2) * t / par, hi = (n / 2) * (t + 1) / par;
    if (lo >= hi && !(t == 0)) return;
    uint64_t ex[1] = {lo};
    Fr x = wn.pow(ex, 1);
    if (t == 0) tw[0] = Fr::one();
    for (size_t i = lo; i < hi; i++) { tw[i] = x; x = x * wn; }
  });
  // every stage: the n/2 butterflies in contiguous per-thread ranges (one fork/join per stage)
  const size_t half = n / 2;
  const size_t parts = half >= 4096 ? (size_t)threads : 1;
  for (int s = 0; s < log_n; s++) {
    const size_t m = size_t(1) << s, step = n / (2 * m);
    parallel_for(parts, (int)parts, [&](size_t t) {
      const size_t lo = half * t / parts, hi = half * (t + 1) / parts;
      for (size_t bf = lo; bf < hi; bf++) {
        const size_t j = bf & (m - 1), k = (bf >> s) * 2 * m;
        Fr x = a[k + j + m] * tw[j * step];
        Fr u = a[k + j];
        a[k + j] = u + x;
        a[k + j + m] = u - x;
      }
    });
  }
  if (inverse) {
    Fr ninv = Fr::from_u64(n).inv();
    parallel_for((size_t)threads, threads, [&](size_t t) { for (size_t i = t; i < n; i += threads) a[i] = a[i] * ninv; });
  }
}

static void coset_scale(std::vector<Fr>& a, const Fr& g, int threads) {
  const size_t n = a.size();
  const size_t chunk = (n + threads - 1) / threads;
  parallel_for((size_t)threads, threads, [&](size_t t) {
    size_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) return;
    uint64_t ex[1] = {b};
    Fr x = g.pow(ex, 1);
    for (size_t i = b; i < e; i++) { a[i] = a[i] * x; x = x * g; }
  });
}

static void transform(std::vector<Fr>& a, int log_n, bool inverse, bool coset, int threads) {
  const Fr g = Fr::from_u64(7);
  if (coset && !inverse) coset_scale(a, g, threads);
  ntt_inplace(a, log_n, inverse, threads);
  if (coset && inverse) coset_scale(a, g.inv(), threads);
}

// ---------------------------------------------------------------------------
// compression (zcash format)
// ---------------------------------------------------------------------------
static void be48(const Fq& a, uint8_t* out) { uint8_t le[48]; a.to_bytes(le); for (int i = 0; i < 48; i++) out[i] = le[47 - i]; }
static void compress_g1(const Jac<Fq>& p, uint8_t out[48]) {
  Fq x, y; bool inf; p.to_affine(&x, &y, &inf);
  if (inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
  be48(x, out); out[0] |= 0x80; if (y.lex_larger()) out[0] |= 0x20;
}
static void compress_g2(const Jac<Fq2>& p, uint8_t out[96]) {
  Fq2 x, y; bool inf; p.to_affine(&x, &y, &inf);
  if (inf) { memset(out, 0, 96); out[0] = 0xC0; return; }
  be48(x.c1, out); be48(x.c0, out + 48); out[0] |= 0x80; if (y.lex_larger()) out[0] |= 0x20;
}

template <class F>
static Jac<F> jac_from_wire(const uint8_t* b) {
  F x, y; bool inf; read_affine<F>(b, &x, &y, &inf);
  return inf ? Jac<F>::inf() : Jac<F>{x, y, F::one()};
}

extern "C" {

int oracle_threads(void) { return n_threads(); }

// data: 2^log_n x 32 B canonical LE, in place
int oracle_ntt_fr(uint8_t* data, uint32_t log_n, int inverse, int coset, int threads) {
  if (threads <= 0) threads = n_threads();
  const size_t n = size_t(1) << log_n;
  std::vector<Fr> a(n);
  const size_t par = n >= 8192 ? (size_t)threads : 1;
  parallel_for(par, (int)par, [&](size_t t) { for (size_t i = n * t / par; i < n * (t + 1) / par; i++) a[i] = Fr::from_bytes(data + 32 * i); });
  transform(a, (int)log_n, inverse != 0, coset != 0, threads);
  parallel_for(par, (int)par, [&](size_t t) { for (size_t i = n * t / par; i < n * (t + 1) / par; i++) a[i].to_bytes(data + 32 * i); });
  return 0;
}

int oracle_msm_g1(const uint8_t* scalars, const uint8_t* bases, uint64_t n, uint8_t out[96], int threads) {
  if (threads <= 0) threads = n_threads();
  write_affine<Fq>(msm<Fq>(scalars, bases, n, threads), out);
  return 0;
}
int oracle_msm_g2(const uint8_t* scalars, const uint8_t* bases, uint64_t n, uint8_t out[192], int threads) {
  if (threads <= 0) threads = n_threads();
  write_affine<Fq2>(msm<Fq2>(scalars, bases, n, threads), out);
  return 0;
}

// h coefficients of the LibsnarkReduction witness map; CSR values canonical LE
int oracle_witness_map(uint32_t n_vars, uint32_t n_pub, uint32_t nc, uint32_t log_n, const uint32_t* const rowptr[3],
                       const uint32_t* const col[3], const uint8_t* const val[3], const uint8_t* z, uint8_t* out_h,
                       int threads) {
  if (threads <= 0) threads = n_threads();
  const size_t N = size_t(1) << log_n;
  std::vector<Fr> zm(n_vars);
  for (uint32_t i = 0; i < n_vars; i++) zm[i] = Fr::from_bytes(z + 32ull * i);
  std::vector<Fr> abc[3];
  for (int m = 0; m < 3; m++) {
    abc[m].assign(N, Fr::zero());
    std::vector<Fr> vals(rowptr[m][nc]);
    for (size_t k = 0; k < vals.size(); k++) vals[k] = Fr::from_bytes(val[m] + 32 * k);
    parallel_for((size_t)threads, threads, [&](size_t t) {
      for (size_t i = t; i < nc; i += threads) {
        Fr acc = Fr::zero();
        for (uint32_t k = rowptr[m][i]; k < rowptr[m][i + 1]; k++) acc = acc + vals[k] * zm[col[m][k]];
        abc[m][i] = acc;
      }
    });
  }
  for (uint32_t j = 0; j < n_pub; j++) abc[0][nc + j] = zm[j];
  for (int m = 0; m < 3; m++) {
    transform(abc[m], (int)log_n, true, false, threads);
    transform(abc[m], (int)log_n, false, true, threads);
  }
  Fr gn = Fr::from_u64(7);
  for (uint32_t i = 0; i < log_n; i++) gn = gn.sqr();
  const Fr zinv = (gn - Fr::one()).inv();
  for (size_t i = 0; i < N; i++) abc[0][i] = (abc[0][i] * abc[1][i] - abc[2][i]) * zinv;
  transform(abc[0], (int)log_n, true, true, threads);
  for (size_t i = 0; i < N; i++) abc[0][i].to_bytes(out_h + 32 * i);
  return 0;
}

// full prover; queries in wire format (l_query has n_vars - n_pub entries, h_query N - 1)
int oracle_groth16_prove(uint32_t n_vars, uint32_t n_pub, uint32_t nc, uint32_t log_n, const uint32_t* const rowptr[3],
                         const uint32_t* const col[3], const uint8_t* const val[3], const uint8_t* alpha_g1,
                         const uint8_t* beta_g1, const uint8_t* beta_g2, const uint8_t* delta_g1,
                         const uint8_t* delta_g2, const uint8_t* a_query, const uint8_t* b_g1_query,
                         const uint8_t* b_g2_query, const uint8_t* h_query, const uint8_t* l_query, const uint8_t* z,
                         const uint8_t* r, const uint8_t* s, uint8_t out_proof[192], int threads) {
  if (threads <= 0) threads = n_threads();
  const size_t N = size_t(1) << log_n;
  std::vector<uint8_t> h(32 * N);
  oracle_witness_map(n_vars, n_pub, nc, log_n, rowptr, col, val, z, h.data(), threads);
  Jac<Fq> h_acc = msm<Fq>(h.data(), h_query, N - 1, threads);
  Jac<Fq> l_acc = msm<Fq>(z + 32ull * n_pub, l_query, n_vars - n_pub, threads);
  Jac<Fq> a_acc = msm<Fq>(z + 32, a_query + 96, n_vars - 1, threads);
  Jac<Fq> b1_acc = msm<Fq>(z + 32, b_g1_query + 96, n_vars - 1, threads);
  Jac<Fq2> b2_acc = msm<Fq2>(z + 32, b_g2_query + 192, n_vars - 1, threads);
  uint64_t rk[4], sk[4], rsk[4];
  memcpy(rk, r, 32); memcpy(sk, s, 32);
  Fr rs = Fr::from_bytes(r) * Fr::from_bytes(s);
  { uint8_t b[32]; rs.to_bytes(b); memcpy(rsk, b, 32); }
  Jac<Fq> d1 = jac_from_wire<Fq>(delta_g1);
  Jac<Fq> g_a = d1.mul(rk, 4).add(jac_from_wire<Fq>(a_query)).add(a_acc).add(jac_from_wire<Fq>(alpha_g1));
  Jac<Fq> g1_b = d1.mul(sk, 4).add(jac_from_wire<Fq>(b_g1_query)).add(b1_acc).add(jac_from_wire<Fq>(beta_g1));
  Jac<Fq2> g2_b = jac_from_wire<Fq2>(delta_g2).mul(sk, 4).add(jac_from_wire<Fq2>(b_g2_query)).add(b2_acc).add(jac_from_wire<Fq2>(beta_g2));
  Jac<Fq> g_c = g_a.mul(sk, 4).add(g1_b.mul(rk, 4)).add(d1.mul(rsk, 4).neg()).add(l_acc).add(h_acc);
  compress_g1(g_a, out_proof);
  compress_g2(g2_b, out_proof + 48);
  compress_g1(g_c, out_proof + 144);
  return 0;
}

}  // extern "C"

// Trusted setup with explicit toxic waste (tau, alpha, beta, gamma, delta), shaped after
// ark_groth16 0.4 generator::generate_parameters_with_qap + LibsnarkReduction::
// instance_map_with_evaluation [not in the reference tree]: u = Lagrange coefficients of the domain
// at tau; a/b/c[col] += u[row] * coeff over the constraint rows; a[i] += u[nc + i] for the instance
// columns; gamma_abc_i = (beta a_i + alpha b_i + c_i) / gamma, l_i = the same / delta,
// h_i = tau^i * Z(tau) / delta for i < N - 1; every query entry = scalar * generator.
// Deliberately different from the product's setup: Lagrange coefficients by the barycentric
// recurrence with one inversion per element batch, 64-bit limbs, Jacobian fixed-base tables.
template <class F>
struct FixedBase {
  std::vector<Jac<F>> t;  // t[w * 256 + d] = d * 2^(8 w) * G
  explicit FixedBase(const Jac<F>& g) : t(32 * 256) {
    Jac<F> base = g;
    for (int w = 0; w < 32; w++) {
      t[w * 256] = Jac<F>::inf();
      for (int d = 1; d < 256; d++) t[w * 256 + d] = t[w * 256 + d - 1].add(base);
      for (int k = 0; k < 8; k++) base = base.dbl();
    }
  }
  Jac<F> mul(const Fr& s) const {
    uint8_t b[32];
    s.to_bytes(b);
    Jac<F> acc = Jac<F>::inf();
    for (int w = 0; w < 32; w++)
      if (b[w]) acc = acc.add(t[w * 256 + b[w]]);
    return acc;
  }
};

static Jac<Fq> g1_gen() {
  static const uint64_t X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
  static const uint64_t Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
  Fq x, y; memcpy(x.l, X, 48); memcpy(y.l, Y, 48);
  return {x.to_mont(), y.to_mont(), Fq::one()};
}
static Jac<Fq2> g2_gen() {
  static const uint64_t X0[6] = {0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull, 0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull};
  static const uint64_t X1[6] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull, 0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull};
  static const uint64_t Y0[6] = {0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull, 0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull};
  static const uint64_t Y1[6] = {0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull, 0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull};
  Fq2 x, y;
  memcpy(x.c0.l, X0, 48); memcpy(x.c1.l, X1, 48); memcpy(y.c0.l, Y0, 48); memcpy(y.c1.l, Y1, 48);
  return {{x.c0.to_mont(), x.c1.to_mont()}, {y.c0.to_mont(), y.c1.to_mont()}, Fq2::one()};
}

extern "C" int oracle_groth16_setup(uint32_t n_vars, uint32_t n_pub, uint32_t nc, uint32_t log_n,
                                    const uint32_t* const rowptr[3], const uint32_t* const col[3],
                                    const uint8_t* const val[3], const uint8_t toxic[160], uint8_t* out_vk,
                                    uint8_t out_beta_g1[96], uint8_t out_delta_g1[96], uint8_t* out_a, uint8_t* out_b1,
                                    uint8_t* out_b2, uint8_t* out_h, uint8_t* out_l, int threads) {
  if (threads <= 0) threads = n_threads();
  const size_t N = size_t(1) << log_n;
  const Fr tau = Fr::from_bytes(toxic), alpha = Fr::from_bytes(toxic + 32), beta = Fr::from_bytes(toxic + 64),
           gamma = Fr::from_bytes(toxic + 96), delta = Fr::from_bytes(toxic + 128);
  // u_i = L_i(tau) = Z(tau) / N * w^i / (tau - w^i)  (ark_poly evaluate_all_lagrange_coefficients)
  Fr tn = tau;
  for (uint32_t i = 0; i < log_n; i++) tn = tn.sqr();
  const Fr zt = tn - Fr::one();
  if (zt.is_zero()) return -1;
  const Fr w = root_of_unity((int)log_n);
  std::vector<Fr> u(N), wi(N);
  {
    Fr x = Fr::one();
    for (size_t i = 0; i < N; i++) { wi[i] = x; x = x * w; }
    const Fr scale = zt * Fr::from_u64(N).inv();
    parallel_for((size_t)threads, threads, [&](size_t t) {
      // chunked batch inversion of (tau - w^i)
      const size_t chunk = (N + threads - 1) / threads, b = t * chunk, e = std::min(N, b + chunk);
      if (b >= e) return;
      std::vector<Fr> pre(e - b);
      Fr run = Fr::one();
      for (size_t i = b; i < e; i++) { pre[i - b] = run; run = run * (tau - wi[i]); }
      Fr inv = run.inv();
      for (size_t i = e; i-- > b;) { u[i] = inv * pre[i - b] * wi[i] * scale; inv = inv * (tau - wi[i]); }
    });
  }
  std::vector<Fr> abc[3];
  for (int m = 0; m < 3; m++) {
    abc[m].assign(n_vars, Fr::zero());
    for (uint32_t i = 0; i < nc; i++)
      for (uint32_t k = rowptr[m][i]; k < rowptr[m][i + 1]; k++)
        abc[m][col[m][k]] = abc[m][col[m][k]] + u[i] * Fr::from_bytes(val[m] + 32ull * k);
  }
  for (uint32_t j = 0; j < n_pub; j++) abc[0][j] = abc[0][j] + u[nc + j];
  const Fr ginv = gamma.inv(), dinv = delta.inv();
  std::vector<Fr> lq(n_vars), hq(N - 1);
  for (uint32_t j = 0; j < n_vars; j++) lq[j] = (beta * abc[0][j] + alpha * abc[1][j] + abc[2][j]) * (j < n_pub ? ginv : dinv);
  {
    Fr t = zt * dinv;
    for (size_t i = 0; i + 1 < N; i++) { hq[i] = t; t = t * tau; }
  }
  const FixedBase<Fq> t1(g1_gen());
  const FixedBase<Fq2> t2(g2_gen());
  parallel_for((size_t)n_vars, threads, [&](size_t j) {
    write_affine<Fq>(t1.mul(abc[0][j]), out_a + 96 * j);
    write_affine<Fq>(t1.mul(abc[1][j]), out_b1 + 96 * j);
    write_affine<Fq2>(t2.mul(abc[1][j]), out_b2 + 192 * j);
    if (j >= n_pub) write_affine<Fq>(t1.mul(lq[j]), out_l + 96 * (j - n_pub));
  });
  parallel_for(N - 1, threads, [&](size_t i) { write_affine<Fq>(t1.mul(hq[i]), out_h + 96 * i); });
  write_affine<Fq>(t1.mul(alpha), out_vk);
  write_affine<Fq2>(t2.mul(beta), out_vk + 96);
  write_affine<Fq2>(t2.mul(gamma), out_vk + 288);
  write_affine<Fq2>(t2.mul(delta), out_vk + 480);
  for (uint32_t j = 0; j < n_pub; j++) write_affine<Fq>(t1.mul(lq[j]), out_vk + 672 + 96ull * j);
  write_affine<Fq>(t1.mul(beta), out_beta_g1);
  write_affine<Fq>(t1.mul(delta), out_delta_g1);
  return 0;
}
