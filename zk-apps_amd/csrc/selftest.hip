// zkmi — host-executed self-test of the device field representation
// (field28.hpp is __host__ __device__): random chains of mul/sqr/add/sub/neg/dbl
// in Fq28 limbs against the 32-bit-limb Fq, plus the zero test on k*p forms.
#include "ctx.hpp"
#include "curve.hpp"
#include "field28.hpp"

using namespace zkmi;

namespace {
struct Rng {
  uint64_t s;
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  Fq fq() {
    Fq a;
    for (int i = 0; i < 12; i += 2) {
      uint64_t v = next();
      a.l[i] = (uint32_t)v;
      a.l[i + 1] = (uint32_t)(v >> 32);
    }
    a.l[11] &= 0x0fffffffu;  // < 2^380 < p
    return a;                // arbitrary residue, read as Montgomery form
  }
};
}  // namespace

#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
extern "C" int32_t zkmi_selftest_fq28(uint64_t seed, uint32_t iters, uint32_t* out_mismatches) {
  if (!out_mismatches) return ZKMI_ERR_BAD_ARG;
  Rng rng{seed};
  uint32_t bad = 0;
  for (uint32_t it = 0; it < iters; it++) {
    Fq a = rng.fq(), b = rng.fq(), c = rng.fq();
    if (it % 7 == 0) b = a;
    if (it % 11 == 0) c = Fq::zero();
    if (it % 13 == 0) a = Fq::one();
    Fq28 A = fq28_from_fq(a), B = fq28_from_fq(b), C = fq28_from_fq(c);
    if (fq_from_fq28(A) != a) bad++;
    // a chain shaped like the XYZZ mixed addition
    Fq u2 = a * b, s2 = c * a.sqr();
    Fq pp_ = u2 - c, r = s2 - b;
    Fq pp = pp_.sqr(), ppp = pp_ * pp, q = c * pp;
    Fq x3 = r.sqr() - ppp - q.dbl();
    Fq y3 = r * (q - x3) - b * ppp;
    Fq z3 = (a + b + c).neg().dbl() * x3;
    Fq28 U2 = A * B, S2 = C * A.sqr();
    Fq28 PP_ = U2 - C, R_ = S2 - B;
    Fq28 PP = PP_.sqr(), PPP = PP_ * PP, Q = C * PP;
    Fq28 X3 = R_.sqr() - PPP - Q.dbl();
    Fq28 Y3 = R_ * (Q - X3) - B * PPP;
    Fq28 Z3 = (A + B + C).neg().dbl() * X3;
    if (fq_from_fq28(X3) != x3) bad++;
    if (fq_from_fq28(Y3) != y3) bad++;
    if (fq_from_fq28(Z3) != z3) bad++;
    if (PP.is_zero() != pp.is_zero()) bad++;
    // zero tests on products and Fq2-style combinations
    Fq28 t0 = A * B, t1 = B * A;
    if (!(t0 - t1).is_zero()) bad++;
    if (!((A - A) * B).is_zero()) bad++;
    if ((A * B).is_zero() != (a * b).is_zero()) bad++;
    Fq28 zero_like = (t0 + t0) - t1 - t1;  // |v| < 4p
    if (!zero_like.is_zero()) bad++;
    if (!Fq28::zero().is_zero() || Fq28::one().is_zero()) bad++;
    // Fq2 over the limb representation (lazy-reduction product/square)
    {
      Fq2 x = {a, b}, y = {c, a * b};
      Fq2_28 X = {A, B}, Y = {C, A * B};
      Fq2 m = x * y, q2 = x.sqr(), chain = (m - q2) * x + y.dbl();
      Fq2_28 M = X * Y, Q2 = X.sqr(), CH = (M - Q2) * X + Y.dbl();
      if (fq_from_fq28(M) != m) bad++;
      if (fq_from_fq28(Q2) != q2) bad++;
      if (fq_from_fq28(CH) != chain) bad++;
      if (!(X * Y - Y * X).is_zero()) bad++;
      if ((X * Y).is_zero() != m.is_zero()) bad++;
    }
    // whole mixed / full additions through curve.hpp in both representations
    {
      Affine<Fq> p1 = {a, b}, p2 = {c, a * c};
      XYZZ<Fq> h = XYZZ<Fq>::from_affine(p1);
      h.madd(p2);
      XYZZ<Fq> h2 = h;
      h2.madd(p1);
      h.add(h2);
      Affine<Fq28> P1 = {A, B}, P2 = {C, A * C};
      XYZZ<Fq28> d = XYZZ<Fq28>::from_affine(P1);
      d.madd(P2);
      XYZZ<Fq28> d2 = d;
      d2.madd(P1);
      d.add(d2);
      if (fq_from_fq28(d.x) != h.x || fq_from_fq28(d.y) != h.y || fq_from_fq28(d.zz) != h.zz ||
          fq_from_fq28(d.zzz) != h.zzz)
        bad++;
      Affine<Fq2> q1 = {{a, b}, {c, a}}, q2 = {{b, c}, {a * b, c}};
      XYZZ<Fq2> g = XYZZ<Fq2>::from_affine(q1);
      g.madd(q2);
      XYZZ<Fq2> g2 = g;
      g2.madd(q1);
      g.add(g2);
      Affine<Fq2_28> Q1 = {{A, B}, {C, A}}, Q2 = {{B, C}, {A * B, C}};
      XYZZ<Fq2_28> e = XYZZ<Fq2_28>::from_affine(Q1);
      e.madd(Q2);
      XYZZ<Fq2_28> e2 = e;
      e2.madd(Q1);
      e.add(e2);
      if (fq_from_fq28(e.x) != g.x || fq_from_fq28(e.y) != g.y || fq_from_fq28(e.zz) != g.zz ||
          fq_from_fq28(e.zzz) != g.zzz)
        bad++;
    }
    // lazy forms feeding a product
    if (fq_from_fq28(A.add_lazy(B) * C.sub_lazy(A)) != (a + b) * (c - a)) bad++;
  }
  *out_mismatches = bad;
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING

// Host self-test of the assembly pool (host_pool.hpp): `callers` threads submit `jobs` jobs each, of 1 .. 64 items and
// widths 1 .. 17, concurrently; every item must run exactly once and every run() must return only after its own items.
#include <atomic>
#include <thread>
#include <vector>
#include "host_pool.hpp"
#ifdef ZKMI_TESTING  // test scaffolding: libzkmi_exp.so only (include/zkmi_testing.h)
extern "C" int32_t zkmi_selftest_host_pool(uint32_t callers, uint32_t jobs, uint32_t* out_mismatches) {
  if (!out_mismatches || callers == 0 || callers > 32) return ZKMI_ERR_BAD_ARG;
  std::atomic<uint32_t> bad{0};
  auto caller = [&](uint32_t c) {
    Rng rng{0x9001u + c};
    for (uint32_t j = 0; j < jobs; j++) {
      const uint32_t n = 1 + (uint32_t)(rng.next() % 64), width = 1 + (uint32_t)(rng.next() % 17);
      std::vector<std::atomic<uint32_t>> hits(n);
      for (auto& h : hits) h.store(0);
      std::atomic<uint64_t> sum{0};
      HostPool::instance().run(n, width, [&](uint32_t i) {
        hits[i].fetch_add(1);
        uint64_t v = i + 1;
        for (int k = 0; k < 200; k++) v = v * 6364136223846793005ull + 1442695040888963407ull;  // a little work
        sum.fetch_add((v & 1) + i + 1);
      });
      uint64_t lo = (uint64_t)n * (n + 1) / 2;
      for (uint32_t i = 0; i < n; i++)
        if (hits[i].load() != 1) bad.fetch_add(1);
      if (sum.load() < lo || sum.load() > lo + n) bad.fetch_add(1);
    }
  };
  std::vector<std::thread> th;
  for (uint32_t c = 1; c < callers; c++) th.emplace_back(caller, c);
  caller(0);
  for (auto& t : th) t.join();
  *out_mismatches = bad.load();
  return ZKMI_OK;
}
#endif  // ZKMI_TESTING
