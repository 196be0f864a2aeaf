// zkmi — device self-test of the quad-split group law (quad.hpp) against the one-lane XYZZ::add of curve.hpp.
// TEST SCAFFOLDING (include/zkmi_testing.h): compiled into libzkmi_exp.so only.
#include "ctx.hpp"
#include "quad.hpp"
#include <stdio.h>

#ifdef ZKMI_TESTING
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
    a.l[11] &= 0x0fffffffu;
    return a;
  }
};

__global__ void k_add_one_lane(const XYZZ<Fq28>* a, const XYZZ<Fq28>* o, XYZZ<Fq28>* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<Fq28> r = a[i];
  r.add(o[i]);
  out[i] = r;
}
template <int XCH, bool DBG>
__global__ void __launch_bounds__(64, 3) k_add_quad(const XYZZ<Fq28>* a, const XYZZ<Fq28>* o, XYZZ<Fq28>* out, uint32_t n, Fq28* dbg) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (i >= n) return;
  XYZZQ<Fq28, XCH> r = XYZZQ<Fq28, XCH>::load(a + i);
  r.template add<DBG>(XYZZQ<Fq28, XCH>::load(o + i), dbg + 24 * (size_t)i);
  r.store(out + i);
}
// wave w sums the 16 points a[16 w ..] (as XYZZ) and the 16 affine points aff[16 w ..] (odd ones negated) into out[2 w], out[2 w + 1]
__global__ void __launch_bounds__(64, 3) k_wave_sum(const XYZZ<Fq28>* a, const Affine<Fq28>* aff, XYZZ<Fq28>* out) {
  const uint32_t w = blockIdx.x, quad = threadIdx.x >> 2;
  XYZZQ<Fq28> s = wave_quad_sum(XYZZQ<Fq28>::load(a + 16 * w + quad));
  if (quad == 0) s.store(out + 2 * w);
  XYZZQ<Fq28> t = wave_quad_sum(XYZZQ<Fq28>::from_affine(aff + 16 * w + quad, (quad & 1u) != 0));
  if (quad == 0) t.store(out + 2 * w + 1);
}

// one pair through add<true>: the intermediate values of the four rounds (diagnostics of a failing self-test)
__global__ void __launch_bounds__(64, 3) k_add_quad_dbg(const XYZZ<Fq28>* a, const XYZZ<Fq28>* o, Fq28* dbg, uint32_t i) {
  if (threadIdx.x >= 4) return;
  XYZZQ<Fq28> r = XYZZQ<Fq28>::load(a + i);
  r.add<true>(XYZZQ<Fq28>::load(o + i), dbg);
  r.store(const_cast<XYZZ<Fq28>*>(a) + i);  // (diagnostics only: the pair is not used again)
}

XYZZ<Fq28> to_dev(const XYZZ<Fq>& p) { return {fq28_from_fq(p.x), fq28_from_fq(p.y), fq28_from_fq(p.zz), fq28_from_fq(p.zzz)}; }
bool same(const XYZZ<Fq28>& d, const XYZZ<Fq>& h) {
  return fq_from_fq28(d.x) == h.x && fq_from_fq28(d.y) == h.y && fq_from_fq28(d.zz) == h.zz && fq_from_fq28(d.zzz) == h.zzz;
}
}  // namespace

// n pairs (a, o) of XYZZ points -- random coordinates (the formulas are identities of the coordinate ring), and every
// eighth pair one of: o = a, o = -a, o = O, a = O, o = a in another representation, o = -a in another representation --
// added by the quad form and by curve.hpp's one-lane form on the device and by the 32-bit-limb host arithmetic; plus
// sums of 16 points over the quads of a wave (XYZZ and affine sources).
extern "C" int32_t zkmi_selftest_quad_add(zkmi_ctx* ctx, uint64_t seed, uint32_t n, uint32_t* out_mismatches) {
  if (!ctx || !out_mismatches || n == 0 || n > (1u << 20)) return ZKMI_ERR_BAD_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ZKMI_ERR_HIP;
  n = (n + 15u) & ~15u;
  Rng rng{seed};
  std::vector<XYZZ<Fq>> ha(n), ho(n);
  std::vector<Affine<Fq>> haff(n);
  for (uint32_t i = 0; i < n; i++) {
    XYZZ<Fq> a = {rng.fq(), rng.fq(), rng.fq(), rng.fq()}, o = {rng.fq(), rng.fq(), rng.fq(), rng.fq()};
    const Fq lam = rng.fq(), l2 = lam.sqr(), l3 = l2 * lam;
    switch (i % 8) {
      case 0: o = a; break;
      case 1: o = a.neg(); break;
      case 2: o = XYZZ<Fq>::infinity(); break;
      case 3: a = XYZZ<Fq>::infinity(); break;
      case 4: o = {a.x * l2, a.y * l3, a.zz * l2, a.zzz * l3}; break;
      case 5: o = {a.x * l2, (a.y * l3).neg(), a.zz * l2, a.zzz * l3}; break;
      default: break;
    }
    if (i % 64 == 6) a = o = XYZZ<Fq>::infinity();
    ha[i] = a;
    ho[i] = o;
    haff[i] = {rng.fq(), rng.fq()};
    if (i % 16 == 9) haff[i] = Affine<Fq>::infinity();
    if (i % 16 == 11) haff[i] = haff[i - 8];        // the tree's first level adds quads q and q + 8 (same sign): a doubling
    if (i % 32 == 13) haff[i] = haff[i - 8].neg();  // P + (-P) at the same level
  }
  std::vector<XYZZ<Fq28>> da(n), dob(n);
  std::vector<Affine<Fq28>> daff(n);
  for (uint32_t i = 0; i < n; i++) {
    da[i] = to_dev(ha[i]);
    dob[i] = to_dev(ho[i]);
    daff[i] = {fq28_from_fq(haff[i].x), fq28_from_fq(haff[i].y)};
    if (haff[i].is_inf()) daff[i] = Affine<Fq28>::infinity();
  }
  XYZZ<Fq28>*ga = nullptr, *go = nullptr, *g1 = nullptr, *gq = nullptr, *gs = nullptr;
  Affine<Fq28>* gaff = nullptr;
  const size_t bytes = sizeof(XYZZ<Fq28>) * n;
  int32_t rc = ZKMI_OK;
  uint32_t bad = 0;
  std::vector<XYZZ<Fq28>> r1(n), rq(n), rs(n / 8);
  if (hipMalloc(&ga, bytes) != hipSuccess || hipMalloc(&go, bytes) != hipSuccess || hipMalloc(&g1, bytes) != hipSuccess ||
      hipMalloc(&gq, bytes) != hipSuccess || hipMalloc(&gs, sizeof(XYZZ<Fq28>) * (n / 8)) != hipSuccess ||
      hipMalloc(&gaff, sizeof(Affine<Fq28>) * n) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  if (hipMemcpy(ga, da.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(go, dob.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(gaff, daff.data(), sizeof(Affine<Fq28>) * n, hipMemcpyHostToDevice) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  hipLaunchKernelGGL(k_add_one_lane, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, g1, n);
  hipLaunchKernelGGL((k_add_quad<0, false>), dim3((4 * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gq, n, (Fq28*)nullptr);
  hipLaunchKernelGGL(k_wave_sum, dim3(n / 16), dim3(64), 0, ctx->stream, ga, gaff, gs);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(r1.data(), g1, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rq.data(), gq, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rs.data(), gs, sizeof(XYZZ<Fq28>) * (n / 8), hipMemcpyDeviceToHost) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  {
    uint32_t bq[8] = {0}, b1[8] = {0};
    for (uint32_t i = 0; i < n; i++) {
      XYZZ<Fq> h = ha[i];
      h.add(ho[i]);
      if (!same(rq[i], h)) bad++, bq[i % 8]++;
      if (!same(r1[i], h)) bad++, b1[i % 8]++;
    }
    if (bad && zkmi::debug_level()) {
      // the same pairs through two variants: exchanges as ds_bpermute instead of DPP; the DPP form with its intermediates stored
      {
        Fq28* gdd = nullptr;
        std::vector<XYZZ<Fq28>> rv(n);
        if (hipMalloc(&gdd, sizeof(Fq28) * 24 * (size_t)n) == hipSuccess) {
          for (int var = 0; var < 3; var++) {
            if (var == 0) hipLaunchKernelGGL((k_add_quad<1, false>), dim3((4 * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gq, n, gdd);
            if (var == 1) hipLaunchKernelGGL((k_add_quad<0, true>), dim3((4 * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gq, n, gdd);
            if (var == 2) hipLaunchKernelGGL((k_add_quad<1, true>), dim3((4 * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gq, n, gdd);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(rv.data(), gq, bytes, hipMemcpyDeviceToHost);
            uint32_t wrong = 0, coord[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < n; i++) {
              XYZZ<Fq> h = ha[i];
              h.add(ho[i]);
              if (!same(rv[i], h)) {
                wrong++;
                coord[0] += fq_from_fq28(rv[i].x) != h.x;
                coord[1] += fq_from_fq28(rv[i].y) != h.y;
                coord[2] += fq_from_fq28(rv[i].zz) != h.zz;
                coord[3] += fq_from_fq28(rv[i].zzz) != h.zzz;
              }
            }
            if (var == 1) {
              // pair 7 of the multi-quad run: are the stored m4 of lanes 1 and 2 right, and is the kernel's y their difference?
              std::vector<Fq28> dd(24);
              (void)hipMemcpy(dd.data(), gdd + 24 * 7, sizeof(Fq28) * 24, hipMemcpyDeviceToHost);
              const Fq28 m41 = dd[5 * 4 + 1], m42 = dd[5 * 4 + 2], ydiff = m41 - m42;
              XYZZ<Fq> h = ha[7];
              h.add(ho[7]);
              fprintf(stderr, "zkmi_selftest_quad_add: multi-quad pair 7: host(m4[1] - m4[2]) %s expected y; kernel y %s host(m4[1] - m4[2])\n",
                      fq_from_fq28(ydiff) == h.y ? "==" : "!=", fq_from_fq28(rv[7].y) == fq_from_fq28(ydiff) ? "==" : "!=");
              for (int l = 0; l < 14; l++)
                fprintf(stderr, "  limb %2d: m4[1] %10d  m4[2] %10d  host diff %10d  kernel y %10d  m4[0] %10d m4[3] %10d\n", l, m41.l[l], m42.l[l], ydiff.l[l],
                        rv[7].y.l[l], dd[5 * 4 + 0].l[l], dd[5 * 4 + 3].l[l]);
            }
            fprintf(stderr, "zkmi_selftest_quad_add: variant %s: %u of %u wrong (x %u, y %u, zz %u, zzz %u)\n",
                    var == 0 ? "bpermute" : var == 1 ? "dpp + stored intermediates" : "bpermute + stored intermediates", wrong, n, coord[0],
                    coord[1], coord[2], coord[3]);
          }
          (void)hipFree(gdd);
        }
        uint32_t coord[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < n; i++) {
          XYZZ<Fq> h = ha[i];
          h.add(ho[i]);
          coord[0] += fq_from_fq28(rq[i].x) != h.x;
          coord[1] += fq_from_fq28(rq[i].y) != h.y;
          coord[2] += fq_from_fq28(rq[i].zz) != h.zz;
          coord[3] += fq_from_fq28(rq[i].zzz) != h.zzz;
        }
        fprintf(stderr, "zkmi_selftest_quad_add: product form: wrong coordinates x %u, y %u, zz %u, zzz %u\n", coord[0], coord[1], coord[2], coord[3]);
      }
      Fq28* gd = nullptr;
      std::vector<Fq28> hd(24);
      const uint32_t i = 7;  // a general pair
      if (hipMalloc(&gd, sizeof(Fq28) * 24) == hipSuccess) {
        (void)hipMemset(gd, 0, sizeof(Fq28) * 24);
        hipLaunchKernelGGL(k_add_quad_dbg, dim3(1), dim3(64), 0, ctx->stream, ga, go, gd, i);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(hd.data(), gd, sizeof(Fq28) * 24, hipMemcpyDeviceToHost);
        (void)hipFree(gd);
        const XYZZ<Fq>&A = ha[i], &O = ho[i];
        const Fq u1 = A.x * O.zz, s1 = A.y * O.zzz, u2 = A.zz * O.x, s2 = A.zzz * O.y;
        const Fq P = u2 - u1, R = s2 - s1, PP = P.sqr(), RR = R.sqr(), zz12 = A.zz * O.zz, zzz12 = A.zzz * O.zzz;
        const Fq PPP = P * PP, Q = u1 * PP, ZZ3 = zz12 * PP, T = zzz12 * PP, X3 = RR - PPP - Q.dbl();
        const Fq want[24] = {u1, s1, u2, s2, P, R, u1 - u2, s1 - s2, PP, RR, zz12, zzz12, PPP, Q, ZZ3, T,
                             X3, X3, X3, X3, X3, R * (Q - X3), s1 * PPP, T * P};
        const char* names[6] = {"m1", "d", "m2", "m3", "x3", "m4"};
        for (int r = 0; r < 6; r++)
          for (int k = 0; k < 4; k++) {
            const bool dontcare = (r == 4 && k >= 2) || (r == 5 && k == 0);
            const bool okv = fq_from_fq28(hd[4 * r + k]) == want[4 * r + k];
            fprintf(stderr, "zkmi_selftest_quad_add: %s lane %d: %s (limb 13 = %d, limb 0 = %d)\n", names[r], k,
                    dontcare ? "-" : okv ? "ok" : "WRONG", hd[4 * r + k].l[13], hd[4 * r + k].l[0]);
          }
      }
      for (int c = 0; c < 8; c++) fprintf(stderr, "zkmi_selftest_quad_add: case %d: quad form %u wrong, one-lane form %u wrong\n", c, bq[c], b1[c]);
    }
  }
  for (uint32_t w = 0; w < n / 16; w++) {
    // the tree order of wave_quad_sum: s = 8, 4, 2, 1 over the quads (the group law is not associative on non-curve
    // coordinates, so the host follows the same order)
    XYZZ<Fq> t[16], u[16];
    for (int q = 0; q < 16; q++) {
      t[q] = ha[16 * w + q];
      Affine<Fq> p = haff[16 * w + q];
      if (q & 1) p = p.neg();
      u[q] = XYZZ<Fq>::from_affine(p);
    }
    for (int s = 8; s >= 1; s >>= 1)
      for (int q = 0; q < s; q++) {
        t[q].add(t[q + s]);
        u[q].add(u[q + s]);
      }
    const bool e1 = !same(rs[2 * w], t[0]), e2 = !same(rs[2 * w + 1], u[0]);
    bad += (e1 ? 1 : 0) + (e2 ? 1 : 0);
    if ((e1 || e2) && zkmi::debug_level() && w < 8) fprintf(stderr, "zkmi_selftest_quad_add: wave %u: XYZZ sum %s, affine sum %s\n", w, e1 ? "WRONG" : "ok", e2 ? "WRONG" : "ok");
  }
out:
  (void)hipFree(ga);
  (void)hipFree(go);
  (void)hipFree(g1);
  (void)hipFree(gq);
  (void)hipFree(gs);
  (void)hipFree(gaff);
  *out_mismatches = bad;
  return rc;
}
#endif  // ZKMI_TESTING
