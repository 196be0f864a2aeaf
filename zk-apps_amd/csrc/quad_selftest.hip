// zkmi — device self-test of the quad-split (G1) and octet-split (G2) group law (quad.hpp) against the one-lane XYZZ::add of
// curve.hpp.  TEST SCAFFOLDING (include/zkmi_testing.h): compiled into libzkmi_exp.so only.
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
  void draw(Fq& x) { x = fq(); }
  void draw(Fq2& x) { x = {fq(), fq()}; }
};

template <class DF>
__global__ void k_add_one_lane(const XYZZ<DF>* a, const XYZZ<DF>* o, XYZZ<DF>* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<DF> r = a[i];
  r.add(o[i]);
  out[i] = r;
}
template <class PT, bool DBG>
__global__ void __launch_bounds__(64, 2) k_add_quad(const typename PT::Point* a, const typename PT::Point* o, typename PT::Point* out, uint32_t n, Fq28* dbg) {
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / PT::LANES;
  if (i >= n) return;
  PT r = PT::load(a + i);
  r.template add<DBG>(PT::load(o + i), dbg + 6 * PT::LANES * (size_t)i);
  r.store(out + i);
}
// wave w sums the PER_WAVE points a[PER_WAVE w ..] (as XYZZ) and as many affine points (odd ones negated) into out[2 w], out[2 w + 1]
template <class PT>
__global__ void __launch_bounds__(64, 2) k_wave_sum(const typename PT::Point* a, const typename PT::APoint* aff, typename PT::Point* out) {
  const uint32_t w = blockIdx.x, quad = threadIdx.x / PT::LANES;
  PT s = wave_quad_sum(PT::load(a + PT::PER_WAVE * w + quad));
  if (quad == 0) s.store(out + 2 * w);
  PT t = wave_quad_sum(PT::from_affine(aff + PT::PER_WAVE * w + quad, (quad & 1u) != 0));
  if (quad == 0) t.store(out + 2 * w + 1);
}

template <class HF, class DF>
XYZZ<DF> to_dev(const XYZZ<HF>& p) {
  return {fq28_from_fq(p.x), fq28_from_fq(p.y), fq28_from_fq(p.zz), fq28_from_fq(p.zzz)};
}
template <class HF, class DF>
bool same(const XYZZ<DF>& d, const XYZZ<HF>& h) {
  return fq_from_fq28(d.x) == h.x && fq_from_fq28(d.y) == h.y && fq_from_fq28(d.zz) == h.zz && fq_from_fq28(d.zzz) == h.zzz;
}

// n pairs (a, o) of XYZZ points -- random coordinates (the formulas are identities of the coordinate ring), and every eighth
// pair one of: o = a, o = -a, o = O, a = O, o = a in another representation, o = -a in another representation -- added by the
// quad / octet form (DPP exchanges, and ds_bpermute exchanges as a cross-check) and by curve.hpp's one-lane form on the
// device and by the 32-bit-limb host arithmetic; plus sums over the points of a wave (XYZZ and affine sources).
template <class HF, class DF, bool EXT2>
int32_t run(zkmi_ctx* ctx, uint64_t seed, uint32_t n, uint32_t* out_bad, const char* name) {
  using PT = XYZZQ<Fq28, 0, EXT2>;
  using PTB = XYZZQ<Fq28, 1, EXT2>;
  constexpr uint32_t PW = PT::PER_WAVE;
  Rng rng{seed};
  std::vector<XYZZ<HF>> ha(n), ho(n);
  std::vector<Affine<HF>> haff(n);
  for (uint32_t i = 0; i < n; i++) {
    XYZZ<HF> a, o;
    rng.draw(a.x), rng.draw(a.y), rng.draw(a.zz), rng.draw(a.zzz);
    rng.draw(o.x), rng.draw(o.y), rng.draw(o.zz), rng.draw(o.zzz);
    HF lam;
    rng.draw(lam);
    const HF l2 = lam.sqr(), l3 = l2 * lam;
    switch (i % 8) {
      case 0: o = a; break;
      case 1: o = a.neg(); break;
      case 2: o = XYZZ<HF>::infinity(); break;
      case 3: a = XYZZ<HF>::infinity(); break;
      case 4: o = {a.x * l2, a.y * l3, a.zz * l2, a.zzz * l3}; break;
      case 5: o = {a.x * l2, (a.y * l3).neg(), a.zz * l2, a.zzz * l3}; break;
      default: break;
    }
    if (i % 64 == 6) a = o = XYZZ<HF>::infinity();
    ha[i] = a;
    ho[i] = o;
    rng.draw(haff[i].x), rng.draw(haff[i].y);
    const uint32_t q = i % PW;
    if (q == PW / 2 + 1) haff[i] = Affine<HF>::infinity();
    if (q == PW / 2 + 3) haff[i] = haff[i - PW / 2];        // the tree's first level adds points q and q + PER_WAVE / 2 (same sign): a doubling
    if (i % (2 * PW) == PW / 2 + 2 + (PW == 8 ? 0 : 3)) haff[i] = haff[i - PW / 2].neg();  // P + (-P) at the same level
  }
  std::vector<XYZZ<DF>> da(n), dob(n);
  std::vector<Affine<DF>> daff(n);
  for (uint32_t i = 0; i < n; i++) {
    da[i] = to_dev<HF, DF>(ha[i]);
    dob[i] = to_dev<HF, DF>(ho[i]);
    daff[i] = {fq28_from_fq(haff[i].x), fq28_from_fq(haff[i].y)};
    if (haff[i].is_inf()) daff[i] = Affine<DF>::infinity();
  }
  XYZZ<DF>*ga = nullptr, *go = nullptr, *g1 = nullptr, *gq = nullptr, *gb = nullptr, *gs = nullptr;
  Affine<DF>* gaff = nullptr;
  const size_t bytes = sizeof(XYZZ<DF>) * n;
  int32_t rc = ZKMI_OK;
  uint32_t bad = 0;
  std::vector<XYZZ<DF>> r1(n), rq(n), rb(n), rs(2 * (n / PW));
  if (hipMalloc(&ga, bytes) != hipSuccess || hipMalloc(&go, bytes) != hipSuccess || hipMalloc(&g1, bytes) != hipSuccess ||
      hipMalloc(&gq, bytes) != hipSuccess || hipMalloc(&gb, bytes) != hipSuccess || hipMalloc(&gs, sizeof(XYZZ<DF>) * 2 * (n / PW)) != hipSuccess ||
      hipMalloc(&gaff, sizeof(Affine<DF>) * n) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  if (hipMemcpy(ga, da.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(go, dob.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(gaff, daff.data(), sizeof(Affine<DF>) * n, hipMemcpyHostToDevice) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  hipLaunchKernelGGL(k_add_one_lane<DF>, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, g1, n);
  hipLaunchKernelGGL((k_add_quad<PT, false>), dim3((PT::LANES * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gq, n, (Fq28*)nullptr);
  hipLaunchKernelGGL((k_add_quad<PTB, false>), dim3((PT::LANES * n + 63) / 64), dim3(64), 0, ctx->stream, ga, go, gb, n, (Fq28*)nullptr);
  hipLaunchKernelGGL(k_wave_sum<PT>, dim3(n / PW), dim3(64), 0, ctx->stream, ga, gaff, gs);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(r1.data(), g1, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rq.data(), gq, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(rb.data(), gb, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(rs.data(), gs, sizeof(XYZZ<DF>) * 2 * (n / PW), hipMemcpyDeviceToHost) != hipSuccess) {
    rc = ZKMI_ERR_HIP;
    goto out;
  }
  {
    uint32_t bq[8] = {0}, bb[8] = {0}, b1[8] = {0}, coord[4] = {0, 0, 0, 0}, wsum[2] = {0, 0};
    for (uint32_t i = 0; i < n; i++) {
      XYZZ<HF> h = ha[i];
      h.add(ho[i]);
      if (!same<HF, DF>(rq[i], h)) {
        bad++, bq[i % 8]++;
        coord[0] += !(fq_from_fq28(rq[i].x) == h.x);
        coord[1] += !(fq_from_fq28(rq[i].y) == h.y);
        coord[2] += !(fq_from_fq28(rq[i].zz) == h.zz);
        coord[3] += !(fq_from_fq28(rq[i].zzz) == h.zzz);
      }
      if (!same<HF, DF>(rb[i], h)) bad++, bb[i % 8]++;
      if (!same<HF, DF>(r1[i], h)) bad++, b1[i % 8]++;
    }
    for (uint32_t w = 0; w < n / PW; w++) {
      // the tree order of wave_quad_sum: s = PER_WAVE / 2 ... 1 over the points (the formulas are not associative on
      // non-curve coordinates, so the host follows the same order)
      XYZZ<HF> t[16], u[16];
      for (uint32_t q = 0; q < PW; q++) {
        t[q] = ha[PW * w + q];
        Affine<HF> p = haff[PW * w + q];
        if (q & 1) p = p.neg();
        u[q] = XYZZ<HF>::from_affine(p);
      }
      for (uint32_t s = PW / 2; s >= 1; s >>= 1)
        for (uint32_t q = 0; q < s; q++) {
          t[q].add(t[q + s]);
          u[q].add(u[q + s]);
        }
      const bool e1 = !same<HF, DF>(rs[2 * w], t[0]), e2 = !same<HF, DF>(rs[2 * w + 1], u[0]);
      bad += (e1 ? 1 : 0) + (e2 ? 1 : 0);
      wsum[0] += e1, wsum[1] += e2;
    }
    if (bad && zkmi::debug_level()) {
      for (int c = 0; c < 8; c++)
        fprintf(stderr, "zkmi_selftest_quad_add[%s]: case %d: DPP form %u wrong, bpermute form %u wrong, one-lane form %u wrong\n", name, c, bq[c], bb[c], b1[c]);
      fprintf(stderr, "zkmi_selftest_quad_add[%s]: DPP form wrong coordinates: x %u, y %u, zz %u, zzz %u; wave sums wrong: XYZZ %u, affine %u of %u\n", name,
              coord[0], coord[1], coord[2], coord[3], wsum[0], wsum[1], n / PW);
    }
  }
out:
  (void)hipFree(ga);
  (void)hipFree(go);
  (void)hipFree(g1);
  (void)hipFree(gq);
  (void)hipFree(gb);
  (void)hipFree(gs);
  (void)hipFree(gaff);
  *out_bad += bad;
  return rc;
}
}  // namespace

extern "C" int32_t zkmi_selftest_quad_add(zkmi_ctx* ctx, uint64_t seed, uint32_t n, uint32_t* out_mismatches) {
  if (!ctx || !out_mismatches || n == 0 || n > (1u << 20)) return ZKMI_ERR_BAD_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return ZKMI_ERR_HIP;
  n = (n + 31u) & ~31u;
  *out_mismatches = 0;
  int32_t rc = run<Fq, Fq28, false>(ctx, seed, n, out_mismatches, "G1 quads");
  if (rc != ZKMI_OK) return rc;
  return run<Fq2, Fq2_28, true>(ctx, seed + 1, n, out_mismatches, "G2 octets");
}
#endif  // ZKMI_TESTING
