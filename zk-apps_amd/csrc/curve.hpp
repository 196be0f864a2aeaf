// zkmi — short-Weierstrass (a = 0) group arithmetic for BLS12-381 G1 (F = Fq)
// and G2 (F = Fq2) in extended-Jacobian "XYZZ" coordinates:
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2;   infinity <=> ZZ == 0.
// Affine infinity is encoded as (0, 0) (not on either curve since b != 0).
//
// Reference locus: none in /root/reference; the
// bucket method only needs a correct group law — the MSM result is a unique
// group element, compared after affine normalisation.
// Formulas: EFD "xyzz" madd-2008-s (8M+2S), add-2008-s (12M+2S), dbl-2008-s-1.
#pragma once
#include <stddef.h>
#include <vector>
#include "field.hpp"

namespace zkmi {

// Building blocks of the XYZZ addition formulas.  The generic forms are plain
// field expressions; field28.hpp overloads them for the lazily reduced device
// limbs (skipped carry sweeps, two products under one Montgomery reduction).
template <class F>
ZK_HD F f_sub_lazy(const F& a, const F& b) { return a - b; }  // result only feeds products
template <class F>
ZK_HD F f_x3(const F& rr, const F& ppp, const F& q) { return rr - ppp - q.dbl(); }
template <class F>
ZK_HD F f_mul_sub_mul(const F& a, const F& b, const F& c, const F& d) { return a * b - c * d; }

template <class F>
struct Affine {
  F x, y;
  ZK_HD static Affine infinity() { return {F::zero(), F::zero()}; }
  ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  ZK_HD Affine neg() const { return {x, y.neg()}; }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  ZK_HD static XYZZ infinity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZK_HD bool is_inf() const { return zz.is_zero(); }
  ZK_HD static XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return infinity();
    return {p.x, p.y, F::one(), F::one()};
  }
  ZK_HD XYZZ neg() const { return {x, y.neg(), zz, zzz}; }

  // rare-path wrappers kept out of line so the hot add path stays compact
  __host__ __device__ __attribute__((noinline)) static XYZZ dbl_affine_slow(Affine<F> p) { return dbl_affine(p); }
  __host__ __device__ __attribute__((noinline)) static XYZZ dbl_slow(XYZZ p) {
    p.dbl_inplace();
    return p;
  }

  ZK_HD static XYZZ dbl_affine(const Affine<F>& p) {
    // mdbl-2008-s-1 with ZZ1 = ZZZ1 = 1
    F u = p.y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = p.x * v;
    F x2 = p.x.sqr();
    F m = x2.dbl() + x2;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * p.y;
    return {x3, y3, v, w};
  }

  ZK_HD void dbl_inplace() {
    if (is_inf()) return;
    F u = y.dbl();
    F v = u.sqr();
    F w = u * v;
    F s = x * v;
    F x2 = x.sqr();
    F m = x2.dbl() + x2;
    F x3 = m.sqr() - s.dbl();
    F y3 = m * (s - x3) - w * y;
    x = x3;
    y = y3;
    zz = v * zz;
    zzz = w * zzz;
  }

  // this += p (affine), complete by case analysis
  ZK_HD void madd(const Affine<F>& p) {
    if (p.is_inf()) return;
    if (is_inf()) {
      x = p.x;
      y = p.y;
      zz = F::one();
      zzz = F::one();
      return;
    }
    // statement order keeps live ranges short (the accumulator, one point and the
    // 64-bit product columns must fit 256 VGPRs for 2 waves per SIMD)
    F pp_ = f_sub_lazy(p.x * zz, x);
    F r = f_sub_lazy(p.y * zzz, y);
    // zero tests are made on products (P = 0 <=> P^2 = 0): exact for every
    // field representation, including the lazily reduced device limbs
    F pp = pp_.sqr();
    F rr = r.sqr();
    if (pp.is_zero()) {
      if (rr.is_zero()) {
        *this = dbl_affine_slow(p);
      } else {
        *this = infinity();
      }
      return;
    }
    F ppp = pp_ * pp;
    zz = zz * pp;
    zzz = zzz * ppp;
    F q = x * pp;
    x = f_x3(rr, ppp, q);
    y = f_mul_sub_mul(r, f_sub_lazy(q, x), y, ppp);
  }

  // this += o
  ZK_HD void add(const XYZZ& o) {
    if (o.is_inf()) return;
    if (is_inf()) {
      *this = o;
      return;
    }
    F u1 = x * o.zz;
    F u2 = o.x * zz;
    F s1 = y * o.zzz;
    F s2 = o.y * zzz;
    F pp_ = f_sub_lazy(u2, u1);
    F r = f_sub_lazy(s2, s1);
    F pp = pp_.sqr();
    F rr = r.sqr();
    if (pp.is_zero()) {
      if (rr.is_zero()) {
        *this = dbl_slow(*this);
      } else {
        *this = infinity();
      }
      return;
    }
    F ppp = pp_ * pp;
    F q = u1 * pp;
    F x3 = f_x3(rr, ppp, q);
    y = f_mul_sub_mul(r, f_sub_lazy(q, x3), s1, ppp);
    x = x3;
    zz = zz * o.zz * pp;
    zzz = zzz * o.zzz * ppp;
  }

  __host__ __device__ Affine<F> to_affine() const {
    if (is_inf()) return Affine<F>::infinity();
    F zi = zzz.inv();            // 1/ZZZ
    F zz_inv = (zi * zz).sqr();  // (ZZ/ZZZ)^2 = 1/ZZ  (since ZZ^3 = ZZZ^2)
    return {x * zz_inv, y * zi};
  }
};

// k * p for a little-endian 32-bit-limb scalar (host-side O(1) steps only)
template <class F>
__host__ __device__ inline XYZZ<F> scalar_mul(const XYZZ<F>& p, const uint32_t* k, int nlimbs) {
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int i = nlimbs - 1; i >= 0; i--) {
    for (int b = 31; b >= 0; b--) {
      acc.dbl_inplace();
      if ((k[i] >> b) & 1) acc.add(p);
    }
  }
  return acc;
}

// ---- host-side helpers of proof assembly (groth16.hip assemble_proof) ----------------------------------
// 4-bit window of a little-endian 32-bit-limb scalar
inline uint32_t scalar_nibble(const uint32_t* k, int w) { return (k[w >> 3] >> ((w & 7) * 4)) & 15u; }

// k * P for a point fixed at key-load time (delta in G1 and G2): 64 windows x 15 multiples, no doublings.
template <class F>
struct FixedBase4 {
  XYZZ<F> tab[64][15];  // tab[w][d-1] = d * 16^w * P
  bool built = false;
  void build(const Affine<F>& p) {
    XYZZ<F> base = XYZZ<F>::from_affine(p);
    for (int w = 0; w < 64; w++) {
      tab[w][0] = base;
      for (int d = 1; d < 15; d++) {
        tab[w][d] = tab[w][d - 1];
        tab[w][d].add(base);
      }
      base = tab[w][14];
      base.add(tab[w][0]);  // 16 * (16^w P)
    }
    built = true;
  }
  XYZZ<F> mul(const uint32_t* k /* 8 limbs */) const {
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int w = 0; w < 64; w++) {
      const uint32_t d = scalar_nibble(k, w);
      if (d) acc.add(tab[w][d - 1]);
    }
    return acc;
  }
};

// The same with 8-bit windows and AFFINE table entries: 32 mixed additions (8M + 2S) per multiplication instead of 64
// complete ones (12M + 2S) -- proof assembly is what a rank's few host CPUs spend their time on when small proofs travel in
// groups (2 700 proofs/s per GPU at 2^14: DESIGN.md section 5), and three of its four scalar multiplications are by points
// fixed at key-load time.  32 x 255 points: 0.78 MB (G1) / 1.57 MB (G2) of host memory per table.
template <class F>
void batch_to_affine(const XYZZ<F>* pts, size_t n, Affine<F>* out);
template <class F>
struct FixedBase8 {
  std::vector<Affine<F>> tab;  // tab[w * 255 + d - 1] = d * 256^w * P
  void build(const Affine<F>& p) {
    std::vector<XYZZ<F>> t(32 * 255);
    XYZZ<F> base = XYZZ<F>::from_affine(p);
    for (int w = 0; w < 32; w++) {
      XYZZ<F>* row = t.data() + (size_t)w * 255;
      row[0] = base;
      for (int d = 1; d < 255; d++) {
        row[d] = row[d - 1];
        row[d].add(base);
      }
      base = row[254];
      base.add(row[0]);  // 256 * (256^w P)
    }
    tab.resize(t.size());
    batch_to_affine(t.data(), t.size(), tab.data());
  }
  XYZZ<F> mul(const uint32_t* k /* 8 limbs */) const {
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (int w = 0; w < 32; w++) {
      const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 255u;
      if (d) acc.madd(tab[(size_t)w * 255 + d - 1]);
    }
    return acc;
  }
};

// One shared inversion for n points (Montgomery's trick on the zzz coordinates; Fq2 inverts through its norm, so a batch
// costs ONE base-field inversion either way): a group of 64 proofs normalises its 192 proof elements together.
template <class F>
void batch_to_affine(const XYZZ<F>* pts, size_t n, Affine<F>* out) {
  std::vector<F> pre(n);
  F run = F::one();
  for (size_t i = 0; i < n; i++) {
    pre[i] = run;
    if (!pts[i].is_inf()) run = run * pts[i].zzz;
  }
  F inv = run.inv();
  for (size_t i = n; i-- > 0;) {
    if (pts[i].is_inf()) {
      out[i] = Affine<F>::infinity();
      continue;
    }
    const F zi = inv * pre[i];  // 1 / zzz_i
    inv = inv * pts[i].zzz;
    const F zz_inv = (zi * pts[i].zz).sqr();  // (zz / zzz)^2 = 1 / zz
    out[i] = {pts[i].x * zz_inv, pts[i].y * zi};
  }
}

// k * P for ONE variable point: 4-bit windows, 255 doublings + <= 64 additions (plain double-and-add: + ~128)
template <class F>
inline XYZZ<F> scalar_mul_w4(const XYZZ<F>& p, const uint32_t* k) {
  XYZZ<F> tp[15];
  tp[0] = p;
  for (int d = 1; d < 15; d++) {
    tp[d] = tp[d - 1];
    tp[d].add(p);
  }
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int w = 63; w >= 0; w--) {
    for (int i = 0; i < 4; i++) acc.dbl_inplace();
    const uint32_t d = scalar_nibble(k, w);
    if (d) acc.add(tp[d - 1]);
  }
  return acc;
}

// a * P + b * Q for 256-bit scalars: one doubling chain, 4-bit windows (Straus)
template <class F>
inline XYZZ<F> scalar_mul2(const XYZZ<F>& p, const uint32_t* a, const XYZZ<F>& q, const uint32_t* b) {
  XYZZ<F> tp[15], tq[15];
  tp[0] = p;
  tq[0] = q;
  for (int d = 1; d < 15; d++) {
    tp[d] = tp[d - 1];
    tp[d].add(p);
    tq[d] = tq[d - 1];
    tq[d].add(q);
  }
  XYZZ<F> acc = XYZZ<F>::infinity();
  for (int w = 63; w >= 0; w--) {
    for (int i = 0; i < 4; i++) acc.dbl_inplace();
    const uint32_t da = scalar_nibble(a, w), db = scalar_nibble(b, w);
    if (da) acc.add(tp[da - 1]);
    if (db) acc.add(tq[db - 1]);
  }
  return acc;
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace zkmi
