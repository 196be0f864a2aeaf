// zkmi — reduced optimal-ate pairing on BLS12-381 (host CPU), used by the
// Groth16 verifier.  The reference's verify locus is
// the mock ZkProof::verify_update (shielder/mocked_zk/src/relations.rs:138-155);
// a real verifier replaces its hash recomputation with this pairing check.
//
// e(P, Q) = conj( f_{|x|,Q}(P) )^((p^12-1)/r),  x = -0xd201000000010000.
// Lines are evaluated on the M-twist in affine coordinates and embedded as the
// sparse element  -yP*xi + (yT - l*xT) v w + (l*xP) v^2 w  (the common factor
// xi in Fq2 is killed by the final exponentiation).
#include <string.h>
#include "ctx.hpp"
#include "pairing.hpp"

namespace zkmi {

Fq6 operator*(const Fq6& a, const Fq6& b) {
  Fq2 t0 = a.a0 * b.a0, t1 = a.a1 * b.a1, t2 = a.a2 * b.a2;
  Fq2 c0 = t0 + ((a.a1 + a.a2) * (b.a1 + b.a2) - t1 - t2).mul_xi();
  Fq2 c1 = (a.a0 + a.a1) * (b.a0 + b.a1) - t0 - t1 + t2.mul_xi();
  Fq2 c2 = (a.a0 + a.a2) * (b.a0 + b.a2) - t0 - t2 + t1;
  return {c0, c1, c2};
}

Fq6 Fq6::inv() const {
  Fq2 c0 = a0.sqr() - (a1 * a2).mul_xi();
  Fq2 c1 = a2.sqr().mul_xi() - a0 * a1;
  Fq2 c2 = a1.sqr() - a0 * a2;
  Fq2 t = (a0 * c0 + (a2 * c1 + a1 * c2).mul_xi()).inv();
  return {c0 * t, c1 * t, c2 * t};
}

Fq12 operator*(const Fq12& a, const Fq12& b) {
  Fq6 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
  Fq6 c1 = (a.c0 + a.c1) * (b.c0 + b.c1) - t0 - t1;
  return {t0 + t1.mul_v(), c1};
}

Fq12 Fq12::inv() const {
  Fq6 t = (c0 * c0 - (c1 * c1).mul_v()).inv();
  return {c0 * t, (c1 * t).neg()};
}

Fq12 Fq12::pow(const uint32_t* e, int nlimbs) const {
  Fq12 res = Fq12::one();
  bool started = false;
  for (int i = nlimbs - 1; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      if (started) res = res.sqr();
      if ((e[i] >> b) & 1) {
        res = started ? res * (*this) : *this;
        started = true;
      }
    }
  return res;
}

static Fq12 line_eval(const Fq2& lam, const G2Affine& t, const G1Affine& p) {
  Fq12 l;
  l.c0 = {Fq2{p.y.neg(), Fq::zero()}.mul_xi(), Fq2::zero(), Fq2::zero()};
  l.c1 = {Fq2::zero(), t.y - lam * t.x, lam.mul_fq(p.x)};
  return l;
}

Fq12 miller_loop(const G1Affine& p, const G2Affine& q) {
  if (p.is_inf() || q.is_inf()) return Fq12::one();
  const uint64_t X_ABS = 0xd201000000010000ull;
  G2Affine t = q;
  Fq12 f = Fq12::one();
  for (int b = 62; b >= 0; b--) {
    // tangent at T
    Fq2 x2 = t.x.sqr();
    Fq2 lam = (x2.dbl() + x2) * t.y.dbl().inv();
    f = f.sqr() * line_eval(lam, t, p);
    Fq2 x3 = lam.sqr() - t.x.dbl();
    Fq2 y3 = lam * (t.x - x3) - t.y;
    t = {x3, y3};
    if ((X_ABS >> b) & 1) {
      Fq2 lam2 = (q.y - t.y) * (q.x - t.x).inv();
      f = f * line_eval(lam2, t, p);
      Fq2 x4 = lam2.sqr() - t.x - q.x;
      Fq2 y4 = lam2 * (t.x - x4) - t.y;
      t = {x4, y4};
    }
  }
  return f.conj();
}

static const uint32_t P_SQUARED[24] = {
    0x1c718e39u, 0x26aa0000u, 0x76382eabu, 0x7ced6b1du, 0x62113cfdu, 0x162c3383u,
    0x3e71b743u, 0x66bf91edu, 0x7091a049u, 0x292e85a8u, 0x86185c7bu, 0x1d68619cu,
    0x0978ef01u, 0xf5314933u, 0x16ddca6eu, 0x50a62cfdu, 0x349e8bd0u, 0x66e59e49u,
    0x0e7046b4u, 0xe2dc90e5u, 0xa22f25e9u, 0x4bd278eau, 0xb8c35fc7u, 0x02a437a4u,
};
// (p^4 - p^2 + 1) / r
static const uint32_t HARD_EXP[40] = {
    0x38e3ba79u, 0xe516c3f4u, 0xe208ccf1u, 0xfa9912aau, 0x335d5b68u, 0x905ce937u,
    0xb0dea236u, 0xc71a2629u, 0x996754c8u, 0x83774940u, 0xb6a1e799u, 0x21d160aeu,
    0xed237db4u, 0x2ed0b283u, 0x6c6f1821u, 0x915c97f3u, 0xde783765u, 0x67f17fcbu,
    0x9096d1b7u, 0x2378b903u, 0x1bdc51dcu, 0x7988f876u, 0x03fc77a1u, 0x20769950u,
    0xa621315bu, 0x827eca0bu, 0x8d63cb9fu, 0xe5a72bceu, 0xc28b6f8au, 0xf68f7764u,
    0xcf081517u, 0x2f230063u, 0x528d6a9au, 0x94506632u, 0xeb996ca3u, 0xd3cde88eu,
    0x195c899eu, 0xc0bd38c3u, 0x3d807d01u, 0x000f686bu,
};

Fq12 final_exponentiation(const Fq12& f) {
  Fq12 f1 = f.conj() * f.inv();               // f^(p^6 - 1)
  Fq12 f2 = f1.pow(P_SQUARED, 24) * f1;       // ^(p^2 + 1)
  return f2.pow(HARD_EXP, 40);                // ^((p^4 - p^2 + 1)/r)
}

void fq12_to_wire(const Fq12& f, uint8_t out[576]) {
  const Fq6* c[2] = {&f.c0, &f.c1};
  int k = 0;
  for (int i = 0; i < 2; i++) {
    const Fq2* a[3] = {&c[i]->a0, &c[i]->a1, &c[i]->a2};
    for (int j = 0; j < 3; j++) {
      fq_to_wire(a[j]->c0, out + 48 * k++);
      fq_to_wire(a[j]->c1, out + 48 * k++);
    }
  }
}

}  // namespace zkmi

extern "C" int32_t zkmi_pairing(const uint8_t g1_affine[96], const uint8_t g2_affine[192], uint8_t out_fq12[576]) {
  using namespace zkmi;
  if (!g1_affine || !g2_affine || !out_fq12) return ZKMI_ERR_BAD_ARG;
  G1Affine p;
  G2Affine q;
  if (!g1_from_wire(g1_affine, &p, true) || !g2_from_wire(g2_affine, &q, true)) return ZKMI_ERR_NON_CANONICAL;
  fq12_to_wire(pairing(p, q), out_fq12);
  return ZKMI_OK;
}
