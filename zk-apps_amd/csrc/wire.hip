// zkmi — wire encodings at the C ABI (SURVEY.md §8b "Data conventions"):
// little-endian canonical field elements, affine points, and the zcash/IETF
// compressed point format used for proofs.  Host-only code.
//
// Reference anchors: Scalar{bytes:[u8;32]} little-endian
// (shielder/mocked_zk/src/scalar.rs:1-30).  The compressed format restates the
// public BLS12-381 serialisation (bit 7 compressed, bit 6 infinity, bit 5 =
// y lexicographically larger; G2: x.c1 || x.c0) — not present in the reference.
#include <string.h>
#include "ctx.hpp"

namespace zkmi {

template <class P>
static bool fp_from_wire(const uint8_t* b, Fp<P>* out) {
  Fp<P> a;
  memcpy(a.l, b, sizeof(a.l));
  // canonical: a < modulus
  bool lt = false;
  for (int i = P::N - 1; i >= 0; i--) {
    if (a.l[i] != P::MOD[i]) {
      lt = a.l[i] < P::MOD[i];
      break;
    }
  }
  if (!lt) return false;
  *out = a.to_mont();
  return true;
}
template <class P>
static void fp_to_wire(const Fp<P>& a, uint8_t* b) {
  Fp<P> c = a.from_mont();
  memcpy(b, c.l, sizeof(c.l));
}

bool fr_from_wire(const uint8_t* b, Fr* out) { return fp_from_wire(b, out); }
void fr_to_wire(const Fr& a, uint8_t* b) { fp_to_wire(a, b); }
bool fq_from_wire(const uint8_t* b, Fq* out) { return fp_from_wire(b, out); }
void fq_to_wire(const Fq& a, uint8_t* b) { fp_to_wire(a, b); }
bool fr_is_canonical(const uint8_t* b) {
  Fr t;
  return fp_from_wire(b, &t);
}

static Fq fq_from_u32(uint32_t v) {
  Fq a = Fq::zero();
  a.l[0] = v;
  return a.to_mont();
}

bool g1_on_curve(const G1Affine& p) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + fq_from_u32(4);
}
bool g2_on_curve(const G2Affine& p) {
  if (p.is_inf()) return true;
  Fq four = fq_from_u32(4);
  Fq2 b = {four, four};
  return p.y.sqr() == p.x.sqr() * p.x + b;
}

static bool all_zero(const uint8_t* b, size_t n) {
  uint8_t acc = 0;
  for (size_t i = 0; i < n; i++) acc |= b[i];
  return acc == 0;
}

bool g1_from_wire(const uint8_t* b, G1Affine* out, bool check_curve) {
  if (all_zero(b, 96)) {
    *out = G1Affine::infinity();
    return true;
  }
  if (!fq_from_wire(b, &out->x) || !fq_from_wire(b + 48, &out->y)) return false;
  return !check_curve || g1_on_curve(*out);
}
void g1_to_wire(const G1Affine& p, uint8_t* b) {
  if (p.is_inf()) {
    memset(b, 0, 96);
    return;
  }
  fq_to_wire(p.x, b);
  fq_to_wire(p.y, b + 48);
}
bool g2_from_wire(const uint8_t* b, G2Affine* out, bool check_curve) {
  if (all_zero(b, 192)) {
    *out = G2Affine::infinity();
    return true;
  }
  if (!fq_from_wire(b, &out->x.c0) || !fq_from_wire(b + 48, &out->x.c1) || !fq_from_wire(b + 96, &out->y.c0) ||
      !fq_from_wire(b + 144, &out->y.c1))
    return false;
  return !check_curve || g2_on_curve(*out);
}
void g2_to_wire(const G2Affine& p, uint8_t* b) {
  if (p.is_inf()) {
    memset(b, 0, 192);
    return;
  }
  fq_to_wire(p.x.c0, b);
  fq_to_wire(p.x.c1, b + 48);
  fq_to_wire(p.y.c0, b + 96);
  fq_to_wire(p.y.c1, b + 144);
}

static void hex48_le(const char* hex, uint8_t out[48]) {
  // hex is 96 big-endian hex digits
  for (int i = 0; i < 48; i++) {
    auto nib = [](char c) -> uint8_t { return c <= '9' ? c - '0' : c - 'a' + 10; };
    out[47 - i] = (uint8_t)((nib(hex[2 * i]) << 4) | nib(hex[2 * i + 1]));
  }
}

G1Affine g1_generator() {
  uint8_t w[96];
  hex48_le("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", w);
  hex48_le("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", w + 48);
  G1Affine g;
  g1_from_wire(w, &g, false);
  return g;
}
G2Affine g2_generator() {
  uint8_t w[192];
  hex48_le("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8", w);
  hex48_le("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e", w + 48);
  hex48_le("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801", w + 96);
  hex48_le("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be", w + 144);
  G2Affine g;
  g2_from_wire(w, &g, false);
  return g;
}

// ---- square roots (p = 3 mod 4) -------------------------------------------
static void fq_exp_limbs(uint32_t* e, int add, int shift) {
  // e = (p + add) >> shift, add in {-3, +1, -1}
  int64_t carry = add;
  uint32_t t[12];
  for (int i = 0; i < 12; i++) {
    int64_t v = (int64_t)FqParams::MOD[i] + carry;
    t[i] = (uint32_t)v;
    carry = v >> 32;
  }
  for (int i = 0; i < 12; i++) {
    uint64_t v = t[i];
    if (i + 1 < 12) v |= (uint64_t)t[i + 1] << 32;
    e[i] = (uint32_t)(v >> shift);
  }
}

static bool fq_sqrt(const Fq& a, Fq* out) {
  uint32_t e[12];
  fq_exp_limbs(e, 1, 2);  // (p+1)/4
  Fq s = a.pow(e, 12);
  if (s.sqr() != a) return false;
  *out = s;
  return true;
}

static Fq2 fq2_pow(const Fq2& a, const uint32_t* e, int n) {
  Fq2 res = Fq2::one();
  for (int i = n - 1; i >= 0; i--)
    for (int b = 31; b >= 0; b--) {
      res = res.sqr();
      if ((e[i] >> b) & 1) res = res * a;
    }
  return res;
}

static bool fq2_sqrt(const Fq2& a, Fq2* out) {
  // Adj & Rodriguez-Henriquez, algorithm 9 (p = 3 mod 4)
  if (a.is_zero()) {
    *out = a;
    return true;
  }
  uint32_t e34[12], e12[12];
  fq_exp_limbs(e34, -3, 2);  // (p-3)/4
  fq_exp_limbs(e12, -1, 1);  // (p-1)/2
  Fq2 a1 = fq2_pow(a, e34, 12);
  Fq2 alpha = a1 * (a1 * a);
  Fq2 a0 = alpha.conj() * alpha;  // alpha^p = conj(alpha)
  Fq2 minus_one = Fq2::one().neg();
  if (a0 == minus_one) return false;
  Fq2 x0 = a1 * a;
  Fq2 res;
  if (alpha == minus_one) {
    Fq2 u = {Fq::zero(), Fq::one()};
    res = u * x0;
  } else {
    Fq2 b = fq2_pow(Fq2::one() + alpha, e12, 12);
    res = b * x0;
  }
  if (res.sqr() != a) return false;
  *out = res;
  return true;
}

static void be48(const Fq& a, uint8_t out[48]) {
  uint8_t le[48];
  fq_to_wire(a, le);
  for (int i = 0; i < 48; i++) out[i] = le[47 - i];
}
static bool from_be48(const uint8_t in[48], uint8_t mask_top, Fq* out) {
  uint8_t le[48];
  for (int i = 0; i < 48; i++) le[i] = in[47 - i];
  le[47] &= mask_top;
  return fq_from_wire(le, out);
}

void g1_compress(const G1Affine& p, uint8_t out[48]) {
  if (p.is_inf()) {
    memset(out, 0, 48);
    out[0] = 0xC0;
    return;
  }
  be48(p.x, out);
  out[0] |= 0x80;
  if (p.y.lex_larger()) out[0] |= 0x20;
}
// Subgroup membership [r]P = O (host, O(255) group operations).  Curve points outside the
// r-order subgroups exist on both curves (cofactors h1, h2 > 1); arkworks' validating
// deserialisation and the zcash format both reject them, and the Miller loop is only defined on
// the subgroups.
bool g1_in_subgroup(const G1Affine& p) {
  if (p.is_inf()) return true;
  return scalar_mul(G1XYZZ::from_affine(p), FrParams::MOD, 8).is_inf();
}
bool g2_in_subgroup(const G2Affine& p) {
  if (p.is_inf()) return true;
  return scalar_mul(G2XYZZ::from_affine(p), FrParams::MOD, 8).is_inf();
}

bool g1_decompress(const uint8_t in[48], G1Affine* out) {
  if (!(in[0] & 0x80)) return false;
  if (in[0] & 0x40) {
    // the one canonical encoding of infinity: 0xC0 followed by zeros (sort flag clear)
    if (in[0] != 0xC0 || !all_zero(in + 1, 47)) return false;
    *out = G1Affine::infinity();
    return true;
  }
  Fq x, y;
  if (!from_be48(in, 0x1F, &x)) return false;
  if (!fq_sqrt(x.sqr() * x + fq_from_u32(4), &y)) return false;
  if (y.lex_larger() != (bool)(in[0] & 0x20)) y = y.neg();
  *out = {x, y};
  return true;
}
void g2_compress(const G2Affine& p, uint8_t out[96]) {
  if (p.is_inf()) {
    memset(out, 0, 96);
    out[0] = 0xC0;
    return;
  }
  be48(p.x.c1, out);
  be48(p.x.c0, out + 48);
  out[0] |= 0x80;
  if (p.y.lex_larger()) out[0] |= 0x20;
}
bool g2_decompress(const uint8_t in[96], G2Affine* out) {
  if (!(in[0] & 0x80)) return false;
  if (in[0] & 0x40) {
    if (in[0] != 0xC0 || !all_zero(in + 1, 95)) return false;
    *out = G2Affine::infinity();
    return true;
  }
  Fq2 x, y;
  if (!from_be48(in, 0x1F, &x.c1)) return false;
  if (!from_be48(in + 48, 0xFF, &x.c0)) return false;
  Fq four = fq_from_u32(4);
  Fq2 b = {four, four};
  if (!fq2_sqrt(x.sqr() * x + b, &y)) return false;
  if (y.lex_larger() != (bool)(in[0] & 0x20)) y = y.neg();
  *out = {x, y};
  return true;
}

// ---- uncompressed big-endian point forms (the Compress::No side of the zcash-style encoding) ----
// G1: x || y (96 B), G2: x.c1 || x.c0 || y.c1 || y.c0 (192 B); byte 0 carries the flags: bit 7 = 0
// (uncompressed), bit 6 = infinity (all other bits and bytes zero), bit 5 unused.
void g1_write_be(const G1Affine& p, uint8_t out[96]) {
  if (p.is_inf()) {
    memset(out, 0, 96);
    out[0] = 0x40;
    return;
  }
  be48(p.x, out);
  be48(p.y, out + 48);
}
bool g1_read_be(const uint8_t in[96], G1Affine* out, bool check_curve) {
  if (in[0] & 0x80) return false;
  if (in[0] & 0x40) {
    if (in[0] != 0x40 || !all_zero(in + 1, 95)) return false;
    *out = G1Affine::infinity();
    return true;
  }
  if (in[0] & 0x20) return false;
  if (!from_be48(in, 0xFF, &out->x) || !from_be48(in + 48, 0xFF, &out->y)) return false;
  return !check_curve || g1_on_curve(*out);
}
void g2_write_be(const G2Affine& p, uint8_t out[192]) {
  if (p.is_inf()) {
    memset(out, 0, 192);
    out[0] = 0x40;
    return;
  }
  be48(p.x.c1, out);
  be48(p.x.c0, out + 48);
  be48(p.y.c1, out + 96);
  be48(p.y.c0, out + 144);
}
bool g2_read_be(const uint8_t in[192], G2Affine* out, bool check_curve) {
  if (in[0] & 0x80) return false;
  if (in[0] & 0x40) {
    if (in[0] != 0x40 || !all_zero(in + 1, 191)) return false;
    *out = G2Affine::infinity();
    return true;
  }
  if (in[0] & 0x20) return false;
  if (!from_be48(in, 0xFF, &out->x.c1) || !from_be48(in + 48, 0xFF, &out->x.c0) || !from_be48(in + 96, 0xFF, &out->y.c1) ||
      !from_be48(in + 144, 0xFF, &out->y.c0))
    return false;
  return !check_curve || g2_on_curve(*out);
}

}  // namespace zkmi
