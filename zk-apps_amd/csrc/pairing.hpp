// zkmi — Fq6 / Fq12 tower and the reduced optimal-ate pairing on BLS12-381
// (host only; SURVEY.md §8a row a11: O(1) per proof, CPU is sufficient).
#pragma once
#include "curve.hpp"

namespace zkmi {

// Fq6 = Fq2[v]/(v^3 - xi), xi = 1 + u
struct Fq6 {
  Fq2 a0, a1, a2;
  static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  bool operator==(const Fq6& o) const { return a0 == o.a0 && a1 == o.a1 && a2 == o.a2; }
  friend Fq6 operator+(const Fq6& a, const Fq6& b) { return {a.a0 + b.a0, a.a1 + b.a1, a.a2 + b.a2}; }
  friend Fq6 operator-(const Fq6& a, const Fq6& b) { return {a.a0 - b.a0, a.a1 - b.a1, a.a2 - b.a2}; }
  Fq6 neg() const { return {a0.neg(), a1.neg(), a2.neg()}; }
  friend Fq6 operator*(const Fq6& a, const Fq6& b);
  Fq6 mul_v() const { return {a2.mul_xi(), a0, a1}; }
  Fq6 inv() const;
};

// Fq12 = Fq6[w]/(w^2 - v)
struct Fq12 {
  Fq6 c0, c1;
  static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
  bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
  friend Fq12 operator*(const Fq12& a, const Fq12& b);
  Fq12 sqr() const { return (*this) * (*this); }
  Fq12 conj() const { return {c0, c1.neg()}; }
  Fq12 inv() const;
  Fq12 pow(const uint32_t* e, int nlimbs) const;
};

Fq12 miller_loop(const G1Affine& p, const G2Affine& q);
Fq12 final_exponentiation(const Fq12& f);
inline Fq12 pairing(const G1Affine& p, const G2Affine& q) { return final_exponentiation(miller_loop(p, q)); }
void fq12_to_wire(const Fq12& f, uint8_t out[576]);

}  // namespace zkmi
