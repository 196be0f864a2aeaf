// zkmi — the complete XYZZ group law with the FORMULA split across a lane quad (device only).
//
// Why: XYZZ + XYZZ in one lane (curve.hpp add / msm_impl.hpp add_generic) keeps two points, the products' 64-bit columns
// and four temporaries live: 230-320 VGPRs.  Every kernel built on it (segment sums, tree sums, redo pass, the summing of
// heavy-bucket partials) is a one-wave-per-SIMD kernel that is not placed while accumulation waves hold 504 of a SIMD's 512
// registers, and its additions are chains of 14 dependent field products.
//
// Here lane q of a quad (threadIdx.x & 3) holds coordinate q of a point -- 0: X, 1: Y, 2: ZZ, 3: ZZZ -- and the addition
// add-2008-s (12M + 2S) runs as FOUR rounds of one product per lane, operands exchanged with DPP quad_perm moves:
//
//   round 1   m1 = a * rot2(o)            lane 0: U1 = X1 ZZ2    1: S1 = Y1 ZZZ2   2: U2 = ZZ1 X2    3: S2 = ZZZ1 Y2
//   round 2   d = rot2(m1) - m1           lane 0: P = U2 - U1    1: R = S2 - S1
//             m2 = {d d, d d, a o, a o}   lane 0: PP             1: RR             2: ZZ1 ZZ2        3: ZZZ1 ZZZ2
//   round 3   m3 = {d, U1, m2, m2} * PP   lane 0: PPP            1: Q = U1 PP      2: ZZ3            3: T = ZZZ1 ZZZ2 PP
//   round 4   X3 = RR - PPP - 2 Q (lanes 0 and 1, no product)
//             m4 = {-, R (Q - X3), S1 PPP, T P}                  1: R (Q - X3)     2: S1 PPP         3: ZZZ3
//             Y3 = m4[1] - m4[2]
//
// A lane's state is ONE coordinate of each point (14 limbs instead of 56) and the chain is 4 products long instead of 14;
// the quad issues 16 products for the 14 the formula needs.  The rare doubling case (o = a) is three more rounds in the same
// style (dbl-2008-s-1) instead of a call, so kernels built on this stay call-free and at the accumulation kernels' register
// count: they are placed beside them.  All branches are quad-uniform (flags are broadcast inside the quad before they are
// tested), which is what DPP needs: a disabled source lane would read as zero.
//
// Value ranges (field28.hpp): products leave (-p/2, 3p/2); the lazy differences d, Q - X3 have limbs below 2^29 and only
// feed products; X3 and Y3 are carried differences of at most four products (|v| < 6p < 16p).
#pragma once
#include "curve.hpp"
#include "field28.hpp"
#include <type_traits>

#if defined(__HIPCC__)
namespace zkmi {

// quad_perm selectors: lane i of a quad reads lane S_i; encoded S_0 | S_1 << 2 | S_2 << 4 | S_3 << 6
constexpr int QP_ROT2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);   // 2 3 0 1
constexpr int QP_SWAP = 1 | (0 << 2) | (3 << 4) | (2 << 6);   // 1 0 3 2
constexpr int QP_B0 = 0;                                      // 0 0 0 0
constexpr int QP_B1 = 1 | (1 << 2) | (1 << 4) | (1 << 6);     // 1 1 1 1
constexpr int QP_B2 = 2 | (2 << 2) | (2 << 4) | (2 << 6);     // 2 2 2 2
constexpr int QP_0023 = 0 | (0 << 2) | (2 << 4) | (3 << 6);   // lane 1 reads lane 0
constexpr int QP_0110 = 0 | (1 << 2) | (1 << 4) | (0 << 6);   // lane 2 reads lane 1, lane 3 reads lane 0
constexpr int QP_0223 = 0 | (2 << 2) | (2 << 4) | (3 << 6);   // lane 1 reads lane 2
constexpr int QP_0003 = 0 | (0 << 2) | (0 << 4) | (3 << 6);   // lanes 1, 2 read lane 0
constexpr int QP_3103 = 3 | (1 << 2) | (0 << 4) | (3 << 6);   // lane 0 reads lane 3, lane 2 reads lane 0

template <int CTRL, class F>
__device__ __forceinline__ F quad_get(const F& a) {
  F r;
#pragma unroll
  for (int i = 0; i < F::NL; i++) r.l[i] = __builtin_amdgcn_mov_dpp(a.l[i], CTRL, 0xF, 0xF, true);
  return r;
}
template <int CTRL>
__device__ __forceinline__ bool quad_flag(bool f) {
  return __builtin_amdgcn_mov_dpp(f ? 1 : 0, CTRL, 0xF, 0xF, true) != 0;
}
template <class F>
__device__ __forceinline__ F quad_sel(bool c, const F& a, const F& b) {
  F r;
#pragma unroll
  for (int i = 0; i < F::NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// keeps the moves that produced `a` as instructions of their own (see XYZZQ::add, Y3)
template <class F>
__device__ __forceinline__ void dpp_fence(F& a) {
#pragma unroll
  for (int i = 0; i < F::NL; i++) asm volatile("" : "+v"(a.l[i]));
}

// one coordinate of an XYZZ point per lane of a quad
// (XCH = 1: the exchanges as ds_bpermute moves instead of DPP -- the self-test's cross-check of the DPP selectors)
//
// EXT2 = true: G2.  A point takes an OCTET of lanes: lanes 0-3 hold the c0 components of X, Y, ZZ, ZZZ and lanes 4-7 the c1
// components (Fq2 = Fq[u] / (u^2 + 1)); `v` is still ONE base-field value per lane.  The exchanges between coordinates are
// the same quad_perm moves (they act inside each quad, i.e. on one component); a field product becomes the lane's component
// of the Fq2 product, which needs the other component of both operands from lane ^ 4: two bank-masked DPP row shifts per limb
// (row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3).  Operands of an Fq2 lane product must be normalised (a column
// sums two partial products: no spare bit), so the differences that the G1 form leaves lazy are carried here.
template <class F, int XCH = 0, bool EXT2 = false>
struct XYZZQ {
  F v;
  static constexpr int LANES = EXT2 ? 8 : 4;          // lanes per point
  static constexpr int PER_WAVE = 64 / LANES;         // points per wave
  using Elem = typename std::conditional<EXT2, Fq2T<F>, F>::type;  // a coordinate in memory
  using Point = XYZZ<Elem>;
  using APoint = Affine<Elem>;
  template <int CTRL>
  __device__ __forceinline__ static F get(const F& a) {
    if constexpr (XCH == 0) {
      return quad_get<CTRL>(a);
    } else {
      const int lane = (int)(threadIdx.x & 63u), k = lane & 3;
      const int src = (lane & ~3) | ((CTRL >> (2 * k)) & 3);
      F r;
#pragma unroll
      for (int i = 0; i < F::NL; i++) r.l[i] = __shfl(a.l[i], src);
      return r;
    }
  }
  template <int CTRL>
  __device__ __forceinline__ static bool flag(bool f) {
    if constexpr (XCH == 0) {
      return quad_flag<CTRL>(f);
    } else {
      const int lane = (int)(threadIdx.x & 63u), k = lane & 3;
      return __shfl(f ? 1 : 0, (lane & ~3) | ((CTRL >> (2 * k)) & 3)) != 0;
    }
  }
  // the value lane ^ 4 holds (the other Fq2 component of the same coordinate); EXT2 only
  __device__ __forceinline__ static int other_word(int x) {
    if constexpr (XCH == 0) {
      const int t = __builtin_amdgcn_update_dpp(x, x, 0x104 /* row_shl:4 */, 0xF, 0x5, false);  // lanes 0-3, 8-11 <- lane + 4
      return __builtin_amdgcn_update_dpp(t, x, 0x114 /* row_shr:4 */, 0xF, 0xA, false);          // lanes 4-7, 12-15 <- lane - 4
    } else {
      return __shfl(x, (int)((threadIdx.x & 63u) ^ 4u));
    }
  }
  __device__ __forceinline__ static F other(const F& a) {
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; i++) r.l[i] = other_word(a.l[i]);
    return r;
  }
  __device__ __forceinline__ static uint32_t q() { return threadIdx.x & 3u; }
  __device__ __forceinline__ static uint32_t comp() { return EXT2 ? (threadIdx.x >> 2) & 1u : 0u; }
  // index of this lane's value among the base-field values of a point / affine point in memory
  __device__ __forceinline__ static uint32_t slot() { return EXT2 ? 2u * q() + comp() : q(); }
  __device__ __forceinline__ static uint32_t aslot() { return EXT2 ? 2u * (q() & 1u) + comp() : (q() & 1u); }

  // ---- the field operations of the formulas: base field, or this lane's component of Fq2 ----
  __device__ __forceinline__ static F fmul(const F& a, const F& b) {
    if constexpr (!EXT2) {
      return F::mul_inline(a, b);
    } else {
      // c0 = a0 b0 - a1 b1 (comp 0: own * own - other * other); c1 = a0 b1 + a1 b0 (comp 1: own * other + other * own):
      // a.v * X + other(a) * Y with X = comp ? other(b) : b, Y = comp ? b : -other(b); one reduction (field28.hpp Fq2P)
      constexpr int NL = F::NL;
      const bool c1 = comp() != 0u;
      int64_t T[2 * NL];
#pragma unroll
      for (int i = 0; i < 2 * NL; i++) T[i] = 0;
      const F pb = other(b);
      {
        const F X = quad_sel(c1, pb, b);
#pragma unroll
        for (int i = 0; i < NL; i++)
#pragma unroll
          for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * X.l[j];
      }
      {
        const F pa = other(a);
        F Y;
#pragma unroll
        for (int i = 0; i < NL; i++) Y.l[i] = c1 ? b.l[i] : -pb.l[i];
#pragma unroll
        for (int i = 0; i < NL; i++)
#pragma unroll
          for (int j = 0; j < NL; j++) T[i + j] += (int64_t)pa.l[i] * Y.l[j];
      }
      return F::reduce(T);
    }
  }
  // a - b / 2 a as operands of a product: lazy in the base field, carried for Fq2 lane products
  __device__ __forceinline__ static F fsub(const F& a, const F& b) {
    if constexpr (EXT2) return a - b;
    else return a.sub_lazy(b);
  }
  __device__ __forceinline__ static F fdbl(const F& a) {
    if constexpr (EXT2) return a.dbl();
    else return a.add_lazy(a);
  }
  // the element this lane holds a component of is zero (a product: exact, field28.hpp is_zero)
  // (the exchange is executed by ALL lanes, never behind a short-circuit: a lane that skipped it would be a disabled DPP
  // source for its partner -- the c1 component of ZZ = 1 is zero, so "mine && other" diverged inside every octet of an
  // affine point: round-6 self-test)
  __device__ __forceinline__ static bool fzero(const F& a) { return both(a.is_zero()); }
  __device__ __forceinline__ static bool both(bool f) {  // f on both components
    if constexpr (EXT2) {
      const int o = other_word(f ? 1 : 0);
      return f & (o != 0);
    } else {
      return f;
    }
  }
  __device__ __forceinline__ static bool either(bool f) {
    if constexpr (EXT2) {
      const int o = other_word(f ? 1 : 0);
      return f | (o != 0);
    } else {
      return f;
    }
  }
  __device__ __forceinline__ static F fone() { return comp() ? F::zero() : F::one(); }

  __device__ __forceinline__ static XYZZQ infinity() { return {F::zero()}; }
  // ZZ = 0 (lane 2 of each quad), known to all lanes of the point
  __device__ __forceinline__ bool is_inf() const { return flag<QP_B2>(fzero(v)); }
  // memory form: x | y | zz | zzz (Fq2: c0 | c1 inside each); the lanes of a point move one contiguous point
  __device__ __forceinline__ static F load_f(const F* s) {
    F r;
    constexpr int W = sizeof(F) / 8;
    const uint2* s2 = reinterpret_cast<const uint2*>(s);
#pragma unroll
    for (int i = 0; i < W; i++) {
      const uint2 w = s2[i];
      r.l[2 * i] = (int32_t)w.x;
      r.l[2 * i + 1] = (int32_t)w.y;
    }
    return r;
  }
  __device__ __forceinline__ static XYZZQ load(const Point* p) { return {load_f(reinterpret_cast<const F*>(p) + slot())}; }
  __device__ __forceinline__ void store(Point* p) const {
    F* d = reinterpret_cast<F*>(p) + slot();
    constexpr int W = sizeof(F) / 8;
    uint2* d2 = reinterpret_cast<uint2*>(d);
#pragma unroll
    for (int i = 0; i < W; i++) d2[i] = make_uint2((uint32_t)v.l[2 * i], (uint32_t)v.l[2 * i + 1]);
  }
  // an affine table entry (+- by the digit's sign) as an XYZZ point: (x, +-y, 1, 1), or infinity for the all-zero entry
  __device__ __forceinline__ static XYZZQ from_affine(const APoint* p, bool negate) {
    const uint32_t k = q();
    XYZZQ r;
    r.v = load_f(reinterpret_cast<const F*>(p) + aslot());  // lanes 2 and 3 load x and y once more
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < F::NL; i++) any |= (uint32_t)r.v.l[i];
    // the OR over what lanes 0 and 1 (of every component) hold decides infinity
    const bool f0 = flag<QP_B0>(any != 0), f1 = flag<QP_B1>(any != 0);
    const bool fin = either(f0 | f1);
    if (negate && k == 1u) r.v = r.v.neg();
    if (k >= 2u) r.v = fone();
    if (!fin) r.v = F::zero();
    return r;
  }

  // this = 2 this (this != O): dbl-2008-s-1 in three rounds
  __device__ __forceinline__ void dbl_nonzero() {
    const uint32_t k = q();
    // round 1   lane 0: X^2    lane 1: V = (2 Y)^2
    const F a1 = quad_sel(k == 1u, fdbl(v), v);
    const F m1 = fmul(a1, a1);
    // M = 3 X^2 (carried: it is squared below), on lanes 0 and 3
    F m = get<QP_B0>(m1);
#pragma unroll
    for (int i = 0; i < F::NL; i++) m.l[i] *= 3;
    m.carry();
    const F vv = get<QP_B1>(m1);
    // round 2   lane 0: S = X V    1: W = (2 Y) V    2: ZZ3 = ZZ V    3: M^2
    const F a2 = quad_sel(k == 3u, m, a1);
    const F b2 = quad_sel(k == 3u, m, vv);
    const F m2 = fmul(a2, b2);
    // X3 = M^2 - 2 S on lane 0
    const F msq = get<QP_3103>(m2);
    F x3;
#pragma unroll
    for (int i = 0; i < F::NL; i++) x3.l[i] = msq.l[i] - 2 * m2.l[i];
    x3.carry();
    const F w = get<QP_B1>(m2);
    // round 3   lane 0: M (S - X3)    1: W Y    3: ZZZ3 = W ZZZ
    const F a3 = quad_sel(k == 0u, m, v);
    const F b3 = quad_sel(k == 0u, fsub(m2, x3), w);
    const F m3 = fmul(a3, b3);
    // Y3 = M (S - X3) - W Y on lane 1
    const F t0 = get<QP_0023>(m3);
    F y3 = t0 - m3;
    v = quad_sel(k == 0u, x3, quad_sel(k == 1u, y3, quad_sel(k == 2u, m2, m3)));
  }

  // this += o, complete (either may be infinity, o = +-this)
  // (DBG: the self-test's view of the intermediate values -- dbg[6 * LANES] per point, slot-major like a point)
  template <bool DBG = false>
  __device__ __forceinline__ void add(const XYZZQ& o, F* dbg = nullptr) {
    const uint32_t k = q();
    if (o.is_inf()) return;  // uniform over the lanes of the point
    if (is_inf()) {
      v = o.v;
      return;
    }
    const F m1 = fmul(v, get<QP_ROT2>(o.v));
    const F d = fsub(get<QP_ROT2>(m1), m1);
    const bool low = k < 2u;
    const F m2 = fmul(quad_sel(low, d, v), quad_sel(low, d, o.v));
    const bool z = fzero(m2);
    if (flag<QP_B0>(z)) {  // P = 0: the same x
      if (flag<QP_B1>(z)) dbl_nonzero();  // R = 0: o = this
      else v = F::zero();                 // o = -this
      return;
    }
    const F pp = get<QP_B0>(m2);
    const F u1 = get<QP_0023>(m1);
    const F m3 = fmul(quad_sel(k == 0u, d, quad_sel(k == 1u, u1, m2)), pp);
    // X3 = RR - PPP - 2 Q on lanes 0 and 1 (lane 0: RR, Q from lane 1; lane 1: PPP from lane 0)
    const F s2 = get<QP_SWAP>(m2), s3 = get<QP_SWAP>(m3);
    const bool l0 = k == 0u;
    F x3;
#pragma unroll
    for (int i = 0; i < F::NL; i++)
      x3.l[i] = (l0 ? s2.l[i] : m2.l[i]) - (l0 ? m3.l[i] : s3.l[i]) - 2 * (l0 ? s3.l[i] : m3.l[i]);
    x3.carry();
    // round 4   lane 1: R (Q - X3)    2: S1 PPP    3: T P
    const F s1_or_p = get<QP_0110>(quad_sel(k == 0u, d, m1));  // lane 2: S1 (lane 1's m1); lane 3: P (lane 0's d)
    const F ppp = get<QP_B0>(m3);
    const F a4 = quad_sel(k == 1u, d, quad_sel(k == 2u, s1_or_p, m3));
    const F b4 = quad_sel(k == 1u, fsub(m3, x3), quad_sel(k == 2u, ppp, s1_or_p));
    const F m4 = fmul(a4, b4);
    // (own - dpp(own)): the fence keeps the DPP move from being folded into the subtraction.  Folded, the compiler emits
    // v_subrev_u32_dpp vD, vM, vM quad_perm:[0,2,2,3] -- and on gfx950 that instruction returned dpp(vM) - vM, the NEGATED
    // difference (self-test of round 6, limb by limb: profiles/r06/experiments/quad_add_subrev_dpp.txt); every other
    // subtraction here has the DPP operand as the minuend (v_sub_u32_dpp), which is right.
    F sub = get<QP_0223>(m4);
    dpp_fence(sub);
    const F y3 = m4 - sub;
    if constexpr (DBG) {
      dbg[0 * LANES + slot()] = m1;
      dbg[1 * LANES + slot()] = d;
      dbg[2 * LANES + slot()] = m2;
      dbg[3 * LANES + slot()] = m3;
      dbg[4 * LANES + slot()] = x3;
      dbg[5 * LANES + slot()] = m4;
    }
    v = quad_sel(k == 0u, x3, quad_sel(k == 1u, y3, quad_sel(k == 2u, m3, m4)));
  }
};

// the points of a wave summed into point 0 (control flow uniform per point; ds_bpermute moves, no LDS allocation)
template <class F, int XCH, bool EXT2>
__device__ __forceinline__ XYZZQ<F, XCH, EXT2> wave_quad_sum(XYZZQ<F, XCH, EXT2> acc) {
  for (int s = 32; s >= XYZZQ<F, XCH, EXT2>::LANES; s >>= 1) {
    XYZZQ<F, XCH, EXT2> o;
#pragma unroll
    for (int i = 0; i < F::NL; i++) o.v.l[i] = __shfl_down(acc.v.l[i], s);
    if ((threadIdx.x & 63u) < (uint32_t)s) acc.add(o);
  }
  return acc;
}

}  // namespace zkmi
#endif  // __HIPCC__
