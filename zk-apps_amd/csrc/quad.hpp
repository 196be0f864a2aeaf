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
template <class F, int XCH = 0>
struct XYZZQ {
  F v;
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
  __device__ __forceinline__ static uint32_t q() { return threadIdx.x & 3u; }
  __device__ __forceinline__ static XYZZQ infinity() { return {F::zero()}; }
  // ZZ = 0 (lane 2), known to all four lanes
  __device__ __forceinline__ bool is_inf() const { return flag<QP_B2>(v.is_zero()); }
  // memory form: XYZZ<F> = x | y | zz | zzz, lane q reads / writes coordinate q (a quad moves one contiguous point)
  __device__ __forceinline__ static XYZZQ load(const XYZZ<F>* p) {
    XYZZQ r;
    const F* s = reinterpret_cast<const F*>(p) + q();
    constexpr int W = sizeof(F) / 8;
    const uint2* s2 = reinterpret_cast<const uint2*>(s);
#pragma unroll
    for (int i = 0; i < W; i++) {
      const uint2 w = s2[i];
      r.v.l[2 * i] = (int32_t)w.x;
      r.v.l[2 * i + 1] = (int32_t)w.y;
    }
    return r;
  }
  __device__ __forceinline__ void store(XYZZ<F>* p) const {
    F* d = reinterpret_cast<F*>(p) + q();
    constexpr int W = sizeof(F) / 8;
    uint2* d2 = reinterpret_cast<uint2*>(d);
#pragma unroll
    for (int i = 0; i < W; i++) d2[i] = make_uint2((uint32_t)v.l[2 * i], (uint32_t)v.l[2 * i + 1]);
  }
  // an affine table entry (+- by the digit's sign) as an XYZZ point: (x, +-y, 1, 1), or infinity for the all-zero entry
  __device__ __forceinline__ static XYZZQ from_affine(const Affine<F>* p, bool negate) {
    const uint32_t k = q();
    XYZZQ r;
    const F* s = reinterpret_cast<const F*>(p) + (k & 1u);
    constexpr int W = sizeof(F) / 8;
    const uint2* s2 = reinterpret_cast<const uint2*>(s);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
      const uint2 w = s2[i];
      r.v.l[2 * i] = (int32_t)w.x;
      r.v.l[2 * i + 1] = (int32_t)w.y;
      any |= w.x | w.y;
    }
    // lanes 2 and 3 loaded x and y once more: the OR over the quad of what lanes 0 and 1 hold decides infinity
    const bool fin = flag<QP_B0>(any != 0) || flag<QP_B1>(any != 0);
    if (negate && k == 1u) r.v = r.v.neg();
    if (k >= 2u) r.v = fin ? F::one() : F::zero();
    if (!fin) r.v = F::zero();
    return r;
  }

  // this = 2 this (this != O): dbl-2008-s-1 in three rounds
  __device__ __forceinline__ void dbl_nonzero() {
    const uint32_t k = q();
    // round 1   lane 0: X^2    lane 1: V = (2 Y)^2
    const F a1 = quad_sel(k == 1u, v.add_lazy(v), v);
    const F m1 = F::mul_inline(a1, a1);
    // M = 3 X^2 (carried: it is squared below), on lanes 0 and 3
    F m = get<QP_B0>(m1);
#pragma unroll
    for (int i = 0; i < F::NL; i++) m.l[i] *= 3;
    m.carry();
    const F vv = get<QP_B1>(m1);
    // round 2   lane 0: S = X V    1: W = (2 Y) V    2: ZZ3 = ZZ V    3: M^2
    const F a2 = quad_sel(k == 3u, m, a1);
    const F b2 = quad_sel(k == 3u, m, vv);
    const F m2 = F::mul_inline(a2, b2);
    // X3 = M^2 - 2 S on lane 0
    const F msq = get<QP_3103>(m2);
    F x3;
#pragma unroll
    for (int i = 0; i < F::NL; i++) x3.l[i] = msq.l[i] - 2 * m2.l[i];
    x3.carry();
    const F w = get<QP_B1>(m2);
    // round 3   lane 0: M (S - X3)    1: W Y    3: ZZZ3 = W ZZZ
    const F a3 = quad_sel(k == 0u, m, v);
    const F b3 = quad_sel(k == 0u, m2.sub_lazy(x3), w);
    const F m3 = F::mul_inline(a3, b3);
    // Y3 = M (S - X3) - W Y on lane 1
    const F t0 = get<QP_0023>(m3);
    F y3 = t0 - m3;
    v = quad_sel(k == 0u, x3, quad_sel(k == 1u, y3, quad_sel(k == 2u, m2, m3)));
  }

  // this += o, complete (either may be infinity, o = +-this)
  // (DBG: the self-test's view of the intermediate values -- dbg[6 * 4] per quad, coordinate-major like a point)
  template <bool DBG = false>
  __device__ __forceinline__ void add(const XYZZQ& o, F* dbg = nullptr) {
    const uint32_t k = q();
    if (o.is_inf()) return;  // quad-uniform
    if (is_inf()) {
      v = o.v;
      return;
    }
    const F m1 = F::mul_inline(v, get<QP_ROT2>(o.v));
    const F d = get<QP_ROT2>(m1).sub_lazy(m1);
    const bool low = k < 2u;
    const F m2 = F::mul_inline(quad_sel(low, d, v), quad_sel(low, d, o.v));
    const bool z = m2.is_zero();
    if (flag<QP_B0>(z)) {  // P = 0: the same x
      if (flag<QP_B1>(z)) dbl_nonzero();  // R = 0: o = this
      else v = F::zero();                      // o = -this
      return;
    }
    const F pp = get<QP_B0>(m2);
    const F u1 = get<QP_0023>(m1);
    const F m3 = F::mul_inline(quad_sel(k == 0u, d, quad_sel(k == 1u, u1, m2)), pp);
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
    const F b4 = quad_sel(k == 1u, m3.sub_lazy(x3), quad_sel(k == 2u, ppp, s1_or_p));
    const F m4 = F::mul_inline(a4, b4);
    // (own - dpp(own)): the fence keeps the DPP move from being folded into the subtraction.  Folded, the compiler emits
    // v_subrev_u32_dpp vD, vM, vM quad_perm:[0,2,2,3] -- and on gfx950 that instruction returned dpp(vM) - vM, the NEGATED
    // difference (self-test of round 6, limb by limb: profiles/r06/experiments/quad_add_subrev_dpp.txt); every other
    // subtraction here has the DPP operand as the minuend (v_sub_u32_dpp), which is right.
    F sub = get<QP_0223>(m4);
    dpp_fence(sub);
    const F y3 = m4 - sub;
    if constexpr (DBG) {
      dbg[0 * 4 + k] = m1;
      dbg[1 * 4 + k] = d;
      dbg[2 * 4 + k] = m2;
      dbg[3 * 4 + k] = m3;
      dbg[4 * 4 + k] = x3;
      dbg[5 * 4 + k] = m4;
    }
    v = quad_sel(k == 0u, x3, quad_sel(k == 1u, y3, quad_sel(k == 2u, m3, m4)));
  }
};

// the quads of a wave summed into quad 0 (quad-uniform control flow; ds_bpermute moves, no LDS allocation)
template <class F, int XCH>
__device__ __forceinline__ XYZZQ<F, XCH> wave_quad_sum(XYZZQ<F, XCH> acc) {
  for (int s = 32; s >= 4; s >>= 1) {
    XYZZQ<F, XCH> o;
#pragma unroll
    for (int i = 0; i < F::NL; i++) o.v.l[i] = __shfl_down(acc.v.l[i], s);
    if ((threadIdx.x & 63u) < (uint32_t)s) acc.add(o);
  }
  return acc;
}

}  // namespace zkmi
#endif  // __HIPCC__
