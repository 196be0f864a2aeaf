// zkmi — Poseidon-5 (x^5) over a 255/254-bit scalar field in the 28-bit limb form.
//
// Parameters fixed by the reference's relations (shielder/relations/src/lib.rs:17-26):
//   T_WIDTH = 5, RATE = 4, R_F = 8, R_P = 56; hashing call = PoseidonHasher::hash_fix_len_array
//   (update_note.rs:100,131; update_account.rs:62; merkle_proof.rs:56).
// The arithmetic itself is in crates that are not in the tree (halo2-base 0.4.1, pse-poseidon
// 0.2.0); what is restated here is their published procedure: Grain-LFSR round
// constants + Cauchy MDS (poseidon.hip), plain ARK -> S-box -> MDS rounds, and the sponge framing
// [2^64, 0, 0, 0, 0] / "+1" padding / output state[1].
//
// One code path for host (witness generation of a single relation instance) and device (batched
// hashing, Merkle trees): everything below is __host__ __device__.
#pragma once
#include "field28.hpp"

namespace zkmi {

constexpr int POS_T = 5, POS_RATE = 4, POS_RF = 8, POS_RP = 56, POS_ROUNDS = POS_RF + POS_RP;

// Montgomery-form constants as they sit in HBM (13.8 KB; uniform addresses -> scalar loads)
template <class F>
struct PoseidonConsts {
  F rc[POS_ROUNDS * POS_T];
  F mds[POS_T * POS_T];  // row-major: new[i] = sum_j mds[5 i + j] * state[j]
  F cap;                 // 2^64, the initial capacity element
  // Equivalent form of the 56 partial rounds (Poseidon paper, appendix B; derived in poseidon.hip):
  // one scalar constant per round on lane 0, a sparse matrix per round (first row + first column,
  // identity elsewhere), one 4x4 block applied after the last partial round, and the leftover
  // constant vector folded into the first full round that follows.
  F pk[POS_RP];               // lane-0 constants
  F prow[POS_RP * POS_T];     // new[0] = sum_j prow[j] * s[j]
  F pcol[POS_RP * (POS_T - 1)];  // new[j+1] = s[j+1] + pcol[j] * s[0]
  F plast[(POS_T - 1) * (POS_T - 1)];  // s[1..] <- plast * s[1..] after the partial rounds
  F rc_tail[POS_T];           // constants of round R_F/2 + R_P including the folded remainder
};

// sum_j m[j] * s[j] with ONE Montgomery reduction: 50 partial products of < 2^56 per column
// stay below 2^62 in the 64-bit accumulators (operands normalised, |v| < 4 p).
template <class F>
ZK_HD F pos_dot5(const F* __restrict__ m, const F* __restrict__ s) {
  constexpr int NL = F::NL;
  int64_t T[2 * NL];
#pragma unroll
  for (int i = 0; i < 2 * NL; i++) T[i] = 0;
#pragma unroll
  for (int j = 0; j < POS_T; j++)
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
      for (int k = 0; k < NL; k++) T[i + k] += (int64_t)m[j].l[i] * s[j].l[k];
  return F::reduce(T);
}

template <class F>
ZK_HD F pos_pow5(const F& x) {
  const F x2 = x.sqr_inline();
  const F x4 = x2.sqr_inline();
  return F::mul_inline(x4, x);
}

template <class F>
ZK_HD void pos_round(F st[POS_T], const F* __restrict__ rc, const F* __restrict__ mds, bool full) {
#pragma unroll
  for (int i = 0; i < POS_T; i++) st[i] = st[i] + rc[i];
  st[0] = pos_pow5(st[0]);
  // (these two loops stay rolled: the bodies are ~1-3 k instructions each)
  if (full) {
    for (int i = 1; i < POS_T; i++) st[i] = pos_pow5(st[i]);
  }
  F nx[POS_T];
  for (int i = 0; i < POS_T; i++) nx[i] = pos_dot5(mds + POS_T * i, st);
#pragma unroll
  for (int i = 0; i < POS_T; i++) st[i] = nx[i];
}

// ---- device form: the state lives in this thread's column of an LDS tile -------------------
// Layout [element][limb][thread] (bank-conflict free), 5 x 10 words x 256 threads = 50 KB per block,
// three blocks per CU.  Every loop over state elements stays rolled and loads its operand just in
// time, so the register footprint is one accumulator set + two elements instead of the whole
// state (the all-in-registers form needs 255 VGPRs or spills ~1.4 KB per lane to scratch).
// No barrier anywhere: a thread only touches its own column.
constexpr int POS_LDS_WORDS = POS_T * Fr28::NL * 256;
#if defined(__HIPCC__)
template <class F>
__device__ __forceinline__ void pos_lds_store(int32_t* col, int elem, const F& v) {
#pragma unroll
  for (int k = 0; k < F::NL; k++) col[(elem * F::NL + k) * 256] = v.l[k];
}
template <class F>
__device__ __forceinline__ F pos_lds_load(const int32_t* col, int elem) {
  F v;
#pragma unroll
  for (int k = 0; k < F::NL; k++) v.l[k] = col[(elem * F::NL + k) * 256];
  return v;
}
template <class F>
__device__ __forceinline__ void pos_mac(int64_t* T, const F& a, const F& b) {
#pragma unroll
  for (int i = 0; i < F::NL; i++)
#pragma unroll
    for (int k = 0; k < F::NL; k++) T[i + k] += (int64_t)a.l[i] * b.l[k];
}
// sum_j row[j] * state[first + j], j < n, operands streamed from LDS
template <class F>
__device__ __forceinline__ F pos_dot_lds(const F* __restrict__ row, const int32_t* col, int first, int n) {
  int64_t T[2 * F::NL];
#pragma unroll
  for (int i = 0; i < 2 * F::NL; i++) T[i] = 0;
#pragma unroll 1
  for (int j = 0; j < n; j++) pos_mac(T, row[j], pos_lds_load<F>(col, first + j));
  return F::reduce(T);
}
template <class F>
__device__ __forceinline__ void pos_round_full_lds(const F* __restrict__ rc, const F* __restrict__ mds, int32_t* col) {
#pragma unroll 1
  for (int i = 0; i < POS_T; i++) pos_lds_store(col, i, pos_pow5(pos_lds_load<F>(col, i) + rc[i]));
  // the five outputs wait in registers (rotated into place) until the inputs are no longer needed
  F nx[POS_T];
#pragma unroll
  for (int i = 0; i < POS_T; i++) nx[i] = F::zero();
#pragma unroll 1
  for (int i = 0; i < POS_T; i++) {
    const F y = pos_dot_lds<F>(mds + POS_T * i, col, 0, POS_T);
    nx[0] = nx[1];
    nx[1] = nx[2];
    nx[2] = nx[3];
    nx[3] = nx[4];
    nx[4] = y;
  }
#pragma unroll
  for (int i = 0; i < POS_T; i++) pos_lds_store(col, i, nx[i]);
}
// the permutation on the LDS-resident state (sparse partial rounds, see PoseidonConsts)
template <class F>
__device__ __forceinline__ void poseidon_permute_lds(int32_t* col, const PoseidonConsts<F>* __restrict__ c) {
  int r = 0;
#pragma unroll 1
  for (; r < POS_RF / 2; r++) pos_round_full_lds<F>(c->rc + POS_T * r, c->mds, col);
#pragma unroll 1
  for (int i = 0; i < POS_RP; i++) {
    const F x0 = pos_pow5(pos_lds_load<F>(col, 0) + c->pk[i]);
    int64_t T[2 * F::NL];
#pragma unroll
    for (int q = 0; q < 2 * F::NL; q++) T[q] = 0;
    pos_mac(T, c->prow[POS_T * i], x0);
#pragma unroll 1
    for (int j = 1; j < POS_T; j++) pos_mac(T, c->prow[POS_T * i + j], pos_lds_load<F>(col, j));
    const F n0 = F::reduce(T);
#pragma unroll 1
    for (int j = 1; j < POS_T; j++)
      pos_lds_store(col, j, F::mul_inline(c->pcol[(POS_T - 1) * i + j - 1], x0) + pos_lds_load<F>(col, j));
    pos_lds_store(col, 0, n0);
  }
  {
    F t[POS_T - 1];
#pragma unroll
    for (int i = 0; i < POS_T - 1; i++) t[i] = F::zero();
#pragma unroll 1
    for (int i = 0; i < POS_T - 1; i++) {
      const F y = pos_dot_lds<F>(c->plast + (POS_T - 1) * i, col, 1, POS_T - 1);
      t[0] = t[1];
      t[1] = t[2];
      t[2] = t[3];
      t[3] = y;
    }
#pragma unroll
    for (int i = 0; i < POS_T - 1; i++) pos_lds_store(col, i + 1, t[i]);
  }
#pragma unroll 1
  for (r = POS_RF / 2 + POS_RP; r < POS_ROUNDS; r++)
    pos_round_full_lds<F>(r == POS_RF / 2 + POS_RP ? c->rc_tail : c->rc + POS_T * r, c->mds, col);
}
// hash_fix_len_array, device form
template <class F>
__device__ __forceinline__ F poseidon_hash_words_lds(const uint32_t* __restrict__ in, uint32_t len,
                                                     const PoseidonConsts<F>* __restrict__ c, int32_t* col) {
  pos_lds_store(col, 0, c->cap);
#pragma unroll 1
  for (int i = 1; i < POS_T; i++) pos_lds_store(col, i, F::zero());
  uint32_t done = 0;
  bool more = true;
#pragma unroll 1
  while (more) {
    const uint32_t take = (len - done) < (uint32_t)POS_RATE ? (len - done) : (uint32_t)POS_RATE;
#pragma unroll 1
    for (uint32_t i = 0; i < take; i++)
      pos_lds_store(col, 1 + i, pos_lds_load<F>(col, 1 + i) + F::from_canonical(in + 8 * (done + i)));
    if (take < (uint32_t)POS_RATE) pos_lds_store(col, 1 + take, pos_lds_load<F>(col, 1 + take) + F::one());
    done += take;
    poseidon_permute_lds<F>(col, c);
    more = (take == (uint32_t)POS_RATE);
  }
  return pos_lds_load<F>(col, 1);
}
#endif

// the definition: 64 x (add round constants, S-box, MDS)
template <class F>
ZK_HD void poseidon_permute_plain(F st[POS_T], const PoseidonConsts<F>* __restrict__ c) {
  int r = 0;
#pragma unroll 1
  for (; r < POS_RF / 2; r++) pos_round(st, c->rc + POS_T * r, c->mds, true);
#pragma unroll 1
  for (; r < POS_RF / 2 + POS_RP; r++) pos_round(st, c->rc + POS_T * r, c->mds, false);
#pragma unroll 1
  for (; r < POS_ROUNDS; r++) pos_round(st, c->rc + POS_T * r, c->mds, true);
}

// s0 * k + t with one reduction of the product (k, s0, t normalised)
template <class F>
ZK_HD F pos_mul_add(const F& k, const F& s0, const F& t) {
  return F::mul_inline(k, s0) + t;
}

// same permutation, partial rounds in the sparse form: 9 instead of 25 products per round
// (register-array form: host witness generation and the host self-test; the kernels run
// poseidon_permute_lds, the same sequence on an LDS-resident state)
template <class F>
ZK_HD void poseidon_permute(F st[POS_T], const PoseidonConsts<F>* __restrict__ c) {
  int r = 0;
#pragma unroll 1
  for (; r < POS_RF / 2; r++) pos_round(st, c->rc + POS_T * r, c->mds, true);
#pragma unroll 1
  for (int i = 0; i < POS_RP; i++) {
    st[0] = pos_pow5(st[0] + c->pk[i]);
    const F n0 = pos_dot5(c->prow + POS_T * i, st);
#pragma unroll
    for (int j = 0; j < POS_T - 1; j++) st[j + 1] = pos_mul_add(c->pcol[(POS_T - 1) * i + j], st[0], st[j + 1]);
    st[0] = n0;
  }
  {
    // s[1..4] <- plast * s[1..4]; rolled, results rotated into place (constant indices only)
    F t[POS_T - 1];
#pragma unroll
    for (int i = 0; i < POS_T - 1; i++) t[i] = F::zero();
#pragma unroll 1
    for (int i = 0; i < POS_T - 1; i++) {
      constexpr int NL = F::NL;
      int64_t T[2 * NL];
#pragma unroll
      for (int q = 0; q < 2 * NL; q++) T[q] = 0;
      const F* __restrict__ row = c->plast + (POS_T - 1) * i;
#pragma unroll
      for (int j = 0; j < POS_T - 1; j++)
#pragma unroll
        for (int a = 0; a < NL; a++)
#pragma unroll
          for (int b = 0; b < NL; b++) T[a + b] += (int64_t)row[j].l[a] * st[j + 1].l[b];
      const F y = F::reduce(T);
      t[0] = t[1];
      t[1] = t[2];
      t[2] = t[3];
      t[3] = y;
    }
#pragma unroll
    for (int i = 0; i < POS_T - 1; i++) st[i + 1] = t[i];
  }
  // second half: the first of its rounds uses the constants with the folded remainder
#pragma unroll 1
  for (r = POS_RF / 2 + POS_RP; r < POS_ROUNDS; r++)
    pos_round(st, r == POS_RF / 2 + POS_RP ? c->rc_tail : c->rc + POS_T * r, c->mds, true);
}

// hash_fix_len_array over `len` canonical 32-byte little-endian inputs (8 words each)
template <class F>
ZK_HD F poseidon_hash_words(const uint32_t* __restrict__ in, uint32_t len, const PoseidonConsts<F>* __restrict__ c) {
  F st[POS_T];
  st[0] = c->cap;
#pragma unroll
  for (int i = 1; i < POS_T; i++) st[i] = F::zero();
  uint32_t done = 0;
  bool more = true;
#pragma unroll 1
  while (more) {
    const uint32_t take = (len - done) < (uint32_t)POS_RATE ? (len - done) : (uint32_t)POS_RATE;
#pragma unroll
    for (int i = 0; i < POS_RATE; i++) {
      if ((uint32_t)i < take) st[1 + i] = st[1 + i] + F::from_canonical(in + 8 * (done + i));
      if ((uint32_t)i == take) st[1 + i] = st[1 + i] + F::one();  // padding behind the last input
    }
    done += take;
    poseidon_permute(st, c);
    // a full last chunk is followed by a padding-only permutation (take == 0 next time round)
    more = (take == (uint32_t)POS_RATE);
  }
  return st[1];
}

// host: constants in canonical form (rc: 64 x 5 x 32 B, mds: 25 x 32 B) for field 0 = BLS12-381 Fr,
// 1 = BN254 Fr; generated once per process
const uint8_t* poseidon_rc_canonical(int field);
const uint8_t* poseidon_mds_canonical(int field);
const PoseidonConsts<Fr28>* poseidon_consts_bls();
const PoseidonConsts<BnFr28>* poseidon_consts_bn();

}  // namespace zkmi
