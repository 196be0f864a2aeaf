/* zkmi without Python and without PyTorch: the bench.py workload and a key-lifecycle churn from a plain C process that
 * binds libzkmi.so to the HIP runtime it was BUILT for (/opt/rocm), the way a Rust host of ZkProof::update_account
 * (shielder/contract/drink_tests/utils/shielder.rs:78-134) would.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/bench_prove.c \
 *       -Lzk-apps_amd -lzkmi -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/zk-apps_amd -Wl,-rpath,/opt/rocm/lib -o bench_prove
 *   ./bench_prove [--log-n 20] [--proofs 20] [--warmup 2] [--churn OPS] [--seed S] [--dump FILE]
 *
 * 1. prints zkmi_hip_versions (build and runtime must agree here: nothing preloads another runtime);
 * 2. update_note (withdraw, Poseidon-5) relation at N = 2^log_n, trusted setup, `proofs + warmup` DISTINCT assignments
 *    generated on the device (zkmi_update_note_witness_batch_dev) from the same seeds bench.py uses, a fresh (r, s) per
 *    proof; the timed region is ONE zkmi_groth16_prove_batch_dev call over `proofs` resident witnesses (three in flight);
 *    every proof is checked by the pairing verifier afterwards; --dump writes the proof bytes (the GPU test compares
 *    them with the Python binding's proofs from the same seeds);
 * 3. --churn OPS: OPS random operations on ONE context -- setups of sizes 2^13..2^17 in random order with random group
 *    sizes, batches with partial groups, single proofs from device and host witnesses, the unsatisfied-witness error
 *    path, generic MSMs (plain and prepared) and NTT round trips in between, short-lived second contexts, keys freed in
 *    random order; every proof must equal, byte for byte, the first proof ever made from the same (size, witness, r, s).
 * Exit code 0 only if everything verified and matched. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zkmi.h"

static const uint8_t FR_MOD[32] = {0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
                                   0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};

/* the generator bench.py draws its inputs from (SplitMix64, rejection-sampled to [0, r)) */
typedef struct {
  uint64_t s;
} splitmix;
static uint64_t sm_next(splitmix* g) {
  g->s += 0x9E3779B97F4A7C15ull;
  uint64_t z = g->s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static int lt_mod(const uint8_t v[32]) {
  for (int i = 31; i >= 0; i--)
    if (v[i] != FR_MOD[i]) return v[i] < FR_MOD[i];
  return 0;
}
static void sm_fr(splitmix* g, uint8_t out[32]) {
  for (;;) {
    for (int i = 0; i < 4; i++) {
      const uint64_t w = sm_next(g);
      memcpy(out + 8 * i, &w, 8); /* little-endian host */
    }
    out[31] &= 0x7f;
    if (lt_mod(out)) return;
  }
}
static void put_u64(zkmi_fr* f, uint64_t v) {
  memset(f->bytes, 0, 32);
  memcpy(f->bytes, &v, 8);
}

/* the withdraw bench.py's relation_and_witness() builds from `seed` (same draws in the same order) */
static void note_update_from_seed(uint64_t seed, zkmi_note_update* in) {
  splitmix g = {seed};
  zkmi_fr tok0, tok1, user;
  memset(in, 0, sizeof(*in));
  sm_fr(&g, tok0.bytes);
  sm_fr(&g, tok1.bytes);
  const uint64_t bal0 = sm_next(&g) >> 1, bal1 = sm_next(&g) >> 1;
  sm_fr(&g, user.bytes);
  put_u64(&in->amount, bal0 >> 3);
  in->token = tok0;
  in->user = user;
  for (int k = 0; k < 3; k++) sm_fr(&g, in->new_note[k].bytes);
  for (int k = 0; k < 3; k++) sm_fr(&g, in->old_note[k].bytes);
  in->tree_height = 10;
  for (int k = 0; k < 10; k++) in->path_shape[k] = (uint8_t)(sm_next(&g) & 1);
  for (int k = 0; k < 10; k++) sm_fr(&g, in->path[k].bytes);
  in->op_priv_user = user;
  in->account[0] = tok0;
  put_u64(&in->account[1], bal0);
  in->account[2] = tok1;
  put_u64(&in->account[3], bal1);
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static uint64_t fnv1a(const uint8_t* p, size_t n, uint64_t h) {
  for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}

#define CHECK(call)                                                                             \
  do {                                                                                          \
    int32_t rc_ = (call);                                                                       \
    if (rc_ != ZKMI_OK) {                                                                       \
      fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, ctx ? zkmi_last_error(ctx) : ""); \
      return 1;                                                                                 \
    }                                                                                           \
  } while (0)
#define HIPCHECK(call)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_));   \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

/* ---- a key with its relation, verifying key and a pool of device-resident assignments -------------------------- */
#define POOL 6      /* distinct assignments per size in the churn */
#define RS_POOL 4   /* distinct (r, s) pairs per size */
#define MAX_KEYS 10
typedef struct {
  uint32_t lg, n_pub;
  zkmi_r1cs* r1;
  zkmi_pk* pk;
  uint8_t* vk;
} zkey;

/* per-size state that outlives the keys: assignments (device + one host copy), blinding pairs, first-seen proofs */
typedef struct {
  int ready;
  void* d_z[POOL];
  uint8_t* h_z0;              /* host copy of assignment 0 (host-witness entry points) */
  uint8_t pub[POOL][6 * 32];  /* the six public inputs of every assignment */
  uint8_t r[RS_POOL][32], s[RS_POOL][32];
  uint8_t first[POOL][RS_POOL][192];
  uint8_t seen[POOL][RS_POOL];
} size_state;

static int toxic_for(uint32_t lg, uint8_t toxic[160]) {
  splitmix g = {0x70C51C00ull + lg};
  for (int k = 0; k < 5; k++) sm_fr(&g, toxic + 32 * k);
  return 0;
}

static int key_create(zkmi_ctx* ctx, uint32_t lg, uint32_t group, zkey* k) {
  memset(k, 0, sizeof(*k));
  k->lg = lg;
  CHECK(zkmi_update_note_r1cs(lg, ZKMI_OP_WITHDRAW, &k->r1));
  CHECK(zkmi_r1cs_shape(k->r1, NULL, &k->n_pub, NULL, NULL));
  uint8_t toxic[160];
  toxic_for(lg, toxic);
  const uint64_t cap = 672 + 96 * (uint64_t)k->n_pub;
  k->vk = malloc(cap);
  CHECK(zkmi_ctx_set_group_size(ctx, group));
  CHECK(zkmi_groth16_setup(ctx, k->r1, toxic, &k->pk, k->vk, cap));
  CHECK(zkmi_ctx_set_group_size(ctx, 0));
  return 0;
}
static void key_free(zkey* k) {
  if (k->pk) (void)zkmi_pk_free(k->pk);
  if (k->r1) (void)zkmi_r1cs_free(k->r1);
  free(k->vk);
  memset(k, 0, sizeof(*k));
}

static int size_prepare(zkmi_ctx* ctx, uint32_t lg, size_state* st) {
  if (st->ready) return 0;
  zkmi_note_update in[POOL];
  int32_t status[POOL];
  for (int i = 0; i < POOL; i++) {
    note_update_from_seed(0xC4000000ull + 64 * lg + (uint64_t)i, &in[i]);
    HIPCHECK(hipMalloc(&st->d_z[i], (size_t)32 << lg));
  }
  CHECK(zkmi_update_note_witness_batch_dev(ctx, lg, ZKMI_OP_WITHDRAW, in, POOL, (void* const*)st->d_z, status));
  for (int i = 0; i < POOL; i++) {
    if (status[i] != ZKMI_OK) {
      fprintf(stderr, "assignment %d of size 2^%u: status %d\n", i, lg, status[i]);
      return 1;
    }
    HIPCHECK(hipMemcpy(st->pub[i], (const uint8_t*)st->d_z[i] + 32, 6 * 32, hipMemcpyDeviceToHost));
  }
  st->h_z0 = malloc((size_t)32 << lg);
  HIPCHECK(hipMemcpy(st->h_z0, st->d_z[0], (size_t)32 << lg, hipMemcpyDeviceToHost));
  splitmix g = {0xB11D0000ull + lg};
  for (int j = 0; j < RS_POOL; j++) {
    sm_fr(&g, st->r[j]);
    sm_fr(&g, st->s[j]);
  }
  st->ready = 1;
  return 0;
}

/* a proof of (assignment w, blinding pair j) of this size: remember the first one, compare all later ones with it */
static int check_proof(const zkey* k, size_state* st, int w, int j, const uint8_t proof[192], const char* how, long op) {
  if (!st->seen[w][j]) {
    memcpy(st->first[w][j], proof, 192);
    st->seen[w][j] = 1;
    const int32_t rc = zkmi_groth16_verify(k->vk, k->n_pub, st->pub[w], proof);
    if (rc != ZKMI_OK) {
      fprintf(stderr, "op %ld: %s proof of 2^%u (w %d, rs %d) does not verify: %d\n", op, how, k->lg, w, j, rc);
      return 1;
    }
    return 0;
  }
  if (memcmp(st->first[w][j], proof, 192) != 0) {
    fprintf(stderr, "op %ld: %s proof of 2^%u (w %d, rs %d) differs from its first occurrence\n", op, how, k->lg, w, j);
    return 1;
  }
  return 0;
}

/* n base points through the PRODUCT's ABI only (the generators of synthetic bases are test scaffolding,
 * include/zkmi_testing.h): 4 096 distinct points G + i Q made on the host once, repeated cyclically up to n and uploaded with
 * zkmi_bases_g1_load.  Repeated bases are valid inputs, and with random scalars two copies of a point now and then meet in
 * one bucket with the same digit: the P + P redo paths get exercised on the way. */
static int make_bases(zkmi_ctx* ctx, uint64_t n, zkmi_bases_g1** out) {
  enum { DISTINCT = 4096 };
  static uint8_t* pts = NULL;
  if (!pts) {
    pts = malloc((size_t)96 * DISTINCT);
    uint8_t q[96], k[32] = {0xEE, 0xFF, 0xC0};
    if (!pts || zkmi_g1_generator(pts) != ZKMI_OK || zkmi_g1_mul(pts, k, q) != ZKMI_OK) return 1;
    for (int i = 1; i < DISTINCT; i++)
      if (zkmi_g1_add(pts + 96 * (size_t)(i - 1), q, pts + 96 * (size_t)i) != ZKMI_OK) return 1;
  }
  uint8_t* all = malloc((size_t)96 * n);
  if (!all) return 1;
  for (uint64_t i = 0; i < n; i++) memcpy(all + 96 * i, pts + 96 * (size_t)(i % DISTINCT), 96);
  const int32_t rc = zkmi_bases_g1_load(ctx, all, n, 0, out);
  free(all);
  return rc == ZKMI_OK ? 0 : 1;
}

static int churn(zkmi_ctx* ctx, long ops, uint64_t seed) {
  enum { LG_MIN = 13, LG_MAX = 17 };
  static size_state sizes[LG_MAX + 1];
  zkey keys[MAX_KEYS];
  int nkeys = 0;
  memset(keys, 0, sizeof(keys));
  splitmix g = {seed};
  long n_setup = 0, n_batch = 0, n_single = 0, n_host = 0, n_err = 0, n_msm = 0, n_ntt = 0, n_ctx2 = 0, n_free = 0, n_proofs = 0, n_big_small = 0;
  uint32_t last_lg = 0;
  /* MSM / NTT fixtures: results must not change over the run */
  const uint64_t msm_n[3] = {(1ull << 14) - 3, (1ull << 18) + 11, (1ull << 21) + 5};
  zkmi_bases_g1* bases[3] = {NULL, NULL, NULL};
  void* d_sc = NULL;
  uint8_t msm_first[3][96];
  int msm_seen[3] = {0, 0, 0}, prepared[3] = {0, 0, 0};
  HIPCHECK(hipMalloc(&d_sc, 32 * msm_n[2]));
  {
    uint8_t* h = malloc(32 * msm_n[2]);
    splitmix q = {seed ^ 0x5ca1a5};
    for (uint64_t i = 0; i < msm_n[2]; i++) {
      for (int w = 0; w < 4; w++) {
        const uint64_t v = sm_next(&q);
        memcpy(h + 32 * i + 8 * w, &v, 8);
      }
      h[32 * i + 31] &= 0x3f;
    }
    HIPCHECK(hipMemcpy(d_sc, h, 32 * msm_n[2], hipMemcpyHostToDevice));
    free(h);
  }
  const double t0 = now_s();
  for (long op = 0; op < ops; op++) {
    const uint32_t kind = (uint32_t)(sm_next(&g) % 100);
    if (nkeys == 0 || (kind < 10 && nkeys < MAX_KEYS)) {
      /* ---- a new key: random size, random forced group size; the first setups force big -> small transitions ---- */
      uint32_t lg = LG_MIN + (uint32_t)(sm_next(&g) % (LG_MAX - LG_MIN + 1));
      if (op < 4) lg = (op & 1) ? 15 : 16;  /* 16 -> 15 first: the sequence of round 2's crash */
      const uint32_t groups[7] = {0, 0, 64, 32, 16, 8, 1};
      if (size_prepare(ctx, lg, &sizes[lg])) return 1;
      if (key_create(ctx, lg, groups[sm_next(&g) % 7], &keys[nkeys])) return 1;
      if (last_lg > lg) n_big_small++;
      last_lg = lg;
      nkeys++;
      n_setup++;
      continue;
    }
    zkey* k = &keys[sm_next(&g) % (uint64_t)nkeys];
    size_state* st = &sizes[k->lg];
    if (kind < 45) {
      /* ---- a batch with a partial last group ---- */
      const uint32_t count = 1 + (uint32_t)(sm_next(&g) % 70);
      const void** zz = malloc(sizeof(void*) * count);
      uint8_t *rr = malloc(32 * count), *ss = malloc(32 * count), *out = malloc(192 * count);
      int* wi = malloc(sizeof(int) * count);
      int* ji = malloc(sizeof(int) * count);
      for (uint32_t i = 0; i < count; i++) {
        wi[i] = (int)(sm_next(&g) % POOL);
        ji[i] = (int)(sm_next(&g) % RS_POOL);
        zz[i] = st->d_z[wi[i]];
        memcpy(rr + 32 * i, st->r[ji[i]], 32);
        memcpy(ss + 32 * i, st->s[ji[i]], 32);
      }
      CHECK(zkmi_groth16_prove_batch_dev(ctx, k->pk, count, zz, rr, ss, out));
      for (uint32_t i = 0; i < count; i++)
        if (check_proof(k, st, wi[i], ji[i], out + 192 * i, "batch", op)) return 1;
      n_proofs += count;
      n_batch++;
      free(zz), free(rr), free(ss), free(out), free(wi), free(ji);
    } else if (kind < 65) {
      /* ---- one proof from a device-resident assignment ---- */
      const int w = (int)(sm_next(&g) % POOL), j = (int)(sm_next(&g) % RS_POOL);
      uint8_t proof[192];
      CHECK(zkmi_groth16_prove_dev(ctx, k->pk, st->d_z[w], st->r[j], st->s[j], proof));
      if (check_proof(k, st, w, j, proof, "single", op)) return 1;
      n_proofs++, n_single++;
    } else if (kind < 72) {
      /* ---- one proof from a host assignment ---- */
      const int j = (int)(sm_next(&g) % RS_POOL);
      uint8_t proof[192];
      CHECK(zkmi_groth16_prove(ctx, k->pk, st->h_z0, st->r[j], st->s[j], proof));
      if (check_proof(k, st, 0, j, proof, "host", op)) return 1;
      n_proofs++, n_host++;
    } else if (kind < 76) {
      /* ---- an assignment that does not satisfy the relation must be refused, and the context must survive it ---- */
      uint8_t proof[192];
      st->h_z0[32 * 100] ^= 1;
      const int32_t rc = zkmi_groth16_prove(ctx, k->pk, st->h_z0, st->r[0], st->s[0], proof);
      st->h_z0[32 * 100] ^= 1;
      if (rc != ZKMI_ERR_UNSATISFIED && rc != ZKMI_ERR_NON_CANONICAL) {
        fprintf(stderr, "op %ld: a broken assignment of 2^%u returned %d\n", op, k->lg, rc);
        return 1;
      }
      n_err++;
    } else if (kind < 84) {
      /* ---- a generic MSM (grows the sort buffers past what grouped keys reserved); prepared bases half of the time ---- */
      const int q = (int)(sm_next(&g) % 3);
      if (!bases[q] && make_bases(ctx, msm_n[q], &bases[q])) {
        fprintf(stderr, "op %ld: could not make %llu base points\n", op, (unsigned long long)msm_n[q]);
        return 1;
      }
      if (!prepared[q] && (sm_next(&g) & 1)) {
        CHECK(zkmi_bases_g1_prepare(ctx, bases[q]));
        prepared[q] = 1;
      }
      uint8_t out[96];
      CHECK(zkmi_msm_g1_dev(ctx, d_sc, msm_n[q], bases[q], out));
      if (!msm_seen[q]) memcpy(msm_first[q], out, 96), msm_seen[q] = 1;
      else if (memcmp(msm_first[q], out, 96) != 0) {
        fprintf(stderr, "op %ld: MSM of %llu terms changed its result\n", op, (unsigned long long)msm_n[q]);
        return 1;
      }
      n_msm++;
    } else if (kind < 90) {
      /* ---- NTT round trip on a host buffer ---- */
      const uint32_t lg = 10 + (uint32_t)(sm_next(&g) % 9);
      const size_t bytes = (size_t)32 << lg;
      uint8_t *a = malloc(bytes), *b = malloc(bytes);
      splitmix q = {sm_next(&g)};
      for (size_t i = 0; i < bytes; i += 8) {
        const uint64_t v = sm_next(&q);
        memcpy(a + i, &v, 8);
      }
      for (size_t i = 31; i < bytes; i += 32) a[i] &= 0x3f;
      memcpy(b, a, bytes);
      const int coset = (int)(sm_next(&g) & 1);
      CHECK(zkmi_ntt_fr(ctx, b, lg, 0, coset));
      CHECK(zkmi_ntt_fr(ctx, b, lg, 1, coset));
      if (memcmp(a, b, bytes) != 0) {
        fprintf(stderr, "op %ld: NTT round trip of 2^%u failed\n", op, lg);
        return 1;
      }
      free(a), free(b);
      n_ntt++;
    } else if (kind < 93) {
      /* ---- a short-lived second context on the same device: one small proof, destroyed again ---- */
      zkmi_ctx* c2 = NULL;
      if (zkmi_ctx_create(0, &c2) != ZKMI_OK) {
        fprintf(stderr, "op %ld: second context\n", op);
        return 1;
      }
      zkey k2;
      size_state* s13 = &sizes[13];
      if (size_prepare(ctx, 13, s13)) return 1;
      {
        zkmi_ctx* ctx_saved = ctx;
        ctx = c2;
        if (key_create(c2, 13, 0, &k2)) return 1;
        uint8_t proof[192];
        CHECK(zkmi_groth16_prove_dev(c2, k2.pk, s13->d_z[1], s13->r[1], s13->s[1], proof));
        ctx = ctx_saved;
        if (check_proof(&k2, s13, 1, 1, proof, "second-context", op)) return 1;
      }
      key_free(&k2);
      (void)zkmi_ctx_destroy(c2);
      n_proofs++, n_ctx2++;
    } else if (nkeys > 1 || kind >= 97) {
      /* ---- free a random key ---- */
      const int i = (int)(sm_next(&g) % (uint64_t)nkeys);
      key_free(&keys[i]);
      keys[i] = keys[nkeys - 1];
      memset(&keys[nkeys - 1], 0, sizeof(zkey));
      nkeys--;
      n_free++;
    }
  }
  for (int i = 0; i < nkeys; i++) key_free(&keys[i]);
  for (int q = 0; q < 3; q++)
    if (bases[q]) (void)zkmi_bases_g1_free(bases[q]);
  (void)hipFree(d_sc);
  for (uint32_t lg = LG_MIN; lg <= LG_MAX; lg++)
    if (sizes[lg].ready) {
      for (int i = 0; i < POOL; i++) (void)hipFree(sizes[lg].d_z[i]);
      free(sizes[lg].h_z0);
    }
  printf("churn: %ld operations in %.1f s: %ld setups (%ld big->small), %ld batches, %ld single, %ld host-witness, %ld refused, %ld MSMs, "
         "%ld NTT round trips, %ld second contexts, %ld keys freed early; %ld proofs, every one equal to its first occurrence: clean\n",
         ops, now_s() - t0, n_setup, n_big_small, n_batch, n_single, n_host, n_err, n_msm, n_ntt, n_ctx2, n_free, n_proofs);
  return 0;
}

int main(int argc, char** argv) {
  uint32_t log_n = 20, proofs = 20, warmup = 2;
  long churn_ops = 0;
  uint64_t seed = 0xC0FFEE;
  const char* dump = NULL;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--log-n") && i + 1 < argc) log_n = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--proofs") && i + 1 < argc) proofs = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--churn") && i + 1 < argc) churn_ops = atol(argv[++i]);
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], NULL, 0);
    else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    else {
      fprintf(stderr, "usage: %s [--log-n L] [--proofs K] [--warmup W] [--churn OPS] [--seed S] [--dump FILE]\n", argv[0]);
      return 2;
    }
  }
  zkmi_ctx* ctx = NULL;
  int32_t hv_build = 0, hv_run = 0;
  int32_t rc = zkmi_ctx_create(0, &ctx);
  if (rc != ZKMI_OK) {
    fprintf(stderr, "zkmi_ctx_create -> %d: no gfx950 device; there is no CPU fallback\n", rc);
    return 2;
  }
  CHECK(zkmi_hip_versions(&hv_build, &hv_run));
  printf("%s; HIP build %d, runtime %d (%s)\n", zkmi_version(), hv_build, hv_run,
         hv_build / 100000 == hv_run / 100000 ? "same release" : "DIFFERENT releases");

  if (proofs > 0) {
    /* ---- the bench.py workload ---- */
    const uint32_t total = proofs + warmup;
    zkmi_r1cs* r1 = NULL;
    zkmi_pk* pk = NULL;
    uint32_t n_pub = 0, n_vars = 0;
    double t = now_s();
    CHECK(zkmi_update_note_r1cs(log_n, ZKMI_OP_WITHDRAW, &r1));
    CHECK(zkmi_r1cs_shape(r1, &n_vars, &n_pub, NULL, NULL));
    splitmix g = {0x5A4B0001ull};
    uint8_t toxic[160];
    for (int k = 0; k < 5; k++) sm_fr(&g, toxic + 32 * k);
    uint8_t* vk = malloc(672 + 96 * (size_t)n_pub);
    CHECK(zkmi_groth16_setup(ctx, r1, toxic, &pk, vk, 672 + 96 * (uint64_t)n_pub));
    const double setup_s = now_s() - t;
    t = now_s();
    zkmi_note_update* in = malloc(sizeof(zkmi_note_update) * total);
    void** d_z = malloc(sizeof(void*) * total);
    int32_t* status = malloc(sizeof(int32_t) * total);
    uint8_t *rr = malloc(32 * (size_t)total), *ss = malloc(32 * (size_t)total), *out = malloc(192 * (size_t)total);
    for (uint32_t i = 0; i < total; i++) {
      note_update_from_seed(0x5A4B0000ull + i, &in[i]);  /* bench.py: seeds 0x5A4B0000 + 16 rank + i */
      HIPCHECK(hipMalloc(&d_z[i], (size_t)32 << log_n));
      sm_fr(&g, rr + 32 * i);
      sm_fr(&g, ss + 32 * i);
    }
    CHECK(zkmi_update_note_witness_batch_dev(ctx, log_n, ZKMI_OP_WITHDRAW, in, total, (void* const*)d_z, status));
    for (uint32_t i = 0; i < total; i++)
      if (status[i] != ZKMI_OK) {
        fprintf(stderr, "assignment %u: status %d\n", i, status[i]);
        return 1;
      }
    CHECK(zkmi_ctx_sync(ctx));
    const double wit_s = now_s() - t;
    if (warmup) CHECK(zkmi_groth16_prove_batch_dev(ctx, pk, warmup, (const void* const*)d_z, rr, ss, out));
    HIPCHECK(hipDeviceSynchronize());
    t = now_s();
    CHECK(zkmi_groth16_prove_batch_dev(ctx, pk, proofs, (const void* const*)(d_z + warmup), rr + 32 * warmup, ss + 32 * warmup,
                                       out + 192 * (size_t)warmup));
    const double el = now_s() - t;
    uint32_t bad = 0;
    uint8_t pub[6 * 32];
    for (uint32_t i = 0; i < total; i++) {
      HIPCHECK(hipMemcpy(pub, (const uint8_t*)d_z[i] + 32, 6 * 32, hipMemcpyDeviceToHost));
      if (zkmi_groth16_verify(vk, n_pub, pub, out + 192 * (size_t)i) != ZKMI_OK) bad++;
    }
    /* one proof alone on an idle GPU: the latency a wallet sees */
    uint8_t one[192];
    HIPCHECK(hipDeviceSynchronize());
    t = now_s();
    CHECK(zkmi_groth16_prove_dev(ctx, pk, d_z[0], rr, ss, one));
    const double lat = now_s() - t;
    if (memcmp(one, out, 192) != 0) {
      fprintf(stderr, "single proof differs from the batch's proof of the same inputs\n");
      bad++;
    }
    printf("{\"program\": \"examples/bench_prove.c\", \"log_n\": %u, \"proofs\": %u, \"warmup\": %u, \"distinct_witnesses\": %u, "
           "\"proofs_per_s\": %.3f, \"ms_per_proof\": %.3f, \"single_proof_latency_ms\": %.3f, \"setup_s\": %.2f, \"witness_gen_s\": %.2f, "
           "\"verified_by_pairing\": %u, \"failed\": %u, \"hip_build\": %d, \"hip_runtime\": %d, \"proofs_fnv1a\": \"%016llx\"}\n",
           log_n, proofs, warmup, total, proofs / el, 1e3 * el / proofs, 1e3 * lat, setup_s, wit_s, total - bad, bad, hv_build, hv_run,
           (unsigned long long)fnv1a(out, 192 * (size_t)total, 0xcbf29ce484222325ull));
    if (dump) {
      FILE* f = fopen(dump, "wb");
      if (!f || fwrite(out, 192, total, f) != total) {
        fprintf(stderr, "cannot write %s\n", dump);
        return 1;
      }
      fclose(f);
    }
    for (uint32_t i = 0; i < total; i++) (void)hipFree(d_z[i]);
    (void)zkmi_pk_free(pk);
    (void)zkmi_r1cs_free(r1);
    free(vk), free(in), free(d_z), free(status), free(rr), free(ss), free(out);
    if (bad) return 1;
  }
  if (churn_ops > 0 && churn(ctx, churn_ops, seed)) return 1;
  (void)zkmi_ctx_destroy(ctx);
  return 0;
}
