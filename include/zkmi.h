/* zkmi — C ABI of the MI355X-native proving backend for the Shielder
 * proof-generation hot path (BLS12-381, Groth16-shaped workload).
 *
 * The reference (/root/reference = Cardinal-Cryptography/zk-apps @ v2) has no
 * FFI of its own.  Each entry point below is what a Rust
 * shim behind the reference's only "prover API" would bind — the inherent
 * methods of mocked_zk::relations::ZkProof
 *   new              shielder/mocked_zk/src/relations.rs:37-55
 *   update_account   shielder/mocked_zk/src/relations.rs:79-98   (prove step,
 *                    called at shielder/contract/drink_tests/utils/shielder.rs:105-114)
 *   verify_creation  shielder/mocked_zk/src/relations.rs:127-136 (contract/lib.rs:56)
 *   verify_update    shielder/mocked_zk/src/relations.rs:138-155 (contract/lib.rs:74)
 * and the arithmetic stages they stand in for.
 * INTEGRATION.md shows the reference-side `extern "C"` block.
 *
 * Conventions
 *   - every function returns int32_t: 0 = ZKMI_OK, negative = error; no
 *     exceptions or aborts cross the ABI; zkmi_last_error() gives a string.
 *   - Fr  : 32-byte little-endian canonical integer < r   (= Scalar{bytes:[u8;32]},
 *           shielder/mocked_zk/src/scalar.rs:1-30)
 *   - Fq  : 48-byte little-endian canonical integer < p
 *   - G1 affine : x || y (96 B);  G2 affine : x.c0 || x.c1 || y.c0 || y.c1 (192 B);
 *     all-zero bytes = point at infinity
 *   - proof : 192 B = compressed A (48) || B (96) || C (48), zcash/IETF big-endian
 *   - caller owns every buffer; the library never frees caller memory; device
 *     state lives behind opaque handles bound to one HIP device; a ctx is not
 *     thread-safe (one ctx per host thread / process / GPU).
 *   - *_dev entry points take pointers into HBM (e.g. torch tensors'
 *     data_ptr()); all other pointers are host memory.
 */
#ifndef ZKMI_H
#define ZKMI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKMI_OK 0
#define ZKMI_ERR_BAD_ARG (-1)
#define ZKMI_ERR_NON_CANONICAL (-2)   /* Fr/Fq >= modulus, point not on curve */
#define ZKMI_ERR_HIP (-3)             /* HIP runtime error (see zkmi_last_error) */
#define ZKMI_ERR_NO_DEVICE (-4)       /* no gfx950 device / HIP extension unusable */
#define ZKMI_ERR_VERIFICATION (-5)    /* = ZkpError::VerificationError   (mocked_zk/src/errors.rs:3-7) */
#define ZKMI_ERR_ACCOUNT_UPDATE (-6)  /* = ZkpError::AccountUpdateError */
#define ZKMI_ERR_OPERATION_COMBINE (-7) /* = ZkpError::OperationCombineError */
#define ZKMI_ERR_UNSATISFIED (-8)     /* witness does not satisfy the relation */
#define ZKMI_ERR_RCCL (-9)            /* RCCL missing or a collective failed (see zkmi_last_error) */

typedef struct zkmi_ctx zkmi_ctx;
typedef struct zkmi_bases_g1 zkmi_bases_g1;
typedef struct zkmi_bases_g2 zkmi_bases_g2;
typedef struct zkmi_r1cs zkmi_r1cs;
typedef struct zkmi_pk zkmi_pk;

/* ---- library / context -------------------------------------------------- */
/* "zkmi 0.1 (gfx950) src:<16 hex digits>": the digits are the sha256 digest of the sources the binary was built from
 * (scripts/src_digest.py; bench.py and smoke() compare it with the files beside the library: `library_matches_sources`). */
const char* zkmi_version(void);
/* Where the host-side numbers of zkmi_host_info came from, e.g. "cpus=16 (cgroup2 /sys/fs/cgroup/cpu.max) ranks=8
 * (LOCAL_WORLD_SIZE) threads=2": the CPU grant is min(logical CPUs, affinity mask, the smallest cgroup quota between the
 * process's own cgroup and the root); the ranks sharing it come from LOCAL_WORLD_SIZE (torchrun), OMPI_COMM_WORLD_LOCAL_SIZE,
 * SLURM_NTASKS_PER_NODE or MPI_LOCALNRANKS. */
const char* zkmi_host_info_string(void);
/* Fingerprint of the library's internal struct layouts (sizes and member offsets of the context, the MSM workspaces and the
 * handle types): two builds of this source tree may only share objects (a context made by one, bases made by the other --
 * what the test suite does with the testing library) when their fingerprints are equal.  out_n = values written / needed. */
int32_t zkmi_abi_layout_probe(uint64_t* out, uint32_t cap, uint32_t* out_n);
/* HIP_VERSION the library was built with / hipRuntimeGetVersion of the runtime it is bound to (diagnostics) */
int32_t zkmi_hip_versions(int32_t* out_build, int32_t* out_runtime);
/* The prover's host side: proof assembly (row a10: O(1) scalar multiplications, compression) runs on one persistent pool
 * of threads per process, sized by what the process may actually use -- min(logical CPUs, affinity mask, cgroup CPU quota)
 * divided by the processes of the job on this node (LOCAL_WORLD_SIZE, as torch.distributed.run exports it), at most 16.
 * ZKMI_HOST_THREADS=<n> (environment) or zkmi_set_host_threads override it; n = 0: the library's choice again.
 * zkmi_host_info: out[0] = CPUs granted, out[1] = local ranks, out[2] = threads per process (the caller's included),
 * out[3] = pool workers started so far.  The batch prover's driving thread sleeps between polls while it waits. */
int32_t zkmi_host_info(uint32_t out[4]);
int32_t zkmi_set_host_threads(uint32_t n);
int32_t zkmi_device_count(int32_t* out_count);
int32_t zkmi_ctx_create(int32_t device, zkmi_ctx** out_ctx);
int32_t zkmi_ctx_destroy(zkmi_ctx* ctx);
const char* zkmi_last_error(const zkmi_ctx* ctx);
int32_t zkmi_ctx_sync(zkmi_ctx* ctx);

/* per-phase HIP-event timers on the ctx stream (bench.py's roofline leg) */
#define ZKMI_PH_MSM_SORT 0
#define ZKMI_PH_MSM_ACCUM_G1 1
#define ZKMI_PH_MSM_REDUCE_G1 2
#define ZKMI_PH_MSM_ACCUM_G2 3
#define ZKMI_PH_MSM_REDUCE_G2 4
#define ZKMI_PH_NTT 5
#define ZKMI_PH_WITNESS 6
#define ZKMI_PH_MISC 7
int32_t zkmi_prof_enable(zkmi_ctx* ctx, int32_t on);
int32_t zkmi_prof_reset(zkmi_ctx* ctx);
int32_t zkmi_prof_get(zkmi_ctx* ctx, int32_t phase, double* out_total_ms, uint64_t* out_launches);

/* ---- row a6: Fr NTT ------------------------------------------------------ */
/* In-place radix-2 transform of 2^log_n Fr elements, natural order in/out.
 * inverse: 0 forward, 1 inverse (scaled by N^-1); coset: 0/1 (generator 7).
 * Replaces ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place + coset
 * variants [not in reference tree; SURVEY.md row a6]. */
int32_t zkmi_ntt_fr(zkmi_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset);
/* Same on a device buffer of 2^log_n x 32 B canonical LE elements (in place). */
int32_t zkmi_ntt_fr_dev(zkmi_ctx* ctx, void* d_data, uint32_t log_n, int32_t inverse, int32_t coset);

/* ---- rows a8 / a9: multi-scalar multiplication --------------------------- */
/* Upload n affine points (wire format) and keep them resident in HBM in
 * Montgomery form.  check != 0 additionally verifies y^2 = x^3 + b on the host. */
int32_t zkmi_bases_g1_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bases_g1** out);
int32_t zkmi_bases_g1_free(zkmi_bases_g1* b);
int32_t zkmi_bases_g2_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bases_g2** out);
int32_t zkmi_bases_g2_free(zkmi_bases_g2* b);
/* Bases used for many MSMs of their full length (an SRS, a proving-key query): build the table 2^(c w) * P_i once
 * (ceil(255 / c) x n points of HBM).  Afterwards zkmi_msm_g{1,2}[_dev] with n == the number of bases run the prover's
 * shared-bucket schedule (12-13 insertions per scalar instead of 16); other n, and the *_windows entry points, keep the
 * windowed schedule.  Same results either way. */
int32_t zkmi_bases_g1_prepare(zkmi_ctx* ctx, zkmi_bases_g1* b);
int32_t zkmi_bases_g2_prepare(zkmi_ctx* ctx, zkmi_bases_g2* b);
int32_t zkmi_bases_g1_read(zkmi_ctx* ctx, const zkmi_bases_g1* b, uint64_t first, uint64_t count, uint8_t* out_affine);
int32_t zkmi_bases_g2_read(zkmi_ctx* ctx, const zkmi_bases_g2* b, uint64_t first, uint64_t count, uint8_t* out_affine);

/* sum_i scalars[i] * bases[i]; scalars are n x 32 B canonical LE in host memory.
 * Replaces ark_ec::VariableBaseMSM::msm_bigint / halo2curves::msm::best_multiexp
 * [not in reference tree; SURVEY.md rows a8, a9]. */
int32_t zkmi_msm_g1(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bases_g1* bases, uint8_t out_affine[96]);
int32_t zkmi_msm_g2(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bases_g2* bases, uint8_t out_affine[192]);
/* scalars already resident in HBM (n x 32 B canonical LE) */
int32_t zkmi_msm_g1_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases, uint8_t out_affine[96]);
int32_t zkmi_msm_g2_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g2* bases, uint8_t out_affine[192]);

/* Multi-GPU split of one large MSM (BASELINE config 3): each
 * rank runs the bucket method over its slice of the points and emits one
 * partial sum per window (nwin x 96 B affine, nwin <= 64); the ranks exchange
 * these few KiB (RCCL all-gather of raw bytes, done by the host layer) and
 * every rank combines locally.  *out_window_bits receives c. */
int32_t zkmi_msm_g1_windows_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                uint64_t plan_n, uint8_t* out_windows_affine, uint32_t* out_nwin, uint32_t* out_window_bits);
int32_t zkmi_msm_g1_combine(const uint8_t* windows_affine, uint32_t n_ranks, uint32_t nwin, uint32_t window_bits,
                            uint8_t out_affine[96]);

/* The same exchange behind the C ABI, one process per GPU over RCCL ("ncclAllGather, dtype ncclUint8,
 * + local EC summation and Horner combine on each rank"): the rank's partial sums stay in HBM, are all-gathered on the
 * reduction stream and combined on every rank.  libzkmi.so resolves RCCL at first use (the copy already in the process,
 * else librccl.so.1): nothing here is needed to load the library on a host without RCCL.
 *   zkmi_comm_unique_id   rank 0 draws the 128-byte id (ncclGetUniqueId); the HOST distributes the bytes to all ranks
 *   zkmi_comm_init        every rank: ncclCommInitRank on ctx's device (collective call)
 *   zkmi_comm_from_nccl   wrap an ncclComm_t the host created itself (same RCCL instance; not destroyed with the handle)
 *   zkmi_msm_g1_allgather_combine  this rank's n terms (plan from plan_n = the GLOBAL number of terms, equal on all ranks),
 *                         all-gather, combination: the FULL result on every rank.  Collective: all ranks call it.
 * ZKMI_ERR_RCCL: RCCL not found, or a call failed (zkmi_last_error).
 * Failure semantics of the two collectives: arguments, the plan and every allocation are checked before anything is
 * launched, so ZKMI_ERR_BAD_ARG never leaves the other ranks waiting PROVIDED all ranks pass consistent arguments (the same
 * plan_n, n <= their bases).  After ZKMI_ERR_HIP / ZKMI_ERR_RCCL from a collective the communicator is unusable and the
 * peers may be blocked inside ncclAllGather: abort all ranks (as after any failed NCCL collective).
 * Lifetime: zkmi_comm_destroy may run before or after zkmi_ctx_destroy of the context the handle was made on.
 * Environment: ZKMI_RCCL_LIB=<file> makes that file the only RCCL candidate (deployments with RCCL outside the loader's
 * search path; a missing file yields ZKMI_ERR_RCCL). */
typedef struct zkmi_comm zkmi_comm;
int32_t zkmi_comm_unique_id(uint8_t out_id[128]);
int32_t zkmi_comm_init(zkmi_ctx* ctx, uint32_t n_ranks, uint32_t rank, const uint8_t id[128], zkmi_comm** out);
int32_t zkmi_comm_from_nccl(zkmi_ctx* ctx, void* nccl_comm, uint32_t n_ranks, uint32_t rank, zkmi_comm** out);
int32_t zkmi_comm_destroy(zkmi_comm* comm);
int32_t zkmi_msm_g1_allgather_combine(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n,
                                      const zkmi_bases_g1* bases, uint64_t plan_n, uint8_t out_affine[96]);

/* The WINDOW split (BASELINE configs[3] as worded: "windows split across 8 GPUs"): every rank holds ALL n scalars and
 * bases and computes the sums of a contiguous range of the plan's windows; the ranks' windows are disjoint, so the
 * exchange is a concatenation.  zkmi_msm_g1_window_range_dev: windows [w_first, w_first + w_count) of the plan of plan_n
 * terms -> w_count x 96 B (the caller combines: zkmi_msm_g1_combine over n_ranks x nwin_total windows with all-zero =
 * infinity entries where a rank owns nothing).  zkmi_msm_g1_window_split_allgather: the collective form over a zkmi_comm
 * (rank k takes windows [k nwin / R, (k + 1) nwin / R)); the full result on every rank.  The POINT split above moves 1 / R
 * of the scalars and bases per rank and is what bench.py measures by default. */
int32_t zkmi_msm_g1_window_range_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                     uint64_t plan_n, uint32_t w_first, uint32_t w_count, uint8_t* out_windows_affine,
                                     uint32_t* out_nwin_total, uint32_t* out_window_bits);
int32_t zkmi_msm_g1_window_split_allgather(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n,
                                           const zkmi_bases_g1* bases, uint8_t out_affine[96]);
/* The 2-D split: n_ranks = P x window_groups; rank k = g * window_groups + q holds the points of group g (d_scalars / bases
 * are its slice of n points, as in the point split; plan_n = the size of the whole MSM) and computes only the windows of
 * range q of window_groups (range q: windows [q nwin / Q, (q + 1) nwin / Q)).  Per rank: 1 / n_ranks of the bucket
 * insertions like both 1-D splits, 1 / P of the points resident, 1 / Q of the buckets to reduce -- the share of a rank's
 * time that the point split leaves unscaled.  window_groups = 1 is the point split, = n_ranks the window split (over a
 * slice that is then the whole input).  window_groups must divide the communicator's size (else ZKMI_ERR_BAD_ARG, before
 * anything is launched).  The full result on every rank. */
int32_t zkmi_msm_g1_split2d_allgather(zkmi_ctx* ctx, zkmi_comm* comm, const void* d_scalars, uint64_t n, const zkmi_bases_g1* bases,
                                      uint64_t plan_n, uint32_t window_groups, uint8_t out_affine[96]);

/* What every rank computes AFTER the all-gather, on caller-supplied slots (host arithmetic; no GPU, no RCCL): partials =
 * n_ranks slots of XYZZ points in the form the reductions leave in HBM (4 x 48-byte LE Montgomery coordinates x, y, zz, zzz;
 * all-zero zz = infinity).  window_split = 0: rank k's slot holds the partial sums of ALL windows over its points (the
 * point split); 1: the partial sums of its windows [k nwin / R, (k + 1) nwin / R); Q >= 2 (Q divides n_ranks): the 2-D split
 * with Q window ranges -- rank k's slot holds the partial sums of window range k mod Q over its point group's points.  zkmi_msm_exchange_layout reports the
 * slot geometry for a plan of plan_n terms: out[0] windows, [1] partial sums per window, [2] points per slot (point
 * split), [3] points per slot (window split over n_ranks), [4] bytes per point, [5] window bits, [6] log2 of the
 * reduction's segment length, [7] log2 of the partitions the partial top window is spread over. */
int32_t zkmi_msm_g1_combine_partials(const uint8_t* partials, uint32_t n_ranks, uint64_t plan_n, int32_t window_split,
                                     uint8_t out_affine[96]);
int32_t zkmi_msm_exchange_layout(uint64_t plan_n, uint32_t n_ranks, uint32_t out[8]);

/* Same split driven from ONE process holding one ctx per GPU (SURVEY.md §8b
 * "zkmi_msm_g1_multi", BASELINE config 3): device d holds counts[d] scalars at
 * d_scalars[d] and the matching slice of the points in bases[d] (loaded on
 * ctxs[d]).  All devices are enqueued before any is waited for, then the
 * per-window partial sums are added on the host.  n_dev <= 64. */
int32_t zkmi_msm_g1_multi(zkmi_ctx* const* ctxs, uint32_t n_dev, const void* const* d_scalars, const uint64_t* counts,
                          const zkmi_bases_g1* const* bases, uint8_t out_affine[96]);

/* The bucket plan the library would use for an MSM of n terms (host logic, no GPU): shared != 0 = the prover's
 * shared-bucket schedule over precomputed tables, 0 = the windowed schedule.  out[0..5] = digit bits c, digits per
 * scalar, partitions (shared) or windows (windowed), buckets per partition/window, log2 of the reduction segment
 * length, heavy-bucket threshold. */
int32_t zkmi_msm_plan_query(uint64_t n, int32_t shared, uint32_t out[6]);

/* ---- group / encoding helpers (host) -------------------------------------- */
int32_t zkmi_g1_compress(const uint8_t affine[96], uint8_t out[48]);
int32_t zkmi_g1_decompress(const uint8_t in[48], uint8_t out_affine[96]);
int32_t zkmi_g2_compress(const uint8_t affine[192], uint8_t out[96]);
int32_t zkmi_g2_decompress(const uint8_t in[96], uint8_t out_affine[192]);
int32_t zkmi_g1_mul(const uint8_t affine[96], const uint8_t scalar[32], uint8_t out_affine[96]);
int32_t zkmi_g2_mul(const uint8_t affine[192], const uint8_t scalar[32], uint8_t out_affine[192]);
int32_t zkmi_g1_add(const uint8_t a[96], const uint8_t b[96], uint8_t out_affine[96]);
int32_t zkmi_g2_add(const uint8_t a[192], const uint8_t b[192], uint8_t out_affine[192]);
/* [r]P = O (points on the curve but outside the prime-order subgroup exist on both curves) */
int32_t zkmi_g1_in_subgroup(const uint8_t affine[96]);   /* ZKMI_OK / ZKMI_ERR_NON_CANONICAL */
int32_t zkmi_g2_in_subgroup(const uint8_t affine[192]);
int32_t zkmi_g1_generator(uint8_t out_affine[96]);
int32_t zkmi_g2_generator(uint8_t out_affine[192]);
/* reduced optimal-ate pairing e(P,Q) as 12 x 48-byte LE Fq coefficients in the
 * tower basis c0.c0.c0, c0.c0.c1, c0.c1.c0, ... (Fq12 = Fq6[w], Fq6 = Fq2[v]) */
int32_t zkmi_pairing(const uint8_t g1_affine[96], const uint8_t g2_affine[192], uint8_t out_fq12[576]);

/* ---- rows a1-a5, a7: relation + witness ---------------------------------- */
/* R1CS in CSR form; column 0 is the constant 1, columns [0, n_pub) are the
 * instance variables; values are 32-byte canonical LE Fr. */
int32_t zkmi_r1cs_create(uint32_t n_vars, uint32_t n_pub, uint32_t n_constraints,
                         const uint32_t* a_rowptr, const uint32_t* a_col, const uint8_t* a_val,
                         const uint32_t* b_rowptr, const uint32_t* b_col, const uint8_t* b_val,
                         const uint32_t* c_rowptr, const uint32_t* c_col, const uint8_t* c_val,
                         zkmi_r1cs** out);
int32_t zkmi_r1cs_free(zkmi_r1cs* r);
/* Semantic inputs of the hash-free chain stand-in of the first builds (the relation itself and its assignment
 * generators are test scaffolding now: include/zkmi_testing.h; the relation with real hashing is zkmi_update_note_*). */
typedef struct {
  uint8_t bytes[32];
} zkmi_fr;
typedef struct {
  zkmi_fr amount, token, user; /* op_pub.into()                                  */
  zkmi_fr old_nullifier;       /* public: old_note.nullifier                     */
  zkmi_fr new_note[4];         /* zk_id, trapdoor, nullifier, account_hash       */
  zkmi_fr old_trapdoor, old_account_hash;
  uint8_t path_shape[10];      /* MerkleProof::path_shape (0/1)                  */
  zkmi_fr path[10];            /* MerkleProof::path                              */
  zkmi_fr old_account[2];      /* TOKENS_NUMBER balances                         */
} zkmi_update_note_input;
/* value mod r (SHA-256 outputs used as Scalars by the mock can exceed r) */
int32_t zkmi_fr_reduce(const uint8_t in[32], uint8_t out[32]);
int32_t zkmi_r1cs_shape(const zkmi_r1cs* r, uint32_t* n_vars, uint32_t* n_pub, uint32_t* n_constraints, uint32_t* log_n);
/* export matrix m (0=A,1=B,2=C) as CSR; pass NULL buffers to query nnz */
int32_t zkmi_r1cs_export(const zkmi_r1cs* r, int32_t m, uint32_t* rowptr, uint32_t* col, uint8_t* val, uint64_t* nnz);
int32_t zkmi_r1cs_is_satisfied(const zkmi_r1cs* r, const uint8_t* z);

/* ---- BN254 (halo2curves::bn256) MSM / NTT, KZG-commit-shaped driver ------ *
 * The reference's relations are halo2 circuits over bn256 (shielder/Cargo.toml:26,
 * shielder/Cargo.lock:436-492); a halo2/KZG prover's hot loops are best_fft and best_multiexp
 * (crates not in tree).  Same kernels as above over the 254-bit fields.
 * Fr / Fq: 32-byte LE canonical; G1 affine: x || y (64 B), all-zero = infinity;
 * curve y^2 = x^3 + 3, generator (1, 2); NTT root = 7^((r-1)/2^28) (bn256::Fr::ROOT_OF_UNITY),
 * coset shift 7 (bn256::Fr::MULTIPLICATIVE_GENERATOR). */
typedef struct zkmi_bn_bases zkmi_bn_bases;
int32_t zkmi_bn254_bases_load(zkmi_ctx* ctx, const uint8_t* affine, uint64_t n, int32_t check, zkmi_bn_bases** out);
int32_t zkmi_bn254_bases_read(zkmi_ctx* ctx, const zkmi_bn_bases* b, uint64_t first, uint64_t count, uint8_t* out);
int32_t zkmi_bn254_bases_free(zkmi_bn_bases* b);
/* Fixed-base preparation of an SRS that is committed to many times (a halo2 prover commits every
 * column against the same ParamsKZG): builds the table 2^(c w) * P_i; MSMs / commitments of exactly
 * b->n terms then use one shared set of buckets (13 instead of 16 insertions per scalar). */
int32_t zkmi_bn254_srs_prepare(zkmi_ctx* ctx, zkmi_bn_bases* b);
int32_t zkmi_bn254_msm_g1(zkmi_ctx* ctx, const uint8_t* scalars, uint64_t n, const zkmi_bn_bases* bases, uint8_t out_affine[64]);
int32_t zkmi_bn254_msm_g1_dev(zkmi_ctx* ctx, const void* d_scalars, uint64_t n, const zkmi_bn_bases* bases, uint8_t out_affine[64]);
int32_t zkmi_bn254_ntt_fr(zkmi_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset);
int32_t zkmi_bn254_ntt_fr_dev(zkmi_ctx* ctx, void* d_data, uint32_t log_n, int32_t inverse, int32_t coset);
/* KZG commitment from evaluations: coefficients = iNTT(evaluations) (written back to d_evals),
 * commitment = MSM(srs, coefficients); srs holds at least 2^log_n points [tau^i] G. */
int32_t zkmi_bn254_kzg_commit_dev(zkmi_ctx* ctx, void* d_evals, uint32_t log_n, const zkmi_bn_bases* srs,
                                  uint8_t out_commitment[64]);
/* KZG opening of one polynomial at one point -- halo2_proofs::arithmetic::eval_polynomial + kate_division and the
 * commitment to the quotient (poly::kzg::multiopen's inner step; halo2_proofs is a git dependency of the reference,
 * shielder/Cargo.toml:26, not in the tree): d_coeffs = n coefficients in HBM (32-byte LE canonical, constant term first),
 * *out_eval = p(zeta), out_proof = commit(q) for q(X) = (p(X) - p(zeta)) / (X - zeta).  srs holds at least n - 1 points
 * [tau^i] G (a prepared SRS is used through its table).  d_quotient: NULL, or n - 1 x 32 B in HBM that receive q. */
int32_t zkmi_bn254_kzg_open_dev(zkmi_ctx* ctx, const void* d_coeffs, uint64_t n, const uint8_t zeta[32], const zkmi_bn_bases* srs,
                                void* d_quotient, uint8_t out_eval[32], uint8_t out_proof[64]);

/* k polynomials of n coefficients each opened at ONE point (a rotation set of poly::kzg::multiopen): d_polys = k device
 * pointers (a host array), out_evals = k x 32 B, out_evals[j] = p_j(zeta); out_proof = commit((f - f(zeta)) / (X - zeta)) for
 * f = sum_j v^j p_j (the verifier folds commitments and evaluations with the same powers of v). */
int32_t zkmi_bn254_kzg_open_many_dev(zkmi_ctx* ctx, const void* const* d_polys, uint32_t k, uint64_t n, const uint8_t zeta[32],
                                     const uint8_t v[32], const zkmi_bn_bases* srs, uint8_t* out_evals, uint8_t out_proof[64]);
/* Grand product of a permutation argument (halo2_proofs::plonk::permutation::prover::commit: batch_invert of the
 * denominators + running product; PLONK's z(X)): d_out[0] = 1, d_out[i] = prod_{j < i} d_num[j] / d_den[j] for i < n,
 * out_total = the product over all n terms.  All arrays: n x 32-byte LE canonical Fr in HBM.  ZKMI_ERR_BAD_ARG on a zero
 * denominator (d_out is then unspecified). */
int32_t zkmi_bn254_grand_product_dev(zkmi_ctx* ctx, const void* d_num, const void* d_den, uint64_t n, void* d_out,
                                     uint8_t out_total[32]);

/* ---- the contract's SHA-256 Merkle tree, batched ------------ *
 * compute_hash / combine_merkle_hash = SHA-256(first.bytes || second.bytes)
 * (shielder/contract/merkle.rs:24-28, shielder/mocked_zk/src/lib.rs:24-28).
 * in: n_hashes x 2 x 32 B, out: n_hashes x 32 B. */
int32_t zkmi_sha256_pairs(zkmi_ctx* ctx, const uint8_t* in, uint64_t n_hashes, uint8_t* out);
int32_t zkmi_sha256_pairs_dev(zkmi_ctx* ctx, const void* d_in, uint64_t n_hashes, void* d_out);
/* MerkleTree<DEPTH> after its first n_filled add_leaf calls (merkle.rs:48-81): d_nodes holds
 * 2^(log_leaves+1) - 1 elements of 32 B, leaves first, then each level, root last; the caller
 * writes the first n_filled leaves.  A node no inserted leaf has touched does not exist in the
 * contract's mapping and reads as Scalar 0 (not as a hash of zeros) - reproduced here. */
int32_t zkmi_sha256_merkle_tree_dev(zkmi_ctx* ctx, void* d_nodes, uint32_t log_leaves, uint64_t n_filled);

/* ---- Poseidon-5 ------------------------------------------ *
 * T_WIDTH = 5, RATE = 4, R_F = 8, R_P = 56, S-box x^5
 * (shielder/relations/src/lib.rs:17-26); hashing = PoseidonHasher::hash_fix_len_array
 * (update_note.rs:100,131; update_account.rs:62; merkle_proof.rs:56): state
 * [2^64,0,0,0,0], inputs added to state[1..], a 1 added behind the last input,
 * an extra padding-only permutation when len % 4 == 0, output state[1].
 * Constants are generated with the Grain LFSR procedure of the Poseidon paper
 * (what halo2-base's OptimizedPoseidonSpec::new::<8,56,0>() runs; crate not in
 * tree).  field: the scalar field of BLS12-381 (the prover's) or of BN254 (the
 * field the reference's relations are written over). */
#define ZKMI_FIELD_BLS12_381_FR 0
#define ZKMI_FIELD_BN254_FR 1
/* constants in canonical LE form: out_rc = 64 x 5 x 32 B (round-major), out_mds = 5 x 5 x 32 B (row-major) */
int32_t zkmi_poseidon_spec(int32_t field, uint8_t* out_rc, uint8_t* out_mds);
/* n_hashes independent hash_fix_len_array calls of `arity` (0..64) inputs each, on the device.
 * in: n_hashes x arity x 32 B canonical LE (< modulus), out: n_hashes x 32 B. */
int32_t zkmi_poseidon_hash_batch(zkmi_ctx* ctx, int32_t field, const uint8_t* in, uint64_t n_hashes, uint32_t arity,
                                 uint8_t* out);
int32_t zkmi_poseidon_hash_batch_dev(zkmi_ctx* ctx, int32_t field, const void* d_in, uint64_t n_hashes, uint32_t arity,
                                     void* d_out);
/* Binary Poseidon Merkle tree (node = hash_fix_len_array([left, right]), merkle_proof.rs:56):
 * d_nodes holds 2^(log_leaves+1) - 1 elements of 32 B; the caller fills the first 2^log_leaves
 * (the leaves); each level is appended behind the previous one, the root is the last element. */
int32_t zkmi_poseidon_merkle_tree_dev(zkmi_ctx* ctx, int32_t field, void* d_nodes, uint32_t log_leaves);
/* Merkle paths of n leaves out of that node array: out_shape = n x log_leaves selector bytes
 * (0: the sibling is the left input, merkle_proof.rs:53-55), out_paths = n x log_leaves x 32 B siblings
 * (host buffers) = MerkleProof::{path_shape, path} of zkmi_note_update. */
int32_t zkmi_poseidon_merkle_paths_dev(zkmi_ctx* ctx, const void* d_nodes, uint32_t log_leaves, const uint32_t* leaf_idx,
                                       uint32_t n, uint8_t* out_shape, uint8_t* out_paths);
/* n Merkle-path recomputations (the value CircuitMerkleProof::verify compares with the root):
 * d_leaves n x 32 B, d_shape n x depth bytes, d_paths n x depth x 32 B -> d_roots n x 32 B. */
int32_t zkmi_poseidon_merkle_roots_dev(zkmi_ctx* ctx, int32_t field, const void* d_leaves, const void* d_shape,
                                       const void* d_paths, uint32_t depth, uint64_t n, void* d_roots);

/* ---- rows a1-a5 with real hashing: the update_note relation ---------------- *
 * update_note_circuit (shielder/relations/src/relations/update_note.rs:106-149) with
 * verify_note_circuit (:91-103), CircuitMerkleProof::verify (merkle_proof.rs:38-61)
 * and update_account_circuit (update_account.rs:68-95) over the concrete account the
 * mock defines (two (token, balance) slots, u128 balances; mocked_zk/src/account.rs),
 * arithmetised as R1CS with Poseidon-5 and padded with the multiplication chain to
 * constraints + publics = 2^log_n, variables = 2^log_n (log_n >= 13).
 * Public inputs, in the order update_note.rs:121,127 fixes:
 *   amount, token, user, new_note_hash, merkle_root, old_note.nullifier. */
#define ZKMI_OP_DEPOSIT 0
#define ZKMI_OP_WITHDRAW 1
#define ZKMI_MAX_TREE_HEIGHT 32
typedef struct {
  zkmi_fr amount, token, user;   /* op_pub                                          */
  zkmi_fr new_note[3];           /* zk_id, trapdoor, nullifier (account_hash derived) */
  zkmi_fr old_note[3];           /* zk_id, trapdoor, nullifier                      */
  /* TREE_HEIGHT is a const generic of the reference (merkle_proof.rs:11); here a run-time field:
   * the first tree_height entries of path_shape / path are used.  0 = ZKMI_MERKLE_TREE_DEPTH (10). */
  uint32_t tree_height;
  uint8_t path_shape[ZKMI_MAX_TREE_HEIGHT]; /* MerkleProof::path_shape              */
  zkmi_fr path[ZKMI_MAX_TREE_HEIGHT];       /* MerkleProof::path                    */
  zkmi_fr op_priv_user;          /* op_priv                                         */
  zkmi_fr account[4];            /* old account: token_0, balance_0, token_1, balance_1 */
} zkmi_note_update;
/* log_n >= 13 (the relation proper has ~5.7 k constraints at height 10); tree_height 1..32 */
int32_t zkmi_update_note_r1cs(uint32_t log_n, int32_t op_kind, zkmi_r1cs** out); /* height 10 */
int32_t zkmi_update_note_r1cs_h(uint32_t log_n, int32_t op_kind, uint32_t tree_height, zkmi_r1cs** out);
/* Full assignment (2^log_n x 32 B) from the semantic inputs; the note/account hashes, the
 * Merkle root and every S-box intermediate are computed here.  out_publics (optional)
 * receives the 6 public inputs.  Returns ZKMI_ERR_ACCOUNT_UPDATE / ZKMI_ERR_OPERATION_COMBINE
 * (the mock's ZkpError variants) when the update is impossible: the assignment is still
 * written but does not satisfy the relation. */
int32_t zkmi_update_note_witness(uint32_t log_n, int32_t op_kind, const zkmi_note_update* in, uint8_t* out_z,
                                 uint8_t* out_publics);

/* SURVEY.md 8f-1 "witness generation on device": n instances at once, one GPU thread per
 * instance walking the same statement sequence (value-only).  `in` = host array of n inputs,
 * d_z_out[i] = device buffer of 2^log_n x 32 B for instance i (ready for zkmi_groth16_prove_dev /
 * _prove_batch_dev), out_status[i] = ZKMI_OK or the mock's error code for that instance. */
int32_t zkmi_update_note_witness_batch_dev(zkmi_ctx* ctx, uint32_t log_n, int32_t op_kind, const zkmi_note_update* in,
                                           uint32_t n, void* const* d_z_out, int32_t* out_status);
/* the same device code executed on the host for one instance (test hook: must equal
 * zkmi_update_note_witness byte for byte) */
int32_t zkmi_update_note_witness_values_host(uint32_t log_n, int32_t op_kind, const zkmi_note_update* in, uint8_t* out_z);

/* ---- rows a7, a10: Groth16 ------------------------------------------------ */
/* Trusted setup with explicit toxic waste tau|alpha|beta|gamma|delta
 * (5 x 32 B), heavy part (fixed-base multiplications) on the device.  The
 * proving key stays resident in HBM; vk_out receives
 *   alpha_g1 (96) | beta_g2 (192) | gamma_g2 (192) | delta_g2 (192) | n_pub x gamma_abc_g1 (96 each). */
int32_t zkmi_groth16_setup(zkmi_ctx* ctx, const zkmi_r1cs* r1cs, const uint8_t toxic[160], zkmi_pk** out_pk,
                           uint8_t* vk_out, uint64_t vk_cap);
/* Proofs of small domains travel through the batch prover in groups (one digit sort and one accumulation launch per
 * query for up to 64 proofs).  group = 0: the library's choice (about 2^20 constraints per group), 1: never group,
 * 2..64: at most that many (clamped to what the key's domain allows).  Applies to keys created on `ctx` AFTER the call. */
int32_t zkmi_ctx_set_group_size(zkmi_ctx* ctx, uint32_t group);
/* Load a proving key from host arrays in wire format (drop-in for a key
 * produced by another Groth16 implementation). */
int32_t zkmi_pk_load(zkmi_ctx* ctx, const zkmi_r1cs* r1cs, const uint8_t alpha_g1[96], const uint8_t beta_g1[96],
                     const uint8_t beta_g2[192], const uint8_t delta_g1[96], const uint8_t delta_g2[192],
                     const uint8_t* a_query, const uint8_t* b_g1_query, const uint8_t* b_g2_query,
                     const uint8_t* h_query, const uint8_t* l_query, zkmi_pk** out_pk);
int32_t zkmi_pk_free(zkmi_pk* pk);
int32_t zkmi_pk_shape(const zkmi_pk* pk, uint32_t* n_vars, uint32_t* n_pub, uint32_t* log_n);
int32_t zkmi_pk_export_g1_elems(const zkmi_pk* pk, uint8_t out_beta_g1[96], uint8_t out_delta_g1[96]);
/* What the prover learned from the last finished proof (group) of this key, and what it decided from it: out[0] = 1 while
 * the B1 MSM is folded into the reduction L and H share (taken over r z; only while the assignments fill at least 9 in 10
 * of their digits -- a witness of bits is cheaper through the sort of z), out[1] = non-zero digits the digit sort of that
 * proof's (group's) assignment(s) placed, out[2] = digits a completely dense assignment would have placed.  Diagnostic:
 * the proof bytes do not depend on it. */
int32_t zkmi_pk_schedule_state(const zkmi_pk* pk, uint64_t out[3]);
/* export one query of a resident key (0=a,1=b_g1,2=b_g2,3=h,4=l) in wire format */
int32_t zkmi_pk_export_query(zkmi_ctx* ctx, const zkmi_pk* pk, int32_t which, uint64_t first, uint64_t count, uint8_t* out);
/* witness -> proof.  z = full assignment (n_vars x 32 B, z[0] = 1); r, s =
 * prover randomness (32 B each, explicit so proofs are reproducible).
 * Replaces ark_groth16::prover::create_proof_with_assignment [not in tree]. */
int32_t zkmi_groth16_prove(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* z, const uint8_t r[32], const uint8_t s[32],
                           uint8_t out_proof[192]);
/* same with the witness already resident in HBM (n_vars x 32 B canonical LE) — the entry point bench.py
 * times.  Every prove entry point checks on the device that the elements are < r, that z[0] = 1 and that
 * the assignment satisfies the relation (ZKMI_ERR_NON_CANONICAL / ZKMI_ERR_UNSATISFIED). */
int32_t zkmi_groth16_prove_dev(zkmi_ctx* ctx, const zkmi_pk* pk, const void* d_z, const uint8_t r[32], const uint8_t s[32],
                               uint8_t out_proof[192]);
/* Batch of n_proofs independent proofs over one resident key (BASELINE config 2):
 * d_z[i] = device pointer to witness i, r/s = n_proofs x 32 B, out = n_proofs x 192 B.
 * Two proofs are kept in flight (GPU works on proof i+1 while the CPU finishes proof i). */
int32_t zkmi_groth16_prove_batch_dev(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t n_proofs, const void* const* d_z,
                                     const uint8_t* r, const uint8_t* s, uint8_t* out_proofs);
/* The same pipeline over HOST witnesses (z[i] = n_vars x 32 B): witness i+1 is uploaded on a copy stream
 * while earlier proofs compute.  Pin the buffers for a truly asynchronous copy.  Elements >= r are
 * detected on the device (ZKMI_ERR_NON_CANONICAL). */
int32_t zkmi_groth16_prove_batch(zkmi_ctx* ctx, const zkmi_pk* pk, uint32_t n_proofs, const uint8_t* const* z,
                                 const uint8_t* r, const uint8_t* s, uint8_t* out_proofs);
/* h-polynomial coefficients only (row a7), N x 32 B canonical LE */
int32_t zkmi_groth16_witness_map(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* z, uint8_t* out_h);
/* BASELINE config 2 inside ONE host process: a batch of independent proofs over n_dev GPUs,
 * proof i -> ctxs[i % n_dev]; pks[d] = the same key set up / loaded on ctxs[d] (replicas); one host thread per
 * device, no collective.  z[i]: host pointer (z_on_device = 0) or device pointer on ctxs[i % n_dev]'s GPU
 * (z_on_device = 1).  r, s: n_proofs x 32 B; out_proofs: n_proofs x 192 B in the caller's order.
 * This is what a Rust host calls where the reference's callers loop over actors and call
 * ZkProof::update_account one by one (shielder/drink_tests/mod.rs:133-207). */
int32_t zkmi_groth16_prove_batch_multi(zkmi_ctx* const* ctxs, const zkmi_pk* const* pks, uint32_t n_dev, uint32_t n_proofs,
                                       const void* const* z, int32_t z_on_device, const uint8_t* r, const uint8_t* s,
                                       uint8_t* out_proofs);

/* row a11: pairing check on the host CPU.  publics excludes the leading 1.
 * Validating, like arkworks' / zcash's deserialisation: proof points must be canonical compressed
 * encodings (one encoding of infinity) of points in the r-order subgroups, and so must the vk's points;
 * anything else returns ZKMI_ERR_NON_CANONICAL, a failed pairing equation ZKMI_ERR_VERIFICATION. */
int32_t zkmi_groth16_verify(const uint8_t* vk, uint32_t n_pub, const uint8_t* publics, const uint8_t proof[192]);

/* ---- keys in arkworks' CanonicalSerialize layout (drop-in for keys made by ark-groth16) -------- *
 * VerifyingKey = alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | Vec<gamma_abc_g1>;
 * ProvingKey = vk | beta_g1 | delta_g1 | Vec a | Vec b_g1 | Vec b_g2 | Vec h | Vec l; Vec = u64 LE length + elements;
 * points in the zcash-style big-endian encoding, compressed (48 / 96 B) or not (96 / 192 B).
 * Layout restated from ark-groth16 0.4 / ark-serialize 0.4 / ark-bls12-381 0.4 [not in the reference tree;
 * oracle/README.md rows 8, 10].  A compressed Proof is this library's 192-byte proof as is. */
int32_t zkmi_ark_vk_read(const uint8_t* buf, uint64_t len, int32_t compressed, uint8_t* out_vk, uint64_t vk_cap,
                         uint32_t* out_n_pub, uint64_t* out_consumed);
/* out may be NULL to query the size; *out_len receives the bytes needed / written */
int32_t zkmi_ark_vk_write(const uint8_t* vk, uint32_t n_pub, int32_t compressed, uint8_t* out, uint64_t cap, uint64_t* out_len);
/* proving key for `r1cs` from arkworks bytes (shape must match: n_vars, n_pub, domain); out_vk optional */
int32_t zkmi_ark_pk_load(zkmi_ctx* ctx, const zkmi_r1cs* r1cs, const uint8_t* buf, uint64_t len, int32_t compressed,
                         int32_t check_curve, zkmi_pk** out_pk, uint8_t* out_vk, uint64_t vk_cap);
int32_t zkmi_ark_pk_write(zkmi_ctx* ctx, const zkmi_pk* pk, const uint8_t* vk, int32_t compressed, uint8_t* out,
                          uint64_t cap, uint64_t* out_len);

/* ---- row a12: the reference's prove/verify surface ------------------------ */
#define ZKMI_MERKLE_TREE_DEPTH 10 /* shielder/mocked_zk/src/lib.rs:16 */
#define ZKMI_TOKENS_NUMBER 2      /* shielder/mocked_zk/src/lib.rs:17 */
typedef struct { uint8_t bytes[32]; } zkmi_scalar;                 /* scalar.rs:1-6 */
typedef struct { zkmi_scalar balances[ZKMI_TOKENS_NUMBER][2]; } zkmi_account; /* account.rs:10-14, (token, balance) */
typedef struct { uint32_t kind; /* 0 Deposit, 1 Withdraw */ uint8_t amount[16]; zkmi_scalar token; zkmi_scalar user; } zkmi_op_pub; /* ops.rs:4-25 */
typedef struct { zkmi_scalar user; } zkmi_op_priv;                 /* ops.rs:28-37 */
typedef struct {                                                   /* relations.rs:14-26 */
  zkmi_scalar id, trapdoor_new, trapdoor_old, nullifier_new;
  zkmi_account acc_old, acc_new;
  zkmi_op_priv op_priv;
  zkmi_scalar merkle_proof[ZKMI_MERKLE_TREE_DEPTH];
  uint32_t merkle_proof_leaf_id;
} zkmi_zkproof;
int32_t zkmi_scalar_from_u128(const uint8_t le16[16], zkmi_scalar* out);      /* scalar.rs:14-24 */
int32_t zkmi_scalar_to_u128(const zkmi_scalar* s, uint8_t out_le16[16]);      /* scalar.rs:26-30 */
int32_t zkmi_account_new(const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], zkmi_account* out); /* account.rs:27-34 */
int32_t zkmi_account_hash(const zkmi_account* a, zkmi_scalar* out);           /* account.rs:16-24 */
int32_t zkmi_account_update(const zkmi_account* a, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv, zkmi_account* out); /* account.rs:36-79 */
int32_t zkmi_note_hash(const zkmi_scalar* id, const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                       const zkmi_scalar* account_hash, zkmi_scalar* out);    /* note.rs:25-40 */
int32_t zkmi_combine_merkle_hash(const zkmi_scalar* first, const zkmi_scalar* second, zkmi_scalar* out); /* lib.rs:24-28 */
int32_t zkmi_operation_combine(const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv);   /* ops.rs:47-63 */
int32_t zkmi_zkproof_new(const zkmi_scalar* id, const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                         const zkmi_op_priv* op_priv, const zkmi_account* acc, zkmi_zkproof* out); /* relations.rs:37-55 */
int32_t zkmi_zkproof_update_account(const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv,
                                    const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                                    const zkmi_scalar merkle_proof[ZKMI_MERKLE_TREE_DEPTH], uint32_t leaf_id,
                                    zkmi_scalar* out_h_note_new, zkmi_zkproof* out_new);  /* relations.rs:79-98 */
int32_t zkmi_zkproof_verify_creation(const zkmi_zkproof* self, const zkmi_scalar* h_note_new,
                                     const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER]);       /* relations.rs:127-136 */
int32_t zkmi_zkproof_verify_update(const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_scalar* h_note_new,
                                   const zkmi_scalar* merkle_root, const zkmi_scalar* nullifier_old); /* relations.rs:138-155 */


/* ---- the same surface with REAL proofs --------------------------------------- *
 * The mock's "proof" is the witness; these entry points produce / check a Groth16 proof of the
 * Poseidon relations instead.  Scalars are arbitrary 32-byte strings in the mock; the relations
 * work over Fr, so every Scalar enters as its value mod r (zkmi_fr_reduce) and `amount` as its
 * u128 value.  h_note_new / merkle_root are Poseidon outputs (canonical Fr) produced by the prover.
 *
 * Creation relation = what verify_creation stands for (relations.rs:127-136, contract/lib.rs:50-58):
 * verify_account_circuit (update_account.rs:52-65) on Account::new(tokens) (account.rs:27-34) +
 * verify_note_circuit (update_note.rs:91-103).  Publics: h_note_new, token_0, token_1. */
typedef struct {
  zkmi_fr tokens[ZKMI_TOKENS_NUMBER];
  zkmi_fr note[3]; /* zk_id, trapdoor, nullifier */
} zkmi_note_create;
int32_t zkmi_create_note_r1cs(uint32_t log_n, zkmi_r1cs** out); /* log_n >= 11 */
int32_t zkmi_create_note_witness(uint32_t log_n, const zkmi_note_create* in, uint8_t* out_z, uint8_t* out_publics /* 3 x 32 B */);
/* ZkProof::new (relations.rs:37-55) + a proof that verify_creation's statement holds */
int32_t zkmi_shielder_prove_creation(zkmi_ctx* ctx, const zkmi_pk* pk_create, const zkmi_zkproof* knowledge,
                                     const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], const uint8_t r[32], const uint8_t s[32],
                                     zkmi_scalar* out_h_note_new, uint8_t out_proof[192]);
/* verify_creation (relations.rs:127-136): ZKMI_OK or ZKMI_ERR_VERIFICATION */
int32_t zkmi_shielder_verify_creation(const uint8_t* vk_create, const zkmi_scalar* h_note_new,
                                      const zkmi_scalar tokens[ZKMI_TOKENS_NUMBER], const uint8_t proof[192]);
/* ZkProof::update_account (relations.rs:79-98): same arguments (merkle_proof = tree_height siblings,
 * leaf first; leaf_id), same error codes (ZKMI_ERR_OPERATION_COMBINE / ZKMI_ERR_ACCOUNT_UPDATE), returns
 * (h_note_new, new ZkProof) plus the recomputed Merkle root and the 192-byte proof.  The key is picked by
 * op_pub->kind (one circuit per operation kind: deposit adds, withdraw subtracts). */
int32_t zkmi_shielder_prove_update(zkmi_ctx* ctx, const zkmi_pk* pk_deposit, const zkmi_pk* pk_withdraw,
                                   const zkmi_zkproof* self, const zkmi_op_pub* op_pub, const zkmi_op_priv* op_priv,
                                   const zkmi_scalar* trapdoor, const zkmi_scalar* nullifier,
                                   const zkmi_scalar* merkle_proof, uint32_t tree_height, uint32_t leaf_id,
                                   const uint8_t r[32], const uint8_t s[32], zkmi_scalar* out_h_note_new,
                                   zkmi_scalar* out_merkle_root, zkmi_zkproof* out_new, uint8_t out_proof[192]);
/* verify_update (relations.rs:138-155, called at contract/lib.rs:74): OpPub -> publics
 * (amount, token, user | h_note_new | merkle_root | nullifier_old: ops.rs:6-25, update_note.rs:121,127) */
int32_t zkmi_shielder_verify_update(const uint8_t* vk_deposit, const uint8_t* vk_withdraw, const zkmi_op_pub* op_pub,
                                    const zkmi_scalar* h_note_new, const zkmi_scalar* merkle_root,
                                    const zkmi_scalar* nullifier_old, const uint8_t proof[192]);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
