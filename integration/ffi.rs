//! `mocked_zk/src/ffi.rs` — bindings to libzkmi.so (include/zkmi.h).  Shown, not compiled (no Rust toolchain in
//! the build image); every function below is exported by the library and exercised through the same C ABI by
//! tests/ (ctypes) and examples/prove_withdraw.c (plain C).
#![allow(non_camel_case_types, dead_code)]
use core::ffi::c_char;

pub const ZKMI_OK: i32 = 0;
pub const ZKMI_ERR_BAD_ARG: i32 = -1;
pub const ZKMI_ERR_NON_CANONICAL: i32 = -2;
pub const ZKMI_ERR_HIP: i32 = -3;
pub const ZKMI_ERR_NO_DEVICE: i32 = -4;
pub const ZKMI_ERR_VERIFICATION: i32 = -5; // ZkpError::VerificationError   (mocked_zk/src/errors.rs:3-7)
pub const ZKMI_ERR_ACCOUNT_UPDATE: i32 = -6; // ZkpError::AccountUpdateError
pub const ZKMI_ERR_OPERATION_COMBINE: i32 = -7; // ZkpError::OperationCombineError
pub const ZKMI_ERR_UNSATISFIED: i32 = -8;

pub const MERKLE_TREE_DEPTH: usize = 10; // mocked_zk/src/lib.rs:16
pub const TOKENS_NUMBER: usize = 2; // mocked_zk/src/lib.rs:17
pub const ZKMI_MAX_TREE_HEIGHT: usize = 32;

#[repr(C)] pub struct zkmi_ctx { _p: [u8; 0] }
#[repr(C)] pub struct zkmi_pk { _p: [u8; 0] }
#[repr(C)] pub struct zkmi_r1cs { _p: [u8; 0] }
#[repr(C)] pub struct zkmi_bases_g1 { _p: [u8; 0] }
#[repr(C)] pub struct zkmi_bases_g2 { _p: [u8; 0] }
#[repr(C)] pub struct zkmi_comm { _p: [u8; 0] }

/// = `Scalar { bytes: [u8; 32] }` (mocked_zk/src/scalar.rs:1-6), little-endian
#[repr(C)] #[derive(Clone, Copy)] pub struct zkmi_scalar { pub bytes: [u8; 32] }
/// a canonical Fr element (< r), little-endian
#[repr(C)] #[derive(Clone, Copy)] pub struct zkmi_fr { pub bytes: [u8; 32] }
/// = `Account { balances: [(Scalar, Scalar); TOKENS_NUMBER] }` (account.rs:10-14): (token, balance)
#[repr(C)] #[derive(Clone, Copy)] pub struct zkmi_account { pub balances: [[zkmi_scalar; 2]; TOKENS_NUMBER] }
/// = `OpPub::{Deposit, Withdraw} { amount: u128, token, user }` (ops.rs:4-25); kind 0 = Deposit, 1 = Withdraw
#[repr(C)] #[derive(Clone, Copy)] pub struct zkmi_op_pub { pub kind: u32, pub amount: [u8; 16], pub token: zkmi_scalar, pub user: zkmi_scalar }
/// = `OpPriv { user }` (ops.rs:28-37)
#[repr(C)] #[derive(Clone, Copy)] pub struct zkmi_op_priv { pub user: zkmi_scalar }
/// = `ZkProof` (relations.rs:14-26), field for field
#[repr(C)] #[derive(Clone, Copy)]
pub struct zkmi_zkproof {
    pub id: zkmi_scalar, pub trapdoor_new: zkmi_scalar, pub trapdoor_old: zkmi_scalar, pub nullifier_new: zkmi_scalar,
    pub acc_old: zkmi_account, pub acc_new: zkmi_account, pub op_priv: zkmi_op_priv,
    pub merkle_proof: [zkmi_scalar; MERKLE_TREE_DEPTH], pub merkle_proof_leaf_id: u32,
}
/// semantic inputs of the update_note relation, in the load order of `UpdateNoteInput::new`
/// (relations/src/relations/update_note.rs:47-88); `tree_height` = the const generic TREE_HEIGHT (merkle_proof.rs:11)
#[repr(C)] #[derive(Clone, Copy)]
pub struct zkmi_note_update {
    pub amount: zkmi_fr, pub token: zkmi_fr, pub user: zkmi_fr,
    pub new_note: [zkmi_fr; 3], pub old_note: [zkmi_fr; 3],
    pub tree_height: u32,
    pub path_shape: [u8; ZKMI_MAX_TREE_HEIGHT], pub path: [zkmi_fr; ZKMI_MAX_TREE_HEIGHT],
    pub op_priv_user: zkmi_fr, pub account: [zkmi_fr; 4],
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct zkmi_note_create { pub tokens: [zkmi_fr; TOKENS_NUMBER], pub note: [zkmi_fr; 3] }

#[link(name = "zkmi")]
extern "C" {
    // context
    pub fn zkmi_version() -> *const c_char;
    pub fn zkmi_hip_versions(out_build: *mut i32, out_runtime: *mut i32) -> i32;
    pub fn zkmi_device_count(out_count: *mut i32) -> i32;
    pub fn zkmi_ctx_create(device: i32, out_ctx: *mut *mut zkmi_ctx) -> i32;
    pub fn zkmi_ctx_destroy(ctx: *mut zkmi_ctx) -> i32;
    pub fn zkmi_last_error(ctx: *const zkmi_ctx) -> *const c_char;
    pub fn zkmi_ctx_sync(ctx: *mut zkmi_ctx) -> i32;

    // relations (rows a1-a5): update_note (one circuit per operation kind) and note creation
    pub fn zkmi_update_note_r1cs(log_n: u32, op_kind: i32, out: *mut *mut zkmi_r1cs) -> i32;
    pub fn zkmi_update_note_r1cs_h(log_n: u32, op_kind: i32, tree_height: u32, out: *mut *mut zkmi_r1cs) -> i32;
    pub fn zkmi_update_note_witness(log_n: u32, op_kind: i32, input: *const zkmi_note_update, out_z: *mut u8, out_publics: *mut u8) -> i32;
    pub fn zkmi_update_note_witness_batch_dev(ctx: *mut zkmi_ctx, log_n: u32, op_kind: i32, input: *const zkmi_note_update, n: u32,
                                              d_z_out: *const *mut core::ffi::c_void, out_status: *mut i32) -> i32;
    pub fn zkmi_create_note_r1cs(log_n: u32, out: *mut *mut zkmi_r1cs) -> i32;
    pub fn zkmi_create_note_witness(log_n: u32, input: *const zkmi_note_create, out_z: *mut u8, out_publics: *mut u8) -> i32;
    pub fn zkmi_r1cs_shape(r: *const zkmi_r1cs, n_vars: *mut u32, n_pub: *mut u32, n_constraints: *mut u32, log_n: *mut u32) -> i32;
    pub fn zkmi_r1cs_free(r: *mut zkmi_r1cs) -> i32;
    pub fn zkmi_fr_reduce(input: *const u8, out: *mut u8) -> i32;

    // keys
    pub fn zkmi_ctx_set_group_size(ctx: *mut zkmi_ctx, group: u32) -> i32;
    pub fn zkmi_groth16_setup(ctx: *mut zkmi_ctx, r1cs: *const zkmi_r1cs, toxic: *const u8, out_pk: *mut *mut zkmi_pk, vk_out: *mut u8, vk_cap: u64) -> i32;
    pub fn zkmi_ark_pk_load(ctx: *mut zkmi_ctx, r1cs: *const zkmi_r1cs, buf: *const u8, len: u64, compressed: i32, check_curve: i32,
                            out_pk: *mut *mut zkmi_pk, out_vk: *mut u8, vk_cap: u64) -> i32;
    pub fn zkmi_ark_vk_read(buf: *const u8, len: u64, compressed: i32, out_vk: *mut u8, vk_cap: u64, out_n_pub: *mut u32, out_consumed: *mut u64) -> i32;
    pub fn zkmi_pk_shape(pk: *const zkmi_pk, n_vars: *mut u32, n_pub: *mut u32, log_n: *mut u32) -> i32;
    pub fn zkmi_host_info(out: *mut u32) -> i32;
    pub fn zkmi_set_host_threads(n: u32) -> i32;
    pub fn zkmi_pk_schedule_state(pk: *const zkmi_pk, out: *mut u64) -> i32;
    pub fn zkmi_pk_free(pk: *mut zkmi_pk) -> i32;

    // witness -> proof (rows a6-a10), proof check (row a11)
    pub fn zkmi_groth16_prove(ctx: *mut zkmi_ctx, pk: *const zkmi_pk, z: *const u8, r: *const u8, s: *const u8, out_proof: *mut u8) -> i32;
    pub fn zkmi_groth16_prove_batch(ctx: *mut zkmi_ctx, pk: *const zkmi_pk, n_proofs: u32, z: *const *const u8, r: *const u8, s: *const u8, out_proofs: *mut u8) -> i32;
    pub fn zkmi_groth16_prove_batch_dev(ctx: *mut zkmi_ctx, pk: *const zkmi_pk, n_proofs: u32, d_z: *const *const core::ffi::c_void,
                                        r: *const u8, s: *const u8, out_proofs: *mut u8) -> i32;
    // one batch over several GPUs from one process: proof i -> ctxs[i % n_dev] (one host thread per device inside)
    pub fn zkmi_groth16_prove_batch_multi(ctxs: *const *mut zkmi_ctx, pks: *const *const zkmi_pk, n_dev: u32, n_proofs: u32,
                                          z: *const *const core::ffi::c_void, z_on_device: i32, r: *const u8, s: *const u8,
                                          out_proofs: *mut u8) -> i32;
    pub fn zkmi_groth16_verify(vk: *const u8, n_pub: u32, publics: *const u8, proof: *const u8) -> i32;

    // the mock's own surface, bit for bit (row a12) — lets call sites migrate one at a time
    pub fn zkmi_account_new(tokens: *const zkmi_scalar, out: *mut zkmi_account) -> i32;
    pub fn zkmi_account_update(a: *const zkmi_account, op_pub: *const zkmi_op_pub, op_priv: *const zkmi_op_priv, out: *mut zkmi_account) -> i32;
    pub fn zkmi_operation_combine(op_pub: *const zkmi_op_pub, op_priv: *const zkmi_op_priv) -> i32;
    pub fn zkmi_zkproof_new(id: *const zkmi_scalar, trapdoor: *const zkmi_scalar, nullifier: *const zkmi_scalar, op_priv: *const zkmi_op_priv,
                            acc: *const zkmi_account, out: *mut zkmi_zkproof) -> i32;
    pub fn zkmi_zkproof_update_account(this: *const zkmi_zkproof, op_pub: *const zkmi_op_pub, op_priv: *const zkmi_op_priv,
                                       trapdoor: *const zkmi_scalar, nullifier: *const zkmi_scalar, merkle_proof: *const zkmi_scalar,
                                       leaf_id: u32, out_h_note_new: *mut zkmi_scalar, out_new: *mut zkmi_zkproof) -> i32;
    pub fn zkmi_zkproof_verify_creation(this: *const zkmi_zkproof, h_note_new: *const zkmi_scalar, tokens: *const zkmi_scalar) -> i32;
    pub fn zkmi_zkproof_verify_update(this: *const zkmi_zkproof, op_pub: *const zkmi_op_pub, h_note_new: *const zkmi_scalar,
                                      merkle_root: *const zkmi_scalar, nullifier_old: *const zkmi_scalar) -> i32;

    // the same surface with real proofs
    pub fn zkmi_shielder_prove_creation(ctx: *mut zkmi_ctx, pk_create: *const zkmi_pk, knowledge: *const zkmi_zkproof, tokens: *const zkmi_scalar,
                                        r: *const u8, s: *const u8, out_h_note_new: *mut zkmi_scalar, out_proof: *mut u8) -> i32;
    pub fn zkmi_shielder_verify_creation(vk_create: *const u8, h_note_new: *const zkmi_scalar, tokens: *const zkmi_scalar, proof: *const u8) -> i32;
    pub fn zkmi_shielder_prove_update(ctx: *mut zkmi_ctx, pk_deposit: *const zkmi_pk, pk_withdraw: *const zkmi_pk, this: *const zkmi_zkproof,
                                      op_pub: *const zkmi_op_pub, op_priv: *const zkmi_op_priv, trapdoor: *const zkmi_scalar,
                                      nullifier: *const zkmi_scalar, merkle_proof: *const zkmi_scalar, tree_height: u32, leaf_id: u32,
                                      r: *const u8, s: *const u8, out_h_note_new: *mut zkmi_scalar, out_merkle_root: *mut zkmi_scalar,
                                      out_new: *mut zkmi_zkproof, out_proof: *mut u8) -> i32;
    pub fn zkmi_shielder_verify_update(vk_deposit: *const u8, vk_withdraw: *const u8, op_pub: *const zkmi_op_pub, h_note_new: *const zkmi_scalar,
                                       merkle_root: *const zkmi_scalar, nullifier_old: *const zkmi_scalar, proof: *const u8) -> i32;

    // Poseidon note tree on the device (the hash the relations use; merkle_proof.rs:56)
    pub fn zkmi_poseidon_hash_batch(ctx: *mut zkmi_ctx, field: i32, input: *const u8, n_hashes: u64, arity: u32, out: *mut u8) -> i32;
    pub fn zkmi_poseidon_merkle_tree_dev(ctx: *mut zkmi_ctx, field: i32, d_nodes: *mut core::ffi::c_void, log_leaves: u32) -> i32;
    pub fn zkmi_poseidon_merkle_paths_dev(ctx: *mut zkmi_ctx, d_nodes: *const core::ffi::c_void, log_leaves: u32, leaf_idx: *const u32, n: u32,
                                          out_shape: *mut u8, out_paths: *mut u8) -> i32;

    // the kernels under the prover, for a caller that keeps its own proof system (rows a6, a8, a9): what
    // ark_poly's Radix2EvaluationDomain::{fft, ifft, coset_*} and ark_ec's VariableBaseMSM::msm_bigint become
    pub fn zkmi_ntt_fr(ctx: *mut zkmi_ctx, data: *mut u8, log_n: u32, inverse: i32, coset: i32) -> i32;
    pub fn zkmi_ntt_fr_dev(ctx: *mut zkmi_ctx, d_data: *mut core::ffi::c_void, log_n: u32, inverse: i32, coset: i32) -> i32;
    pub fn zkmi_bases_g1_load(ctx: *mut zkmi_ctx, affine: *const u8, n: u64, check: i32, out: *mut *mut zkmi_bases_g1) -> i32;
    pub fn zkmi_bases_g2_load(ctx: *mut zkmi_ctx, affine: *const u8, n: u64, check: i32, out: *mut *mut zkmi_bases_g2) -> i32;
    pub fn zkmi_bases_g1_prepare(ctx: *mut zkmi_ctx, b: *mut zkmi_bases_g1) -> i32;
    pub fn zkmi_bases_g2_prepare(ctx: *mut zkmi_ctx, b: *mut zkmi_bases_g2) -> i32;
    pub fn zkmi_bases_g1_free(b: *mut zkmi_bases_g1) -> i32;
    pub fn zkmi_bases_g2_free(b: *mut zkmi_bases_g2) -> i32;
    pub fn zkmi_msm_g1(ctx: *mut zkmi_ctx, scalars: *const u8, n: u64, bases: *const zkmi_bases_g1, out_affine: *mut u8) -> i32;
    pub fn zkmi_msm_g2(ctx: *mut zkmi_ctx, scalars: *const u8, n: u64, bases: *const zkmi_bases_g2, out_affine: *mut u8) -> i32;
    pub fn zkmi_msm_g1_dev(ctx: *mut zkmi_ctx, d_scalars: *const core::ffi::c_void, n: u64, bases: *const zkmi_bases_g1, out_affine: *mut u8) -> i32;
    pub fn zkmi_msm_g2_dev(ctx: *mut zkmi_ctx, d_scalars: *const core::ffi::c_void, n: u64, bases: *const zkmi_bases_g2, out_affine: *mut u8) -> i32;
    // one MSM split by points over ranks (BASELINE config 3): per-window partials, all-gathered by the caller, combined anywhere
    pub fn zkmi_msm_g1_windows_dev(ctx: *mut zkmi_ctx, d_scalars: *const core::ffi::c_void, n: u64, bases: *const zkmi_bases_g1, plan_n: u64,
                                   out_windows_affine: *mut u8, out_nwin: *mut u32, out_window_bits: *mut u32) -> i32;
    pub fn zkmi_msm_g1_combine(windows_affine: *const u8, n_ranks: u32, nwin: u32, window_bits: u32, out_affine: *mut u8) -> i32;
    // the same exchange over RCCL, one process per GPU (rank 0 draws the id, the host distributes its 128 bytes)
    pub fn zkmi_msm_g1_window_range_dev(ctx: *mut zkmi_ctx, d_scalars: *const core::ffi::c_void, n: u64, bases: *const zkmi_bases_g1, plan_n: u64,
                                        w_first: u32, w_count: u32, out_windows_affine: *mut u8, out_nwin_total: *mut u32, out_window_bits: *mut u32) -> i32;
    pub fn zkmi_msm_g1_window_split_allgather(ctx: *mut zkmi_ctx, comm: *mut zkmi_comm, d_scalars: *const core::ffi::c_void, n: u64,
                                              bases: *const zkmi_bases_g1, out_affine: *mut u8) -> i32;
    pub fn zkmi_msm_g1_split2d_allgather(ctx: *mut zkmi_ctx, comm: *mut zkmi_comm, d_scalars: *const core::ffi::c_void, n: u64,
                                         bases: *const zkmi_bases_g1, plan_n: u64, window_groups: u32, out_affine: *mut u8) -> i32;
    pub fn zkmi_comm_unique_id(out_id: *mut u8) -> i32;
    pub fn zkmi_comm_init(ctx: *mut zkmi_ctx, n_ranks: u32, rank: u32, id: *const u8, out: *mut *mut zkmi_comm) -> i32;
    pub fn zkmi_comm_from_nccl(ctx: *mut zkmi_ctx, nccl_comm: *mut core::ffi::c_void, n_ranks: u32, rank: u32, out: *mut *mut zkmi_comm) -> i32;
    pub fn zkmi_comm_destroy(comm: *mut zkmi_comm) -> i32;
    pub fn zkmi_msm_g1_allgather_combine(ctx: *mut zkmi_ctx, comm: *mut zkmi_comm, d_scalars: *const core::ffi::c_void, n: u64,
                                         bases: *const zkmi_bases_g1, plan_n: u64, out_affine: *mut u8) -> i32;
    // what every rank computes behind the all-gather, on caller-supplied slots (host arithmetic), and the slot geometry
    pub fn zkmi_msm_g1_combine_partials(partials: *const u8, n_ranks: u32, plan_n: u64, window_split: i32, out_affine: *mut u8) -> i32;
    pub fn zkmi_msm_exchange_layout(plan_n: u64, n_ranks: u32, out: *mut u32) -> i32;
}
