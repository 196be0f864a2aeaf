//! `mocked_zk/src/relations.rs` with real proofs: the four methods the callers use keep their names, arity and
//! `ZkpError` behaviour (reference: shielder/mocked_zk/src/relations.rs:36-155; callers
//! contract/drink_tests/utils/shielder.rs:60,105-114 and contract/lib.rs:56,74).  Shown, not compiled.
//!
//! What changes for the callers: `update_account` additionally returns the 192-byte proof (it replaces the witness
//! struct as the object submitted to the contract), and the verify side needs the verifying keys, which the contract
//! would hold in storage next to `supported_tokens` (contract/lib.rs:29-35).
use crate::{errors::ZkpError, ffi::*, ops::{OpPriv, OpPub, Operation}, Scalar, MERKLE_TREE_DEPTH, TOKENS_NUMBER};

pub struct Prover { ctx: *mut zkmi_ctx, pk_create: *mut zkmi_pk, pk_deposit: *mut zkmi_pk, pk_withdraw: *mut zkmi_pk }
pub struct VerifyingKeys { pub create: Vec<u8>, pub deposit: Vec<u8>, pub withdraw: Vec<u8> }
pub type ProofBytes = [u8; 192];

fn err(rc: i32) -> ZkpError {
    match rc {
        ZKMI_ERR_ACCOUNT_UPDATE => ZkpError::AccountUpdateError,
        ZKMI_ERR_OPERATION_COMBINE => ZkpError::OperationCombineError,
        _ => ZkpError::VerificationError,
    }
}
fn sc(s: &Scalar) -> zkmi_scalar { zkmi_scalar { bytes: s.bytes } }
fn op(o: &OpPub) -> zkmi_op_pub {
    let (kind, amount, token, user) = match *o {
        OpPub::Deposit { amount, token, user } => (0u32, amount, token, user),
        OpPub::Withdraw { amount, token, user } => (1u32, amount, token, user),
    };
    zkmi_op_pub { kind, amount: amount.to_le_bytes(), token: sc(&token), user: sc(&user) }
}

impl super::ZkProof {
    // `new` is unchanged (relations.rs:37-55): it only stores the caller's knowledge.

    /// relations.rs:79-98 + the proof.  `rs` = the prover's randomness (r || s).
    pub fn update_account_proved(&self, p: &Prover, operation: Operation, trapdoor: Scalar, nullifier: Scalar,
                                 merkle_proof: [Scalar; MERKLE_TREE_DEPTH], merkle_proof_leaf_id: u32, rs: &[u8; 64])
                                 -> Result<(Scalar, Scalar, Self, ProofBytes), ZkpError> {
        let this: zkmi_zkproof = self.to_ffi();
        let (mut h, mut root) = (zkmi_scalar { bytes: [0; 32] }, zkmi_scalar { bytes: [0; 32] });
        let mut next = this;
        let mut proof = [0u8; 192];
        let path: Vec<zkmi_scalar> = merkle_proof.iter().map(sc).collect();
        let rc = unsafe {
            zkmi_shielder_prove_update(p.ctx, p.pk_deposit, p.pk_withdraw, &this, &op(&operation.op_pub),
                                       &zkmi_op_priv { user: sc(&operation.op_priv.user) }, &sc(&trapdoor), &sc(&nullifier),
                                       path.as_ptr(), MERKLE_TREE_DEPTH as u32, merkle_proof_leaf_id, rs.as_ptr(), rs[32..].as_ptr(),
                                       &mut h, &mut root, &mut next, proof.as_mut_ptr())
        };
        if rc != ZKMI_OK { return Err(err(rc)); }
        Ok((Scalar::from_bytes(h.bytes), Scalar::from_bytes(root.bytes), Self::from_ffi(&next), proof))
    }
}

/// relations.rs:127-136: the contract's add_note (contract/lib.rs:50-58)
pub fn verify_creation(vk: &VerifyingKeys, proof: &ProofBytes, h_note_new: Scalar, tokens_list: [Scalar; TOKENS_NUMBER]) -> Result<(), ZkpError> {
    let t: Vec<zkmi_scalar> = tokens_list.iter().map(sc).collect();
    match unsafe { zkmi_shielder_verify_creation(vk.create.as_ptr(), &sc(&h_note_new), t.as_ptr(), proof.as_ptr()) } {
        ZKMI_OK => Ok(()),
        rc => Err(err(rc)),
    }
}

/// relations.rs:138-155: the contract's update_note (contract/lib.rs:63-78)
pub fn verify_update(vk: &VerifyingKeys, proof: &ProofBytes, op_pub: OpPub, h_note_new: Scalar, merkle_root: Scalar, nullifier_old: Scalar)
                     -> Result<(), ZkpError> {
    match unsafe {
        zkmi_shielder_verify_update(vk.deposit.as_ptr(), vk.withdraw.as_ptr(), &op(&op_pub), &sc(&h_note_new), &sc(&merkle_root),
                                    &sc(&nullifier_old), proof.as_ptr())
    } {
        ZKMI_OK => Ok(()),
        rc => Err(err(rc)),
    }
}
