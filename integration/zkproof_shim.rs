//! `mocked_zk/src/relations.rs` with real proofs: a DROP-IN replacement of `ZkProof`.  The five public methods keep
//! the reference's names, argument lists and return types exactly (reference: shielder/mocked_zk/src/relations.rs:36-155),
//! so the call sites compile unchanged:
//!     ZkProof::new(id, trapdoor, nullifier, op_priv, acc)                      drink_tests/utils/shielder.rs:60
//!     proof.update_account(operation, trapdoor, nullifier, merkle_proof, leaf_id)   drink_tests/utils/shielder.rs:105-114
//!     proof.verify_creation(h_note_new, tokens)                                 contract/lib.rs:56
//!     proof.verify_update(op_pub, h_note_new, merkle_root, nullifier_old)        contract/lib.rs:74
//! Shown, not compiled (no Rust toolchain in the build image); tests/test_cpu_host.py checks the method names and
//! arities against the reference file when it is present, and every `zkmi_*` call against include/zkmi.h.
//!
//! What is different underneath:
//!  * the value carries the 192-byte Groth16 proof (`proof`) next to the caller's knowledge.  The knowledge fields are
//!    what `update_account` needs on the wallet side; what travels to the contract is SCALE-encoded like before, and a
//!    production deployment would strip everything but `proof` from the encoding (the verify methods read nothing else);
//!  * hashes are the relation's Poseidon hashes (relations/src/relations/update_note.rs:100,131), not the mock's SHA-256;
//!  * the proving side needs a process-global `Prover` (GPU context + three proving keys), the verifying side the three
//!    verifying keys: `install_prover` / `install_verifying_keys` are called once at start-up.  Inside an ink! contract
//!    the verify calls would go through a chain extension to the same two C functions (host CPU only, no GPU).
use crate::{account::Account, errors::ZkpError, ffi::*, ops::{OpPriv, OpPub, Operation}, Scalar, MERKLE_TREE_DEPTH, TOKENS_NUMBER};
use std::sync::OnceLock;

pub struct Prover { ctx: *mut zkmi_ctx, pk_create: *mut zkmi_pk, pk_deposit: *mut zkmi_pk, pk_withdraw: *mut zkmi_pk }
unsafe impl Send for Prover {}
unsafe impl Sync for Prover {}  // guarded by PROVER_LOCK: a zkmi_ctx serves one host thread at a time
pub struct VerifyingKeys { pub create: Vec<u8>, pub deposit: Vec<u8>, pub withdraw: Vec<u8> }

static PROVER: OnceLock<Prover> = OnceLock::new();
static PROVER_LOCK: std::sync::Mutex<()> = std::sync::Mutex::new(());
static VERIFYING_KEYS: OnceLock<VerifyingKeys> = OnceLock::new();

pub fn install_prover(p: Prover) -> Result<(), Prover> { PROVER.set(p) }
pub fn install_verifying_keys(k: VerifyingKeys) -> Result<(), VerifyingKeys> { VERIFYING_KEYS.set(k) }

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
/// the scalar-field modulus r of BLS12-381, little-endian
const FR_MODULUS_LE: [u8; 32] = [
    0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
    0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73,
];
/// one element UNIFORM on [0, r): rejection sampling of 255-bit strings (acceptance r / 2^255, about 0.905), as Fr::rand does --
/// Groth16's zero-knowledge argument needs the blinding scalars uniform on the whole field, and clearing the top two
/// bits would only ever reach the lower 55 % of it
fn fresh_fr() -> [u8; 32] {
    loop {
        let mut v = [0u8; 32];
        getrandom::getrandom(&mut v).expect("os randomness");
        v[31] &= 0x7f;
        // v < r, compared from the most significant byte down
        let mut below = false;
        for i in (0..32).rev() {
            if v[i] != FR_MODULUS_LE[i] {
                below = v[i] < FR_MODULUS_LE[i];
                break;
            }
        }
        if below {
            return v;
        }
    }
}
/// prover randomness r || s: two independent uniform field elements
fn fresh_rs() -> [u8; 64] {
    let mut rs = [0u8; 64];
    rs[..32].copy_from_slice(&fresh_fr());
    rs[32..].copy_from_slice(&fresh_fr());
    rs
}

/// serde for the 192 proof bytes: the reference derives `serde::Serialize, serde::Deserialize` on `ZkProof`
/// (relations.rs:15) and serde implements neither for `[u8; 192]` (its array impls stop at 32), so the derive needs this
/// `with`-module (the `serde-big-array` crate's `BigArray` does the same; written out here so the shim adds no dependency).
/// Wire form: a 192-tuple of u8, i.e. exactly what `[u8; N]` serialises to for N <= 32.
mod proof_bytes_serde {
    use serde::{de::{Error, SeqAccess, Visitor}, ser::SerializeTuple, Deserializer, Serializer};
    pub fn serialize<S: Serializer>(v: &[u8; 192], s: S) -> Result<S::Ok, S::Error> {
        let mut t = s.serialize_tuple(192)?;
        for b in v { t.serialize_element(b)?; }
        t.end()
    }
    pub fn deserialize<'de, D: Deserializer<'de>>(d: D) -> Result<[u8; 192], D::Error> {
        struct V;
        impl<'de> Visitor<'de> for V {
            type Value = [u8; 192];
            fn expecting(&self, f: &mut core::fmt::Formatter) -> core::fmt::Result { f.write_str("192 proof bytes") }
            fn visit_seq<A: SeqAccess<'de>>(self, mut a: A) -> Result<[u8; 192], A::Error> {
                let mut out = [0u8; 192];
                for (i, o) in out.iter_mut().enumerate() { *o = a.next_element()?.ok_or_else(|| A::Error::invalid_length(i, &self))?; }
                Ok(out)
            }
        }
        d.deserialize_tuple(192, V)
    }
}

#[ink::scale_derive(Encode, Decode, TypeInfo)]
#[derive(Debug, Clone, Copy, serde::Serialize, serde::Deserialize)]  // the reference's derives (relations.rs:14-15)
pub struct ZkProof {
    id: Scalar,
    trapdoor_new: Scalar,
    trapdoor_old: Scalar,
    nullifier_new: Scalar,
    acc_old: Account,
    acc_new: Account,
    op_priv: OpPriv,
    merkle_proof: [Scalar; MERKLE_TREE_DEPTH],
    merkle_proof_leaf_id: u32,
    /// Groth16 proof of the relation this value was produced by (creation or update), compressed A || B || C
    #[serde(with = "proof_bytes_serde")]
    proof: [u8; 192],
    /// the Merkle root the last update was proved against (what the caller passes to the contract's update_note)
    merkle_root: Scalar,
}

impl ZkProof {
    fn knowledge(&self) -> zkmi_zkproof {
        zkmi_zkproof {
            id: sc(&self.id), trapdoor_new: sc(&self.trapdoor_new), trapdoor_old: sc(&self.trapdoor_old),
            nullifier_new: sc(&self.nullifier_new), acc_old: self.acc_old.to_ffi(), acc_new: self.acc_new.to_ffi(),
            op_priv: zkmi_op_priv { user: sc(&self.op_priv.user) },
            merkle_proof: self.merkle_proof.map(|s| sc(&s)), merkle_proof_leaf_id: self.merkle_proof_leaf_id,
        }
    }
    fn from_knowledge(k: &zkmi_zkproof, proof: [u8; 192], merkle_root: Scalar) -> Self {
        Self {
            id: Scalar::from_bytes(k.id.bytes), trapdoor_new: Scalar::from_bytes(k.trapdoor_new.bytes),
            trapdoor_old: Scalar::from_bytes(k.trapdoor_old.bytes), nullifier_new: Scalar::from_bytes(k.nullifier_new.bytes),
            acc_old: Account::from_ffi(&k.acc_old), acc_new: Account::from_ffi(&k.acc_new),
            op_priv: OpPriv { user: Scalar::from_bytes(k.op_priv.user.bytes) },
            merkle_proof: k.merkle_proof.map(|s| Scalar::from_bytes(s.bytes)), merkle_proof_leaf_id: k.merkle_proof_leaf_id,
            proof, merkle_root,
        }
    }

    /// relations.rs:37-55, same arguments.  Also proves the note-creation relation (update_note.rs:91-103 +
    /// update_account.rs:52-65) for the freshly created account, so that `verify_creation` has something to check.
    pub fn new(id: Scalar, trapdoor: Scalar, nullifier: Scalar, op_priv: OpPriv, acc: Account) -> Self {
        let mut this = Self {
            id, trapdoor_new: trapdoor, nullifier_new: nullifier, acc_new: acc, trapdoor_old: 0_u128.into(), acc_old: acc,
            op_priv, merkle_proof: [0_u128.into(); MERKLE_TREE_DEPTH], merkle_proof_leaf_id: 0,
            proof: [0u8; 192], merkle_root: 0_u128.into(),
        };
        let p = PROVER.get().expect("install_prover() first");
        let _g = PROVER_LOCK.lock().unwrap();
        let tokens: [zkmi_scalar; TOKENS_NUMBER] = core::array::from_fn(|t| sc(&acc.balances[t].0));
        let rs = fresh_rs();
        let mut h = zkmi_scalar { bytes: [0; 32] };
        let rc = unsafe {
            zkmi_shielder_prove_creation(p.ctx, p.pk_create, &this.knowledge(), tokens.as_ptr(), rs.as_ptr(), rs[32..].as_ptr(),
                                         &mut h, this.proof.as_mut_ptr())
        };
        assert_eq!(rc, ZKMI_OK, "creation proof");  // `new` is infallible in the reference
        this
    }

    /// relations.rs:79-98, same arguments and return type: (h_note_new, the knowledge after the update).
    pub fn update_account(&self, operation: Operation, trapdoor: Scalar, nullifier: Scalar,
                          merkle_proof: [Scalar; MERKLE_TREE_DEPTH], merkle_proof_leaf_id: u32) -> Result<(Scalar, Self), ZkpError> {
        let p = PROVER.get().ok_or(ZkpError::VerificationError)?;
        let _g = PROVER_LOCK.lock().unwrap();
        let this = self.knowledge();
        let (mut h, mut root) = (zkmi_scalar { bytes: [0; 32] }, zkmi_scalar { bytes: [0; 32] });
        let mut next = this;
        let mut proof = [0u8; 192];
        let path: [zkmi_scalar; MERKLE_TREE_DEPTH] = merkle_proof.map(|s| sc(&s));
        let rs = fresh_rs();
        let rc = unsafe {
            zkmi_shielder_prove_update(p.ctx, p.pk_deposit, p.pk_withdraw, &this, &op(&operation.op_pub),
                                       &zkmi_op_priv { user: sc(&operation.op_priv.user) }, &sc(&trapdoor), &sc(&nullifier),
                                       path.as_ptr(), MERKLE_TREE_DEPTH as u32, merkle_proof_leaf_id, rs.as_ptr(), rs[32..].as_ptr(),
                                       &mut h, &mut root, &mut next, proof.as_mut_ptr())
        };
        if rc != ZKMI_OK { return Err(err(rc)); }
        Ok((Scalar::from_bytes(h.bytes), Self::from_knowledge(&next, proof, Scalar::from_bytes(root.bytes))))
    }

    /// relations.rs:100-108 (sic: the reference spells it `acccount`), same arguments: the mock's account arithmetic
    pub fn verify_acccount_update(&self, op: Operation, h_acc_old: Scalar) -> Result<Account, ZkpError> {
        let mut out = self.acc_old.to_ffi();
        let rc = unsafe { zkmi_account_update(&self.acc_old.to_ffi(), &self::op(&op.op_pub), &zkmi_op_priv { user: sc(&op.op_priv.user) }, &mut out) };
        if rc != ZKMI_OK { return Err(err(rc)); }
        if self.acc_old.hash() != h_acc_old { return Err(ZkpError::VerificationError); }
        Ok(Account::from_ffi(&out))
    }

    /// relations.rs:127-136, same arguments: the contract's add_note (contract/lib.rs:50-58)
    pub fn verify_creation(&self, h_note_new: Scalar, tokens_list: [Scalar; TOKENS_NUMBER]) -> Result<(), ZkpError> {
        let vk = VERIFYING_KEYS.get().ok_or(ZkpError::VerificationError)?;
        let t: [zkmi_scalar; TOKENS_NUMBER] = tokens_list.map(|s| sc(&s));
        match unsafe { zkmi_shielder_verify_creation(vk.create.as_ptr(), &sc(&h_note_new), t.as_ptr(), self.proof.as_ptr()) } {
            ZKMI_OK => Ok(()),
            rc => Err(err(rc)),
        }
    }

    /// relations.rs:138-155, same arguments: the contract's update_note (contract/lib.rs:63-78).
    ///
    /// PRECONDITION on `merkle_root` (where the drop-in is NOT a no-op for the caller): the relation proves membership of
    /// the old note under a POSEIDON tree -- `CircuitMerkleProof::verify`, merkle_proof.rs:38-61, the same hash as the note
    /// hashes -- while the mock contract keeps a SHA-256 tree (`contract/merkle.rs:48-106`) and the callers pass ITS root
    /// (`drink_tests/utils/shielder.rs:85-91, 116-127`; the mock's `verify_update` recomputes that root with SHA-256,
    /// relations.rs:147-153).  With real proofs the root handed in here must be the Poseidon root the proof was made
    /// against: `update_account` returns it next to the proof (`ZkProof::merkle_root()`), `zkmi_poseidon_merkle_root_dev` /
    /// the contract-side tree must be switched to Poseidon-5.  A SHA-256 root fails with
    /// `ZkpError::VerificationError` -- it is a different public input, never a silent accept.
    pub fn verify_update(&self, op_pub: OpPub, h_note_new: Scalar, merkle_root: Scalar, nullifier_old: Scalar) -> Result<(), ZkpError> {
        let vk = VERIFYING_KEYS.get().ok_or(ZkpError::VerificationError)?;
        match unsafe {
            zkmi_shielder_verify_update(vk.deposit.as_ptr(), vk.withdraw.as_ptr(), &op(&op_pub), &sc(&h_note_new), &sc(&merkle_root),
                                        &sc(&nullifier_old), self.proof.as_ptr())
        } {
            ZKMI_OK => Ok(()),
            rc => Err(err(rc)),
        }
    }

    /// not in the reference: the Merkle root the last `update_account` was proved against and the raw proof bytes
    pub fn merkle_root(&self) -> Scalar { self.merkle_root }
    pub fn proof_bytes(&self) -> &[u8; 192] { &self.proof }
}

/// A batch of updates over all GPUs of the node (BASELINE config 2): what a service does instead of looping over
/// `update_account`; witnesses come from zkmi_update_note_witness, proofs return in the caller's order.
pub fn prove_batch_multi(ctxs: &[*mut zkmi_ctx], pks: &[*const zkmi_pk], witnesses: &[&[u8]], rs: &[[u8; 64]]) -> Result<Vec<[u8; 192]>, ZkpError> {
    let n = witnesses.len();
    let z: Vec<*const core::ffi::c_void> = witnesses.iter().map(|w| w.as_ptr() as *const _).collect();
    let r: Vec<u8> = rs.iter().flat_map(|x| x[..32].to_vec()).collect();
    let s: Vec<u8> = rs.iter().flat_map(|x| x[32..].to_vec()).collect();
    let mut out = vec![[0u8; 192]; n];
    let rc = unsafe {
        zkmi_groth16_prove_batch_multi(ctxs.as_ptr(), pks.as_ptr(), ctxs.len() as u32, n as u32, z.as_ptr(), 0, r.as_ptr(), s.as_ptr(),
                                       out.as_mut_ptr() as *mut u8)
    };
    if rc != ZKMI_OK { return Err(err(rc)); }
    Ok(out)
}
