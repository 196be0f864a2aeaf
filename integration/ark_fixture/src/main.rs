//! Pins the GPU prover against arkworks itself.  Reads `relation.bin` + `witness.bin` (written by
//! scripts/export_relation_for_arkworks.py: the R1CS matrices this repository proves, in CSR form, and a satisfying
//! assignment), runs ark-groth16's setup and prover with fixed (r, s), and writes
//!   pk_uncompressed.bin   ProvingKey::serialize_uncompressed      (loaded by zkmi_ark_pk_load)
//!   vk_uncompressed.bin   VerifyingKey::serialize_uncompressed
//!   rs.bin                r || s, 32-byte little-endian each
//!   proof.bin             Proof::serialize_compressed (192 bytes)  (compared with zkmi_groth16_prove's output)
//! Not built in this repository's image (no cargo); see integration/README.md.
use ark_bls12_381::{Bls12_381, Fr};
use ark_ff::PrimeField;
use ark_groth16::{prover::create_proof_with_reduction_and_matrices, r1cs_to_qap::LibsnarkReduction, Groth16};
use ark_relations::{lc, r1cs::{ConstraintMatrices, ConstraintSynthesizer, ConstraintSystemRef, LinearCombination, SynthesisError, Variable}};
use ark_serialize::CanonicalSerialize;
use ark_std::rand::{rngs::StdRng, SeedableRng};
use std::{env, fs, path::Path};

struct Csr { rowptr: Vec<u32>, col: Vec<u32>, val: Vec<Fr> }
struct Relation { n_vars: usize, n_pub: usize, nc: usize, m: [Csr; 3], z: Vec<Fr> }

fn u32_at(b: &[u8], off: &mut usize) -> u32 { let v = u32::from_le_bytes(b[*off..*off + 4].try_into().unwrap()); *off += 4; v }
fn fr_at(b: &[u8], off: &mut usize) -> Fr { let v = Fr::from_le_bytes_mod_order(&b[*off..*off + 32]); *off += 32; v }

fn read(dir: &Path) -> Relation {
    let b = fs::read(dir.join("relation.bin")).unwrap();
    let mut off = 0;
    let (n_vars, n_pub, nc) = (u32_at(&b, &mut off) as usize, u32_at(&b, &mut off) as usize, u32_at(&b, &mut off) as usize);
    let mut mats = Vec::new();
    for _ in 0..3 {
        let rowptr: Vec<u32> = (0..=nc).map(|_| u32_at(&b, &mut off)).collect();
        let nnz = *rowptr.last().unwrap() as usize;
        let col: Vec<u32> = (0..nnz).map(|_| u32_at(&b, &mut off)).collect();
        let val: Vec<Fr> = (0..nnz).map(|_| fr_at(&b, &mut off)).collect();
        mats.push(Csr { rowptr, col, val });
    }
    let w = fs::read(dir.join("witness.bin")).unwrap();
    let mut o = 0;
    let z = (0..n_vars).map(|_| fr_at(&w, &mut o)).collect();
    let c = mats.pop().unwrap(); let bb = mats.pop().unwrap(); let a = mats.pop().unwrap();
    Relation { n_vars, n_pub, nc, m: [a, bb, c], z }
}

/// the relation as an arkworks circuit: column 0 = Variable::One, columns [1, n_pub) instance, the rest witness
impl ConstraintSynthesizer<Fr> for &Relation {
    fn generate_constraints(self, cs: ConstraintSystemRef<Fr>) -> Result<(), SynthesisError> {
        let mut vars = vec![Variable::One];
        for j in 1..self.n_pub { vars.push(cs.new_input_variable(|| Ok(self.z[j]))?); }
        for j in self.n_pub..self.n_vars { vars.push(cs.new_witness_variable(|| Ok(self.z[j]))?); }
        let row = |m: &Csr, i: usize| -> LinearCombination<Fr> {
            let mut l = lc!();
            for k in m.rowptr[i] as usize..m.rowptr[i + 1] as usize { l = l + (m.val[k], vars[m.col[k] as usize]); }
            l
        };
        for i in 0..self.nc { cs.enforce_constraint(row(&self.m[0], i), row(&self.m[1], i), row(&self.m[2], i))?; }
        Ok(())
    }
}

fn main() {
    let dir = env::args().nth(1).expect("usage: ark_fixture <dir with relation.bin and witness.bin>");
    let dir = Path::new(&dir);
    let rel = read(dir);
    let mut rng = StdRng::seed_from_u64(0x5A4B);
    let pk = Groth16::<Bls12_381>::generate_random_parameters_with_reduction(&rel, &mut rng).unwrap();
    let (r, s) = (Fr::from(0x1234_5678_9abc_def1u64), Fr::from(0x0fed_cba9_8765_4321u64));
    // the matrices arkworks derives from the circuit (the prover's view), then its prover with explicit (r, s)
    let cs = ark_relations::r1cs::ConstraintSystem::<Fr>::new_ref();
    (&rel).generate_constraints(cs.clone()).unwrap();
    cs.finalize();
    let matrices: ConstraintMatrices<Fr> = cs.to_matrices().unwrap();
    let full: Vec<Fr> = rel.z.clone();
    let proof = create_proof_with_reduction_and_matrices::<Bls12_381, LibsnarkReduction>(
        &pk, r, s, &matrices, rel.n_pub, rel.nc, &full).unwrap();
    assert!(Groth16::<Bls12_381>::verify_proof(&ark_groth16::prepare_verifying_key(&pk.vk), &proof, &rel.z[1..rel.n_pub]).unwrap());
    let mut buf = Vec::new(); pk.serialize_uncompressed(&mut buf).unwrap(); fs::write(dir.join("pk_uncompressed.bin"), &buf).unwrap();
    buf.clear(); pk.vk.serialize_uncompressed(&mut buf).unwrap(); fs::write(dir.join("vk_uncompressed.bin"), &buf).unwrap();
    buf.clear(); r.serialize_compressed(&mut buf).unwrap(); s.serialize_compressed(&mut buf).unwrap(); fs::write(dir.join("rs.bin"), &buf).unwrap();
    buf.clear(); proof.serialize_compressed(&mut buf).unwrap(); assert_eq!(buf.len(), 192); fs::write(dir.join("proof.bin"), &buf).unwrap();
    println!("wrote pk_uncompressed.bin, vk_uncompressed.bin, rs.bin, proof.bin to {}", dir.display());
}
