"""GPU parity at BASELINE.json's full sizes, byte for byte against the C++ oracle (oracle/zkmi_oracle.cpp, the
multi-threaded restatement: seconds per case on the GPU box's host cores), plus the key-lifecycle churn.

The smaller-size sweeps, golden fixtures and property tests live in test_gpu_parity.py; this file closes the gap the
round-2 verdict named: NTT at 2^20 / 2^22 in all four modes, G1 MSM at 2^20 (uniform and witness-like), G2 MSM at 2^18
(oracle) and 2^22 (closed form), full proof bytes at 2^18 and 2^20 over the ORACLE's own trusted setup."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import bls12_381 as ec
from oracle.bls12_381 import R

pytestmark = pytest.mark.gpu


def frs(vals):
    return b"".join(ec.fr_to_bytes(v) for v in vals)


def _canonical_bytes(n, seed):
    """n uniformly random canonical scalars (< 2^254 < r) as wire bytes."""
    a = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F
    return a


def _witness_like(n, seed):
    """40 % zero, 20 % one, 10 % below 2^16, 30 % uniform."""
    rng = np.random.default_rng(seed)
    a = _canonical_bytes(n, seed + 1)
    kind = rng.random(n)
    a[kind < 0.4] = 0
    one = (kind >= 0.4) & (kind < 0.6)
    a[one] = 0
    a[one, 0] = 1
    small = (kind >= 0.6) & (kind < 0.7)
    a[small, 2:] = 0
    return a


@pytest.mark.parametrize("lg", [20, 22])
def test_ntt_full_size_all_modes_vs_cpp_oracle(ctx, lg):
    """Row a6 at BASELINE's sizes (2^20: configs 1-2, 2^22: config 4): forward, inverse, coset forward, coset inverse,
    every output byte against the C++ oracle (= ark-poly's Radix2EvaluationDomain as restated in oracle/ntt.py)."""
    from oracle import cpp as ocpp

    ocpp.build()
    x = _canonical_bytes(1 << lg, 100 + lg).tobytes()
    for inverse in (False, True):
        for coset in (False, True):
            got = ctx.ntt(x, lg, inverse=inverse, coset=coset)
            want = ocpp.ntt(x, lg, inverse=inverse, coset=coset)
            assert got == want, (lg, inverse, coset)


@pytest.mark.parametrize("kind", ["uniform", "witness_like"])
def test_msm_g1_2p20_vs_cpp_oracle(ctx, zk, kind):
    """Row a8 at config 1's size: 2^20 terms, uniform scalars and the witness-like mix (heavy buckets: 20 % of all
    points share one bucket per window), against the C++ oracle's windowed Pippenger; and the same MSM over prepared
    bases (the prover's shared-bucket schedule)."""
    from oracle import cpp as ocpp

    ocpp.build()
    n = 1 << 20
    sc = (_canonical_bytes(n, 7) if kind == "uniform" else _witness_like(n, 8)).tobytes()
    b = ctx.bases_g1_synthetic(n)
    want = ocpp.msm_g1(sc, b.read(0, n))
    assert ctx.msm_g1(sc, b) == want
    # the same terms as ONE slice under the plan of a 2^24-term MSM (20-bit windows, the two-level sort of the big plans): the
    # witness-like mix puts 210 000 records into one fine partition there -- the oversized-partition kernels (k_big_*)
    import torch

    d = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    w, nwin, cb = ctx.msm_g1_windows_dev(d.data_ptr(), n, b, 1 << 24)
    assert (nwin, cb) == (13, 20)
    assert zk.msm_g1_combine(w, 1, nwin, cb) == want
    b.prepare()
    assert ctx.msm_g1(sc, b) == want
    b.free()


def test_msm_g2_2p18_vs_cpp_oracle(ctx):
    """Row a9: G2 MSM of 2^18 terms (Fq2 lane-pair kernels) against the C++ oracle, plain and prepared."""
    from oracle import cpp as ocpp

    ocpp.build()
    n = 1 << 18
    sc = _canonical_bytes(n, 9).tobytes()
    b = ctx.bases_g2_synthetic(n)
    want = ocpp.msm_g2(sc, b.read(0, n))
    assert ctx.msm_g2(sc, b) == want
    b.prepare()
    assert ctx.msm_g2(sc, b) == want
    b.free()


def test_msm_g2_2p22_closed_form(ctx):
    """Config 4's G2 MSM size on its own (2^22 terms, 896 MiB of algorithmic bytes): with P_i = G2 + i Q the result
    is (sum s_i) G2 + (sum i s_i) Q, computed by the Python oracle."""
    import torch

    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(422)
    raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    raw[:, 31] &= 0x3F
    idx = torch.arange(n, dtype=torch.int64, device="cuda")
    s0 = raw.to(torch.int64).sum(dim=0).cpu().tolist()
    s1 = (raw.to(torch.int64) * idx[:, None]).sum(dim=0).cpu().tolist()
    tot = sum(v << (8 * k) for k, v in enumerate(s0)) % R
    wtot = sum(v << (8 * k) for k, v in enumerate(s1)) % R
    b = ctx.bases_g2_synthetic(n)
    torch.cuda.synchronize()
    got = ctx.msm_g2_dev(raw.data_ptr(), n, b)
    want = ec.pt_add(ec.Fq2, ec.g2_mul(tot), ec.g2_mul(wtot, ec.g2_mul(0xC0FFEE)))
    assert got == ec.g2_to_bytes(want)
    b.free()
    del raw
    torch.cuda.empty_cache()


def _oracle_vk_dict(vk, n_pub):
    return {
        "alpha_g1": ec.g1_from_bytes(vk[:96]),
        "beta_g2": ec.g2_from_bytes(vk[96:288]),
        "gamma_g2": ec.g2_from_bytes(vk[288:480]),
        "delta_g2": ec.g2_from_bytes(vk[480:672]),
        "gamma_abc_g1": [ec.g1_from_bytes(vk[672 + 96 * i: 768 + 96 * i]) for i in range(n_pub)],
    }


def test_msm_g2_2p22_vs_cpp_oracle(ctx):
    """Row a9 at BASELINE config 4's size: the G2 MSM of 2^22 terms (896 MiB of algorithmic bytes; above 2^21 terms the
    digit sort is the record sort, not the fine-partition one) against the C++ oracle's windowed Pippenger over Fq2, plain
    and over prepared bases -- beside the closed form below, which is independent Python arithmetic."""
    from oracle import cpp as ocpp

    ocpp.build()
    n = 1 << 22
    sc = _canonical_bytes(n, 22).tobytes()
    b = ctx.bases_g2_synthetic(n)
    want = ocpp.msm_g2(sc, b.read(0, n))
    assert ctx.msm_g2(sc, b) == want
    b.prepare()
    assert ctx.msm_g2(sc, b) == want
    b.free()


def _proof_bytes_against_the_oracle(ctx, zk, r1, wit, publics, seed):
    """vk + 192 proof bytes of the product (its setup, its GPU prover: single and batch entry points) against the C++
    oracle's own setup + prover from the same toxic waste, witness, r and s; the proof must verify under the oracle's key."""
    import torch
    from oracle import cpp as ocpp

    ocpp.build()
    rng = ec.SplitMix64(seed)
    toxic = frs([rng.fr() for _ in range(5)])
    pk, vk = ctx.groth16_setup(r1, toxic)
    mats = [r1.export(m) for m in range(3)]
    ovk, okey = ocpp.groth16_setup(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, toxic)
    assert vk == ovk
    r_, s_ = ec.fr_to_bytes(rng.fr()), ec.fr_to_bytes(rng.fr())
    want = ocpp.groth16_prove(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, okey, wit, r_, s_)
    d = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert ctx.groth16_prove_dev(pk, d.data_ptr(), r_, s_) == want
    assert ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * 3, [r_] * 3, [s_] * 3) == [want] * 3
    assert zk.groth16_verify(ovk, frs(publics), want) is True
    bad = list(publics)
    bad[0] = (bad[0] + 1) % R
    assert zk.groth16_verify(ovk, frs(bad), want) is False
    pk.free()
    return want


@pytest.mark.parametrize("which", ["deposit", "withdraw", "creation"])
def test_proof_bytes_vs_cpp_oracle_at_the_relations_own_sizes(ctx, zk, which):
    """BASELINE configs[0] is the DEPOSIT relation at 2^14 (mocked_zk/src/ops.rs:6-25: Deposit / Withdraw); the byte-parity
    tests of the big sizes all prove withdraw.  Here: deposit and withdraw at 2^14 and the creation relation (what
    ZkProof::verify_creation stands for, relations.rs:127-136) at 2^12, each against the oracle's own setup + prover."""
    from test_cpu_host import _note_update_case
    from oracle import poseidon as ps

    if which == "creation":
        lg = 12
        r1 = zk.create_note_r1cs(lg)
        rng = ec.SplitMix64(777)
        tok, note = (rng.fr(), rng.fr()), (rng.fr(), rng.fr(), rng.fr())
        wit, publics = zk.create_note_witness(lg, zk.note_create(tok, note))
        assert publics[0] == ps.note_hash(note[0], note[1], note[2], ps.hash_fix_len([tok[0], 0, tok[1], 0]))
    else:
        lg, op_kind = 14, 0 if which == "deposit" else 1
        r1 = zk.update_note_r1cs(lg, op_kind)
        inp, publics = _note_update_case(zk, 9100 + op_kind, op_kind, slot=1 - op_kind)  # (withdraw from the 1 000 balance)
        wit, pub, rc = zk.update_note_witness(lg, op_kind, inp)
        assert rc == 0 and pub == publics
    _proof_bytes_against_the_oracle(ctx, zk, r1, wit, publics, 0x5A4B0100 + lg + len(which))
    r1.free()


@pytest.mark.parametrize("op_kind", [0, 1])
def test_update_note_proof_from_an_oracle_side_witness(ctx, zk, op_kind):
    """Relation AND prover on inputs the product did not produce (2^13, the relation's real size): the loaded values come
    from oracle/relation_witness.py (UpdateNoteInput::new's order, update_note.rs:47-88, every hash by oracle/poseidon.py), the
    remaining variables from its generic solver over the exported matrices -- which raises if the constraint system does not
    compute the oracle's hashes; the assignment must satisfy the oracle's R1CS evaluator, equal the product generator's
    bytes (the solution is unique), and the GPU proof over it must equal the oracle prover's bytes."""
    from oracle import relation_witness as rw
    from test_cpu_host import _oracle_r1cs

    lg, height = 13, 10
    r1 = zk.update_note_r1cs(lg, op_kind)
    rng = ec.SplitMix64(0xB0B + op_kind)
    tok = [rng.fr(), rng.fr()]
    bal, amount, slot = [5000, 123], 321, 1 - op_kind  # (deposit into slot 1, withdraw from slot 0)
    new_note, old_note = (rng.fr(), rng.fr(), rng.fr()), (rng.fr(), rng.fr(), rng.fr())
    user = rng.fr()
    shape = [rng.next() & 1 for _ in range(height)]
    path = [rng.fr() for _ in range(height)]
    loaded = rw.update_note_loaded(op_kind, amount, tok[slot], user, new_note, old_note, shape, path, user, (tok[0], bal[0], tok[1], bal[1]))
    orc = _oracle_r1cs(r1)
    z_int = rw.solve(r1.n_vars, orc.A, orc.B, orc.C, loaded)
    assert orc.is_satisfied(z_int)
    wit = b"".join(v.to_bytes(32, "little") for v in z_int)
    # the product's generator on the same instance: the same bytes
    inp = zk.note_update(amount, tok[slot], user, new_note, old_note, shape, path, user, (tok[0], bal[0], tok[1], bal[1]))
    w_prod, pub, rc = zk.update_note_witness(lg, op_kind, inp)
    assert rc == 0 and w_prod == wit and pub == loaded[1:7]
    _proof_bytes_against_the_oracle(ctx, zk, r1, wit, loaded[1:7], 0x5A4B0200 + op_kind)
    r1.free()


@pytest.mark.parametrize("lg", [18, 20, 22])
def test_proof_bytes_vs_cpp_oracle_over_the_oracle_side_key(ctx, zk, lg):
    """Rows a7 + a10 at the headline size (2^20) and at BASELINE config 4's (2^22: "G2 MSM + pairing ... full Groth16 proof",
    on the reference's update_note relation with Poseidon hashing, not the chain stand-in): the ORACLE runs its own trusted
    setup (oracle_groth16_setup, pinned to the Python oracle on the N = 128 golden key) and its own prover; the product runs
    its setup and its GPU prover from the same toxic waste, witness, r and s.  Verifying keys and the 192 proof bytes must
    be identical, and the proof must pass the pairing check under the oracle's key -- in the product's verifier and, at
    2^22, in the Python oracle's own (oracle/groth16.verify: polynomial-basis Fq12, no code shared with the product)."""
    import torch
    from oracle import cpp as ocpp
    from test_cpu_host import _note_update_case

    ocpp.build()
    r1 = zk.update_note_r1cs(lg, 1)
    rng = ec.SplitMix64(0x5A4B0000 + lg)
    toxic = frs([rng.fr() for _ in range(5)])
    pk, vk = ctx.groth16_setup(r1, toxic)
    mats = [r1.export(m) for m in range(3)]
    ovk, okey = ocpp.groth16_setup(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, toxic)
    assert vk == ovk
    # spot-check the queries against the oracle's (the 2^16 test compares them in full)
    n, N = r1.n_vars, 1 << lg
    for which, name, cnt, w in ((0, "a_query", n, 96), (1, "b_g1_query", n, 96), (2, "b_g2_query", n, 192),
                                (3, "h_query", N - 1, 96), (4, "l_query", n - r1.n_pub, 96)):
        for first in (0, cnt // 2, cnt - 64):
            assert pk.export_query(which, first, 64) == okey[name][w * first: w * (first + 64)], (name, first)
    inp, publics = _note_update_case(zk, 9000 + lg, 1)
    wit, _, _ = zk.update_note_witness(lg, 1, inp)
    r_, s_ = ec.fr_to_bytes(rng.fr()), ec.fr_to_bytes(rng.fr())
    want = ocpp.groth16_prove(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, okey, wit, r_, s_)
    d = torch.frombuffer(bytearray(wit), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    assert ctx.groth16_prove_dev(pk, d.data_ptr(), r_, s_) == want
    assert ctx.groth16_prove_batch_dev(pk, [d.data_ptr()] * 3, [r_] * 3, [s_] * 3) == [want] * 3
    assert zk.groth16_verify(ovk, frs(publics), want) is True
    if lg == 22:
        from oracle import groth16 as g16

        assert g16.verify(_oracle_vk_dict(ovk, r1.n_pub), list(publics), g16.proof_from_bytes(want))
        bad = list(publics)
        bad[-1] = (bad[-1] + 1) % R
        assert not g16.verify(_oracle_vk_dict(ovk, r1.n_pub), bad, g16.proof_from_bytes(want))
    pk.free()
    r1.free()
    del d, okey, mats
    torch.cuda.empty_cache()


def test_grouped_key_survives_a_bigger_msm_on_the_same_context(ctx, zk):
    """A small key proves in groups of 64 and owns 2^20 buckets of the shared sort buffers; a 2^21-term MSM on the
    same context needs more ENTRIES than that key reserved but fewer buckets.  The buffers must only ever grow
    (round-2 advice: reserve() re-allocated from the new n alone and the next grouped batch failed)."""
    import torch

    import bench

    lg = 14
    r1, wits = bench.relation_and_witness(zk, "poseidon", lg, [141, 142])
    rng = ec.SplitMix64(1414)
    pk, vk = ctx.groth16_setup(r1, frs([rng.fr() for _ in range(5)]))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    rs = [ec.fr_to_bytes(rng.fr()) for _ in range(2)]
    ss = [ec.fr_to_bytes(rng.fr()) for _ in range(2)]
    torch.cuda.synchronize()
    idx = [i % 2 for i in range(130)]
    args = ([d[j].data_ptr() for j in idx], [rs[j] for j in idx], [ss[j] for j in idx])
    before = ctx.groth16_prove_batch_dev(pk, *args)
    n = 1 << 21
    g = torch.Generator(device="cuda").manual_seed(21)
    raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    raw[:, 31] &= 0x3F
    b = ctx.bases_g1_synthetic(n)
    torch.cuda.synchronize()
    first = ctx.msm_g1_dev(raw.data_ptr(), n, b)
    b.prepare()
    assert ctx.msm_g1_dev(raw.data_ptr(), n, b) == first
    after = ctx.groth16_prove_batch_dev(pk, *args)
    assert after == before
    assert zk.groth16_verify(vk, wits[0][32: 32 * r1.n_pub], after[0]) is True
    assert ctx.groth16_prove_dev(pk, d[1].data_ptr(), rs[1], ss[1]) == before[1]
    b.free()
    pk.free()
    r1.free()
    del raw
    torch.cuda.empty_cache()


def test_key_lifecycle_churn_short():
    """scripts/churn.py for a few hundred operations in a child process (a fatal signal must fail the test, not the
    session): shuffled key sizes with big -> small transitions, forced group sizes, batch / single / host proofs,
    error path, generic MSMs up to 2^22 terms, NTTs, a second context; every proof byte-identical to its first
    occurrence.  The long run (10^4 operations) is recorded in DESIGN.md section 8."""
    env = dict(os.environ, ZKMI_BACKTRACE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "churn.py"), "--ops", "400", "--seed", "3", "--max-log-n", "20",
                        "--watchdog", "1500"], capture_output=True, text=True, timeout=1700, cwd=ROOT, env=env)
    assert p.returncode == 0 and "CHURN OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def test_prove_batch_multi_over_two_contexts(ctx, zk):
    """zkmi_groth16_prove_batch_multi (BASELINE config 2 behind the C ABI): proof i runs on ctxs[i % n_dev] with that
    context's replica of the key, one host thread per device.  Devices 0 and 1 where the box has two GPUs, two
    contexts on device 0 otherwise.  Same bytes as the single-context batch prover, from host and device witnesses."""
    import torch

    import bench

    lg = 14
    r1, wits = bench.relation_and_witness(zk, "poseidon", lg, [71, 72, 73])
    rng = ec.SplitMix64(7171)
    toxic = frs([rng.fr() for _ in range(5)])
    dev1 = 1 if torch.cuda.device_count() > 1 else 0
    ctx1 = zk.context(dev1)
    pk0, vk0 = ctx.groth16_setup(r1, toxic)
    pk1, vk1 = ctx1.groth16_setup(r1, toxic)
    assert vk0 == vk1
    n = 7
    idx = [i % 3 for i in range(n)]
    rs = [ec.fr_to_bytes(rng.fr()) for _ in range(n)]
    ss = [ec.fr_to_bytes(rng.fr()) for _ in range(n)]
    d0 = [torch.frombuffer(bytearray(w), dtype=torch.uint8).to("cuda:0") for w in wits]
    d1 = [torch.frombuffer(bytearray(w), dtype=torch.uint8).to(f"cuda:{dev1}") for w in wits]
    pin = [torch.frombuffer(bytearray(w), dtype=torch.uint8).pin_memory() for w in wits]
    torch.cuda.synchronize()
    want = ctx.groth16_prove_batch_dev(pk0, [d0[j].data_ptr() for j in idx], rs, ss)
    for i in (0, n - 1):
        assert zk.groth16_verify(vk0, wits[idx[i]][32: 32 * r1.n_pub], want[i]) is True
    got_host = zk.groth16_prove_batch_multi([ctx, ctx1], [pk0, pk1], [pin[j].data_ptr() for j in idx], False, rs, ss)
    assert got_host == want
    ptrs = [(d0 if i % 2 == 0 else d1)[j].data_ptr() for i, j in enumerate(idx)]
    assert zk.groth16_prove_batch_multi([ctx, ctx1], [pk0, pk1], ptrs, True, rs, ss) == want
    # a single device through the same entry point, and the argument checks
    assert zk.groth16_prove_batch_multi([ctx], [pk0], [d0[j].data_ptr() for j in idx], True, rs, ss) == want
    with pytest.raises(Exception):
        zk.groth16_prove_batch_multi([ctx, ctx], [pk0, pk0], ptrs, True, rs, ss)  # one context twice
    pk0.free()
    pk1.free()
    ctx1.close()
    r1.free()


def _bench(args, timeout=900):
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_msm26_workload_small():
    """bench.py --workload msm26 (BASELINE config 3 in the bench contract) at a reduced size on one GPU: the line
    carries the contract's keys, strong scaling, and the closed-form check of the MSM result."""
    p, lines = _bench(["--workload", "msm26", "--msm-log-n", "22", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    o = lines[0]
    assert o["n_gpus"] == 1 and o["scaling"] == "strong" and o["matches_closed_form_on_every_rank"] is True
    assert o["unit"] == "GB/s" and o["value"] > 0 and o["roofline"]["bound"] == "hbm" and 0 < o["roofline"]["frac"] < 1


def test_bench_gpus_gt_1_spawns_ranks_or_fails():
    """`python bench.py --gpus 2` without a launcher starts two fresh ranks.  With two GPUs the line must say
    n_gpus = 2; on a one-GPU box the second rank cannot bind a device and the whole command must fail without
    printing any line (round 2 printed an n_gpus = 1 line here)."""
    import torch

    p, lines = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "14", "--no-secondary", "--no-cpu-baseline"])
    assert "starting 2 ranks" in p.stderr
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["verified_by_pairing"] is True, p.stderr[-3000:]
    else:
        assert p.returncode != 0 and not lines, p.stdout[-2000:]


@pytest.mark.parametrize("world,workload", [(2, "proofs"), (8, "proofs"), (2, "msm26"), (8, "msm26")])
def test_bench_multi_rank_path_as_a_shared_gpu_dry_run(world, workload):
    """bench.py's N > 1 branch -- the exact command the driver's scaling step runs -- executed before an 8-GPU node meets it:
    `bench.py --gpus N --shared-gpu-dry-run` spawns N ranks BEFORE touching HIP (spawn_ranks), every rank proves / owns its
    share on device 0, the barriers and the max-over-ranks run over gloo, the library's exchange over tests/fake_rccl, and
    rank 0 alone emits ONE JSON line with every contract key, `cpu_baseline` included; the line is marked and its metric name
    cannot be mistaken for a scaling point.  LOCAL_WORLD_SIZE reaches the library: a rank's host threads = its share of
    the CPUs the box grants."""
    args = ["--gpus", str(world), "--shared-gpu-dry-run", "--warmup", "1", "--no-secondary", "--cpu-sample-log-n", "14"]
    if workload == "proofs":
        args += ["--steps", "3", "--log-n", "14"]
    else:
        args += ["--workload", "msm26", "--steps", "1", "--msm-log-n", "20", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-3000:])
    assert "starting %d ranks" % world in p.stderr
    o = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in o, key
    assert o["dry_run_shared_gpu"] is True and o["metric"].startswith("DRY_RUN_%d_ranks_sharing_one_gpu__" % world) and o["physical_gpus"] == 1
    assert o["n_gpus"] == world and o["value"] > 0
    if workload == "proofs":
        assert o["verified_by_pairing"] is True and o["scaling"] == "weak"
        assert o["cpu_baseline"]["kind"] == "port" and o["cpu_baseline"]["proof_bytes_match_gpu"] is True
        h = o["host"]
        assert h["local_ranks"] == world and h["threads"] == max(1, min(16, h["cpus_granted"] // world)), h
        assert abs(o["value"] - world * o["steps"] / (o["ms_per_step"] * o["steps"] / 1e3)) < 1e-6 * o["value"]
    else:
        assert o["matches_closed_form_on_every_rank"] is True and o["scaling"] == "strong"
    # a launcher with another world size is refused, dry run or not
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--shared-gpu-dry-run"], capture_output=True, text=True, env=env2, timeout=120)
    assert p2.returncode == 2 and "refusing" in p2.stderr and "{" not in p2.stdout


VARIANTS = [
    {},
    {"ZKMI_NTT_RB": "0"}, {"ZKMI_NTT_RB": "1"}, {"ZKMI_NTT_RB": "2"}, {"ZKMI_NTT_RB": "3"}, {"ZKMI_NTT_RB": "4"}, {"ZKMI_NTT_RB": "5"},
    {"ZKMI_NTT_LOCAL3": "1", "ZKMI_WITNESS_BATCH": "0"},
    {"ZKMI_SORT_FINE": "0"}, {"ZKMI_SORT_FINE": "2"}, {"ZKMI_SORT_STAGE": "0"},
    {"ZKMI_HEAVY_ON": "0"}, {"ZKMI_HEAVY_ON": "1"},
    {"ZKMI_ACCUM": "0", "ZKMI_ACCUM_G2": "0"}, {"ZKMI_ACCUM": "2", "ZKMI_ACCUM_G2": "3"}, {"ZKMI_AUX_SPLIT": "0", "ZKMI_SORT_SIDE": "1"},
    {"ZKMI_LH_MERGE": "0"}, {"ZKMI_LH_MERGE": "0", "ZKMI_AUX_SPLIT": "0"}, {"ZKMI_LH_MERGE": "1"}, {"ZKMI_LH_MERGE": "1", "ZKMI_LH_MERGE_GROUPS": "1"}, {"ZKMI_RB1_FOLD": "0"}, {"ZKMI_RB1_FOLD": "1"}, {"ZKMI_RB1_STREAM": "1"}, {"ZKMI_RB1_FOLD_ALWAYS": "1"},
    {"ZKMI_HEAVY_NC": "0"}, {"ZKMI_HEAVY_NC": "0", "ZKMI_HEAVY_ON": "0"}, {"ZKMI_HOST_WAIT": "0"},
    {"ZKMI_BIG_SORT": "0"}, {"ZKMI_BIG_FINE_LOG": "8"}, {"ZKMI_BIG_FINE_LOG": "9"}, {"ZKMI_WIN_TWO_LEVEL": "30", "ZKMI_SOLO_EVENT_ORDER": "0"}, {"ZKMI_WIN_TWO_LEVEL": "17"}, {"ZKMI_BIG_WSTAGE": "0", "ZKMI_SPLIT_PLAIN_RANK": "1"}, {"ZKMI_SORT_BIG": "1"},
    {"ZKMI_SOLO_FUSE_H": "0", "ZKMI_SOLO_G2_EARLY": "0"}, {"ZKMI_SOLO_MAX_LOG": "12", "ZKMI_G2_TREE_SPLIT": "0"}, {"ZKMI_SPREAD": "0"}, {"ZKMI_SOLO_SPLIT": "0"}, {"ZKMI_FORCE_MULTI": "1"}, {"ZKMI_FORCE_MULTI": "1", "ZKMI_SOLO_SPLIT": "0"},
    {"ZKMI_QUAD": "0", "ZKMI_QUAD_BATCH": "0", "ZKMI_QUAD_G2": "0", "ZKMI_QUAD_G2_BATCH": "0"}, {"ZKMI_QUAD": "5", "ZKMI_QUAD_BATCH": "10", "ZKMI_QUAD_G2": "10", "ZKMI_QUAD_G2_BATCH": "5"},
    {"ZKMI_QUAD": "10", "ZKMI_QUAD_BATCH": "5", "ZKMI_QUAD_G2": "5", "ZKMI_QUAD_G2_BATCH": "10"}, {"ZKMI_QUAD_BATCH": "15", "ZKMI_QUAD_G2_BATCH": "15", "ZKMI_HEAVY_NC": "0"}, {"ZKMI_QUAD": "31", "ZKMI_QUAD_G2": "31"}, {"ZKMI_SOLO_SPLIT_G2": "0", "ZKMI_HEAVY_DEFER": "1"},
]
EXP_LIB = os.path.join(ROOT, "zk-apps_amd", "libzkmi_exp.so")


def _variant_digest(env_extra, lib=None):
    env = dict(os.environ, **env_extra)
    if lib:
        env["ZKMI_LIB"] = lib
    else:
        env.pop("ZKMI_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "variant_check.py")], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    line = [l for l in p.stdout.splitlines() if l.startswith("VARIANT_DIGEST")]
    assert p.returncode == 0 and line, (env_extra, lib, p.stdout[-1500:], p.stderr[-3000:])
    return line[0].split()[1]


def test_ab_switches_do_not_change_any_result():
    """The product library has one schedule and reads no tuning switch (csrc/tune.hpp); the A/B library libzkmi_exp.so
    (`make experiments`) carries the switches and the retired kernels they select (register-blocked NTT passes,
    record / direct-scatter digit sorts, where the heavy-bucket kernels run, first-generation accumulation kernels, stream
    layouts, separate L / H bucket sets).  Every variant of it must give the product's bytes: scripts/variant_check.py (NTTs
    of five sizes in four modes, four MSMs, 79 proofs at 2^13 and 2^17) runs in a child process per variant -- the switches
    are read once per process -- and the digests are compared.  The product library is run WITH a hostile environment too:
    it must not react to any of the switches."""
    assert os.path.exists(EXP_LIB), "zk-apps_amd/libzkmi_exp.so missing: run __graft_entry__.build() (make experiments)"
    want = _variant_digest({})
    hostile = {k: v for d in VARIANTS for k, v in d.items()}
    assert _variant_digest(hostile) == want, "the product library reacted to a tuning variable"
    digests = {str(v): _variant_digest(v, EXP_LIB) for v in VARIANTS}
    assert set(digests.values()) == {want}, (want, digests)
