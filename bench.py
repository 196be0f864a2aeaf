#!/usr/bin/env python3
"""bench.py — Shielder-withdraw-shaped Groth16 proofs/s at N = 2^20 on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1
launched under torch.distributed.run, one rank per GPU.  A "step" is one
witness -> proof pass (7 Fr NTTs of 2^20, 4 G1 MSMs + 1 G2 MSM of 2^20 - 1
terms, proof assembly) with the proving key and the witness already resident in
HBM.  Independent proofs shard across ranks with no data-path collective
(weak scaling); the only collectives are the timing barrier and the MAX over
ranks of the elapsed time.

Started WITHOUT a launcher and with --gpus N > 1, this process never touches the
GPU: it starts `python -m torch.distributed.run --nproc-per-node N ... bench.py`
as a child (fresh ranks) and exits with its code, so an N-GPU request can never
be answered with a one-GPU line.  A world size that differs from --gpus is an error.

`--workload msm26` is BASELINE config 3 instead: ONE G1 MSM of 2^26 points split
by points over the ranks, per-window partial sums exchanged with an RCCL
all-gather and combined on every rank (strong scaling: the total work is fixed).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed
inside libzkmi on its launch stream) and, at N = 1, `cpu_baseline` (the in-repo
C++ oracle prover on the host cores, bounded sample).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from zkmi_loader import load_pkg  # noqa: E402

torch = None  # imported in main(), after the decision to spawn ranks (the parent of a spawn never loads it)
dist = None

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr_bytes(self):
        r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < r:
                return v.to_bytes(32, "little")


def withdraw_input(z, seed):
    """The semantic inputs of one withdraw (zkmi_note_update), drawn from SplitMix64(seed); examples/bench_prove.c
    draws the same ones in the same order."""
    rng = SplitMix64(seed)
    f = lambda: int.from_bytes(rng.fr_bytes(), "little")
    tok = (f(), f())
    bal = (rng.next() >> 1, rng.next() >> 1)
    amount = bal[0] >> 3
    user = f()
    return z.note_update(amount, tok[0], user, (f(), f(), f()), (f(), f(), f()), [rng.next() & 1 for _ in range(10)],
                         [f() for _ in range(10)], user, (tok[0], bal[0], tok[1], bal[1]))


def relation_and_witness(z, relation, log_n, seeds):
    """(r1cs, [witness bytes per seed]), assignments generated on the HOST.  "poseidon": the reference's update_note
    relation with real Poseidon-5 hashing (withdraw), padded to 2^log_n; "chain": the hash-free stand-in of the first builds."""
    if relation == "chain":
        return z.shielder_r1cs(log_n), [z.shielder_witness(log_n, s) for s in seeds]
    r1 = z.update_note_r1cs(log_n, 1)
    return r1, [z.update_note_witness(log_n, 1, withdraw_input(z, s))[0] for s in seeds]


MAX_DISTINCT = 64  # distinct resident assignments per rank (32 MiB each at 2^20); longer runs cycle through them


def resident_witnesses(z, ctx, relation, log_n, seeds):
    """(r1cs, [device tensors]): one DISTINCT assignment per seed, resident in HBM before the timed region.  The
    update_note relation generates them on the device (zkmi_update_note_witness_batch_dev: SURVEY.md 8f-1), the chain
    stand-in on the host."""
    if relation == "chain":
        r1, wits = relation_and_witness(z, relation, log_n, seeds)
        return r1, [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    r1 = z.update_note_r1cs(log_n, 1)
    bufs = [torch.empty(32 << log_n, dtype=torch.uint8, device="cuda") for _ in seeds]
    st = ctx.update_note_witness_batch_dev(log_n, 1, [withdraw_input(z, s) for s in seeds], [b.data_ptr() for b in bufs])
    assert all(x == 0 for x in st), st
    return r1, bufs


def source_revision(z=None):
    """What was measured: the last commit that touched the kernels / C ABI ('+dirty' when the tree differs from it) where
    a git checkout is present, and always a digest of those sources themselves (the GPU box receives a snapshot without
    .git) -- the same digest scripts/pmc_summary.py stamps into the PMC summary, so a profile that is older than the
    library shows."""
    out = {"git": None, "csrc_sha256_16": csrc_digest()}
    if z is not None:
        # the digest compiled INTO the library that ran (zkmi_version()) against the files on this box: a stale .so beside
        # fresh sources reads false here
        out["library_src_sha256_16"] = z.src_digest()
        out["library_matches_sources"] = out["library_src_sha256_16"] == out["csrc_sha256_16"]
    try:
        h = subprocess.run(["git", "log", "-1", "--format=%h", "--", "zk-apps_amd/csrc", "include"], cwd=ROOT, capture_output=True,
                           text=True, timeout=10).stdout.strip()
        d = subprocess.run(["git", "status", "--porcelain", "--", "zk-apps_amd/csrc", "include"], cwd=ROOT, capture_output=True,
                           text=True, timeout=10).stdout.strip()
        out["git"] = (h + ("+dirty" if d else "")) or None
    except Exception:
        pass
    return out


def csrc_digest():
    """sha256 (16 hex digits) over the kernel / C-ABI sources (scripts/src_digest.py: the one definition, also compiled
    into the library by the Makefile and returned by zkmi_version())"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import src_digest
    finally:
        sys.path.pop(0)
    return src_digest.csrc_digest(ROOT)


def cpu_kernel_twins(z, ctx, log_n):
    """BASELINE.md section 2's per-kernel CPU twins at the full size: the oracle's G1 MSM (arkworks
    msm_bigint shape) and Fr NTT over 2^log_n terms on the host cores."""
    from oracle import cpp as ocpp  # the checker; only the cpu_baseline leg may touch oracle/

    import numpy as np

    n = 1 << log_n
    raw = bytearray(np.random.default_rng(0x5A4B00C1).bytes(32 * n))
    for i in range(31, 32 * n, 32):
        raw[i] &= 0x3F  # canonical
    raw = bytes(raw)
    b = ctx.bases_g1_synthetic(n)
    pts = b.read(0, n)
    t0 = time.time()
    want = ocpp.msm_g1(raw, pts)
    t_msm = time.time() - t0
    ok = ctx.msm_g1(raw, b) == want
    b.free()
    t_ntt = None
    for _ in range(2):  # the first call also pays for the 32 MiB ctypes staging buffers
        t0 = time.time()
        ocpp.ntt(raw, log_n)
        t_ntt = min(t_ntt or 1e9, time.time() - t0)
    return {"msm_g1_cpu": {"n": n, "seconds": t_msm, "GBps": 128.0 * n / t_msm / 1e9, "equals_gpu_result": ok},
            "ntt_fr_cpu": {"n": n, "seconds": t_ntt, "GBps": 64.0 * n / t_ntt / 1e9}}


def cpu_baseline(z, ctx, sample_log_n, full_log_n, relation):
    """In-repo C++ oracle prover ("port") on the host cores, on a bounded sample: one full proof at 2^sample_log_n
    (default: the full size, N = 2^20), scaled linearly in N to 2^full_log_n when smaller."""
    from oracle import cpp as ocpp  # the checker; only this leg may touch oracle/

    ocpp.build()
    r1, (wit,) = relation_and_witness(z, relation, sample_log_n, [0x5A4B0000])
    rng = SplitMix64(0x5A4B00C0)
    toxic = b"".join(rng.fr_bytes() for _ in range(5))
    pk, vk = ctx.groth16_setup(r1, toxic)
    n, N = r1.n_vars, 1 << r1.log_n
    g = z  # wire-format key for the CPU prover
    # alpha/beta/delta in wire format from the vk + host multiplications
    alpha_g1, beta_g2, delta_g2 = vk[:96], vk[96:288], vk[480:672]
    beta_g1 = g.g1_mul(g.g1_generator(), toxic[64:96])
    delta_g1 = g.g1_mul(g.g1_generator(), toxic[128:160])
    key = {
        "alpha_g1": alpha_g1, "beta_g1": beta_g1, "beta_g2": beta_g2, "delta_g1": delta_g1, "delta_g2": delta_g2,
        "a_query": pk.export_query(0, 0, n), "b_g1_query": pk.export_query(1, 0, n),
        "b_g2_query": pk.export_query(2, 0, n), "h_query": pk.export_query(3, 0, N - 1),
        "l_query": pk.export_query(4, 0, n - r1.n_pub),
    }
    mats = [r1.export(m) for m in range(3)]
    r, s = rng.fr_bytes(), rng.fr_bytes()
    proof_cpu, dt = ocpp.groth16_prove_timed(r1.n_vars, r1.n_pub, r1.n_constraints, r1.log_n, mats, key, wit, r, s)
    proof_gpu = ctx.groth16_prove(pk, wit, r, s)
    pk.free()
    scale = float(1 << (full_log_n - sample_log_n))
    twins = cpu_kernel_twins(z, ctx, full_log_n)
    return {
        "value": 1.0 / (dt * scale),
        "unit": "proofs/s",
        "cores": ocpp.threads(),
        "kind": "port",
        "sample": f"one full proof at N=2^{sample_log_n} with the in-repo C++ oracle prover (arkworks-algorithm restatement, "
                  f"not arkworks; MSMs split over (window x point-chunk) tasks, one fork per NTT stage) took {dt:.2f} s on "
                  f"{ocpp.threads()} threads" + (f"; scaled x{int(scale)} (linear in N) to N=2^{full_log_n}" if scale != 1 else ""),
        "proof_bytes_match_gpu": proof_cpu == proof_gpu,
        "kernel_twins_at_full_size": twins,
    }


class stdout_to_stderr:
    """fd-level redirect of stdout to stderr for the duration of the block, C stdio buffers included"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # text a C library printed sits in ITS stdout buffer: flush while fd 1 is still stderr
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def _need_torch():
    global torch, dist
    if torch is None:
        import torch as _t
        import torch.distributed as _d

        torch, dist = _t, _d


def git_blob_hash(data):
    """`git hash-object` of a byte string: ties a figure read from a committed summary to that file's version."""
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(path, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary.

    PMC counters cannot be read from inside the timed process, so the separate `--pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` passes of this same command (scripts/profile.sh) are summarised into a JSON
    file that travels with the repo; this returns (FETCH_SIZE + WRITE_SIZE) * 1024 for one launch.
    The kernel gathers 112/224-byte table entries with per-lane loads (64-byte fabric requests), so the
    guide's x2 correction for 128-byte coalesced requests is NOT applied.
    A missing file or kernel yields traffic = None with a note (the timing above it stays valid)."""
    pmc_traffic.valu = pmc_traffic.name = pmc_traffic.total_valu = pmc_traffic.source = None
    pmc_traffic.held_clock_ghz = pmc_traffic.alone_ms = pmc_traffic.upper_bound = None
    if path == "none":
        return None, "PMC summary lookup disabled (--pmc-summary none)"
    try:
        with open(os.path.join(ROOT, path), "rb") as f:
            raw = f.read()
        rows = json.loads(raw)
    except (OSError, ValueError) as e:
        return None, "PMC summary %s unreadable (%s): traffic not reported" % (path, e)
    pmc_traffic.source = {"file": path, "git_blob": git_blob_hash(raw)}
    # wave-instructions of one proof = sum over the per-proof kernels of (avg per dispatch x dispatches per proof);
    # the summary's metadata row says how many proofs its passes ran
    meta = next((r for r in rows if r.get("kernel") == "__meta__"), None)
    if meta and meta.get("csrc_sha256_16"):
        # the sources the profiled library was built from against the sources this run measures
        pmc_traffic.source["profiled_csrc_sha256_16"] = meta["csrc_sha256_16"]
        pmc_traffic.source["same_sources_as_this_run"] = meta["csrc_sha256_16"] == csrc_digest()
    if meta and meta.get("proofs"):
        # (setup-time kernels, and the generation of the resident assignments in front of the timed region)
        one_time = ("k_build_table", "k_fixed_base", "k_bases_convert", "k_bitrev_points", "k_power_table", "k_bitrev_copy",
                    "k_update_note_values")
        tot = 0.0
        for r in rows:
            if r.get("kernel", "").startswith(one_time) or "SQ_INSTS_VALU_avg_per_dispatch" not in r:
                continue
            tot += r["SQ_INSTS_VALU_avg_per_dispatch"] * r["SQ_INSTS_VALU_dispatches"] / meta["proofs"]
        pmc_traffic.total_valu = tot
    for r in rows:
        # `kernel` is a name prefix: the accumulation kernel's template arguments depend on the build's defaults
        if r.get("kernel", "").replace(" ", "").startswith(kernel.replace(" ", "")) and "SQ_INSTS_VALU_avg_per_dispatch" in r:
            pmc_traffic.name = r["kernel"]
            fetch = r.get("FETCH_SIZE_avg_per_dispatch")
            write = r.get("WRITE_SIZE_avg_per_dispatch")
            if fetch is None or write is None:
                break
            pmc_traffic.valu = r.get("SQ_INSTS_VALU_avg_per_dispatch")
            pmc_traffic.held_clock_ghz = r.get("held_clock_ghz_avg_per_dispatch")
            pmc_traffic.alone_ms = (r.get("alone_ns_avg_per_dispatch") or 0) / 1e6 or None
            # The guide's gfx950 correction (FETCH_SIZE tallies the 128-B requests of a wide COALESCED read -- 16 B per lane,
            # adjacent lanes adjacent addresses -- at 64 B: double it) applies to streaming reads.  This kernel's reads are
            # per-lane gathers of 112- / 224-byte table entries, 16 B per lane and instruction at 64 unrelated addresses: 64-B
            # requests, counted exactly -- so `traffic` is the UNCORRECTED sum, and `traffic_upper_bound` = 2 x FETCH + WRITE
            # is what it would be if every request were a mis-tallied 128-B one; the truth lies between, near the lower value.
            pmc_traffic.upper_bound = (2.0 * fetch + write) * 1024.0
            return (fetch + write) * 1024.0, (
                "bytes/launch = (FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB) from %s (separate rocprofv3 --pmc passes of "
                "this command). UNCORRECTED for the gfx950 FETCH_SIZE half-count: that correction applies to wide coalesced "
                "streaming reads (128-B requests tallied at 64 B), these are per-lane gathers of table entries (64-B requests, "
                "counted exactly); upper bound if every request were half-counted: traffic_upper_bound = 2 x FETCH + WRITE"
                % (fetch, write, path))
    return None, ("kernel %r has no FETCH_SIZE / WRITE_SIZE row in %s (stale summary: refresh with scripts/profile.sh): "
                  "traffic not reported" % (kernel, path))


pmc_traffic.valu = None
pmc_traffic.name = None
pmc_traffic.total_valu = None
pmc_traffic.source = None
pmc_traffic.held_clock_ghz = None
pmc_traffic.alone_ms = None
pmc_traffic.upper_bound = None
# 1024 SIMDs x 2.4 GHz / 4 cycles: v_mad_u64_u32 / v_mad_i64_i32 (77 % of the kernel's instructions) issue once per
# 4 cycles per SIMD (scripts/ubench.hip); under this load the chip holds ~1.95 GHz, so ~500 G/s is what is attainable
VALU_ISSUE_PEAK = 614.4e9
# The same ceiling for the accumulation loop's ACTUAL opcode mix: its 4 619 VALU instructions per addition cost 0.972
# multiply-add slots on average (simple VOP2 operations -- 8 % of the mix -- issue in 0.55 of a slot, 64-bit shifts / adds,
# v_mul_lo and VOP3 forms in 1.05: scripts/isa_mix_report.py over scripts/ubench_ops.hip's measured issue costs,
# profiles/r05/experiments/isa_mix_k_accum_g1_nc.txt)
VALU_MIX_SLOTS_PER_INSTR = 0.972
VALU_ISSUE_PEAK_ACTUAL_MIX = VALU_ISSUE_PEAK / VALU_MIX_SLOTS_PER_INSTR


def secondary_measurements(z, ctx, log_n):
    """SURVEY.md 8d items beside the headline: one whole G1 MSM (digit sort + bucket accumulation + reduction +
    host combine) of 2^log_n terms alone on the chip, with uniform scalars and with the witness-like mix
    (40 % zero, 20 % one, 10 % < 2^16, 30 % uniform)."""
    _need_torch()
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(0x5A4B)
    uni = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    uni[:, 31] &= 0x3F
    kind = torch.rand(n, device="cuda", generator=g)
    mix = uni.clone()
    small = (kind >= 0.6) & (kind < 0.7)
    mix[small, 2:] = 0
    mix[(kind >= 0.4) & (kind < 0.6)] = 0
    mix[(kind >= 0.4) & (kind < 0.6), 0] = 1
    mix[kind < 0.4] = 0
    b = ctx.bases_g1_synthetic(n)
    out = {}
    for name, sc in (("uniform", uni), ("witness_like", mix)):
        ctx.msm_g1_dev(sc.data_ptr(), n, b)  # warm-up (allocations, plans)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.msm_g1_dev(sc.data_ptr(), n, b)
            times.append(time.perf_counter() - t0)
        t = sorted(times)[len(times) // 2]
        out[name] = {"ms": 1e3 * t, "GBps": 128.0 * n / t / 1e9, "frac_of_hbm_peak": 128.0 * n / t / 1e9 / HBM_PEAK_GBS}
    b.free()
    return {"n": n, "what": "zkmi_msm_g1_dev end to end (sort + accumulate + reduce + host window combine), windowed schedule, "
            "scalars resident, median of 5, wall clock", **out}


def ntt_alone(z, ctx, log_n, pmc_path):
    """BASELINE configs[1] names a 2^20 NTT beside the MSM: one Fr transform of 2^log_n resident elements alone on the chip,
    forward and coset-inverse (the two shapes the prover runs), through the C ABI (zkmi_ntt_dev: canonical 32-byte elements
    in place).  Algorithmic bytes 2 x 32 N; wall clock around a synchronised call, median of 9."""
    _need_torch()
    N = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(0x5A4B0007)
    x = torch.randint(0, 256, (N, 32), dtype=torch.uint8, device="cuda", generator=g)
    x[:, 31] &= 0x3F
    out = {"n": N, "what": "zkmi_ntt_dev in place over resident canonical elements (conversion to limb form, the two passes, "
           "conversion back), alone on the chip, median of 9, wall clock around ctx.sync()"}
    for name, kw in (("forward", {}), ("coset_inverse", {"inverse": True, "coset": True})):
        ctx.ntt_dev(x.data_ptr(), log_n, **kw)
        ctx.sync()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            ctx.ntt_dev(x.data_ptr(), log_n, **kw)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        out[name] = {"ms": 1e3 * t, "GBps": 64.0 * N / t / 1e9, "frac_of_hbm_peak": 64.0 * N / t / 1e9 / HBM_PEAK_GBS}
    # the pass kernel's own figures from the PMC summary (alone on the chip: scripts/pmc_summary.py), when it has them
    try:
        rows = json.loads(open(os.path.join(ROOT, pmc_path)).read()) if pmc_path != "none" else []
    except (OSError, ValueError):
        rows = []
    for r in rows:
        # (the full-size pass kernel: 512 threads since round 6, 1 024 before)
        if r.get("kernel", "").startswith("k_ntt_pass") and ("512" in r["kernel"] or "1024" in r["kernel"]) and r.get("alone_ns_avg_per_dispatch") and log_n == 20:
            ms = r["alone_ns_avg_per_dispatch"] / 1e6
            out["pass_kernel"] = {"kernel": r["kernel"], "alone_ms_per_launch": ms,
                                  "valu_frac_of_issue_peak": r.get("SQ_INSTS_VALU_avg_per_dispatch", 0) / (ms * 1e-3) / VALU_ISSUE_PEAK,
                                  "note": "one of the two passes of a transform, from the committed PMC summary (SQ_INSTS_VALU / duration alone)"}
            break
    return out


def config4_block(z, ctx, relation, steps=6):
    """BASELINE configs[4]: full Groth16 proofs at 2^22 (G2 MSM of 2^22 - 1 terms, pairing check) on one GPU -- the batch
    rate over `steps` proofs and the G2 MSM alone (zkmi_msm_g2_dev over the key-sized synthetic bases): 224 n algorithmic
    bytes."""
    _need_torch()
    lg = 22
    t0 = time.time()
    r1, d_wits = resident_witnesses(z, ctx, relation, lg, [0x5A4B4000, 0x5A4B4001])
    rng = SplitMix64(0x5A4B0044)
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    setup_s = time.time() - t0
    rs = [(rng.fr_bytes(), rng.fr_bytes()) for _ in range(steps)]
    ptrs = [d_wits[i % 2].data_ptr() for i in range(steps)]
    ctx.groth16_prove_batch_dev(pk, ptrs[:2], [a for a, _ in rs[:2]], [b for _, b in rs[:2]])
    ctx.sync()
    t0 = time.perf_counter()
    proofs = ctx.groth16_prove_batch_dev(pk, ptrs, [a for a, _ in rs], [b for _, b in rs])
    ctx.sync()
    dt = time.perf_counter() - t0
    pub = bytes(d_wits[(steps - 1) % 2][32 : 32 * r1.n_pub].cpu().numpy().tobytes())
    ok = z.groth16_verify(vk, pub, proofs[-1])
    lat = []
    for i in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.groth16_prove_dev(pk, ptrs[i % 2], rs[i][0], rs[i][1])
        lat.append(time.perf_counter() - t0)
    pk.free()
    r1.free()
    del d_wits
    torch.cuda.empty_cache()
    n = (1 << lg) - 1
    g = torch.Generator(device="cuda").manual_seed(0x5A4B0045)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    sc[:, 31] &= 0x3F
    b = ctx.bases_g2_synthetic(n)
    ctx.msm_g2_dev(sc.data_ptr(), n, b)
    ts = []
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.msm_g2_dev(sc.data_ptr(), n, b)
        ts.append(time.perf_counter() - t0)
    b.free()
    tg = sorted(ts)[1]
    return {"log_n": lg, "proofs": steps, "proofs_per_s": steps / dt, "ms_per_proof": 1e3 * dt / steps,
            "single_proof_latency_ms": 1e3 * sorted(lat)[1], "verified_by_pairing": bool(ok), "setup_seconds": setup_s,
            "algorithmic_GBps_whole_proof": 1184.0 * (1 << lg) * steps / dt / 1e9,
            "msm_g2_alone": {"n": n, "ms": 1e3 * tg, "GBps": 224.0 * n / tg / 1e9, "frac_of_hbm_peak": 224.0 * n / tg / 1e9 / HBM_PEAK_GBS,
                             "what": "zkmi_msm_g2_dev end to end over plain bases (windowed schedule), uniform scalars, median of 3"}}


def small_domain_rate(z, ctx, relation, log_n=14, count=512):
    """BASELINE config 0's size (2^14, the relation's natural size): `count` independent proofs through the same batch
    entry point, which at this size moves groups of 64 proofs through one sort / one accumulation launch per query."""
    _need_torch()
    r1, wits = relation_and_witness(z, relation, log_n, [0x5A4B0100, 0x5A4B0101])
    rng = SplitMix64(0x5A4B0102)
    pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
    d = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
    rs = [(rng.fr_bytes(), rng.fr_bytes()) for _ in range(2)]
    torch.cuda.synchronize()
    idx = [i % 2 for i in range(count)]
    args = ([d[j].data_ptr() for j in idx], [rs[j][0] for j in idx], [rs[j][1] for j in idx])
    ctx.groth16_prove_batch_dev(pk, *[a[:128] for a in args])
    ctx.sync()
    t0 = time.perf_counter()
    c0 = cpu_seconds()
    proofs = ctx.groth16_prove_batch_dev(pk, *args)
    ctx.sync()
    cpu_s = cpu_seconds() - c0
    dt = time.perf_counter() - t0
    ok = all(z.groth16_verify(vk, wits[j][32 : 32 * r1.n_pub], proofs[i]) for i, j in ((0, 0), (count - 1, (count - 1) % 2)))
    lat = []  # one proof at a time on an idle GPU (the wallet's case): witness resident -> 192 bytes on the host
    for i in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.groth16_prove_dev(pk, d[i % 2].data_ptr(), rs[i % 2][0], rs[i % 2][1])
        lat.append(time.perf_counter() - t0)
    pk.free()
    r1.free()
    return {"log_n": log_n, "proofs": count, "proofs_per_s": count / dt, "ms_per_proof": 1e3 * dt / count, "verified_by_pairing": bool(ok),
            "single_proof_latency_ms": 1e3 * sorted(lat)[2],
            # what the host spends per proof (assembly pool + driving thread + HIP runtime), and what eight such ranks would
            # ask of the node's CPUs at this rate (the GPU box grants 16: DESIGN.md section 5)
            "host_cpu_s_per_proof": cpu_s / count,
            "host_cpus_busy": cpu_s / dt,
            "host_cpus_busy_x8_ranks": 8 * cpu_s / dt,
            "note": "same entry point (zkmi_groth16_prove_batch_dev); groups of up to 64 proofs share one digit sort, one "
                    "accumulation launch per query and batched NTT passes"}


def spawn_ranks(n_gpus):
    """--gpus N > 1 without a launcher: start N fresh ranks under torch.distributed.run and pass their exit code on.
    This process has not touched the GPU (torch is not even imported) and never will."""
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: starting %d ranks: %s" % (n_gpus, n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, cwd=ROOT)


def cpu_seconds():
    """user + system CPU time of this process, all threads (the library's assembly pool and the HIP runtime's included)"""
    import resource

    ru = resource.getrusage(resource.RUSAGE_SELF)
    return ru.ru_utime + ru.ru_stime


def timed_region(ctx, use_dist, fn):
    """barrier + synchronize on both sides of fn(); returns (result, elapsed seconds = MAX over ranks).
    timed_region.cpu_s = this rank's CPU seconds inside fn() (before the closing barrier: a rank that waits for the others
    inside an RCCL barrier spins, which is the launcher's cost, not the prover's)."""
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    c0 = cpu_seconds()
    res = fn()
    ctx.sync()
    timed_region.cpu_s = cpu_seconds() - c0
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=timed_region.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return res, elapsed


timed_region.reduce_device = "cuda"  # "cpu" under --shared-gpu-dry-run (gloo)


R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def run_msm26(args, pkg, z, ctx, rank, world, use_dist):
    """BASELINE config 3: one G1 MSM of 2^msm_log_n points, split by POINTS over the ranks (SURVEY.md 8e's preferred
    partition; BASELINE.json words it as a windows split -- see DESIGN.md section 6 for why points).  Every rank sorts and
    accumulates its n / world points with the window width planned from the global n and emits one partial sum per window;
    the ranks all-gather the device-resident partial sums over RCCL (zkmi_msm_g1_allgather_combine: ncclAllGather of
    ncclUint8 behind the C ABI) and every rank adds and Horner-combines them (EC addition is not an RCCL reduction op: this
    IS the all-reduce).  A step = one whole MSM; value = algorithmic GB/s of the job."""
    par = __import__("zk_apps_amd.parallel", fromlist=["x"])
    n = 1 << args.msm_log_n
    a, b = par.shard_units(n, rank, world)
    m = b - a
    g = torch.Generator(device="cuda").manual_seed(0x5A4B0003 + rank)
    raw = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda", generator=g)
    raw[:, 31] &= 0x3F
    assert args.msm_log_n <= 27  # 255 * i summed over 2^27 terms stays below 2^63
    idx = torch.arange(a, b, dtype=torch.int64, device="cuda")
    sums = torch.stack([raw.to(torch.int64).sum(dim=0), (raw.to(torch.int64) * idx[:, None]).sum(dim=0)]).contiguous()
    del idx
    if use_dist:
        sums = sums.to(timed_region.reduce_device)  # (gloo under --shared-gpu-dry-run)
        allsums = [torch.empty_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
    else:
        allsums = [sums]
    tot = wtot = 0
    for t in allsums:
        v = t.cpu().tolist()
        tot += sum(x << (8 * k) for k, x in enumerate(v[0]))
        wtot += sum(x << (8 * k) for k, x in enumerate(v[1]))
    if args.split == "windows":
        # every rank over ALL points: rebuild the other ranks' scalars (same generators) and keep the whole base sequence
        parts = []
        for k in range(world):
            ak, bk = par.shard_units(n, k, world)
            if k == rank:
                parts.append(raw)
            else:
                gk = torch.Generator(device="cuda").manual_seed(0x5A4B0003 + k)
                t = torch.randint(0, 256, (bk - ak, 32), dtype=torch.uint8, device="cuda", generator=gk)
                t[:, 31] &= 0x3F
                parts.append(t)
        raw = torch.cat(parts).contiguous() if world > 1 else raw
        del parts
        m = n
        bases = ctx.bases_g1_synthetic(n)
    else:
        bases = ctx.bases_g1_synthetic_range(a, m)

    def one():
        if args.split == "windows":
            return ctx.msm_g1_window_split_allgather(comm, raw.data_ptr(), n, bases)
        if args.split == "2d":
            return ctx.msm_g1_split2d_allgather(comm, raw.data_ptr(), m, bases, n, args.window_groups)
        return ctx.msm_g1_allgather_combine(comm, raw.data_ptr(), m, bases, n)

    # the exchange behind the C ABI: zkmi_comm (RCCL; torch.distributed only carries rank 0's 128-byte id), partial sums
    # all-gathered from HBM on the reduction stream, combined on every rank -- also at one rank (a world of 1)
    # (RCCL prints a version banner on STDOUT when its first communicator comes up: sent to stderr, stdout carries the one JSON line)
    with stdout_to_stderr():
        if use_dist:
            comm = par.rccl_comm(z, ctx)
        else:
            comm = ctx.comm_init(1, 0, z.comm_unique_id())
        one()  # first collective (workspaces, RCCL channels)

    for _ in range(max(1, args.warmup)):  # the first call allocates the workspaces
        one()
    ctx.prof_enable(True)
    ctx.prof_reset()
    got, elapsed = timed_region(ctx, use_dist, lambda: [one() for _ in range(args.steps)][-1])
    ctx.prof_enable(False)
    G = z.g1_generator()
    Q = z.g1_mul(G, (0xC0FFEE).to_bytes(32, "little"))
    want = z.g1_add(z.g1_mul(G, (tot % R_MOD).to_bytes(32, "little")), z.g1_mul(Q, (wtot % R_MOD).to_bytes(32, "little")))
    ok = got == want
    if use_dist:
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=timed_region.reduce_device)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item())
    comm.free()
    phases = {k: ctx.prof_get(k) for k in pkg.PHASES}
    ms_tot, launches = phases["msm_accum_g1"]
    # (the pipelined form accumulates an MSM's windows in two launches: the roofline leg is per MSM, i.e. both of them)
    avg_ms = ms_tot / max(1, args.steps)
    achieved = 128.0 * m / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    sec = elapsed / args.steps
    out = {
        "metric": "g1_msm_2^%d_points_algorithmic_GBps" % args.msm_log_n,
        "value": 128.0 * n / sec / 1e9,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * sec,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int: signed 28-bit limbs in 32-bit words, 64-bit column accumulators (Fq 381-bit Montgomery)",
        "data": "synthetic",
        "config": {"workload": "single G1 MSM of 2^%d points (BASELINE configs[3]), split by %s over %d GPU(s), RCCL all-gather of "
                               "per-window partial sums + local combine" % (args.msm_log_n, args.split, world),
                   "points_per_rank": m, "curve": "BLS12-381", "scalars": "uniform < 2^254", "bases": "P_i = G + i*[0xC0FFEE]G"},
        "matches_closed_form_on_every_rank": ok,
        "points_per_s": n / sec,
        "frac_of_hbm_peak": 128.0 * n / sec / 1e9 / (HBM_PEAK_GBS * world),
        "phase_ms_per_msm": {k: v[0] / args.steps for k, v in phases.items()},
        "roofline": {"kernel": "k_accum_g1_nc (rank 0's share: %d points)" % m, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     # HIP-event time of the accumulation phase per MSM (one launch in the product; the A/B library's pipelined
                     # form brackets two launches per MSM: `launches_per_msm`)
                     "avg_ms_per_msm": avg_ms, "launches": launches,
                     "launches_per_msm": launches / max(1, args.steps),
                     "algorithmic_bytes_per_launch": 128 * m,
                     "limiter": "VALU integer issue (384-bit Montgomery products), see DESIGN.md 4.1"},
    }
    # HBM traffic and instruction count of the accumulation launch from the committed PMC summary of THIS workload
    # (scripts/profile.sh: separate --pmc passes of `bench.py --workload msm26 --steps 1`), one rank only
    if world == 1 and args.msm_pmc_summary != "none" and args.split == "points":
        traffic, note = pmc_traffic(args.msm_pmc_summary, "k_accum_g1_nc")
        out["roofline"]["traffic"] = traffic
        out["roofline"]["traffic_upper_bound"] = pmc_traffic.upper_bound
        out["roofline"]["traffic_note"] = note
        out["roofline"]["traffic_source"] = pmc_traffic.source
        if pmc_traffic.valu and avg_ms > 0:
            rate = pmc_traffic.valu / (avg_ms * 1e-3)
            out["roofline_valu"] = {"kernel": pmc_traffic.name, "bound": "valu", "achieved": rate / 1e9, "peak": VALU_ISSUE_PEAK / 1e9,
                                    "unit": "G wave-instr/s", "frac": rate / VALU_ISSUE_PEAK, "wave_insts_per_launch": pmc_traffic.valu}
    out["source_revision"] = source_revision(z)
    bases.free()
    if not ok:
        print("bench.py: MSM result differs from the closed form", file=sys.stderr)
    return out, 0 if ok else 1


def mark_dry_run(out, world):
    """--shared-gpu-dry-run: the line must not be mistaken for a scaling point -- all ranks shared ONE GPU."""
    out["dry_run_shared_gpu"] = True
    out["metric"] = "DRY_RUN_%d_ranks_sharing_one_gpu__%s" % (world, out["metric"])
    out["physical_gpus"] = 1
    out["note_dry_run"] = ("all %d ranks ran on device 0 over gloo + the shared-memory all-gather double (tests/fake_rccl): the value exercises the "
                           "multi-rank code path of bench.py and says nothing about scaling, RCCL or xGMI" % world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 48 (proofs: a timed region of ~0.85 s; the ring of three proofs fills and drains once per run) / 3 (msm26)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["proofs", "msm26"], default="proofs",
                    help="proofs = BASELINE configs[1]/[2] (headline); msm26 = configs[3], one 2^26-point G1 MSM split over the ranks")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--msm-log-n", type=int, default=26)
    ap.add_argument("--window-groups", type=int, default=2, help="msm26 --split 2d: window ranges Q (ranks = point groups x Q)")
    ap.add_argument("--split", choices=["points", "windows", "2d"], default="points",
                    help="msm26: how the MSM is cut over the ranks -- by POINTS (SURVEY.md 8e's preferred partition, 1/N of the scalars and "
                         "bases per rank; default) or by WINDOWS (BASELINE configs[3] as worded: every rank holds all points)")
    ap.add_argument("--cpu-sample-log-n", type=int, default=20)
    ap.add_argument("--no-secondary", action="store_true", help="skip the H2D-inclusive and whole-MSM measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--relation", choices=["poseidon", "chain"], default="poseidon")
    ap.add_argument("--pmc-summary", default="profiles/r06/pmc_summary_bench_steps3.json")
    ap.add_argument("--msm-pmc-summary", default="profiles/r06/pmc_summary_msm26_steps1.json")
    ap.add_argument("--shared-gpu-dry-run", action="store_true",
                    help="TEST MODE for boxes with ONE GPU: the N ranks of --gpus N all use device 0 (RCCL refuses that, so the process group "
                         "is gloo and the library's exchange runs over the all-gather double tests/fake_rccl).  Exercises the real spawn, "
                         "rank-0-only emit, max-over-ranks timing and the world-size refusal; the line is marked dry_run_shared_gpu and its "
                         "metric name says so -- it is NOT a scaling point")
    args = ap.parse_args()
    for name in ("pmc_summary", "msm_pmc_summary"):
        # (the newest committed summary; an older round's stands in until scripts/profile.sh has been run for this one --
        # `traffic_source.same_sources_as_this_run` then reads false)
        v = getattr(args, name)
        if v != "none" and not os.path.exists(os.path.join(ROOT, v)) and "profiles/r06/" in v:
            setattr(args, name, v.replace("profiles/r06/", "profiles/r05/"))
    if args.steps is None:
        args.steps = 48 if args.workload == "proofs" else 3
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        return spawn_ranks(args.gpus)  # before anything touches the GPU
    # From here on file descriptor 1 is stderr and the one JSON line goes to a private duplicate of the real stdout:
    # libraries chat on stdout whenever they like (RCCL prints a version banner when a communicator comes up -- during the
    # first collective, from C stdio, sometimes after the call that caused it has returned), and the contract is ONE line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a line for another GPU count"
              % (args.gpus, world), file=sys.stderr)
        return 2
    _need_torch()
    dry = args.shared_gpu_dry_run
    if dry:
        # every rank on device 0; the library's zkmi_comm exchange over the shared-memory double (absolute path, our own file:
        # csrc/comm.hip accepts nothing else from the environment)
        local_rank = 0
        fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
        if not os.path.exists(fake):
            print("bench.py --shared-gpu-dry-run: %s not built (make -C tests/fake_rccl)" % fake, file=sys.stderr)
            return 2
        os.environ["ZKMI_RCCL_LIB"] = fake
    torch.cuda.set_device(local_rank)
    # launched by torch.distributed.run (any world size, also 1): RCCL process group, used only for the
    # barriers around the timed region and the max-over-ranks of the elapsed time -- no data-path collective
    use_dist = launched and "MASTER_ADDR" in os.environ
    timed_region.reduce_device = "cpu" if dry else "cuda"
    if use_dist and dry:
        dist.init_process_group("gloo")
    elif use_dist:
        # RCCL prints a version banner on STDOUT when its first communicator comes up (at the first collective): bring it
        # up here with fd 1 pointed at stderr, so that stdout carries nothing but rank 0's one JSON line
        with stdout_to_stderr():
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()

    pkg = load_pkg()
    z = pkg.Zkmi()
    ctx = z.context(local_rank)
    if args.workload == "msm26":
        out, rc = run_msm26(args, pkg, z, ctx, rank, world, use_dist)
        if dry:
            mark_dry_run(out, world)
        if rank == 0:
            emit(out)
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return rc
    log_n = args.log_n
    N = 1 << log_n

    # --- one-time preparation (not timed): relation, trusted setup on the GPU --
    t0 = time.time()
    # one DISTINCT assignment and one fresh (r, s) per proof of the run (warm-up included), all resident before the timed
    # region; seeds differ per rank.  (Rounds 1-3 alternated two assignments and two blinding pairs.)
    n_distinct = min(args.steps + args.warmup, MAX_DISTINCT)
    r1, d_wits = resident_witnesses(z, ctx, args.relation, log_n, [0x5A4B0000 + 4096 * rank + i for i in range(n_distinct)])
    rng = SplitMix64(0x5A4B0001)
    toxic = b"".join(rng.fr_bytes() for _ in range(5))
    pk, vk = ctx.groth16_setup(r1, toxic)
    setup_s = time.time() - t0
    rs = [(rng.fr_bytes(), rng.fr_bytes()) for _ in range(n_distinct)]
    torch.cuda.synchronize()

    def run(first, count):
        """`count` proofs as one pipelined batch (three in flight on the GPU)."""
        idx = [(first + i) % n_distinct for i in range(count)]
        return ctx.groth16_prove_batch_dev(pk, [d_wits[j].data_ptr() for j in idx], [rs[j][0] for j in idx], [rs[j][1] for j in idx])

    if args.warmup:
        run(0, args.warmup)

    ctx.prof_enable(True)
    ctx.prof_reset()
    proofs, elapsed = timed_region(ctx, use_dist, lambda: run(args.warmup, args.steps))
    proof = proofs[-1] if proofs else None
    ctx.prof_enable(False)

    # correctness of what was timed: the first and the last proof must pass the pairing verifier (publics = z[1 .. n_pub))
    def publics_of(j):
        return bytes(d_wits[j][32 : 32 * r1.n_pub].cpu().numpy().tobytes())

    first_j, last_j = args.warmup % n_distinct, (args.warmup + args.steps - 1) % n_distinct
    verified = bool(proofs) and z.groth16_verify(vk, publics_of(last_j), proof) and z.groth16_verify(vk, publics_of(first_j), proofs[0])

    phases = {k: ctx.prof_get(k) for k in pkg.PHASES}
    n_msm = r1.n_vars - 1
    kernels = {
        "msm_accum_g1": {"bytes": 128 * n_msm},   # n x (32 + 96) B
        "msm_accum_g2": {"bytes": 224 * n_msm},   # n x (32 + 192) B
        "ntt": {"bytes": 7 * 64 * N},             # one launch group = 7 transforms x 2 x 32 N
    }
    dom = max(("msm_accum_g1", "msm_accum_g2"), key=lambda k: phases[k][0])
    ms_tot, launches = phases[dom]
    avg_ms = ms_tot / max(1, launches)
    achieved = kernels[dom]["bytes"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kname = "k_accum_g1_" if dom == "msm_accum_g1" else "k_accum_g2_"
    traffic, traffic_note = pmc_traffic(args.pmc_summary, kname)
    roofline = {
        "kernel": pmc_traffic.name or kname,
        # achieved / peak / frac are the HBM figures BASELINE.json's metric asks for (algorithmic bytes / launch time),
        # so `bound` names that roofline; what actually limits the kernel is the integer multiply-add issue rate:
        # `limiter` + `valu_issue` below
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_upper_bound": pmc_traffic.upper_bound,
        "traffic_note": traffic_note,
        "traffic_source": pmc_traffic.source,
        "avg_launch_ms": avg_ms,
        "launches": launches,
        "algorithmic_bytes_per_launch": kernels[dom]["bytes"],
        "limiter": "valu-issue",
        "note": "bucket accumulation is VALU-integer bound (384-bit Montgomery products), not HBM bound; see DESIGN.md",
    }
    if pmc_traffic.valu and avg_ms > 0:
        # secondary fraction SURVEY.md 8(d) asks for: integer-issue rate of the same kernel
        rate = pmc_traffic.valu / (avg_ms * 1e-3)
        roofline["valu_issue"] = {
            "wave_insts_per_launch": pmc_traffic.valu,
            "achieved": rate / 1e9,
            "peak": VALU_ISSUE_PEAK / 1e9,
            "unit": "G wave-instr/s",
            "frac": rate / VALU_ISSUE_PEAK,
            "note": "SQ_INSTS_VALU (PMC summary) / live avg launch time; the kernel shares the chip with the reductions, "
            "NTTs and the G2 accumulation running on other streams, so its own rate understates chip utilisation",
        }

    out = {
        "metric": "shielder_withdraw_groth16_proofs_per_sec_2^%d_constraints" % log_n,
        "value": world * args.steps / elapsed,
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int: signed 28-bit limbs in 32-bit words, 64-bit column accumulators (Fr 255-bit / Fq 381-bit Montgomery)",
        "data": "synthetic",
        "config": {
            "workload": "single withdraw-shaped Groth16 proof, N=2^%d (7 Fr NTTs, 4 G1 MSM + 1 G2 MSM of 2^%d-1 terms), 1 proof per step per GPU"
            % (log_n, log_n),
            "relation": "update_note (withdraw) with Poseidon-5 hashing, chain-padded to N" if args.relation == "poseidon"
            else "hash-free chain stand-in",
            "curve": "BLS12-381",
            "independent_proofs_per_rank": args.steps,
            "distinct_witnesses_per_rank": min(args.steps, n_distinct),
            "fresh_r_s_per_proof": True,
            "proofs_in_flight_per_gpu": 3,
            # nothing is skipped: all five MSMs run in full; what is shared is the REDUCTION of the three queries that only
            # ever appear summed in C -- the proof bytes equal the oracle's prover's
            "msm_schedule": "A | B1 (over r*z) + L + H reduced together | B2 (G2)",
        },
        "verified_by_pairing": bool(verified),
        "setup_seconds": setup_s,
        "proofs_in_flight": 3,
        "phase_ms_per_proof": {k: v[0] / args.steps for k, v in phases.items()},
        "roofline": roofline,
        "hip_versions": dict(zip(("build", "runtime"), z.hip_versions())),
        "source_revision": source_revision(z),
        # the host side (rank 0): CPUs the process is granted / ranks on the node / threads its assembly pool uses
        # (csrc/host_pool.hpp), CPU seconds per proof inside the timed region and the CPUs that keeps busy
        "host": dict(z.host_info(), host_cpu_s_per_proof=timed_region.cpu_s / max(1, args.steps),
                     host_cpus_busy=timed_region.cpu_s / elapsed if elapsed > 0 else None),
    }
    if rank == 0 and not args.no_secondary:
        # PCIe-inclusive rate (SURVEY.md 8d "end-to-end proofs/s includes witness upload"): the same K proofs from
        # PINNED HOST witnesses; upload i+1 runs on the copy stream while proofs i-1 and i compute.  Never `value`.
        h_wits = [w.cpu().pin_memory() for w in d_wits]
        idx = [(args.warmup + i) % n_distinct for i in range(args.steps)]
        ctx.groth16_prove_batch_host(pk, [h_wits[j].data_ptr() for j in idx[:2]], [rs[j][0] for j in idx[:2]], [rs[j][1] for j in idx[:2]])
        ctx.sync()
        t0 = time.perf_counter()
        hp = ctx.groth16_prove_batch_host(pk, [h_wits[j].data_ptr() for j in idx], [rs[j][0] for j in idx], [rs[j][1] for j in idx])
        ctx.sync()
        dt = time.perf_counter() - t0
        out["value_incl_h2d"] = {"value": args.steps / dt, "unit": "proofs/s per GPU", "ms_per_step": 1e3 * dt / args.steps,
                                 "same_proof_bytes_as_resident_run": hp == proofs,
                                 "note": "witnesses in pinned host memory (32 MiB each at 2^20), uploaded on a copy stream "
                                         "overlapped with the previous proofs; rank 0 only, after the timed region"}
    if rank == 0 and world == 1 and not args.no_secondary:
        # latency of ONE proof on an otherwise idle GPU (nothing to overlap with): witness resident -> 192 bytes on the host
        lat = []
        for i in range(5):
            ctx.sync()
            t0 = time.perf_counter()
            j = (args.warmup + (i % 2) * 4) % n_distinct  # i = 4 ends on the batch's first proof
            one = ctx.groth16_prove_dev(pk, d_wits[j].data_ptr(), rs[j][0], rs[j][1])
            lat.append(time.perf_counter() - t0)
        out["single_proof_latency_ms"] = {"median": 1e3 * sorted(lat)[2], "min": 1e3 * min(lat),
                                          "same_bytes_as_batch": one == proofs[0] if args.steps >= 1 else None,
                                          "note": "zkmi_groth16_prove_dev, one proof in flight, wall clock incl. host assembly"}
        pk.free()
        pk = None
        out["msm_g1_end_to_end"] = secondary_measurements(z, ctx, log_n)
        out["small_domain"] = small_domain_rate(z, ctx, args.relation)
    # whole-proof issue rate: the chip-level figure the per-kernel rate understates (kernels of five streams overlap)
    if pmc_traffic.total_valu:
        out["roofline"]["valu_issue_whole_proof"] = {
            "wave_insts_per_proof": pmc_traffic.total_valu,
            "achieved": pmc_traffic.total_valu * out["value"] / world / 1e9,
            "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instr/s",
            "frac": pmc_traffic.total_valu * out["value"] / world / VALU_ISSUE_PEAK,
            "note": "sum over all kernels of SQ_INSTS_VALU per proof (PMC summary) x proofs/s per GPU",
        }
    if "valu_issue" in out["roofline"]:
        # the roofline that actually bounds the dominant kernel, as an object of its own: `roofline` keeps the HBM figures
        # BASELINE.json's metric asks for (bound = "hbm"), a consumer that classifies kernels by `bound` reads this one
        vi = out["roofline"]["valu_issue"]
        held = pmc_traffic.held_clock_ghz
        out["roofline_valu"] = {"kernel": out["roofline"]["kernel"], "bound": "valu", "achieved": vi["achieved"],
                                # peak for the kernel's ACTUAL opcode mix at 2.4 GHz (631.8); peak_flat = every instruction one 4-cycle slot (614.4)
                                "peak": VALU_ISSUE_PEAK_ACTUAL_MIX / 1e9, "peak_flat": vi["peak"],
                                "unit": vi["unit"], "frac": vi["achieved"] * 1e9 / VALU_ISSUE_PEAK_ACTUAL_MIX, "frac_of_flat_peak": vi["frac"],
                                "whole_proof_frac": out["roofline"].get("valu_issue_whole_proof", {}).get("frac"),
                                # the clock the chip holds under this kernel alone (GRBM_GUI_ACTIVE / 8 / duration, PMC summary) and
                                # both ceilings at THAT clock: what the fractions above read against the silicon's sustained clock
                                "held_clock_ghz": held,
                                "alone": None if not (held and pmc_traffic.alone_ms and pmc_traffic.valu) else {
                                    "ms": pmc_traffic.alone_ms, "achieved": pmc_traffic.valu / pmc_traffic.alone_ms / 1e6,
                                    "frac_of_peak_at_2.4GHz": pmc_traffic.valu / pmc_traffic.alone_ms / 1e6 / (VALU_ISSUE_PEAK_ACTUAL_MIX / 1e9),
                                    "frac_of_peak_at_held_clock": pmc_traffic.valu / pmc_traffic.alone_ms / 1e6 / (VALU_ISSUE_PEAK_ACTUAL_MIX / 1e9 * held / 2.4)}}
    if rank == 0 and world == 1 and not args.no_secondary and log_n == 20:
        # the other BASELINE configs' one-GPU figures in the same driver-run line: configs[1]'s NTT alone,
        # configs[3] on ONE GPU (a step of --workload msm26: the same function), configs[4] at 2^22
        if pk is not None:
            pk.free()
            pk = None
        del d_wits
        torch.cuda.empty_cache()
        out["ntt_2p20"] = ntt_alone(z, ctx, 20, args.pmc_summary)
        saved = {k: getattr(pmc_traffic, k) for k in ("valu", "name", "total_valu", "source", "held_clock_ghz", "alone_ms", "upper_bound")}
        margs = argparse.Namespace(**vars(args))
        margs.steps, margs.warmup, margs.split, margs.msm_log_n = 2, 1, "points", 26
        t0 = time.time()
        m26, rc26 = run_msm26(margs, pkg, z, ctx, 0, 1, False)
        for k, v in saved.items():
            setattr(pmc_traffic, k, v)
        out["msm26_n1"] = {"ms": m26["ms_per_step"], "GBps": m26["value"], "frac_of_hbm_peak": m26["frac_of_hbm_peak"],
                           "matches_closed_form": m26["matches_closed_form_on_every_rank"], "phase_ms_per_msm": m26["phase_ms_per_msm"],
                           "roofline": m26["roofline"], "block_seconds": time.time() - t0,
                           "what": "BASELINE configs[3] on ONE GPU: one G1 MSM of 2^26 points (8 GiB algorithmic), the body of "
                                   "`bench.py --workload msm26`, 2 timed steps"}
        torch.cuda.empty_cache()
        t0 = time.time()
        out["config4"] = config4_block(z, ctx, args.relation)
        out["config4"]["block_seconds"] = time.time() - t0
    if dry:
        mark_dry_run(out, world)
    if rank == 0 and not args.no_cpu_baseline:
        # (at world > 1 too: the other ranks are past their timed region -- the barrier inside timed_region -- and only wait
        # in destroy_process_group; a line without it reads as unmeasured)
        out["cpu_baseline"] = cpu_baseline(z, ctx, min(args.cpu_sample_log_n, log_n), log_n, args.relation)
    if rank == 0:
        emit(out)
    if pk is not None:
        pk.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    return 0 if verified else 1


if __name__ == "__main__":
    sys.exit(main())
