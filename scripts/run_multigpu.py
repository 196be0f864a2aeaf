#!/usr/bin/env python3
"""Multi-GPU legs of BASELINE.json, one process per GPU over RCCL:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/run_multigpu.py [--config 2|3|both] [--log-n 20] [--proofs 64] [--msm-log-n 26]

config 2  `--proofs` independent withdraw proofs at N = 2^log_n, sharded over the ranks with
          parallel.shard_units (no data-path collective); every proof is checked by the pairing verifier,
          the ranks all-gather their 192-byte proofs so that rank 0 holds the whole batch.
config 3  one G1 MSM of 2^msm_log_n points split by points: each rank generates its slice of the synthetic
          bases and scalars in HBM, runs the bucket method, and the ranks exchange the per-window partial sums
          (parallel.msm_g1_split_dev: RCCL all-gather of nwin x 96 B + local combine); the result must equal
          the closed form sum_i s_i (G + i Q) = [sum s_i] G + [sum i s_i] Q on every rank.

Rank 0 prints one JSON line per config.  Also runs with world size 1 (plain `python scripts/run_multigpu.py`).

--one-gpu (TEST MODE, for boxes with a single GPU): all ranks share GPU 0 -- which RCCL refuses ("duplicate GPU") -- so
torch.distributed runs over gloo and libzkmi's exchange over the all-gather double of tests/fake_rccl (ZKMI_RCCL_LIB).
Everything else is the same code: every rank owns its slice / its proofs, plans from the global size, exchanges its
partial sums through zkmi_comm and combines.  The lines then say `n_gpus: 1, ranks: N`; timings mean nothing.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="both", choices=["2", "3", "both"])
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--proofs", type=int, default=64)
    ap.add_argument("--msm-log-n", type=int, default=26)
    ap.add_argument("--window-groups", type=int, default=0,
                    help="config 3 also as a 2-D split: ranks = point groups x this many window ranges (0 = 2 where the world size is even)")
    ap.add_argument("--legs", default="py,abi,windows,2d,prepared",
                    help="config 3: which forms of the MSM to run -- py (partials through torch.distributed), abi (zkmi_msm_g1_allgather_combine), "
                         "windows (window split, <= 2^22 points), 2d (zkmi_msm_g1_split2d_allgather), prepared (digit tables per rank)")
    ap.add_argument("--one-gpu", action="store_true", help="TEST MODE: all ranks share GPU 0 (gloo + tests/fake_rccl), see the docstring")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_gpu = args.one_gpu
    if one_gpu:
        local_rank = 0
        fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
        if not os.path.exists(fake):
            sys.exit("run_multigpu.py --one-gpu: %s not built (make -C tests/fake_rccl)" % fake)
        os.environ["ZKMI_RCCL_LIB"] = fake  # read by libzkmi.so when it first resolves RCCL
    torch.cuda.set_device(local_rank)
    if "MASTER_ADDR" not in os.environ:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ.setdefault("MASTER_PORT", "29533")
    if one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    ddev = torch.device("cpu") if one_gpu else torch.device("cuda")  # where the tensors of torch.distributed collectives live
    gpus = {"n_gpus": 1 if one_gpu else world, "ranks": world, "one_gpu_world": one_gpu}

    from zkmi_loader import load_pkg
    from bench import SplitMix64, relation_and_witness

    pkg = load_pkg()
    par = __import__("zk_apps_amd.parallel", fromlist=["x"])
    z = pkg.Zkmi()
    ctx = z.context(local_rank)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()

    if args.config in ("2", "both"):
        lo, hi = par.shard_units(args.proofs, rank, world)
        # same key on every rank (replicated: 13 GB of 288 GB at 2^20), distinct witnesses per proof
        r1, wits = relation_and_witness(z, "poseidon", args.log_n, [0x5A4B2000 + i for i in range(lo, min(hi, lo + 2))])
        rng = SplitMix64(0x5A4B0001)
        pk, vk = ctx.groth16_setup(r1, b"".join(rng.fr_bytes() for _ in range(5)))
        d_wits = [torch.frombuffer(bytearray(w), dtype=torch.uint8).cuda() for w in wits]
        prng = SplitMix64(0x5A4B3000 + rank)
        rs = [prng.fr_bytes() for _ in range(hi - lo)]
        ss = [prng.fr_bytes() for _ in range(hi - lo)]
        barrier()
        t0 = time.perf_counter()
        proofs = ctx.groth16_prove_batch_dev(pk, [d_wits[i % len(d_wits)].data_ptr() for i in range(hi - lo)], rs, ss) if hi > lo else []
        ctx.sync()
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ddev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ok = all(z.groth16_verify(vk, wits[i % len(wits)][32 : 32 * r1.n_pub], p) for i, p in enumerate(proofs))
        # every rank proves the same number of proofs when world divides --proofs; pad for the gather otherwise
        per = (args.proofs + world - 1) // world
        blob = b"".join(proofs) + bytes(192 * (per - len(proofs)))
        gathered = par.allgather_bytes(blob)
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ddev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if rank == 0:
            n_got = sum(1 for g in gathered for k in range(per) if any(g[192 * k : 192 * k + 192]))
            print(json.dumps({"config": 2, "workload": "%d independent withdraw proofs at N=2^%d over %d rank(s)" % (args.proofs, args.log_n, world),
                              "proofs": args.proofs, "proofs_gathered": n_got, "all_verified": bool(okt.item()),
                              "seconds": float(dt.item()), "proofs_per_s": args.proofs / float(dt.item()), **gpus}), flush=True)
        assert okt.item() == 1
        pk.free()
        r1.free()
        del d_wits
        torch.cuda.empty_cache()

    if args.config in ("3", "both"):
        n = 1 << args.msm_log_n
        a, b = par.shard_units(n, rank, world)
        m = b - a
        g = torch.Generator(device="cuda").manual_seed(1000 + rank)
        raw = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda", generator=g)
        raw[:, 31] &= 0x3F
        # closed form needs sum s_i and sum i s_i over the GLOBAL index i = a + j
        assert args.msm_log_n <= 27  # 255 * i summed over 2^27 terms stays below 2^63
        idx = torch.arange(a, b, dtype=torch.int64, device="cuda")
        s0 = raw.to(torch.int64).sum(dim=0)
        s1 = (raw.to(torch.int64) * idx[:, None]).sum(dim=0)
        sums = torch.stack([s0, s1]).contiguous().to(ddev)
        allsums = [torch.empty_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        tot = wtot = 0
        for t in allsums:
            v = t.cpu().tolist()
            tot += sum(x << (8 * k) for k, x in enumerate(v[0]))
            wtot += sum(x << (8 * k) for k, x in enumerate(v[1]))
        tot %= R
        wtot %= R
        legs = set(args.legs.split(","))
        bases = ctx.bases_g1_synthetic_range(a, m)
        G = z.g1_generator()
        Q = z.g1_mul(G, (0xC0FFEE).to_bytes(32, "little"))
        want = z.g1_add(z.g1_mul(G, tot.to_bytes(32, "little")), z.g1_mul(Q, wtot.to_bytes(32, "little")))
        got = None
        if "py" in legs:
            par.msm_g1_split_dev(z, ctx, raw.data_ptr(), m, bases, n)  # untimed: workspace allocation
            barrier()
            t0 = time.perf_counter()
            got = par.msm_g1_split_dev(z, ctx, raw.data_ptr(), m, bases, n)
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ddev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            okt = torch.tensor([1 if got == want else 0], dtype=torch.int32, device=ddev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if rank == 0:
                print(json.dumps({"config": 3, "workload": "G1 MSM of 2^%d points split by points over %d rank(s), RCCL all-gather of per-window partials"
                                  % (args.msm_log_n, world), "matches_closed_form_on_every_rank": bool(okt.item()), "seconds": float(dt.item()),
                                  "algorithmic_GBps": 128.0 * n / float(dt.item()) / 1e9, **gpus}), flush=True)
            assert okt.item() == 1
        # the same exchange behind the C ABI: zkmi_comm (RCCL communicator created from rank 0's 128-byte id) +
        # zkmi_msm_g1_allgather_combine (ncclAllGather of the device-resident partial sums, combination on every rank)
        comm = par.rccl_comm(z, ctx)
        okt = None
        if "abi" in legs:
            ctx.msm_g1_allgather_combine(comm, raw.data_ptr(), m, bases, n)  # untimed
            barrier()
            t0 = time.perf_counter()
            got_c = ctx.msm_g1_allgather_combine(comm, raw.data_ptr(), m, bases, n)
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ddev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            okt = torch.tensor([1 if got_c == want and (got is None or got_c == got) else 0], dtype=torch.int32, device=ddev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        # BASELINE configs[3] as worded: the WINDOWS split over the ranks, every rank over ALL points (it needs every scalar
        # and base resident: only run where that fits this script's budget), RCCL all-gather of the window partials
        okw = None
        if args.msm_log_n <= 22 and "windows" in legs:
            gall = [torch.Generator(device="cuda").manual_seed(1000 + k) for k in range(world)]
            parts = []
            for k in range(world):
                ak, bk = par.shard_units(n, k, world)
                t = torch.randint(0, 256, (bk - ak, 32), dtype=torch.uint8, device="cuda", generator=gall[k])
                t[:, 31] &= 0x3F
                parts.append(t)
            raw_all = torch.cat(parts).contiguous()
            bases_all = ctx.bases_g1_synthetic(n)
            got_w = ctx.msm_g1_window_split_allgather(comm, raw_all.data_ptr(), n, bases_all)
            okw = torch.tensor([1 if got_w == want else 0], dtype=torch.int32, device=ddev)
            dist.all_reduce(okw, op=dist.ReduceOp.MIN)
            bases_all.free()
            del raw_all
        # the 2-D split (zkmi_msm_g1_split2d_allgather): world = P point groups x Q window ranges; rank g Q + q holds point
        # group g's slice -- generated here from the same per-rank seeds as the 1-D slices it is made of -- and reduces only
        # the windows of range q
        ok2 = None
        Q = args.window_groups or (2 if world % 2 == 0 else 1)
        if world % Q == 0 and Q > 1 and "2d" in legs:
            P = world // Q
            gidx, qidx = rank // Q, rank % Q
            ga, gb = par.shard_units(n, gidx, P)
            parts = []
            for k in range(world):  # the 1-D slices [a_k, b_k) that overlap this group's points, same generators as above
                ak, bk = par.shard_units(n, k, world)
                lo_, hi_ = max(ak, ga), min(bk, gb)
                if lo_ >= hi_:
                    continue
                gk = torch.Generator(device="cuda").manual_seed(1000 + k)
                t = torch.randint(0, 256, (bk - ak, 32), dtype=torch.uint8, device="cuda", generator=gk)
                t[:, 31] &= 0x3F
                parts.append(t[lo_ - ak : hi_ - ak])
            raw_g = torch.cat(parts).contiguous()
            bases_g = ctx.bases_g1_synthetic_range(ga, gb - ga)
            ctx.msm_g1_split2d_allgather(comm, raw_g.data_ptr(), gb - ga, bases_g, n, Q)  # untimed
            barrier()
            t0 = time.perf_counter()
            got_2d = ctx.msm_g1_split2d_allgather(comm, raw_g.data_ptr(), gb - ga, bases_g, n, Q)
            barrier()
            dt2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ddev)
            dist.all_reduce(dt2, op=dist.ReduceOp.MAX)
            ok2 = torch.tensor([1 if got_2d == want else 0], dtype=torch.int32, device=ddev)
            dist.all_reduce(ok2, op=dist.ReduceOp.MIN)
            bases_g.free()
            del raw_g, parts
            if rank == 0:
                print(json.dumps({"config": 3, "workload": "the same MSM as a 2-D split: %d point groups x %d window ranges (zkmi_msm_g1_split2d_allgather)" % (P, Q),
                                  "matches_closed_form_on_every_rank": bool(ok2.item()), "seconds": float(dt2.item()),
                                  "algorithmic_GBps": 128.0 * n / float(dt2.item()) / 1e9, **gpus}), flush=True)
            assert ok2.item() == 1
        comm.free()
        if rank == 0 and okw is not None:
            print(json.dumps({"config": 3, "workload": "the same MSM with its WINDOWS split over the ranks (zkmi_msm_g1_window_split_allgather)",
                              "matches_closed_form_on_every_rank": bool(okw.item()), **gpus}), flush=True)
        assert okw is None or okw.item() == 1
        if rank == 0 and okt is not None:
            print(json.dumps({"config": 3, "workload": "the same MSM through zkmi_msm_g1_allgather_combine (RCCL behind the C ABI)",
                              "matches_closed_form_and_python_path_on_every_rank": bool(okt.item()), "seconds": float(dt.item()),
                              "algorithmic_GBps": 128.0 * n / float(dt.item()) / 1e9, **gpus}), flush=True)
        assert okt is None or okt.item() == 1
        if "prepared" not in legs:
            bases.free()
            ctx.close()
            dist.destroy_process_group()
            return
        # the same split against PREPARED bases (an SRS serves many MSMs: zkmi_bases_g1_prepare once per rank): every
        # rank's share is then one point (shared-bucket schedule), the ranks all-gather 96 bytes each and add
        bases.prepare()
        ctx.msm_g1_dev(raw.data_ptr(), m, bases)  # untimed: the workspaces of the shared-bucket plan are allocated here
        barrier()
        t0 = time.perf_counter()
        mine = ctx.msm_g1_dev(raw.data_ptr(), m, bases)
        parts = par.allgather_bytes(mine)
        got2 = parts[0]
        for p in parts[1:]:
            got2 = z.g1_add(got2, p)
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ddev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        okt = torch.tensor([1 if got2 == want else 0], dtype=torch.int32, device=ddev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(json.dumps({"config": 3, "workload": "the same MSM against prepared bases (digit tables per rank), RCCL all-gather of one point per rank",
                              "matches_closed_form_on_every_rank": bool(okt.item()), "seconds": float(dt.item()),
                              "algorithmic_GBps": 128.0 * n / float(dt.item()) / 1e9, **gpus}), flush=True)
        assert okt.item() == 1
        bases.free()

    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
