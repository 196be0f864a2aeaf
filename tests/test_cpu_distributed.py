"""world_size-2 gloo tests of the N>1 paths: proof sharding and
the MSM point-split exchange (all-gather of per-window partials + local combine
through the product's zkmi_msm_g1_combine)."""
import os
import socket
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _signed_digits(k, c, nwin):
    out, carry = [], 0
    for w in range(nwin):
        d = ((k >> (c * w)) & ((1 << c) - 1)) + carry
        if d > (1 << (c - 1)):
            d -= 1 << c
            carry = 1
        else:
            carry = 0
        out.append(d)
    assert carry == 0
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zkmi_loader import load_pkg
    from oracle import bls12_381 as ec

    pkg = load_pkg()
    par = __import__("zk_apps_amd.parallel", fromlist=["x"])
    zk = pkg.Zkmi()
    # --- proof sharding
    lo, hi = par.shard_units(7, rank, world)
    spans = par.allgather_bytes(bytes([lo, hi]))
    # --- MSM split: n = 24 points, rank r owns [12 r, 12 r + 12)
    n, c = 24, 8
    nwin = 255 // c + 1
    rng = ec.SplitMix64(99)
    scalars = [rng.fr() for _ in range(n)]
    bases = ec.synthetic_bases_g1(n)
    a, b = par.shard_units(n, rank, world)
    wins = []
    for w in range(nwin):
        acc = None
        for i in range(a, b):
            d = _signed_digits(scalars[i], c, nwin)[w]
            acc = ec.pt_add(ec.Fq, acc, ec.pt_mul(ec.Fq, bases[i], d))
        wins.append(ec.g1_to_bytes(acc))
    got = par.msm_g1_combine_ranks(zk, b"".join(wins), nwin, c)
    exp = ec.g1_to_bytes(ec.msm_naive(ec.Fq, scalars, bases))
    # --- the exchange of the prepared-bases form (scripts/run_multigpu.py config 3, second leg): every rank holds the
    # complete sum over its slice as ONE point; all-gather of 96 bytes per rank, then product-side additions
    mine = ec.g1_to_bytes(ec.msm_naive(ec.Fq, scalars[a:b], bases[a:b]))
    parts = par.allgather_bytes(mine)
    tot = parts[0]
    for p in parts[1:]:
        tot = zk.g1_add(tot, p)
    q.put((rank, spans, got == exp and tot == exp))
    dist.destroy_process_group()


def test_gloo_world2_shard_and_msm_exchange():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, spans, ok in res:
        assert ok, f"rank {rank}: combined MSM differs from the oracle"
        assert [tuple(s) for s in spans] == [(0, 4), (4, 7)]
