"""Multi-GPU host layer.  One process per GPU; torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

* Independent proofs: `shard_units` — no data-path collective at all.
* One large MSM split by points (BASELINE config 3): each rank runs the bucket
  method over its slice and emits one partial sum per window; the ranks
  all-gather these nwin x 96 B (a few KiB, latency-bound — EC addition is not an
  RCCL reduction op, so the "all-reduce of partial bucket sums" is an all-gather
  of raw bytes followed by a local EC sum + Horner combine on every rank).
"""
import torch
import torch.distributed as dist


def shard_units(total, rank, world):
    """Contiguous block partition of `total` independent units (proofs)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allgather_bytes(payload: bytes, device=None, group=None):
    """All-gather equally sized byte strings; returns the list indexed by rank."""
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in out]


def msm_g1_combine_ranks(zk, windows_local: bytes, nwin: int, window_bits: int, group=None):
    """Exchange per-window partial sums and combine locally on every rank."""
    parts = allgather_bytes(windows_local, group=group)
    return zk.msm_g1_combine(b"".join(parts), len(parts), nwin, window_bits)


def msm_g1_split_dev(zk, ctx, d_scalars_ptr, n_local, bases_local, n_global, group=None):
    """Rank-local slice of a point-split MSM on the GPU + the exchange step (host bytes through torch.distributed:
    what the gloo CPU test exercises; on GPUs prefer rccl_comm + ctx.msm_g1_allgather_combine below)."""
    windows, nwin, cbits = ctx.msm_g1_windows_dev(d_scalars_ptr, n_local, bases_local, n_global)
    return msm_g1_combine_ranks(zk, windows, nwin, cbits, group=group)


def rccl_comm(zk, ctx, group=None):
    """A zkmi_comm (RCCL communicator behind the C ABI, include/zkmi.h) spanning the ranks of `group`: rank 0 draws the
    unique id, torch.distributed only carries its 128 bytes; the MSM exchange itself never touches Python --
    ctx.msm_g1_allgather_combine(comm, ...) gathers the device-resident partial sums with ncclAllGather and combines."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [zk.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return ctx.comm_init(world, rank, box[0])
