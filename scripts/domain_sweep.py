#!/usr/bin/env python3
"""Proof rate of the batch entry point (zkmi_groth16_prove_batch_dev) for every domain size 2^lo .. 2^hi, witnesses
resident, first and last proof of each batch verified by pairing.  Usage: python scripts/domain_sweep.py [lo [hi]]
(defaults 12 21).  Output: one line per size; kept under profiles/rNN/domain_sweep.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (relation_and_witness, SplitMix64, small_domain_rate)
import torch  # noqa: E402


def main():
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 21
    z = bench.load_pkg().Zkmi(os.environ.get("ZKMI_LIB"))
    ctx = z.context(0)
    print(f"{'log_n':>5} {'proofs':>7} {'proofs/s':>10} {'ms/proof':>9} {'1-proof latency ms':>19}  verified")
    for lg in range(lo, hi + 1):
        count = max(12, min(1024, (1 << 24) >> lg))
        r = bench.small_domain_rate(z, ctx, "poseidon" if lg >= 13 else "chain", lg, count)
        print(f"{lg:>5} {r['proofs']:>7} {r['proofs_per_s']:>10.1f} {r['ms_per_proof']:>9.3f} {r['single_proof_latency_ms']:>19.2f}  {r['verified_by_pairing']}", flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
