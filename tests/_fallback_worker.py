"""Rank body for test_bench_falls_back_to_gloo_when_rccl_is_unavailable (launched by torch.distributed.run)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class NoRccl:
    @classmethod
    def from_environment(cls):
        raise RuntimeError("bodge_hip: ncclCommInitRank failed: simulated")


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    strict = len(sys.argv) > 2 and sys.argv[2] == "strict"
    comm, description = bench.make_communicator(NoRccl, world, rank, "vectors", allow_gloo=not strict)
    assert isinstance(comm, bench.HostReductions)
    comm.barrier()
    total = comm.allreduce_sum(np.arange(4.0) + rank)
    biggest = comm.allreduce_max(np.array([float(rank)]))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump({"sum": total.tolist(), "max": biggest.tolist(), "description": description}, fh)
    from bodge_amd.rendezvous import store_from_environment

    store_from_environment().finish()


if __name__ == "__main__":
    main()
