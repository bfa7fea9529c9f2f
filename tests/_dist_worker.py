"""Worker for tests/test_distributed_cpu.py: one rank of a gloo group (CPU only).

The product's RCCL collective cannot run without GPUs, so the all-reduce is
played here by torch.distributed/gloo; everything else is the real host logic:
`shard_vectors` decides which start vectors a rank owns and
`free_energy_series` turns the reduced moments into F.  The per-rank moments
come from the CPU oracle.
"""

import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import bodge_amd  # noqa: E402
import systems  # noqa: E402
from bodge_amd import chebyshev  # noqa: E402
from bodge_amd.observables import shard_vectors  # noqa: E402
from oracle import cheb_ref  # noqa: E402


class GlooComm:
    """Stands in for bodge_amd.solver.Communicator (same attributes and methods)."""

    def __init__(self):
        self.rank, self.n_ranks = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, values):
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()


def main():
    out_path, total, moments = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    comm = GlooComm()
    system = systems.swave_square(bodge_amd, L=10, zeeman=0.05)
    bsr = system.matrix("bsr")
    scale = chebyshev.spectral_bound(*[system.bsr_arrays()[k] for k in (0, 2)])
    first, count = shard_vectors(total, comm)
    start = cheb_ref.random_block(bsr.shape[0], 0, range(first, first + count), cheb_ref.VEC_Z4)
    local = cheb_ref.moments(bsr, scale, moments, start).sum(axis=1)
    mu = comm.allreduce_sum(local) / total
    value = chebyshev.free_energy_series(mu, scale, 0.5)
    if comm.rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"free_energy": value, "mu": mu.tolist(), "world": comm.n_ranks}, fh)
    with open(f"{out_path}.rank{comm.rank}", "w") as fh:
        json.dump({"first": first, "count": count}, fh)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
