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


def slab_main(out_path, total, moments):
    """Row slabs: every rank advances all vectors on its own rows; halo rows of t_n travel by
    gloo send/recv following the product's `SlabPlan`; local products are scipy matvecs."""
    import scipy.sparse as sp

    from bodge_amd import slab

    comm = GlooComm()
    system = systems.random_periodic(bodge_amd, shape=(6, 4, 3), seed=5)  # periodic: wrap-around halos
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    bounds = slab.partition_rows(system.lattice.size, comm.n_ranks, slab.lattice_granule(system.lattice))
    plan = slab.build_plan(indptr, indices, data, bounds, comm.rank)
    local = sp.bsr_matrix((plan.data, plan.indices, plan.indptr), shape=(4 * plan.n_own, 4 * plan.n_cols))
    full = cheb_ref.random_block(system.shape[0], 0, range(total), cheb_ref.VEC_Z4).reshape(-1, 4, total)

    def exchange(vec):  # vec: (n_cols, 4, R); fill the halo rows from the peers
        sends = []
        for peer, rows in zip(plan.peers, plan.send_rows):
            sends.append(dist.isend(torch.from_numpy(np.ascontiguousarray(vec[rows]).view(np.float64)), peer))
        for peer, off, count in zip(plan.peers, plan.recv_offset, plan.recv_count):
            buf = torch.empty((count, 4, 2 * total), dtype=torch.float64)
            dist.recv(buf, peer)
            vec[off : off + count] = buf.numpy().view(np.complex128)
        for req in sends:
            req.wait()

    def apply(vec):
        exchange(vec)
        out = np.zeros_like(vec)
        out[: plan.n_own] = (local @ vec.reshape(-1, total)).reshape(plan.n_own, 4, total)
        return out

    own = slice(0, plan.n_own)
    t_prev = np.zeros((plan.n_cols, 4, total), dtype=complex)
    t_prev[own] = full[plan.row0 : plan.row1]
    t_cur = apply(t_prev) / scale
    steps = moments // 2
    d, e = np.empty((steps, total)), np.empty((steps, total))
    dot = lambda a, b: np.einsum("iar,iar->r", a[own].conj(), b[own]).real
    d[0], e[0] = dot(t_prev, t_prev), dot(t_cur, t_prev)
    for n in range(1, steps):
        t_next = apply(t_cur) * (2.0 / scale) - t_prev
        d[n], e[n] = dot(t_cur, t_cur), dot(t_next, t_cur)
        t_prev, t_cur = t_cur, t_next
    mu = comm.allreduce_sum(chebyshev.dots_to_moments(d, e).sum(axis=1)) / total
    if comm.rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"mu": mu.tolist(), "world": comm.n_ranks, "scale": scale}, fh)
    with open(f"{out_path}.rank{comm.rank}", "w") as fh:
        json.dump({"rows": [plan.row0, plan.row1], "halo": plan.halo_rows, "peers": plan.peers}, fh)


def main():
    out_path, total, moments = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    if len(sys.argv) > 4 and sys.argv[4] == "slab":
        slab_main(out_path, total, moments)
        dist.destroy_process_group()
        return
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
