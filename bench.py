#!/usr/bin/env python3
"""Headline benchmark: Chebyshev recurrence steps/s on the 4N x 4N BdG Hamiltonian.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], the configuration the roofline target is
quoted on): CubicLattice((1000,1000,1)), H_ii = 3σ0 - 0.05σ3, Δ_ii = -0.1 iσ2,
bonds -σ0, open boundaries -> 4M x 4M complex128 BSR matrix (1.28 GB).  Every
GPU holds the whole matrix and advances its own 8 stochastic-trace vectors
(weak scaling); one "step" is one launch of the fused recurrence kernel
t_{n+1} = (2/a) H t_n - t_{n-1} (+ both dot products) on those vectors.  The
timed region is one whole moment calculation of 2K moments - start-vector
generation, K launches, the dot reductions, and for N > 1 the RCCL all-reduce
of the moment vector - with the matrix already resident in HBM.

`value` = (N x vectors-per-GPU x K) / wall seconds = vector-steps per second.
`roofline.achieved` = algorithmic bytes of one launch / mean launch time from
HIP events on the library's stream.  `cpu_baseline` times the C + OpenMP
restatement (oracle/cheb_c.c) on this host on a bounded sample, with the
single-core scipy.sparse restatement beside it.

For N > 1 the driver starts one process per GPU with torch.distributed.run;
only its environment variables are used (RANK, LOCAL_RANK, WORLD_SIZE,
MASTER_PORT) - the collective is RCCL through the C ABI, not torch.  Only if the RCCL
communicator cannot be created do the ranks fall back to torch.distributed (gloo) for the three
scalar reductions of this script; config.collective says which one ran.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_system(shape, model="swave", zeeman=0.05, gap=0.1, mu=3.0):
    """Synthetic Hamiltonians of SURVEY §8d: "swave" = README model (+Zeeman), "dwave" = config 4."""
    import bodge_amd as ba

    lattice = ba.CubicLattice(tuple(shape))
    system = ba.Hamiltonian(lattice)
    with system as (H, Δ):
        if model == "swave":
            H.set_sites(mu * ba.σ0 - zeeman * ba.σ3)
            Δ.set_sites(-gap * ba.jσ2)
            H.set_bonds(-1.0 * ba.σ0)
        elif model == "dwave":
            pairs = lattice.bond_array(coords=True)
            H.set_sites(mu * ba.σ0)
            H.set_bonds(-1.0 * ba.σ0)
            Δ.set_bonds(-gap * ba.dwave()(pairs[:, 0], pairs[:, 1]))
        else:
            raise ValueError(model)
    return system



# BASELINE.json's metric, verbatim
BASELINE_METRIC = "Chebyshev SpMV steps/s + achieved HBM GB/s, 4N\u00d74N BdG H; free_energy wall-time"


class GlooReductions:
    """Stand-in for the RCCL communicator in the bench's three reductions (sum of the moment
    vector, max of the elapsed time, barrier), over torch.distributed's gloo backend.  Used only
    if the RCCL communicator cannot be created; the JSON line then says so in config.collective."""

    def __init__(self):
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        if not dist.is_initialized():
            from bodge_amd.solver import _StdoutToStderr

            with _StdoutToStderr():  # gloo announces its peers on stdout; stdout carries the JSON line only
                dist.init_process_group("gloo", init_method="env://")
                dist.barrier()

    def _reduce(self, values, op):
        t = self._torch.from_numpy(np.array(values, dtype=np.float64))
        self._dist.all_reduce(t, op=op)
        return t.numpy()

    def allreduce_sum(self, values):
        return self._reduce(values, self._dist.ReduceOp.SUM)

    def allreduce_max(self, values):
        return self._reduce(values, self._dist.ReduceOp.MAX)

    def barrier(self):
        self._dist.barrier()

    def finish(self):
        """torch brings its own ROCm runtime next to the one libbodge_hip.so loaded; their exit-time
        destructors collide (double free at interpreter shutdown), so leave without running them."""
        self._dist.barrier()
        self._dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def make_communicator(communicator_cls, world: int, rank: int, mode: str):
    """(communicator or None, description).  RCCL through the library; if that raises on this rank
    the ranks agree through status files (a rank cannot fall back alone) and use gloo instead."""
    if world <= 1:
        return communicator_cls.from_environment(), "none (single process)"
    tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    base = os.path.join(os.environ.get("BODGE_AMD_RDZV_DIR", "/tmp"), f"bodge_amd_bench_{tag}")
    comm, error = None, ""
    try:
        comm = communicator_cls.from_environment()
    except Exception as exc:  # noqa: BLE001 - any failure means "no RCCL on this rank"
        error = f"{type(exc).__name__}: {exc}"
    import atexit

    own = f"{base}_rank{rank}.status"
    with open(f"{own}.tmp", "w") as fh:
        fh.write("ok" if comm is not None else error or "failed")
    os.replace(f"{own}.tmp", own)
    atexit.register(lambda: os.path.exists(own) and os.unlink(own))
    deadline = time.time() + 300
    states = []
    for r in range(world):
        path = f"{base}_rank{r}.status"
        while not (os.path.exists(path) and os.path.getsize(path) > 0):
            if time.time() > deadline:
                sys.exit(f"rank {rank}: no communicator status from rank {r}")
            time.sleep(0.05)
        with open(path) as fh:
            states.append(fh.read())
    if all(state == "ok" for state in states):
        return comm, "rccl (ncclAllReduce of the moments inside bdg_cheb_moments)"
    if mode == "slab":
        sys.exit(f"rank {rank}: slab mode needs RCCL send/recv; communicator states: {states}")
    reason = next(state for state in states if state != "ok")
    return GlooReductions(), f"gloo fallback on the host (RCCL communicator failed: {reason[:200]})"


def measured_traffic(kernel: str, shape, vectors: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py), or None.

    Counters cannot be read from inside the timed run; the table is keyed by kernel
    configuration and workload so a stale entry is never attached to a different kernel.
    """
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        table = json.load(fh)
    entry = table.get(f"{kernel}|{'x'.join(str(v) for v in shape)} R={vectors}")  # swave/vectors-mode entries
    return entry["traffic_bytes_per_launch"] if entry else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="recurrence launches in the timed call (2x moments)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--lattice", default="1000,1000,1")
    ap.add_argument("--vectors-per-gpu", type=int, default=8)
    ap.add_argument("--vector-kind", default="rademacher", choices=["rademacher", "z4"])
    ap.add_argument("--lanes", type=int, default=0, help="override lanes per block row (tuning)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--temperature", type=float, default=0.5)
    ap.add_argument("--model", default="swave", choices=["swave", "dwave"])
    ap.add_argument("--mode", default="vectors", choices=["vectors", "slab"],
                    help="vectors: H replicated, start vectors sharded (weak scaling, headline); "
                         "slab: lattice planes sharded with per-step halo exchange (strong scaling, config 4)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from bodge_amd import backend, build, chebyshev
    from bodge_amd.solver import VEC_RADEMACHER, VEC_Z4, Communicator, DeviceSolver

    if rank == 0 and not os.path.exists(build.LIBRARY):
        build.build_library()
    backend.load()
    backend.require_device()
    comm, collective = make_communicator(Communicator, world, rank, args.mode)

    shape = [int(v) for v in args.lattice.split(",")]
    t0 = time.perf_counter()
    system = build_system(shape, args.model)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    t_build = time.perf_counter() - t0
    device = local % backend.device_count()
    kind = VEC_RADEMACHER if args.vector_kind == "rademacher" else VEC_Z4
    r_local = args.vectors_per_gpu
    if args.mode == "slab":
        # every rank advances the same r_local vectors on its own slab of x-planes
        from bodge_amd import slab

        bounds = slab.partition_rows(system.lattice.size, world, slab.lattice_granule(system.lattice))
        plan = slab.build_plan(indptr, indices, data, bounds, rank)
        solver = DeviceSolver.from_slab_plan(plan, comm=comm, device=device)  # (slab mode needs RCCL)
        first = 0
    else:
        solver = DeviceSolver(indptr, indices, data, device=device)
        solver.set_lattice_shape(shape)
        first = rank * r_local
    if args.lanes:
        solver.set_lanes_per_row(args.lanes)

    def run(steps):
        if isinstance(comm, GlooReductions):  # RCCL unavailable: local moments, summed on the host
            return comm.allreduce_sum(
                solver.moments_random(scale, 2 * steps, r_local, seed=0, first_id=first, kind=kind, comm=None))
        return solver.moments_random(scale, 2 * steps, r_local, seed=0, first_id=first, kind=kind, comm=comm)

    if args.warmup > 0:
        run(args.warmup)
    if comm is not None:
        comm.barrier()
    t0 = time.perf_counter()
    mu = run(args.steps)
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    perf = solver.perf()

    # Further passes with optimisations switched off, reported beside the headline and never as
    # `value`: blocks streamed from HBM instead of the LDS dictionary, and complex arithmetic.
    def alternative(env):
        for k, v in env.items():
            os.environ[k] = v
        if comm is not None:
            comm.barrier()
        t0 = time.perf_counter()
        run(args.steps)
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = float(comm.allreduce_max(np.array([dt]))[0])
        pf = solver.perf()
        for k in env:
            del os.environ[k]
        launch = pf["kernel_ms"] / max(1, pf["launches"])
        return {
            "value": r_local * (1 if args.mode == "slab" else args.gpus) * args.steps / dt,
            "unit": "steps/s",
            "launch_ms": launch,
            "bytes_per_launch": pf["bytes_per_launch"],
            "achieved_GBps": pf["bytes_per_launch"] / (launch * 1e-3) / 1e9,
            "frac": pf["bytes_per_launch"] / (launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }

    streamed_pass = alternative({"BODGE_AMD_DICT": "0"}) if perf["dict_blocks"] else None
    complex_pass = alternative({"BODGE_AMD_DICT": "0", "BODGE_AMD_REAL": "0"}) if perf["real_arithmetic"] else None

    if rank != 0:
        if isinstance(comm, GlooReductions):
            comm.finish()
        return

    total_vectors = r_local * (1 if args.mode == "slab" else args.gpus)
    value = total_vectors * args.steps / elapsed
    launch_ms = perf["kernel_ms"] / max(1, perf["launches"])
    achieved = perf["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9
    free_energy = chebyshev.free_energy_series(mu / total_vectors, scale, args.temperature)

    kernel_name = ("cheb_step_dict" if perf["dict_blocks"] else
                   "cheb_step_pipelined" if perf["pipelined"] else "cheb_step") + (
        f"<{'Real' if perf['real_arithmetic'] else 'Complex'}{'PH' if perf['ph_packed'] else ''}Mode,"
        f"{perf['lanes_per_row']}>")
    record = {
        "metric": BASELINE_METRIC,
        "metric_detail": "value = Chebyshev vector-steps/s (one step = t_{n+1} = 2 H t_n / a - t_{n-1} on one vector, fused "
                         "with the two dot products); achieved HBM GB/s in roofline.achieved; wall time in free_energy_wall_s",
        "value": value,
        "unit": "steps/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.mode == "slab" else "weak",
        "vs_baseline": None,
        "dtype": "f64" if perf["real_arithmetic"] else "c128",
        "data": "synthetic",
        "config": {
            "workload": f"CubicLattice({tuple(shape)}) {args.model}, {2 * args.steps}-moment stochastic-trace "
                        + (f"free_energy, {r_local} vectors, x-plane slabs with halo exchange" if args.mode == "slab"
                           else f"free_energy, {r_local} vectors/GPU, H replicated, vectors sharded"),
            "n_sites": int(system.lattice.size),
            "nnzb": int(indices.size),
            "moments": 2 * args.steps,
            "vectors_per_gpu": r_local,
            "vector_kind": args.vector_kind,
            "spectral_scale": scale,
            "parallelism": f"{args.mode} x{args.gpus}",
            "collective": collective,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(kernel_name, shape, r_local),
            "kernel": kernel_name,
            "launch_ms": launch_ms,
            "bytes_per_launch": perf["bytes_per_launch"],
            "grid": perf["grid"],
            "lds_bytes": perf["lds_bytes"],
            "strip_rows": perf["strip_rows"],
            "distinct_blocks": perf["dict_blocks"],
        },
        "free_energy_wall_s": elapsed,
        "free_energy_estimate": free_energy,
        "host_assembly_s": t_build,
        "streamed_blocks_kernels": streamed_pass,
        "complex128_kernels": complex_pass,
    }

    if args.cpu_seconds > 0 and args.gpus == 1:
        from oracle import cheb_c, cheb_ref

        bsr = system.matrix("bsr")
        logical = os.cpu_count() or 2
        short = max(3.0, args.cpu_seconds / 3)
        real = bool(perf["real_arithmetic"])
        start = cheb_ref.random_block(bsr.shape[0], 0, range(r_local), kind)
        # headline CPU number: the C + OpenMP restatement in the arithmetic the GPU headline used
        # (real when imag(H) = 0), at the better of two thread counts - so the CPU side is not
        # handicapped by interpreter overhead, storage format, dtype or thread count
        best = None
        for threads in sorted({min(16, logical), min(64, logical)}):
            cheb_c.set_threads(threads)
            rate, steps, _ = cheb_c.time_recurrence(bsr, scale, start, seconds=short, real=real)
            if best is None or rate > best[0]:
                best = (rate, steps, threads)
        extra = {}
        if real:
            cheb_c.set_threads(best[2])
            extra["c_openmp_complex128_steps_per_s"] = cheb_c.time_recurrence(bsr, scale, start, seconds=short)[0]
        # the numpy/scipy restatement the parity tests use, one core, and its variants
        extra["scipy_bsr_1core_steps_per_s"] = cheb_ref.time_recurrence(bsr, scale, r_local, seconds=short, kind=kind)[0]
        extra["scipy_csr_1core_steps_per_s"] = cheb_ref.time_recurrence(bsr, scale, r_local, seconds=short, kind=kind, fmt="csr")[0]
        if real:
            extra["scipy_csr_real_1core_steps_per_s"] = cheb_ref.time_recurrence(
                bsr, scale, r_local, seconds=short, kind=kind, fmt="csr", real=True)[0]
        record["cpu_baseline"] = {
            "value": best[0],
            "unit": "steps/s",
            "cores": best[2],
            "kind": "port",
            "sample": f"{best[1]} timed block-steps of the same {r_local} vectors on the same H: C + OpenMP "
                      f"restatement (oracle/cheb_c.c), {'float64' if real else 'complex128'} arithmetic, "
                      f"{best[2]} threads (host has {logical} logical cores)",
            "other_cpu_variants": extra,
        }
    else:
        record["cpu_baseline"] = None
    print(json.dumps(record), flush=True)
    if isinstance(comm, GlooReductions):
        comm.finish()


if __name__ == "__main__":
    main()
