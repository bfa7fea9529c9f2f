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
`roofline.achieved` = algorithmic bytes of the launches in the timed call / the
HIP-event time from the first of them to the end of the last (`window_ms`; the
call's two lane groups run side by side on two streams of the library, so two
launches are in flight at a time and each lasts `launch_ms`).  The bytes are what each launch has
to move (`bdg_perf.bytes_moved`): a full launch reads two buffers and writes two
(`bytes_full_launch`), the first sweep of a run reads none (t_0 is generated in
registers, t_{-1} = 0) and the last launch stores none (nothing reads the vectors
of the last step; the call returns moments), so short runs average less per launch
(`bytes_per_launch`).  `roofline.effective_GBps` prices the same time at a full
launch's bytes per step.  `cpu_baseline` times the C + OpenMP
restatement (oracle/cheb_c.c) on this host on a bounded sample, with the
single-core scipy.sparse restatement beside it.

For N > 1 the driver starts one process per GPU with torch.distributed.run;
only its environment variables are used (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT) - the collective is RCCL through the C ABI, not torch.  If the RCCL communicator
cannot be created the run FAILS (exit code 3): a scaling number that did not go through RCCL is not
a result.  `--allow-gloo` (rehearsals on a box with fewer GPUs than ranks) lets the ranks carry the
three scalar reductions of this script over torch.distributed's gloo backend instead;
config.collective and config.rccl_ranks always say what actually ran.

The default kernel on this workload is the three-steps-per-sweep form (bodge_amd/csrc/sweep.hpp,
cheb_sweep3): one launch advances every vector by THREE recurrence steps, so K steps are about
K/3 launches and `roofline` is per launch of that kernel; the two-step and one-step kernels are
timed beside it (`two_step_kernels`, `one_step_kernels`, and with the block dictionary and / or
real arithmetic switched off `streamed_blocks_one_step_kernels`, `complex128_one_step_kernels`,
`complex128_sweep_kernels`).  Two more matrices of the same size are assembled and timed at N = 1:
a random on-site potential and gap amplitude (`streamed_blocks_kernels`: 10^6 distinct diagonal
blocks, real) and an exchange field of varying direction (`complex128_kernels`: the same, complex
blocks) - position-dependent Hamiltonians, which take the three-step sweep that streams the
on-site blocks.  None of these is ever `value`.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# ROCm serves the HIP streams of a process from four hardware queues by default; a fifth stream shares one.  A rank of a
# multi-GPU run has the null stream, the library's main and side stream (the two lane groups of the call side by side) and
# RCCL's own: so that main and side never end up on ONE queue (which serialises the lane groups: profiles/r04_stream_pool.log
# shows that loss for a second handle), ranks of such a run ask for eight before the runtime starts.  At N = 1 the setting
# changes nothing (scratch/r4_hwq.sh: 105.9-106.0 against 104.2-106.2 k) and is not made.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_system(shape, model="swave", zeeman=0.05, gap=0.1, mu=3.0, seed=11):
    """Synthetic Hamiltonians of SURVEY §8d: "swave" = README model (+Zeeman), "dwave" = config 4;
    "potential" / "texture" = the s-wave model with position-dependent on-site terms (a random
    potential and gap amplitude: real; an exchange field of varying direction: complex blocks) -
    10^6 distinct diagonal blocks, what per-site fills (ref hamiltonian.py:102-118) make of it; "ssd" = every term of the
    s-wave model scaled by the reference's sine-squared envelope (bond blocks position dependent too); "peierls" =
    the s-wave model with a uniform phase on the x bonds (translation invariant, genuinely complex).
    The Hermiticity test of the closing `with` block is made on the host: the process must not touch
    the GPU before the CPU baseline has forked its workers."""
    import bodge_amd as ba

    lattice = ba.CubicLattice(tuple(shape))
    system = ba.Hamiltonian(lattice)
    system.hermiticity_check = "host"
    rng = np.random.default_rng(seed)
    sites = lattice.size
    with system as (H, Δ):
        if model == "swave":
            H.set_sites(mu * ba.σ0 - zeeman * ba.σ3)
            Δ.set_sites(-gap * ba.jσ2)
            H.set_bonds(-1.0 * ba.σ0)
        elif model == "potential":
            H.set_sites((mu + rng.uniform(-0.5, 0.5, sites))[:, None, None] * ba.σ0 - zeeman * ba.σ3)
            Δ.set_sites(-rng.uniform(0.5 * gap, 1.5 * gap, sites)[:, None, None] * ba.jσ2)
            H.set_bonds(-1.0 * ba.σ0)
        elif model == "texture":
            th, ph = rng.uniform(0, np.pi, sites)[:, None, None], rng.uniform(0, 2 * np.pi, sites)[:, None, None]
            H.set_sites(mu * ba.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * ba.σ1 + np.sin(th) * np.sin(ph) * ba.σ2 + np.cos(th) * ba.σ3))
            Δ.set_sites(-gap * ba.jσ2)
            H.set_bonds(-1.0 * ba.σ0)
        elif model == "ssd":  # the reference's sine-squared deformation (ref hamiltonian.py:488-531) of the s-wave model
            φ = ba.ssd(system)
            coords = np.stack(np.unravel_index(np.arange(sites), tuple(shape)), axis=-1)
            pairs = lattice.bond_array(coords=True)
            on_site, on_bond = φ(coords, coords)[:, None, None], φ(pairs[:, 0], pairs[:, 1])[:, None, None]
            H.set_sites(on_site * (mu * ba.σ0 - zeeman * ba.σ3))
            Δ.set_sites(-gap * on_site * ba.jσ2)
            H.set_bonds(-on_bond * ba.σ0)
        elif model == "peierls":
            pairs = lattice.bond_array(axis=0, coords=True)  # directed x bonds: a phase one way, its conjugate back
            phase = np.where(pairs[:, 1, 0] > pairs[:, 0, 0], np.exp(0.3j), np.exp(-0.3j))
            H.set_sites(mu * ba.σ0 - zeeman * ba.σ3)
            Δ.set_sites(-gap * ba.jσ2)
            H.set_bonds(-phase[:, None, None] * ba.σ0, axis=0)
            for axis in (1, 2):
                if shape[axis] > 1:
                    H.set_bonds(-1.0 * ba.σ0, axis=axis)
        elif model == "landau":  # the texture model in a magnetic field: Landau-gauge Peierls phases exp(±i B y) on the x bonds
            th, ph = rng.uniform(0, np.pi, sites)[:, None, None], rng.uniform(0, 2 * np.pi, sites)[:, None, None]
            H.set_sites(mu * ba.σ0 - 0.3 * (np.sin(th) * np.cos(ph) * ba.σ1 + np.sin(th) * np.sin(ph) * ba.σ2 + np.cos(th) * ba.σ3))
            Δ.set_sites(-gap * ba.jσ2)
            pairs = lattice.bond_array(axis=0, coords=True)
            flux = 0.0123 * pairs[:, 0, 1]
            phase = np.exp(1j * np.where(pairs[:, 1, 0] > pairs[:, 0, 0], flux, -flux))
            H.set_bonds(-phase[:, None, None] * ba.σ0, axis=0)
            for axis in (1, 2):
                if shape[axis] > 1:
                    H.set_bonds(-1.0 * ba.σ0, axis=axis)
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


class HostReductions:
    """Stand-in for the RCCL communicator in this script's three reductions (sum of the moment
    vector, max of the elapsed time, barrier), carried by the launch's TCP rendezvous store on the
    host.  Only reachable with --allow-gloo when the RCCL communicator cannot be created (a
    rehearsal of the launch path with more ranks than GPUs); the JSON line then says so in
    config.collective and reports rccl_ranks = 0, and no credit is claimed for such a number."""

    def __init__(self, store):
        self._store = store

    def _gather(self, values):
        arr = np.ascontiguousarray(values, dtype=np.float64)
        return np.stack([np.frombuffer(blob, dtype=np.float64) for blob in self._store.gather(arr.tobytes())])

    def allreduce_sum(self, values):
        return self._gather(values).sum(axis=0)

    def allreduce_max(self, values):
        return self._gather(values).max(axis=0)

    def barrier(self):
        self._store.barrier()


def make_communicator(communicator_cls, world: int, rank: int, mode: str, allow_gloo: bool = False, store=None):
    """(communicator or None, description).  RCCL through the library.  If that fails on any rank
    the whole run exits non-zero, unless `allow_gloo`: then the ranks agree through the rendezvous
    store (a rank cannot fall back alone) and carry the script's reductions over the host."""
    if world <= 1:
        return communicator_cls.from_environment(), "none (single process)"
    comm, error = None, ""
    try:
        comm = communicator_cls.from_environment()
    except Exception as exc:  # noqa: BLE001 - any failure means "no RCCL on this rank"
        error = f"{type(exc).__name__}: {exc}"
    if store is None:
        from bodge_amd.rendezvous import store_from_environment

        store = store_from_environment()
    states = [blob.decode("utf-8", "replace") for blob in store.gather(("ok" if comm is not None else (error or "failed")).encode())]
    if all(state == "ok" for state in states):
        return comm, "rccl (ncclAllReduce of the moments inside bdg_cheb_moments)"
    reason = next(state for state in states if state != "ok")
    if not allow_gloo or mode == "slab":
        print(f"rank {rank}: the RCCL communicator could not be created on every rank ({reason[:300]}); no number "
              "is reported (--allow-gloo rehearses the launch path with host-side reductions)", file=sys.stderr, flush=True)
        store.finish(10.0)
        sys.exit(3)
    return HostReductions(store), f"host fallback over the TCP rendezvous store (RCCL communicator failed: {reason[:200]})"


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


def cpu_baseline(system, scale, r_local, kind, seconds):
    """The CPU restatements timed on this host (rank 0, N = 1 only), BEFORE the process touches the
    GPU: the whole-host scipy number forks worker processes, which must not inherit a live HIP
    runtime.  Headline = C + OpenMP restatement (oracle/cheb_c.c) in the arithmetic the GPU
    headline uses (float64 when imag(H) = 0 and the vectors are real), at the better of two thread
    counts - so the CPU side is not handicapped by interpreter overhead, storage format, dtype or
    thread count."""
    from oracle import cheb_c, cheb_ref

    bsr = system.matrix("bsr")
    logical = os.cpu_count() or 2
    usable = cpu_share()
    short = max(2.0, seconds / 4)
    real = (not bsr.data.imag.any()) and kind == cheb_ref.VEC_RADEMACHER
    start = cheb_ref.random_block(bsr.shape[0], 0, range(r_local), kind)
    # thread sweep; matrix and vectors placed by first touch from the compute threads (NUMA)
    sweep = {}
    best = None
    # (counts up to the CPUs this process may keep busy; two below the quota as well: a quota is CPU time, and the
    # interpreter's own thread counts against it)
    # (counts up to the CPUs this process may keep busy; two below the quota as well: a quota is CPU time, and the
    # interpreter's own thread counts against it.  Each count with the threads bound one per physical core, spread
    # over the cores of ONE memory node (cheb_c.spread_cpus: 16 threads over both sockets of a GPU box run at 178
    # steps/s, over one node at 1070-1080, profiles/r03_numa_probe.log), and unbound (where the scheduler put them:
    # anything between 130 and 1100 steps/s, profiles/r03_cpu_probe.log) - the baseline is the best of the lot.)
    def thread_sweep(tag):
        nonlocal best
        for threads in sorted({max(1, min(t, usable)) for t in (16, 64, 128, usable, usable - 2)}):
            cheb_c.set_threads(threads)
            for pin in (True, False):
                rate, steps, _ = cheb_c.time_recurrence(bsr, scale, start, seconds=max(1.5, short / 3), real=real, numa=True, pin=pin)
                sweep[f"{threads}{' bound' if pin else ''}{tag}"] = rate
                if best is None or rate > best[0]:
                    best = (rate, steps, threads, pin)

    thread_sweep("")
    gbps = best[0] / r_local * cheb_c.step_bytes(bsr, r_local, real) / 1e9
    extra = {"c_openmp_thread_sweep_steps_per_s": sweep, "c_openmp_achieved_GBps": gbps}
    if real:
        cheb_c.set_threads(best[2])
        extra["c_openmp_complex128_steps_per_s"] = cheb_c.time_recurrence(bsr, scale, start, seconds=short / 2, numa=True, pin=best[3])[0]
    # the numpy/scipy restatement the parity tests use: one core, its variants, and the whole host
    # (P forked workers with one vector each on the shared matrix, P = physical cores, BASELINE.md §5 ii)
    extra["scipy_bsr_1core_steps_per_s"] = cheb_ref.time_recurrence(bsr, scale, r_local, seconds=short / 2, kind=kind)[0]
    extra["scipy_csr_1core_steps_per_s"] = cheb_ref.time_recurrence(bsr, scale, r_local, seconds=short / 2, kind=kind, fmt="csr")[0]
    if real:
        extra["scipy_csr_real_1core_steps_per_s"] = cheb_ref.time_recurrence(
            bsr, scale, r_local, seconds=short / 2, kind=kind, fmt="csr", real=True)[0]
    workers = max(1, min(physical_cores(usable), usable, 128))
    extra["scipy_bsr_whole_host_steps_per_s"] = cheb_ref.time_recurrence_processes(bsr, scale, 1, workers, seconds=short)
    extra["scipy_bsr_whole_host_processes"] = workers
    # the same thread sweep once more, half a minute later: the host is shared, and a whole sweep has come out five times
    # slower than the next one (200 against 1045 steps/s on two boxes of the same kind)
    thread_sweep(" (second round)")
    gbps = best[0] / r_local * cheb_c.step_bytes(bsr, r_local, real) / 1e9
    extra["c_openmp_achieved_GBps"] = gbps
    return {
        "value": best[0],
        "unit": "steps/s",
        "cores": best[2],
        "kind": "port",
        "sample": f"{best[1]} timed block-steps of the same {r_local} vectors on the same H: C + OpenMP "
                  f"restatement (oracle/cheb_c.c), {'float64' if real else 'complex128'} arithmetic, best of "
                  f"{list(sweep)} thread configurations = {best[2]} threads{', bound one per physical core, spread over the cores of one memory node' if best[3] else ', unbound'} "
                  f"(host: {logical} logical cores, {usable} usable by this process - affinity mask and cgroup quota; pages of "
                  f"matrix and vectors placed by first touch from the compute threads; {gbps:.0f} GB/s of algorithmic traffic)",
        "other_cpu_variants": extra,
    }


def user_facing_wall_times(system, shape):
    """Wall times of what a user of the reference calls (ref tutorial.qmd:76-79, 131-134; hamiltonian.py:254, :173, :324),
    through `bodge_amd`'s Python API with the matrix already assembled: the 512-moment, 64-vector stochastic free energy of
    the headline matrix (first call = with the upload of the device copy, then repeated), `diagonalize()` of the 30x30 and
    50x50 rungs of config 5's ladder with the largest deviation from the reference's own eigenvalues (tests/golden, made by
    tests/golden/make_golden.py from the reference), and an `ldos()` sweep of 13 energies on a 64x64 lattice."""
    import warnings

    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        call = dict(method="chebyshev", moments=512, vectors=64, trace="stochastic")
        t0 = time.perf_counter()
        first = system.free_energy(0.5, **call)
        t1 = time.perf_counter()
        again = system.free_energy(0.5, **call)
        t2 = time.perf_counter()
        out["free_energy_512x64_wall_s"] = {"first_call": t1 - t0, "repeated": t2 - t1, "value": again, "same_value": first == again,
                                            "workload": f"system.free_energy(0.5, method='chebyshev', moments=512, vectors=64, trace='stochastic') on CubicLattice({tuple(shape)})"}
        for mirror in system._devices.values():  # (the 1.3 GB device copy of the headline matrix is not needed any more)
            mirror.close()
        system._devices = {}
        golden = None
        path = os.path.join(ROOT, "tests", "golden", "reference_arrays.npz")
        if os.path.exists(path):
            golden = np.load(path)
        out["diagonalize_wall_s"] = {}
        for L in (30, 50):
            small = build_system((L, L, 1), "swave")
            t0 = time.perf_counter()
            energies, states = small.diagonalize()
            t1 = time.perf_counter()
            small.diagonalize()
            t2 = time.perf_counter()
            entry = {"first_call": t1 - t0, "repeated": t2 - t1, "n": 4 * L * L, "eigenpairs": int(len(energies)),
                     "states_shape": list(states.shape)}
            key = f"swave{L}_zeeman/eigenvalues"
            if golden is not None and key in golden.files:
                ref = np.sort(golden[key])
                ref = ref[ref > 0] if len(ref) != len(energies) else ref
                if len(ref) == len(energies):
                    entry["max_abs_deviation_from_reference"] = float(np.abs(np.sort(energies) - ref).max())
            out["diagonalize_wall_s"][f"({L},{L},1)"] = entry
        lattice64 = build_system((64, 64, 1), "swave", gap=0.2)
        energies13 = list(np.linspace(-0.3, 0.3, 13))
        t0 = time.perf_counter()
        rho = lattice64.ldos((32, 32, 0), energies13)
        t1 = time.perf_counter()
        lattice64.ldos((32, 32, 0), energies13)
        t2 = time.perf_counter()
        out["ldos_wall_s"] = {"first_call": t1 - t0, "repeated": t2 - t1, "energies": 13, "lattice": "(64,64,1)",
                              "min_density": float(np.min(rho)), "max_density": float(np.max(rho))}
    return out


def cpu_share() -> int:
    """CPUs this process can actually keep busy: its affinity mask, cut by the cgroup's CPU quota
    (a container on a 256-thread host may own 16 of them; threads beyond the quota are throttled,
    measured 28 instead of 1065 steps/s with 256 threads on such a box)."""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                fields = fh.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    usable = min(usable, max(1, int(float(fields[0]) / float(fields[1]) + 0.5)))
            else:
                quota = int(fields[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = int(fh.read())
                if quota > 0:
                    usable = min(usable, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return usable


def physical_cores(usable: int) -> int:
    """Distinct (package, core) pairs among the CPUs this process may use (SMT siblings count once)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for cpu in cpus:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            with open(base + "physical_package_id") as a, open(base + "core_id") as b:
                seen.add((a.read().strip(), b.read().strip()))
        return max(1, len(seen))
    except OSError:
        return max(1, usable // 2)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="recurrence steps in the timed call (2x moments)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--lattice", default="1000,1000,1")
    ap.add_argument("--vectors-per-gpu", type=int, default=8)
    ap.add_argument("--vector-kind", default="rademacher", choices=["rademacher", "z4"])
    ap.add_argument("--lanes", type=int, default=0, help="override lanes per block row (tuning)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--temperature", type=float, default=0.5)
    ap.add_argument("--user-calls", type=int, default=1, help="0 skips the wall times of free_energy / diagonalize / ldos through the Python API")
    ap.add_argument("--model", default="swave", choices=["swave", "dwave", "potential", "texture", "ssd", "peierls", "landau"])
    ap.add_argument("--mode", default="vectors", choices=["vectors", "slab"],
                    help="vectors: H replicated, start vectors sharded (weak scaling, headline); "
                         "slab: lattice planes sharded with per-step halo exchange (strong scaling, config 4)")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="if RCCL cannot be initialised, do the script's reductions on the host instead of failing "
                         "(rehearsal of the launch path only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    from bodge_amd import backend, build, chebyshev, solver as solver_module
    from bodge_amd.solver import VEC_RADEMACHER, VEC_Z4, Communicator, DeviceSolver

    # rank 0 (re)builds a missing or stale library; the others wait for a current one
    store = None
    if world > 1:
        from bodge_amd.rendezvous import store_from_environment

        store = store_from_environment()
    if rank == 0:
        build.build_library()
    if store is not None:
        store.barrier()
    usable = cpu_share()
    t_rccl = time.perf_counter()
    if world > 1:
        # The first RCCL call reads a 573 MB shared object; from cold storage that takes minutes.  Start the
        # read now (file I/O on a background thread of the library, no GPU call) so that it overlaps with the
        # host assembly and the upload, and share the host's cores between the ranks of this node.
        backend.load()
        solver_module.prefetch_rccl_library()
        backend.set_option("BODGE_AMD_HOST_THREADS", max(1, min(32, usable // world)))

    shape = [int(v) for v in args.lattice.split(",")]
    kind = VEC_RADEMACHER if args.vector_kind == "rademacher" else VEC_Z4
    r_local = args.vectors_per_gpu
    t0 = time.perf_counter()
    system = build_system(shape, args.model)
    indptr, indices, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    t_build = time.perf_counter() - t0
    assert not system._devices, "the host assembly must not create a device mirror (CPU baseline forks workers next)"

    cpu_record = None
    if args.cpu_seconds > 0 and args.gpus == 1 and rank == 0:
        cpu_record = cpu_baseline(system, scale, r_local, kind, args.cpu_seconds)  # before any GPU call

    backend.load()
    backend.require_device()
    t_wait = time.perf_counter()
    if world > 1:  # progress lines while the shared object arrives (a silent wait looks like a hang)
        while not solver_module.rccl_library_ready(20.0):
            print(f"rank {rank}: waiting for librccl.so to arrive from storage, {time.perf_counter() - t_rccl:.0f} s so far",
                  file=sys.stderr, flush=True)
    comm, collective = make_communicator(Communicator, world, rank, args.mode, args.allow_gloo, store)
    rccl_load_s = {"since_prefetch_start": time.perf_counter() - t_rccl, "waited_after_assembly": time.perf_counter() - t_wait} if world > 1 else None
    rccl = isinstance(comm, Communicator)
    # what RCCL itself spans: rank count and the PCI bus id of every rank's device
    rccl_ranks, rank_devices = 0, [None] * world
    if rccl:
        info = comm.info()
        rccl_ranks = info["n_ranks"]
        packed = [int(part, 16) for part in info["pci_bus_id"].replace(".", ":").split(":")]  # domain:bus:device.function
        code = float((packed[0] << 16) | (packed[1] << 8) | (packed[2] << 3) | packed[3])
        slots = np.zeros(world)
        slots[rank] = code
        for r, value in enumerate(comm.allreduce_sum(slots)):
            v = int(value)
            rank_devices[r] = f"{v >> 16:04x}:{(v >> 8) & 0xFF:02x}:{(v >> 3) & 0x1F:02x}.{v & 7}"

    device = local % backend.device_count()
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
        if isinstance(comm, HostReductions):  # RCCL unavailable: local moments, summed on the host
            return comm.allreduce_sum(
                solver.moments_random(scale, 2 * steps, r_local, seed=0, first_id=first, kind=kind, comm=None))
        return solver.moments_random(scale, 2 * steps, r_local, seed=0, first_id=first, kind=kind, comm=comm)

    def timed(steps):
        if comm is not None:
            comm.barrier()
        t0 = time.perf_counter()
        mu = run(steps)
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = float(comm.allreduce_max(np.array([dt]))[0])
        return mu, dt, solver.perf()

    total_vectors = r_local * (1 if args.mode == "slab" else args.gpus)

    def kernel_label(pf):
        mode = f"{'Real' if pf['real_arithmetic'] else 'Complex'}{'PH' if pf['ph_packed'] else ''}Mode"
        if pf["steps_per_launch"] == 3:
            tag = {0: "", 1: ",onsite-streamed", 2: ",all-blocks-streamed"}[pf["onsite_streamed"]]
            return f"cheb_sweep3<{mode},{pf['lanes_per_row']}{tag}>"
        if pf["steps_per_launch"] == 2:
            return f"cheb_sweep<{mode},{pf['lanes_per_row']}>"
        if pf["rolling"]:
            return f"cheb_roll3<{mode},{pf['lanes_per_row']}>"
        family = "cheb_step_dict" if pf["dict_blocks"] else "cheb_step_pipelined" if pf["pipelined"] else "cheb_step"
        return f"{family}<{mode},{pf['lanes_per_row']}>"

    # Further passes with optimisations switched off, reported beside the headline and never as
    # `value`: one recurrence step per launch, blocks streamed from HBM instead of the LDS
    # dictionary, and complex arithmetic (the reference's own dtype).
    def launch_record(pf, dt, vectors, steps, lattice_shape):
        launch = pf["kernel_ms"] / max(1, pf["launches"])
        return {
            "value": vectors * steps / dt,
            "unit": "steps/s",
            "kernel": kernel_label(pf),
            "launch_ms": launch,
            "steps_per_launch": pf["steps_per_launch"],
            "vectors_per_launch": pf["vectors_per_launch"],
            "bytes_per_launch": pf["bytes_moved"] / max(1, pf["launches"]),
            "bytes_full_launch": pf["bytes_per_launch"],
            "streams": pf["streams"],
            "window_ms": pf["window_ms"],
            "achieved_GBps": pf["bytes_moved"] / (pf["window_ms"] * 1e-3) / 1e9,
            "frac": pf["bytes_moved"] / (pf["window_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": measured_traffic(kernel_label(pf), lattice_shape, vectors),
        }

    def alternative(env):
        with backend.options(**env):  # (library switches for this pass only; bdg_set_option, not os.environ)
            _, dt, pf = timed(args.steps)
        return launch_record(pf, dt, total_vectors, args.steps, shape)

    def other_matrix(model, vec_kind):
        """A second Hamiltonian of the same lattice through the same call (N = 1 only): its default
        route, and the one-step kernels on it beside that."""
        other = build_system(shape, model)
        o_indptr, o_indices, o_data = other.bsr_arrays()
        o_scale = chebyshev.spectral_bound(o_indptr, o_data)
        with DeviceSolver(o_indptr, o_indices, o_data, device=device) as dev:
            dev.set_lattice_shape(shape)
            out = {}
            for name, env in (("default", {}), ("one_step", {"BODGE_AMD_SWEEP": "0"})):
                with backend.options(**env):
                    # (these passes start from an idle GPU - the matrix has just been assembled on the host - and the
                    # sweep kernels feel the shader clock for ~50 ms: a longer warm-up than the headline's W steps)
                    dev.moments_random(o_scale, 2 * max(args.warmup, 63), r_local, seed=0, kind=vec_kind)
                    # (... and then some: 63 steps are 5-12 ms of GPU work, after a second of host assembly; at the driver's
                    # --steps 20 the timed call is 2-5 ms long and ran 30 % below the rate of the same call in a loop)
                    settle = time.perf_counter() + 0.15
                    while time.perf_counter() < settle:
                        dev.moments_random(o_scale, 2 * 63, r_local, seed=0, kind=vec_kind)
                    t_start = time.perf_counter()
                    dev.moments_random(o_scale, 2 * args.steps, r_local, seed=0, kind=vec_kind)
                    dt = time.perf_counter() - t_start
                    out[name] = launch_record(dev.perf(), dt, r_local, args.steps, shape)
            pf = dev.perf()
        record = out["default"]
        record["one_step"] = out["one_step"]
        record["workload"] = {"potential": "random on-site potential and gap amplitude (10^6 distinct diagonal blocks), real",
                              "texture": "exchange field of varying direction on every site (10^6 distinct diagonal blocks), complex",
                              "ssd": "every term scaled by the reference's ssd() envelope (on-site AND bond blocks position dependent), real",
                              "landau": "exchange field of varying direction on every site and Landau-gauge Peierls phases on the x bonds (2000 distinct complex bond blocks), complex"}[model]
        record["distinct_blocks_total"] = "> 256" if pf["dict_skipped"] == 1 else pf["dict_blocks"]
        return record

    # Order: the comparison passes run first, the headline last.  The multi-step sweep kernels are
    # sensitive to the shader clock, which takes ~50 ms of continuous load to settle after idle
    # (launch time 270 -> 247 us between an 8-step and a 512-step warm-up); the
    # comparison passes are bandwidth-bound and lose 1-2 % in that phase.  The headline's own W
    # warm-up steps still directly precede its K timed steps.
    if args.warmup > 0:
        run(args.warmup)
    probe = solver.perf()  # which kernel family the default route takes on this matrix
    position_pass = other_matrix("potential", VEC_RADEMACHER) if world == 1 and args.model == "swave" else None
    texture_pass = other_matrix("texture", VEC_Z4) if world == 1 and args.model == "swave" else None
    ssd_pass = other_matrix("ssd", VEC_RADEMACHER) if world == 1 and args.model == "swave" else None
    landau_pass = other_matrix("landau", VEC_Z4) if world == 1 and args.model == "swave" else None
    complex_one_step = (alternative({"BODGE_AMD_SWEEP": "0", "BODGE_AMD_DICT": "0", "BODGE_AMD_REAL": "0"})
                        if probe["real_arithmetic"] else None)
    complex_sweep = alternative({"BODGE_AMD_REAL": "0"}) if probe["real_arithmetic"] and probe["steps_per_launch"] >= 2 else None
    streamed_one_step = alternative({"BODGE_AMD_SWEEP": "0", "BODGE_AMD_DICT": "0"}) if probe["dict_blocks"] else None
    one_step_pass = alternative({"BODGE_AMD_SWEEP": "0"}) if probe["steps_per_launch"] >= 2 or probe["rolling"] else None
    two_step_pass = alternative({"BODGE_AMD_SWEEP_STEPS": "2"}) if probe["steps_per_launch"] == 3 else None
    if args.warmup > 0:
        run(args.warmup)
    mu, elapsed, perf = timed(args.steps)

    if rank != 0:
        if store is not None:
            store.finish()
        return

    # The calls a user of the reference makes, through the Python API, timed as wall time after the headline (never
    # `value`; N = 1 only): BASELINE's metric ends "free_energy wall-time" (VERDICT r3 item 4).
    user_calls = user_facing_wall_times(system, shape) if world == 1 and args.model == "swave" and args.user_calls else None

    value = total_vectors * args.steps / elapsed
    launch_ms = perf["kernel_ms"] / max(1, perf["launches"])
    # algorithmic bytes of the launches inside the event window / their time.  bytes_moved is what the launches of
    # this call have to move: the first sweep reads no vectors (t_0 is generated, t_{-1} = 0) and the last launch
    # stores none (nothing reads them), so the average launch is lighter than a full one (bytes_full_launch)
    # The two lane groups of the call (2 x 4 vectors) run side by side on two streams, their launches filling each
    # other's idle starts and ends: `window_ms` is the HIP-event time from the first launch to the end of the last on
    # either stream, `launch_ms` the duration of one launch on its own stream (kernel_ms / launches - what rocprofv3
    # reports per kernel; two are in flight at a time, so launches x launch_ms = streams x window_ms).
    achieved = perf["bytes_moved"] / (perf["window_ms"] * 1e-3) / 1e9
    free_energy = chebyshev.free_energy_series(mu / total_vectors, scale, args.temperature)
    kernel_name = kernel_label(perf)
    model_label = {"swave": "s-wave+Zeeman", "dwave": "d-wave"}[args.model]
    record = {
        "metric": BASELINE_METRIC,
        "metric_detail": "value = Chebyshev vector-steps/s (one step = t_{n+1} = 2 H t_n / a - t_{n-1} on one vector, fused "
                         "with the two dot products); achieved HBM GB/s in roofline.achieved (per launch of "
                         "roofline.kernel, which advances every vector by roofline.steps_per_launch steps); "
                         "wall time in free_energy_wall_s",
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
            "workload": f"CubicLattice({tuple(shape)}) {model_label}, {2 * args.steps}-moment stochastic-trace "
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
            "rccl_ranks": rccl_ranks,        # ncclCommCount of the communicator the moments were reduced over (0 = none)
            "rank_devices": rank_devices,    # PCI bus id of every rank's GPU, gathered over that communicator
            "rccl_load_s": rccl_load_s,      # N > 1: librccl.so read (started before the host assembly) + communicator creation
            "host_threads_per_rank": max(1, min(32, usable // world)) if world > 1 else None,
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
            "streams": perf["streams"],
            "window_ms": perf["window_ms"],
            "steps_per_launch": perf["steps_per_launch"],
            "vectors_per_launch": perf["vectors_per_launch"],
            "x_neighbours_in_registers": bool(perf["steps_per_launch"] >= 2 or perf["rolling"]),
            "launches": perf["launches"],
            "bytes_per_launch": perf["bytes_moved"] / max(1, perf["launches"]),
            "bytes_full_launch": perf["bytes_per_launch"],
            # the same time priced at the algorithmic bytes of a full launch per `steps_per_launch` steps (what the
            # steps would cost without the savings at the two ends of a run): vector-steps in the event window x
            # bytes per vector-step / kernel time
            "effective_GBps": perf["bytes_per_launch"] / max(1, perf["steps_per_launch"]) * args.steps
                              * (r_local / perf["vectors_per_launch"]) / (perf["window_ms"] * 1e-3) / 1e9,
            "grid": perf["grid"],
            "lds_bytes": perf["lds_bytes"],
            "strip_rows": perf["strip_rows"],
            "distinct_blocks": perf["dict_blocks"],
        },
        "pass_order": "comparison passes (other matrices, complex128, streamed, one-step, two-step) first, then W warm-up "
                      "steps and the K timed steps of the headline kernel",
        "free_energy_wall_s": elapsed,
        "free_energy_estimate": free_energy,
        "user_facing_calls": user_calls,
        "host_assembly_s": t_build,
        "two_step_kernels": two_step_pass,
        "one_step_kernels": one_step_pass,
        "streamed_blocks_kernels": position_pass,
        "complex128_kernels": texture_pass,
        "streamed_bonds_kernels": ssd_pass,
        "complex128_bonds_kernels": landau_pass,
        "complex128_sweep_kernels": complex_sweep,
        "streamed_blocks_one_step_kernels": streamed_one_step,
        "complex128_one_step_kernels": complex_one_step,
        "cpu_baseline": cpu_record,
    }
    print(json.dumps(record), flush=True)
    if store is not None:
        store.finish()


if __name__ == "__main__":
    main()
