"""Device-resident Hamiltonian: thin object wrapper over the C ABI handle.

`DeviceSolver` owns one `bdg_system*` (one BSR matrix in the HBM of one GPU).
`Communicator` owns one RCCL rank.  All numerics happen inside the library;
this module only marshals numpy buffers.
"""

from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import backend
from .backend import VEC_RADEMACHER, VEC_Z4  # noqa: F401  (re-exported)


def _default_device() -> int:
    device = int(os.environ.get("BODGE_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return device % backend.device_count()


DENSE_LIBRARY_FROM = 2048  # 4N above which bdg_eigh_dense WITH all eigenvectors goes to rocSOLVER (kJacobiLimit in the library)


def prefetch_dense_library() -> None:
    """Start bringing the rocSOLVER / rocBLAS shared objects into the page cache in the background
    (`bdg_dense_prefetch`; file I/O only).  From cold storage the first dense eigensolve above
    4N = 2048 otherwise waits minutes for them."""
    backend.check(backend.load().bdg_dense_prefetch())


def prefetch_rccl_library() -> None:
    """The same for the RCCL shared object (573 MB) that the first `Communicator` loads."""
    backend.check(backend.load().bdg_rccl_prefetch())


def rccl_library_ready(timeout: float = 0.0) -> bool:
    ready = C.c_int32(0)
    backend.check(backend.load().bdg_rccl_prefetch_wait(float(timeout), C.byref(ready)))
    return bool(ready.value)


def dense_library_ready(timeout: float = 0.0) -> bool:
    """True once the background read started by `prefetch_dense_library` has finished; waits up
    to `timeout` seconds for it."""
    ready = C.c_int32(0)
    backend.check(backend.load().bdg_dense_prefetch_wait(float(timeout), C.byref(ready)))
    return bool(ready.value)


class DeviceSolver:
    def __init__(self, indptr, indices, data, device: int | None = None, n_cols: int | None = None,
                 row_offset: int = 0):
        """Upload a BSR matrix.  `n_cols` / `row_offset` describe a row slab (see `from_slab_plan`)."""
        lib = backend.load()
        backend.require_device()
        if device is None:
            device = _default_device()
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.complex128)
        self.n_sites = int(indptr.size - 1)
        self.n_cols = self.n_sites if n_cols is None else int(n_cols)
        self.n_blocks = int(indices.size)
        self.dim = 4 * self.n_sites
        self.device = device
        if data.size != 16 * self.n_blocks:
            raise ValueError("block data does not match the index arrays")
        self._lib = lib
        self._handle = C.c_void_p()
        self._keepalive = None
        backend.check(
            lib.bdg_create_slab(
                device, self.n_sites, self.n_cols, self.n_blocks, backend.as_i32p(indptr),
                backend.as_i32p(indices), backend.as_f64p(data.view(np.float64)), int(row_offset),
                C.byref(self._handle),
            )
        )

    @classmethod
    def from_slab_plan(cls, plan, comm: "Communicator | None" = None, device: int | None = None):
        """One row slab (`bodge_amd.slab.SlabPlan`).  With `comm` the halo exchange uses RCCL
        send/recv between ranks; without it the handle must be put into a `SlabGroup`."""
        solver = cls(plan.indptr, plan.indices, plan.data, device=device, n_cols=plan.n_cols,
                     row_offset=plan.row0)
        n_peers = len(plan.peers)
        peer_rank = np.asarray(plan.peers, dtype=np.int32)
        send_count = np.asarray([rows.size for rows in plan.send_rows], dtype=np.int64)
        send_rows = (np.concatenate(plan.send_rows) if n_peers else np.zeros(0)).astype(np.int64)
        recv_col = np.asarray(plan.recv_offset, dtype=np.int64)
        recv_count = np.asarray(plan.recv_count, dtype=np.int64)
        backend.check(
            solver._lib.bdg_slab_set_exchange(
                solver._handle, comm._handle if comm is not None else None, n_peers,
                backend.as_i32p(peer_rank), backend.as_i64p(send_count), backend.as_i64p(send_rows),
                backend.as_i64p(recv_col), backend.as_i64p(recv_count),
            )
        )
        solver._keepalive = comm
        return solver

    @classmethod
    def from_hamiltonian(cls, system, device: int | None = None, drop_zero_blocks: bool = True):
        # (no read-ahead of the rocSOLVER object any more: from 4N > 512 on eigenvalues and eigenvectors come from
        # the library's own tridiagonalisation route; rocSOLVER is reached only through BODGE_AMD_EIGH)
        indptr, indices, data = system.bsr_arrays(drop_zero_blocks=drop_zero_blocks)
        solver = cls(indptr, indices, data, device=device)
        from .lattice import CubicLattice

        if isinstance(system.lattice, CubicLattice):
            solver.set_lattice_shape(system.lattice.shape)
        return solver

    def set_lattice_shape(self, shape) -> None:
        """Geometry hint (performance only): rows are numbered z + Lz*(y + Ly*x)."""
        lx, ly, lz = (int(v) for v in shape)
        backend.check(self._lib.bdg_set_lattice_shape(self._handle, lx, ly, lz))

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.bdg_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------- kernels
    def spmv(self, x: np.ndarray) -> np.ndarray:
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        x = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1)
        if x.size != self.dim:
            raise ValueError(f"expected a vector of {self.dim} entries")
        y = np.empty_like(x)
        backend.check(
            self._lib.bdg_spmv(self._handle, backend.as_f64p(x.view(np.float64)),
                               backend.as_f64p(y.view(np.float64)))
        )
        return y

    def random_vector(self, seed: int, vec_id: int, kind: int = VEC_RADEMACHER) -> np.ndarray:
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        v = np.empty(self.dim, dtype=np.complex128)
        backend.check(
            self._lib.bdg_random_vector(self._handle, seed, vec_id, kind, backend.as_f64p(v.view(np.float64)))
        )
        return v

    def dots_random(self, scale, n_steps, n_vectors, seed=0, first_id=0, kind=VEC_RADEMACHER):
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        d = np.empty((n_steps, n_vectors))
        e = np.empty((n_steps, n_vectors))
        backend.check(
            self._lib.bdg_cheb_dots_random(
                self._handle, float(scale), n_steps, n_vectors, seed, first_id, kind,
                backend.as_f64p(d), backend.as_f64p(e),
            )
        )
        return d, e

    def dots_unit(self, scale, n_steps, rows):
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        d = np.empty((n_steps, rows.size))
        e = np.empty((n_steps, rows.size))
        backend.check(
            self._lib.bdg_cheb_dots_unit(
                self._handle, float(scale), n_steps, rows.size, backend.as_i64p(rows),
                backend.as_f64p(d), backend.as_f64p(e),
            )
        )
        return d, e

    def moments_random(self, scale, n_moments, n_vectors, seed=0, first_id=0, kind=VEC_RADEMACHER,
                       comm: "Communicator | None" = None) -> np.ndarray:
        """Σ_r <v_r|T_m(H/scale)|v_r> for m < n_moments (summed over ranks if `comm`)."""
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        mu = np.empty(n_moments)
        backend.check(
            self._lib.bdg_cheb_moments(
                self._handle, comm._handle if comm is not None else None, float(scale), n_moments,
                n_vectors, seed, first_id, kind, backend.as_f64p(mu),
            )
        )
        return mu

    def moments_unit(self, scale, n_moments, rows) -> np.ndarray:
        """(n_moments, len(rows)) array of <e_row|T_m(H/scale)|e_row>."""
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        mu = np.empty((n_moments, rows.size))
        backend.check(
            self._lib.bdg_cheb_diag_moments(
                self._handle, float(scale), n_moments, rows.size, backend.as_i64p(rows), backend.as_f64p(mu)
            )
        )
        return mu

    def lanczos_begin(self, n_vectors: int, seed: int = 0, first_id: int = 0, kind: int = VEC_RADEMACHER,
                      max_iter: int = 10000) -> None:
        """Start n_vectors independent Lanczos processes on H^2 (see `lanczos_advance`)."""
        backend.check(self._lib.bdg_lanczos_begin(self._handle, n_vectors, seed, first_id, kind, max_iter))
        self._lanczos_vectors = n_vectors

    def lanczos_advance(self, n_iter: int):
        """Run n_iter more iterations; returns (alpha, beta) of shape (n_iter, n_vectors): diagonal and
        off-diagonal (beta_{j+1}) of the tridiagonal representation of H^2."""
        if not getattr(self, "_lanczos_vectors", 0):
            raise ValueError("bodge_hip: lanczos_begin has not been called on this handle")
        alpha = np.empty((n_iter, self._lanczos_vectors))
        beta = np.empty((n_iter, self._lanczos_vectors))
        backend.check(self._lib.bdg_lanczos_advance(self._handle, n_iter, backend.as_f64p(alpha), backend.as_f64p(beta)))
        return alpha, beta

    def lanczos_ritz_vectors(self, coef: np.ndarray) -> np.ndarray:
        """Second pass of a freshly begun Lanczos process: `coef[j, l, r]` are the coordinates of
        level l's Ritz vector of start vector r in the Lanczos basis; returns y[l, r, :] (4N complex)."""
        n_vectors = getattr(self, "_lanczos_vectors", 0)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        if not n_vectors or coef.ndim != 3 or coef.shape[2] != n_vectors:
            raise ValueError("bodge_hip: coefficients must be (iterations, levels, start vectors of lanczos_begin)")
        out = np.empty((coef.shape[1], n_vectors, self.dim), dtype=np.complex128)
        backend.check(self._lib.bdg_lanczos_ritz_vectors(self._handle, coef.shape[0], coef.shape[1], backend.as_f64p(coef),
                                                        backend.as_f64p(out.view(np.float64))))
        return out

    def lanczos_ritz_pairs(self, coef: np.ndarray, eps: np.ndarray, k: int, rank_tol: float = 1e-4):
        """Second pass + Rayleigh-Ritz on the device: the k lowest states found in the levels whose Ritz
        coordinates `coef[j, l, r]` and eigenvalue estimates `eps[l]` are given.  Returns
        (values (k',), vectors (k', 4N), n_found) with k' = min(k, n_found); only these cross PCIe."""
        n_vectors = getattr(self, "_lanczos_vectors", 0)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        if not n_vectors or coef.ndim != 3 or coef.shape[2] != n_vectors or eps.shape != (coef.shape[1],):
            raise ValueError("bodge_hip: coefficients must be (iterations, levels, start vectors of lanczos_begin), eps (levels,)")
        values = np.empty(k)
        vectors = np.empty((k, self.dim), dtype=np.complex128)
        found = C.c_int32(0)
        backend.check(self._lib.bdg_lanczos_ritz_pairs(
            self._handle, coef.shape[0], coef.shape[1], backend.as_f64p(coef), backend.as_f64p(eps), float(rank_tol), k,
            C.byref(found), backend.as_f64p(values), backend.as_f64p(vectors.view(np.float64))))
        self._lanczos_vectors = 0  # (the library ended the run)
        kept = min(k, found.value)
        return values[:kept], vectors[:kept], found.value

    def eigh(self, vectors: bool = True):
        """All eigenvalues ascending (and ALL eigenvectors as columns).  Eigenvalues alone: the library's own kernels
        at every size (Jacobi up to 4N = 512, tridiagonalisation + bisection above, through a band from 5000 rows for
        real matrices).  With vectors: own Jacobi kernels up to 4N = 2048, rocSOLVER above (dsyevd when imag(H) = 0,
        else zheevd) - `eigh_above` is the route `diagonalize()` takes and needs no library."""
        self._lanczos_vectors = 0  # any other use of the handle ends a Lanczos run (library: lanczos_free)
        w = np.empty(self.dim)
        if not vectors:
            backend.check(self._lib.bdg_eigh_dense(self._handle, backend.as_f64p(w), None))
            return w, None
        z = np.empty((self.dim, self.dim), dtype=np.complex128)  # column-major on return
        backend.check(
            self._lib.bdg_eigh_dense(self._handle, backend.as_f64p(w), backend.as_f64p(z.view(np.float64)))
        )
        return w, z.T

    def eigh_above(self, lower_bound: float = 0.0):
        """All eigenvalues ascending, and the eigenvectors (as columns) of those above `lower_bound` - what
        `diagonalize()` keeps.  From 4N > 512 on the library's own tridiagonalisation route (no rocSOLVER)."""
        self._lanczos_vectors = 0
        w = np.empty(self.dim)
        count = C.c_int64(0)
        capacity = self.dim // 2 + 8
        for _ in range(2):
            z = np.empty((capacity, self.dim), dtype=np.complex128)
            status = self._lib.bdg_eigh_dense_above(self._handle, float(lower_bound), capacity, backend.as_f64p(w),
                                                    C.byref(count), backend.as_f64p(z.view(np.float64)))
            if status == 0 or count.value <= capacity:
                break
            capacity = count.value  # (an asymmetric spectrum: more than half of the eigenvalues above the bound)
        backend.check(status)
        return w, z[: count.value].T

    def hermiticity_defect(self) -> float:
        """max |H - H^†| over the stored entries of the uploaded matrix (device-side twin of the
        check at reference hamiltonian.py:121-122)."""
        out = np.zeros(1)
        backend.check(self._lib.bdg_hermiticity_defect(self._handle, backend.as_f64p(out)))
        return float(out[0])

    def set_lanes_per_row(self, lanes: int) -> None:
        backend.check(self._lib.bdg_set_lanes_per_row(self._handle, lanes))

    def perf(self) -> dict:
        rec = backend.Perf()
        backend.check(self._lib.bdg_perf_query(self._handle, C.byref(rec)))
        return {name: getattr(rec, name) for name, _ in backend.Perf._fields_}


# ---------------------------------------------------------------------------
class SlabGroup:
    """Several row slabs of one matrix driven in lock step by this process.

    The slabs may sit on one GPU (how the slab path is validated on a single-GPU box) or on
    several GPUs of the node (`devices`); halo rows move by device-to-device copies.
    """

    def __init__(self, indptr, indices, data, n_slabs: int, granule: int = 1, devices=None, lattice_shape=None):
        from . import slab

        n_rows = len(indptr) - 1
        self.bounds = slab.partition_rows(n_rows, n_slabs, granule)
        self.plans = [slab.build_plan(indptr, indices, data, self.bounds, r) for r in range(n_slabs)]
        devices = [None] * n_slabs if devices is None else list(devices)
        self.members = [DeviceSolver.from_slab_plan(p, device=d) for p, d in zip(self.plans, devices)]
        if lattice_shape is not None and granule == lattice_shape[1] * lattice_shape[2]:
            # geometry hint per slab (a stack of whole x-planes): lets a 3-D slab run the stencil kernel that
            # reads its neighbours' boundary planes in place instead of exchanging halo rows
            for member, plan in zip(self.members, self.plans):
                member.set_lattice_shape((plan.n_own // granule, lattice_shape[1], lattice_shape[2]))
        self._lib = backend.load()
        self._handle = C.c_void_p()
        handles = (C.c_void_p * n_slabs)(*[m._handle for m in self.members])
        backend.check(self._lib.bdg_group_create(handles, n_slabs, C.byref(self._handle)))
        self.dim = 4 * n_rows

    @classmethod
    def from_hamiltonian(cls, system, n_slabs: int, devices=None):
        from . import slab

        indptr, indices, data = system.bsr_arrays()
        shape = getattr(system.lattice, "shape", None)
        return cls(indptr, indices, data, n_slabs, slab.lattice_granule(system.lattice), devices,
                   lattice_shape=tuple(shape) if shape is not None and len(shape) == 3 else None)

    def dots_random(self, scale, n_steps, n_vectors, seed=0, first_id=0, kind=VEC_RADEMACHER):
        d = np.empty((n_steps, n_vectors))
        e = np.empty((n_steps, n_vectors))
        backend.check(
            self._lib.bdg_group_dots_random(
                self._handle, float(scale), n_steps, n_vectors, seed, first_id, kind,
                backend.as_f64p(d), backend.as_f64p(e),
            )
        )
        return d, e

    def dots_unit(self, scale, n_steps, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        d = np.empty((n_steps, rows.size))
        e = np.empty((n_steps, rows.size))
        backend.check(
            self._lib.bdg_group_dots_unit(
                self._handle, float(scale), n_steps, rows.size, backend.as_i64p(rows),
                backend.as_f64p(d), backend.as_f64p(e),
            )
        )
        return d, e

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.bdg_group_destroy(self._handle)
            self._handle = C.c_void_p()
            for member in self.members:
                member.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _StdoutToStderr:
    """RCCL prints a version banner on stdout at communicator creation; programs that emit
    machine-readable stdout (bench.py's single JSON line) must not see it."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Communicator:
    """One RCCL rank.  The 128-byte unique id travels through `bodge_amd.rendezvous` (TCP)."""

    def __init__(self, rank: int, n_ranks: int, device: int, unique_id: bytes):
        lib = backend.load()
        self._lib = lib
        self.rank, self.n_ranks, self.device = rank, n_ranks, device
        self._handle = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        with _StdoutToStderr():
            backend.check(lib.bdg_comm_init(device, buf, n_ranks, rank, C.byref(self._handle)))

    @staticmethod
    def new_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        with _StdoutToStderr():
            backend.check(backend.load().bdg_comm_unique_id(buf))
        return bytes(buf)

    _sequence = 0  # communicators created by this process through `from_environment`

    @classmethod
    def from_environment(cls, timeout: float = 300.0) -> "Communicator | None":
        """Build the communicator of a `torch.distributed.run`-style launch.

        Uses RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT.  Rank 0 publishes the
        RCCL unique id through the launch's TCP rendezvous (`bodge_amd.rendezvous`: no files, a
        per-launch nonce on every message); the others block on it.  Returns None for a
        single-process run.
        """
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1 and os.environ.get("BODGE_AMD_FORCE_COMM") != "1":
            return None  # (the override builds a one-rank communicator: exercises the RCCL path on one GPU)
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", rank))
        # The RCCL object is 573 MB; from cold storage its first read takes minutes.  Every rank
        # streams it into the page cache (shared: one read serves all) BEFORE the id exchange, so
        # that the exchange's timeout covers the exchange and not rank 0's library load.
        prefetch_rccl_library()
        waited = 0
        while not rccl_library_ready(30.0):
            waited += 30
            if rank == 0:
                print(f"[bodge_amd] reading the RCCL library from cold storage ... {waited} s", file=sys.stderr, flush=True)
            if waited >= 1800:
                raise RuntimeError("the RCCL shared object could not be read within 30 minutes")
        if world <= 1:
            uid = cls.new_unique_id()
        else:
            from .rendezvous import store_from_environment

            store = store_from_environment(timeout)
            key = f"rccl_unique_id/{cls._sequence}"
            cls._sequence += 1
            if rank == 0:
                uid = cls.new_unique_id()
                store.set(key, uid)
            else:
                uid = store.get(key, timeout)
            if len(uid) != 128:
                raise RuntimeError(f"rank {rank}: rendezvous returned {len(uid)} bytes instead of an RCCL id")
        # one process per GPU; more ranks than GPUs wrap around.  RCCL refuses two ranks on one device ("invalid usage"),
        # but only after its bootstrap and topology search - tens of seconds per rank: the ranks compare (host, device)
        # over the rendezvous store first and give the same answer at once
        count = backend.device_count()
        device = local % count if count > 0 else local
        if world > 1:
            import socket

            places = [blob.decode() for blob in store.gather(f"{socket.gethostname()}/{device}".encode())]
            clash = sorted({place for place in places if places.count(place) > 1})
            if clash:
                raise RuntimeError(f"rank {rank}: {world} ranks but some share a GPU ({', '.join(clash)}): RCCL refuses duplicate devices")
        comm = cls(rank, world, device, uid)
        comm.barrier()
        return comm

    def info(self) -> dict:
        """{'n_ranks': ncclCommCount, 'rank', 'device', 'pci_bus_id'} as the library and RCCL report them."""
        n_ranks, rank, device = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        bus = C.create_string_buffer(32)
        backend.check(self._lib.bdg_comm_info(self._handle, C.byref(n_ranks), C.byref(rank), C.byref(device), bus))
        return {"n_ranks": n_ranks.value, "rank": rank.value, "device": device.value,
                "pci_bus_id": bus.value.decode("ascii", "replace")}

    def allreduce_sum(self, values: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(values, dtype=np.float64).copy()
        backend.check(self._lib.bdg_comm_allreduce_sum(self._handle, backend.as_f64p(out), out.size))
        return out

    def allreduce_max(self, values: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(values, dtype=np.float64).copy()
        backend.check(self._lib.bdg_comm_allreduce_max(self._handle, backend.as_f64p(out), out.size))
        return out

    def barrier(self) -> None:
        self.allreduce_sum(np.zeros(1))

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.bdg_comm_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
