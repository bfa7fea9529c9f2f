"""Tight-binding BdG Hamiltonian: assembly on the host, observables on the GPU.

Boundary being reproduced (reference `bodge/hamiltonian.py`):

* storage: one BSR matrix, 4x4 complex128 blocks in the basis (e↑, e↓, h↑, h↓)
  per site, int32 `indices/indptr`, canonical order; the skeleton holds the
  diagonal plus every bond and periodic-edge block, zero or not  (ref :37-67)
* `with system as (H, Δ)`: two mappings keyed by coordinate pairs; on exit the
  2x2 values are expanded with particle-hole symmetry (H -> [+H, -H*] on the
  block diagonal) and Hermitian symmetry (Δ_ij in the upper right of block
  (i,j), Δ_ij^† in the lower left of block (j,i)); non-Hermitian result ->
  RuntimeError                                                     (ref :70-126)
* `matrix(format)`, `index(row, col)`                              (ref :129-170)
* `diagonalize`, `free_energy`, `ldos` signatures and results      (ref :173-387)

What is different by design: assembly is vectorised (block lookup by binary
search over sorted keys instead of one `np.where` per term; bulk setters on the
H/Δ mappings), and the three observables never run on the CPU - they call the
HIP library through `bodge_amd.backend` and raise if it is unavailable.
"""

from __future__ import annotations

from typing import Callable

import numpy as np
import scipy.sparse as sp

from .common import Coord, Coords, Index, Indices, Matrix, jσ2, σ, typecheck, π
from .lattice import CubicLattice, Lattice


# `with` blocks that wrote at least this many blocks have their Hermiticity test made on the GPU
# (bdg_hermiticity_defect) when one is present; smaller ones - every reference-sized lattice - on
# the host, where the chunked comparison takes milliseconds.
DEVICE_HERMITICITY_MIN_BLOCKS = 500_000


def _gpu_present() -> bool:
    try:
        from . import backend

        return backend.device_count() > 0
    except (RuntimeError, OSError):
        return False


def _host_library():
    """The shared library for its CPU-thread assembly helpers (bdg_host_*), or None: without it
    (not built, or BODGE_AMD_HOST_NATIVE=0) the same passes are made with numpy.  This concerns the
    host-side assembly only - the observables have no such alternative."""
    import os

    if os.environ.get("BODGE_AMD_HOST_NATIVE", "1") == "0":
        return None
    try:
        from . import backend

        return backend.load()
    except (RuntimeError, OSError):
        return None


def _fill_terms(data: np.ndarray, ids: np.ndarray, values: np.ndarray, kind: int, touched: np.ndarray | None) -> None:
    """Scatter 2x2 spin matrices into the 4x4 Nambu blocks `ids` (ref :102-118).

    kind 0: hopping  blk[0:2,0:2] = v, blk[2:4,2:4] = -v*;  kind 1: pairing  blk[0:2,2:4] = v;
    kind 2: pairing term seen from the transposed block  blk[2:4,0:2] = v^†.
    `values` is (len(ids), 2, 2) or one (2, 2) matrix for every id."""
    lib = _host_library()
    if lib is None:
        if touched is not None:
            touched[ids] = 1
        if kind == 0:
            data[ids, 0:2, 0:2] = values
            data[ids, 2:4, 2:4] = -values.conj()
        elif kind == 1:
            data[ids, 0:2, 2:4] = values
        else:
            data[ids, 2:4, 0:2] = np.swapaxes(values.conj(), -1, -2)
        return
    from . import backend

    ids = np.ascontiguousarray(ids, dtype=np.int64)
    values = np.ascontiguousarray(values, dtype=np.complex128)
    backend.check(lib.bdg_host_fill_terms(
        backend.as_f64p(data), len(data), backend.as_i64p(ids), len(ids), backend.as_f64p(values),
        1 if values.ndim == 3 else 0, kind, None if touched is None else backend.as_u8p(touched)))


def _cubic_skeleton(shape) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(indptr, indices, block_rows) of the block skeleton of a cubic lattice: the diagonal, every
    bond and every opposite-face "edge" pair, zero or not (ref :37-67) - written down directly
    instead of sorting the pair list.  Relative to site i = z + Lz (y + Ly x) the partners sit at
    fixed offsets, which in ascending order are
        -(Lx-1)Sx, -Sx, -(Ly-1)Sy, -Sy, -(Lz-1), -1, 0, +1, +(Lz-1), +Sy, +(Ly-1)Sy, +Sx, +(Lx-1)Sx
    (Sx = Ly Lz, Sy = Lz), each present where the site has that partner.  An axis of extent 2 has
    its edge pair coincide with its bond, an axis of extent 1 has it coincide with the diagonal."""
    Lx, Ly, Lz = (int(v) for v in shape)
    n = Lx * Ly * Lz
    extent, stride = (Lx, Ly, Lz), (Ly * Lz, Lz, 1)
    i = np.arange(n, dtype=np.int64)
    coord = (i // stride[0], (i // Lz) % Ly, i % Lz)
    offsets, present = [], []
    for a in (0, 1, 2):
        if extent[a] >= 3:
            offsets.append(-(extent[a] - 1) * stride[a])
            present.append(coord[a] == extent[a] - 1)
        if extent[a] >= 2:
            offsets.append(-stride[a])
            present.append(coord[a] >= 1)
    offsets.append(0)
    present.append(np.ones(n, dtype=bool))
    for a in (2, 1, 0):
        if extent[a] >= 2:
            offsets.append(stride[a])
            present.append(coord[a] <= extent[a] - 2)
        if extent[a] >= 3:
            offsets.append((extent[a] - 1) * stride[a])
            present.append(coord[a] == 0)
    mask = np.stack(present, axis=1)
    columns = i[:, None] + np.array(offsets, dtype=np.int64)[None, :]
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(mask.sum(axis=1), out=indptr[1:])
    block_rows = np.broadcast_to(i[:, None], mask.shape)[mask]
    return indptr, columns[mask].astype(np.int32), block_rows


# ---------------------------------------------------------------------------
class TermTable(dict):
    """Mapping {(coord_i, coord_j): 2x2 matrix} handed out by `with system as ...`.

    It is a plain dict for everything the reference API does.  The extra
    methods queue whole-lattice assignments as arrays; they are applied, in call
    order, before the individually keyed entries when the `with` block closes.
    """

    def __init__(self, lattice: Lattice):
        super().__init__()
        self._lattice = lattice
        self._bulk: list[tuple[np.ndarray, np.ndarray, np.ndarray]] = []

    def _queue(self, rows: np.ndarray, cols: np.ndarray, values) -> None:
        values = np.asarray(values, dtype=np.complex128)
        if values.shape[-2:] != (2, 2):
            raise ValueError("Expected 2x2 spin matrices (or an array of them)")
        if values.ndim == 2:
            values = values.copy()  # one matrix for every term of the batch
        elif values.ndim != 3 or values.shape[0] != len(rows):
            raise ValueError(f"Expected {len(rows)} matrices, got {values.shape[0]}")
        self._bulk.append((np.asarray(rows, np.int64), np.asarray(cols, np.int64), values))

    def set_sites(self, values) -> None:
        """Assign term[i, i] for every site; `values` is 2x2 or (N, 2, 2) in index order."""
        n = np.arange(self._lattice.size, dtype=np.int64)
        self._queue(n, n, values)

    def set_bonds(self, values, axis=None) -> None:
        """Assign term[i, j] for every directed bond, ordered as `lattice.bond_array(axis)`."""
        pairs = self._lattice.bond_array(axis)
        self._queue(pairs[:, 0], pairs[:, 1], values)

    def set_edges(self, values, axis=None) -> None:
        """Assign term[i, j] for every directed edge pair, ordered as `lattice.edge_array(axis)`."""
        pairs = self._lattice.edge_array(axis)
        self._queue(pairs[:, 0], pairs[:, 1], values)


# ---------------------------------------------------------------------------
class Hamiltonian:
    """4N x 4N Bogoliubov-de Gennes matrix on a lattice of N sites."""

    def __init__(self, lattice: Lattice):
        if not isinstance(lattice, Lattice):
            raise TypeError("Hamiltonian(lattice): expected a Lattice instance")
        self.lattice: Lattice = lattice
        n = lattice.size
        self.shape: Indices = (4 * n, 4 * n)

        if type(lattice) is CubicLattice:
            indptr, indices, block_rows = _cubic_skeleton(lattice.shape)
            keys = block_rows * n + indices
        else:
            rows, cols = self._skeleton_pairs()
            keys = np.unique(np.concatenate([rows * n + cols, cols * n + rows]))
            block_rows = keys // n
            indices = (keys - block_rows * n).astype(np.int32)
            indptr = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(np.bincount(block_rows, minlength=n), out=indptr[1:])

        data = np.zeros((len(keys), 4, 4), dtype=np.complex128)
        self._matrix = sp.bsr_matrix((data, indices, indptr), shape=self.shape, blocksize=(4, 4))
        self._data: Matrix = self._matrix.data

        # Sorted (row, col) keys: block lookup is a binary search.
        self._keys = keys
        self._mirror_ids: np.ndarray | None = None

        # Device-side state (created lazily; invalidated whenever terms change).
        self._revision = 0
        self._devices: dict = {}
        self._device_revision = -1
        self._memo: dict = {}
        self._recheck_all = False  # set when a Hermiticity check failed: the next one covers everything
        # Where the Hermiticity test of a closing `with` block runs: "auto" = on the GPU for large
        # fills when one is present, "host" = always on the host (a process that must not touch the
        # GPU yet, e.g. before forking CPU workers), "device" = always on the GPU.
        self.hermiticity_check = "auto"
        self._memo_revision = -1

    @property
    def _mirror(self) -> np.ndarray:
        """`_mirror[k]` = position of the transposed block of block k (host Hermiticity check)."""
        if self._mirror_ids is None:
            n = self.lattice.size
            block_rows = self._keys // n
            self._mirror_ids = np.searchsorted(self._keys, (self._keys - block_rows * n) * n + block_rows)
        return self._mirror_ids

    def _skeleton_pairs(self) -> tuple[np.ndarray, np.ndarray]:
        lattice = self.lattice
        if isinstance(lattice, CubicLattice):
            diag = np.arange(lattice.size, dtype=np.int64)
            links = np.concatenate([lattice.bond_array(), lattice.edge_array()], axis=0)
            return np.concatenate([diag, links[:, 0]]), np.concatenate([diag, links[:, 1]])
        pairs = np.array([(lattice[a], lattice[b]) for a, b in lattice], dtype=np.int64)
        pairs = pairs.reshape(-1, 2)
        return pairs[:, 0], pairs[:, 1]

    # ----------------------------------------------------------- with-block API
    def __enter__(self) -> tuple[dict, dict]:
        self._hopp = TermTable(self.lattice)
        self._pair = TermTable(self.lattice)
        return self._hopp, self._pair

    def _block_ids(self, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
        wanted = rows * self.lattice.size + cols
        found = np.searchsorted(self._keys, wanted)
        found = np.minimum(found, len(self._keys) - 1)
        if np.any(self._keys[found] != wanted):
            raise IndexError("Term refers to a pair of sites that the lattice does not connect")
        return found

    def _dict_to_arrays(self, table: TermTable):
        """The individually keyed entries of a `with` block as (rows, cols, values) arrays."""
        if not table:
            return None
        lat = self.lattice
        if type(lat) is CubicLattice:
            # reference-style scripts assign one term per site / bond in a Python loop (ref :70-89 and
            # README.md:73-86); turning the keys into site indices must not cost another loop with a
            # type-checked `lattice[coord]` per key
            try:
                keys = np.array(list(table.keys()))
                vals = np.array(list(table.values()), dtype=np.complex128)
            except (TypeError, ValueError):
                keys = vals = None
            # (anything but integer coordinates and 2x2 values goes the per-key way below and meets its checks)
            if (keys is not None and keys.dtype.kind in "iu" and keys.shape == (len(table), 2, 3)
                    and vals.shape == (len(table), 2, 2)):
                keys = keys.astype(np.int64, copy=False)
                outside = (keys < 0) | (keys >= np.asarray(lat.shape, dtype=np.int64))
                if outside.any():
                    coord = keys.reshape(-1, 3)[np.flatnonzero(outside.reshape(-1, 3).any(axis=1))[0]]
                    raise ValueError(f"Coordinate {tuple(int(v) for v in coord)} out of bounds")
                flat = lat._flatten(keys)
                return flat[:, 0].copy(), flat[:, 1].copy(), vals
        rows = np.fromiter((lat[i] for i, _ in table.keys()), dtype=np.int64, count=len(table))
        cols = np.fromiter((lat[j] for _, j in table.keys()), dtype=np.int64, count=len(table))
        vals = np.empty((len(table), 2, 2), dtype=np.complex128)
        for n, v in enumerate(table.values()):
            vals[n] = v
        return rows, cols, vals

    def __exit__(self, exc_type, exc_val, exc_tb):
        data = self._data
        touched = np.zeros(len(data), dtype=np.uint8)
        for table, is_pairing in ((self._hopp, False), (self._pair, True)):
            batches = list(table._bulk)
            keyed = self._dict_to_arrays(table)
            if keyed is not None:
                batches.append(keyed)
            for rows, cols, vals in batches:
                k = self._block_ids(rows, cols)
                if is_pairing:
                    twin = k if rows is cols else self._block_ids(cols, rows)
                    _fill_terms(data, k, vals, 1, touched)
                    _fill_terms(data, twin, vals, 2, None)
                else:
                    _fill_terms(data, k, vals, 0, touched)
        del self._hopp
        del self._pair
        self._revision += 1

        # Blocks this `with` did not write were checked when they were written (and start out zero),
        # unless that earlier check failed and left the matrix in a non-Hermitian state.
        touched = touched.view(bool)
        everything = self._recheck_all or bool(touched.all())
        if self.hermiticity_check not in ("auto", "host", "device"):
            raise ValueError("hermiticity_check must be 'auto', 'host' or 'device'")
        large = int(np.count_nonzero(touched)) >= DEVICE_HERMITICITY_MIN_BLOCKS
        if self.hermiticity_check == "device" or (self.hermiticity_check == "auto" and large and _gpu_present()):
            # large fills: the matrix goes to the GPU now (the observables need it there anyway)
            # and is compared with its conjugate transpose on the device, block against block
            defect = self._solver().hermiticity_defect()
        else:
            defect = self._hermiticity_defect(None if everything else np.flatnonzero(touched))
        self._recheck_all = defect > 1e-6
        if self._recheck_all:
            raise RuntimeError("The constructed Hamiltonian is not Hermitian!")

    def _hermiticity_defect(self, blocks: np.ndarray | None = None, chunk: int = 8192) -> float:
        """max |H - H^†| over stored entries, block (i,j) against block (j,i)^†, for the given block
        ids (default: all).

        Same criterion as the reference's sparse `M - M.getH()` (ref :121-122),
        evaluated in cache-sized chunks so a 10^6-site matrix takes seconds.
        """
        data, mirror, worst = self._data, self._mirror, 0.0
        if blocks is None:
            for lo in range(0, len(data), chunk):
                part = data[lo : lo + chunk]
                twin = data[mirror[lo : lo + chunk]]
                diff = part - twin.conj().transpose(0, 2, 1)
                worst = max(worst, float(np.abs(diff).max()))
            return worst
        for lo in range(0, len(blocks), chunk):
            ids = blocks[lo : lo + chunk]
            diff = data[ids] - data[mirror[ids]].conj().transpose(0, 2, 1)
            worst = max(worst, float(np.abs(diff).max(initial=0.0)))
        return worst

    def has_symmetric_spectrum(self, tol: float = 0.0) -> bool:
        """True if H = -τx H* τx, i.e. every block is [[A, B], [-B*, -A*]]: the spectrum is then
        symmetric about zero.  The assembly guarantees the diagonal part (ref :106-108); the
        off-diagonal part holds when the pairing obeys fermionic antisymmetry Δ_ij = -Δ_ji^T
        (singlet on-site terms, odd-parity triplet bond terms).  The Chebyshev free energy
        relies on it; the dense path and the LDOS do not."""
        return bool(self._block_scan()["ph_defect"] <= tol)

    def gershgorin_bound(self) -> float:
        """max over scalar rows of Σ|H_rc| ≥ ‖H‖ (1.0 for an all-zero matrix): the number
        `chebyshev.spectral_bound(indptr, data, pad=1.0)` returns for the stored blocks."""
        bound = self._block_scan()["row_sum_max"]
        return bound if bound > 0 or np.isnan(bound) else 1.0

    def _block_scan(self) -> dict:
        """One pass over the stored blocks, cached until the next `with`: which blocks are all-zero,
        the particle-hole defect, the Gershgorin bound (bdg_host_scan_blocks; numpy without the library)."""
        def scan() -> dict:
            data, indptr = self._data, self._matrix.indptr
            lib = _host_library()
            if lib is None:
                from .chebyshev import spectral_bound

                nonzero = np.any(data.reshape(len(data), 16) != 0, axis=1)
                a = np.abs(data[:, 2:4, 2:4] + data[:, 0:2, 0:2].conj()).max(initial=0.0)
                b = np.abs(data[:, 2:4, 0:2] + data[:, 0:2, 2:4].conj()).max(initial=0.0)
                bound = spectral_bound(indptr, data, pad=1.0) if np.any(nonzero) else 0.0
                return {"nonzero": nonzero, "n_nonzero": int(np.count_nonzero(nonzero)), "ph_defect": float(max(a, b)),
                        "row_sum_max": float(bound)}
            from . import backend
            import ctypes as C

            nonzero = np.empty(len(data), dtype=np.uint8)
            count, defect, bound, real = C.c_int64(0), C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
            backend.check(lib.bdg_host_scan_blocks(
                backend.as_f64p(data), backend.as_i32p(indptr), len(indptr) - 1, backend.as_u8p(nonzero),
                C.byref(count), C.byref(defect), C.byref(bound), C.byref(real)))
            return {"nonzero": nonzero.view(bool), "n_nonzero": int(count.value), "ph_defect": float(defect.value),
                    "row_sum_max": float(bound.value)}

        return self._memoized("block_scan", scan)

    # ------------------------------------------------------------------ export
    def matrix(self, format: str = "dense"):
        if format == "bsr":
            out = self._matrix.copy()
            out.eliminate_zeros()
            return out
        if format == "csr":
            out = self._matrix.tocsr()
            out.eliminate_zeros()
            return out
        if format == "csc":
            out = self._matrix.tocsc()
            out.eliminate_zeros()
            return out
        if format == "dense":
            return self._matrix.todense()
        raise RuntimeError("Requested matrix format is not yet supported")

    @typecheck
    def index(self, row: Coord, col: Coord) -> Index:
        i = np.array([self.lattice[row]], dtype=np.int64)
        j = np.array([self.lattice[col]], dtype=np.int64)
        return Index(self._block_ids(i, j)[0])

    def bsr_arrays(self, drop_zero_blocks: bool = True):
        """(indptr int32, indices int32, data complex128 (nnzb,4,4)) for the device.

        With drop_zero_blocks the triple equals `matrix("bsr")`'s (zero blocks
        removed, ref :142-143) but is produced without a scipy round trip.
        """
        indptr, indices, data = self._matrix.indptr, self._matrix.indices, self._data
        if not drop_zero_blocks:
            return indptr.copy(), indices.copy(), data.copy()
        scan = self._block_scan()
        keep, n = scan["nonzero"], self.lattice.size
        lib = _host_library()
        if lib is None:
            block_rows = (self._keys // n)[keep]
            new_ptr = np.zeros(n + 1, dtype=np.int32)
            np.cumsum(np.bincount(block_rows, minlength=n), out=new_ptr[1:])
            return new_ptr, indices[keep].copy(), np.ascontiguousarray(data[keep])
        from . import backend

        new_ptr = np.empty(n + 1, dtype=np.int32)
        new_indices = np.empty(scan["n_nonzero"], dtype=np.int32)
        new_data = np.empty((scan["n_nonzero"], 4, 4), dtype=np.complex128)
        backend.check(lib.bdg_host_compact_blocks(
            backend.as_f64p(data), backend.as_i32p(indices), backend.as_i32p(indptr), n, backend.as_u8p(keep.view(np.uint8)),
            backend.as_f64p(new_data), backend.as_i32p(new_indices), backend.as_i32p(new_ptr)))
        return new_ptr, new_indices, new_data

    def _memoized(self, name: str, compute):
        """Value of `compute()` cached until the next `with` block rewrites the matrix."""
        if self._memo_revision != self._revision:
            self._memo = {}
            self._memo_revision = self._revision
        if name not in self._memo:
            self._memo[name] = compute()
        return self._memo[name]

    # ------------------------------------------------------------- observables
    def _solver(self, lane=0, device: int | None = None):
        """Device mirror of the current matrix (re-uploaded after every `with`).

        `lane` > 0 gives further independent mirrors (own stream and buffers): mid-size exact
        traces drive two of them from two host threads to overlap their launch latencies.
        `device` puts the mirror on that GPU of this process (`free_energy(devices=[...])`
        keeps one mirror per listed entry); default: `BODGE_AMD_DEVICE` / `LOCAL_RANK` / 0.
        """
        from .solver import DeviceSolver

        if self._device_revision != self._revision:
            for mirror in self._devices.values():
                mirror.close()
            self._devices = {}
            self._device_revision = self._revision
        key = lane if device is None else (lane, int(device))
        if key not in self._devices:
            self._devices[key] = DeviceSolver.from_hamiltonian(self, device=device)
        return self._devices[key]

    def diagonalize(self, cuda: bool = False, format: str = "reshape"):
        """Positive-energy eigenpairs (E, v[n, site, α]) or raw (E, X[:, n]).

        Same contract as the reference (:173-251); `cuda` is accepted for call
        compatibility - the dense Hermitian eigensolve always runs on the GPU.
        """
        from .observables import diagonalize

        return diagonalize(self, format=format)

    @typecheck
    def free_energy(self, temperature: float = 0.0, cuda: bool = False, **options) -> float:
        """Landau free energy F = -1/2 Σ ε - T Σ log(1 + exp(-ε/T)) over ε > 0 (ref :254-321).

        Keyword-only `options` choose the device algorithm; see
        `bodge_amd.observables.free_energy`.
        """
        from .observables import free_energy

        return free_energy(self, temperature, **options)

    def lowest_eigenvalues(self, k: int = 1, **options) -> Matrix:
        """The k smallest distinct positive eigenvalues (the gap and the levels above it) by a
        Lanczos process on the GPU - for lattices where `diagonalize()` does not fit.  Not part
        of the reference API; see `bodge_amd.observables.lowest_eigenvalues`."""
        from .observables import lowest_eigenvalues

        return lowest_eigenvalues(self, k, **options)

    def lowest_eigenpairs(self, k: int = 1, **options):
        """The k lowest positive eigenvalues with multiplicities and their eigenvectors (layouts of
        `diagonalize`), by two passes of the device Lanczos process - `E, v = diagonalize()` cut to
        its first k states, for lattices where the dense solve does not fit.  Not part of the
        reference API; see `bodge_amd.observables.lowest_eigenpairs`."""
        from .observables import lowest_eigenpairs

        return lowest_eigenpairs(self, k, **options)

    def ldos(self, site: Coord, energies, **options) -> Matrix:
        """Local density of states at `site` for the given energies (ref :324-387)."""
        from .observables import ldos

        return ldos(self, site, energies, **options)


# ---------------------------------------------------------------------------
# Order-parameter helpers (reference hamiltonian.py:390-531).  Pure 2x2 host
# algebra evaluated once per term while the matrix is being assembled.
def swave() -> Callable:
    """Spin structure of singlet s-wave pairing: always iσ2."""

    def spin_structure(*_):
        return jσ2

    return spin_structure


def pwave(dvector: str) -> Callable:
    """Spin structure of triplet p-wave pairing for a d-vector expression.

    `dvector` is an expression in e_x, e_y, e_z (spin axes), p_x, p_y, p_z
    (momentum axes) and their j-prefixed imaginary versions, e.g.
    "(p_x + jp_y) * (e_x + je_y)".  The result maps a bond (i, j) to
    [d(δ)·σ] iσ2 / 2 with δ = j - i.
    """
    names = {}
    for axis, label in enumerate("xyz"):
        unit = np.zeros((3, 1))
        unit[axis, 0] = 1.0
        names[f"e_{label}"] = unit
        names[f"je_{label}"] = 1j * unit
        names[f"p_{label}"] = unit.T
        names[f"jp_{label}"] = 1j * unit.T
    tensor = np.asarray(eval(dvector, {"__builtins__": {}}, names), dtype=np.complex128)
    if tensor.shape != (3, 3):
        raise ValueError("d-vector expression must combine one spin and one momentum factor")

    # gap[p] = Σ_k tensor[k, p] σ_k iσ2 / 2, contracted with the bond direction later.
    gap = np.einsum("kp,kab,bc->pac", tensor, σ, jσ2) / 2

    def spin_structure(i: Coord, j: Coord) -> Matrix:
        step = np.subtract(j, i)
        return np.tensordot(step, gap, axes=(-1, 0))

    return spin_structure


def dwave() -> Callable:
    """Spin structure of d_{x²-y²} singlet pairing: (δx² - δy²)/|δ|² · iσ2.

    Accepts single coordinates or arrays of coordinates (..., 3); in the latter
    case the result has shape (..., 2, 2).
    """

    def spin_structure(i, j) -> Matrix:
        step = np.subtract(j, i)
        weight = (step[..., 0] ** 2 - step[..., 1] ** 2) / (np.sum(step**2, axis=-1) + 1e-16)
        return np.multiply.outer(weight, jσ2)

    return spin_structure


def ssd(system: Hamiltonian) -> Callable:
    """Sine-squared deformation envelope φ(i, j) ∈ [0, 1] for `system`'s lattice."""
    centre = (np.array(system.lattice.shape, dtype=float) - 1) / 2
    radius = float(np.linalg.norm(centre))

    def envelope(i: Coord, j: Coord):
        midpoint = (np.asarray(i, dtype=float) + np.asarray(j, dtype=float)) / 2 - centre
        distance = np.linalg.norm(midpoint, axis=-1)
        return 0.5 * (1 + np.cos(π * distance / (radius + 0.5)))

    return envelope
