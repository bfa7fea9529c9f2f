"""Row-slab domain decomposition of the BSR Hamiltonian (BASELINE config 4).

The lattice index puts x slowest (reference lattice.py:108), so cutting the
block rows into contiguous ranges at plane boundaries gives every rank a slab
of whole x-planes; the columns a slab touches outside its own rows are the
neighbouring planes (and, with periodic edge terms, the planes at the far end).

Everything here is host-side index bookkeeping in numpy.  Each rank derives its
own plan from the replicated global index arrays - no communication is needed
to set the exchange up:

    own rows        global [row0, row1)           -> local rows / columns 0 .. n_own-1
    halo columns    sorted global ids it reads    -> local columns n_own .. n_cols-1
    send lists      for each peer, which of my local rows that peer reads
    recv slices     for each peer, where its rows land in my halo region
                    (contiguous, because owners are contiguous global ranges)

The device side (`bdg_create_slab`, `bdg_slab_set_exchange`, halo pack/unpack
kernels, RCCL send/recv or same-process copies) consumes exactly these arrays.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def partition_rows(n_rows: int, n_parts: int, granule: int = 1) -> np.ndarray:
    """Boundaries (n_parts+1,) of contiguous row ranges, cut at multiples of `granule`.

    For a cubic lattice pass granule = Ly*Lz so that every part is a stack of
    whole x-planes (parts differ by at most one plane).
    """
    if n_rows % granule:
        raise ValueError("granule must divide the number of rows")
    units = n_rows // granule
    if n_parts < 1 or n_parts > units:
        raise ValueError(f"cannot cut {units} planes into {n_parts} slabs")
    base, extra = divmod(units, n_parts)
    sizes = np.full(n_parts, base, dtype=np.int64)
    sizes[:extra] += 1
    return np.concatenate([[0], np.cumsum(sizes)]) * granule


@dataclass
class SlabPlan:
    rank: int
    n_ranks: int
    row0: int
    row1: int
    indptr: np.ndarray  # int32 (n_own + 1)
    indices: np.ndarray  # int32, local column ids
    data: np.ndarray  # complex128 (nnzb_local, 4, 4)
    col_global: np.ndarray  # int64 (n_cols,): global block row behind every local column
    peers: list[int] = field(default_factory=list)
    send_rows: list[np.ndarray] = field(default_factory=list)  # per peer: my local rows it needs
    recv_offset: list[int] = field(default_factory=list)  # per peer: first local column of its rows
    recv_count: list[int] = field(default_factory=list)

    @property
    def n_own(self) -> int:
        return self.row1 - self.row0

    @property
    def n_cols(self) -> int:
        return int(self.col_global.size)

    @property
    def halo_rows(self) -> int:
        return self.n_cols - self.n_own


def _needs(indptr, indices, bounds, q):
    """Sorted global columns that part q reads outside its own row range."""
    lo, hi = int(bounds[q]), int(bounds[q + 1])
    cols = indices[indptr[lo] : indptr[hi]]
    outside = cols[(cols < lo) | (cols >= hi)]
    return np.unique(outside).astype(np.int64)


def build_plan(indptr, indices, data, bounds, rank: int, self_exchange: bool = False) -> SlabPlan:
    """Local matrix and exchange lists of `rank` for the row partition `bounds`.

    self_exchange=True (testing aid) keeps a single-rank plan's periodic-wrap
    columns in the halo region and lists the rank as its own peer, so that the
    pack / send / recv / unpack machinery runs even with one rank.
    """
    indptr = np.asarray(indptr)
    indices = np.asarray(indices)
    bounds = np.asarray(bounds, dtype=np.int64)
    n_ranks = len(bounds) - 1
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n_own = hi - lo

    k0, k1 = int(indptr[lo]), int(indptr[hi])
    cols = indices[k0:k1].astype(np.int64)
    if self_exchange:
        if n_ranks != 1:
            raise ValueError("self_exchange is a single-rank testing mode")
        # pretend the first and last quarter of the rows live on "another" rank when
        # they are referenced from the opposite end (periodic wrap blocks)
        owner_rows = np.repeat(np.arange(lo, hi), np.diff(indptr[lo : hi + 1]))
        far = np.abs(cols - owner_rows) > n_own // 2
        halo = np.unique(cols[far])
    else:
        far = (cols < lo) | (cols >= hi)
        halo = np.unique(cols[far])
    col_global = np.concatenate([np.arange(lo, hi, dtype=np.int64), halo])

    local_cols = np.where(far, n_own + np.searchsorted(halo, cols), cols - lo)
    plan = SlabPlan(
        rank=rank, n_ranks=n_ranks, row0=lo, row1=hi,
        indptr=(indptr[lo : hi + 1] - k0).astype(np.int32),
        indices=local_cols.astype(np.int32),
        data=np.ascontiguousarray(np.asarray(data)[k0:k1]),
        col_global=col_global,
    )

    if self_exchange:
        plan.peers = [0]
        plan.send_rows = [(halo - lo).astype(np.int64)]
        plan.recv_offset = [n_own]
        plan.recv_count = [int(halo.size)]
        return plan

    owner_of_halo = np.searchsorted(bounds, halo, side="right") - 1
    for q in range(n_ranks):
        if q == rank:
            continue
        theirs = _needs(indptr, indices, bounds, q)
        to_send = theirs[(theirs >= lo) & (theirs < hi)] - lo
        mine = np.nonzero(owner_of_halo == q)[0]
        if to_send.size == 0 and mine.size == 0:
            continue
        plan.peers.append(q)
        plan.send_rows.append(to_send.astype(np.int64))
        plan.recv_offset.append(n_own + (int(mine[0]) if mine.size else 0))
        plan.recv_count.append(int(mine.size))
    return plan


def lattice_granule(lattice) -> int:
    """Rows per x-plane of a CubicLattice (1 for anything else)."""
    shape = getattr(lattice, "shape", None)
    return int(shape[1] * shape[2]) if shape is not None and len(shape) == 3 else 1
