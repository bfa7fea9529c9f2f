"""Host-side slab decomposition (no GPU): the local matrices plus the exchange lists must
reproduce the global matrix-vector product, including periodic wrap-around halos."""

import numpy as np
import pytest
import scipy.sparse as sp

import systems
from bodge_amd import slab


def _emulate_spmv(indptr, indices, data, bounds, x, self_exchange=False):
    """y = H x assembled from per-rank local products with halos filled through the plans."""
    n_ranks = len(bounds) - 1
    plans = [slab.build_plan(indptr, indices, data, bounds, r, self_exchange) for r in range(n_ranks)]
    xs = []
    for p in plans:
        local = np.zeros((p.n_cols, 4), dtype=complex)
        local[: p.n_own] = x.reshape(-1, 4)[p.row0 : p.row1]
        xs.append(local)
    for p in plans:  # every rank packs what each peer asked for; the peer unpacks at its offset
        for peer, rows in zip(p.peers, p.send_rows):
            q = plans[peer]
            slot = q.peers.index(p.rank)
            assert q.recv_count[slot] == rows.size
            dst = slice(q.recv_offset[slot], q.recv_offset[slot] + rows.size)
            assert np.array_equal(q.col_global[dst], rows + p.row0)
            xs[peer][dst] = xs[p.rank][rows]
    y = np.empty_like(x)
    for p, local in zip(plans, xs):
        mat = sp.bsr_matrix((p.data, p.indices, p.indptr), shape=(4 * p.n_own, 4 * p.n_cols))
        y[4 * p.row0 : 4 * p.row1] = mat @ local.reshape(-1)
    return y, plans


@pytest.mark.parametrize("name,n_ranks", [("random357", 3), ("random357", 1), ("dwave8", 4), ("dwave8", 8),
                                          ("swave20", 5), ("chain128", 8)])
def test_slabs_reproduce_global_spmv(api, name, n_ranks):
    spec = systems.CATALOG[name]
    system = spec["build"](api, **spec["kwargs"])
    indptr, indices, data = system.bsr_arrays()
    bounds = slab.partition_rows(system.lattice.size, n_ranks, slab.lattice_granule(system.lattice))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(system.shape[0]) + 1j * rng.standard_normal(system.shape[0])
    y, plans = _emulate_spmv(indptr, indices, data, bounds, x)
    assert np.allclose(y, system.matrix("bsr") @ x, rtol=1e-13, atol=1e-13)
    assert sum(p.n_own for p in plans) == system.lattice.size
    if n_ranks > 1:
        plane = slab.lattice_granule(system.lattice)
        periodic = name == "random357"
        for p in plans:
            inner = 0 < p.rank < n_ranks - 1
            expected = 2 * plane if (inner or periodic) else plane
            assert p.halo_rows == expected  # one plane per x-neighbour (wrap included when periodic)


def test_self_exchange_mode_routes_wrap_blocks_through_the_halo(api):
    system = systems.random_periodic(api, shape=(6, 4, 3), seed=5)
    indptr, indices, data = system.bsr_arrays()
    bounds = np.array([0, system.lattice.size])
    rng = np.random.default_rng(1)
    x = rng.standard_normal(system.shape[0]) + 0j
    y, plans = _emulate_spmv(indptr, indices, data, bounds, x, self_exchange=True)
    assert plans[0].halo_rows == 2 * 12 and plans[0].peers == [0]
    assert np.allclose(y, system.matrix("bsr") @ x, rtol=1e-13, atol=1e-13)


def test_partition_rows():
    assert slab.partition_rows(100, 8, 10).tolist() == [0, 20, 40, 50, 60, 70, 80, 90, 100]
    assert slab.partition_rows(12, 3).tolist() == [0, 4, 8, 12]
    with pytest.raises(ValueError):
        slab.partition_rows(100, 11, 10)
    with pytest.raises(ValueError):
        slab.partition_rows(101, 2, 10)
