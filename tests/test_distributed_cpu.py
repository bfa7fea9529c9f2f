"""N > 1 path on the CPU: two gloo ranks must reproduce the single-rank moments and F."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import systems
from bodge_amd import chebyshev
from bodge_amd.observables import shard_vectors
from oracle import cheb_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeComm:
    def __init__(self, rank, n_ranks):
        self.rank, self.n_ranks = rank, n_ranks


@pytest.mark.parametrize("total,world", [(64, 8), (10, 4), (7, 2), (8, 8), (5, 1)])
def test_shards_partition_the_vectors(total, world):
    spans = [shard_vectors(total, FakeComm(r, world)) for r in range(world)]
    ids = [i for first, count in spans for i in range(first, first + count)]
    assert ids == list(range(total))
    counts = [c for _, c in spans]
    assert max(counts) - min(counts) <= 1
    assert shard_vectors(total, None) == (0, total)
    with pytest.raises(ValueError):
        shard_vectors(2, FakeComm(3, 4))


def _free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_gloo_ranks_match_single_rank(api, tmp_path):
    total, moments = 6, 64
    out = tmp_path / "result.json"
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(total), str(moments),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert proc.returncode == 0, proc.stderr[-2000:]
    got = json.loads(out.read_text())
    assert got["world"] == 2
    spans = [json.loads((tmp_path / f"result.json.rank{r}").read_text()) for r in range(2)]
    assert [(s["first"], s["count"]) for s in spans] == [(0, 3), (3, 3)]

    system = systems.swave_square(api, L=10, zeeman=0.05)
    bsr = system.matrix("bsr")
    indptr, _, data = system.bsr_arrays()
    scale = chebyshev.spectral_bound(indptr, data)
    start = cheb_ref.random_block(bsr.shape[0], 0, range(total), cheb_ref.VEC_Z4)
    mu = cheb_ref.moments(bsr, scale, moments, start).sum(axis=1) / total
    assert np.allclose(got["mu"], mu, rtol=0, atol=1e-12 * bsr.shape[0])
    assert np.isclose(got["free_energy"], chebyshev.free_energy_series(mu, scale, 0.5), rtol=1e-12)


@pytest.mark.timeout(600)
def test_two_gloo_ranks_slab_exchange_matches_single_rank(api, tmp_path):
    """Slab mode on 2 CPU ranks: the product's SlabPlan drives gloo halo exchange; moments must
    equal the undivided oracle run."""
    total, moments = 3, 48
    out = tmp_path / "slab.json"
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(total), str(moments), "slab",
    ]
    proc = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True,
                          text=True, timeout=540)
    assert proc.returncode == 0, proc.stderr[-2000:]
    got = json.loads(out.read_text())
    ranks = [json.loads((tmp_path / f"slab.json.rank{r}").read_text()) for r in range(2)]
    assert [r["rows"] for r in ranks] == [[0, 36], [36, 72]]
    assert all(r["halo"] == 24 and r["peers"] == [1 - i] for i, r in enumerate(ranks))

    system = systems.random_periodic(api, shape=(6, 4, 3), seed=5)
    bsr = system.matrix("bsr")
    start = cheb_ref.random_block(bsr.shape[0], 0, range(total), cheb_ref.VEC_Z4)
    mu = cheb_ref.moments(bsr, got["scale"], moments, start).sum(axis=1) / total
    assert np.allclose(got["mu"], mu, rtol=0, atol=1e-12 * bsr.shape[0])


@pytest.mark.timeout(600)
def test_rccl_rendezvous_under_torchrun(tmp_path):
    """The unique-id hand-off that precedes ncclCommInitRank, run by 4 real ranks started the way
    the driver starts bench.py (torch.distributed.run): every rank must receive rank 0's 128 bytes."""
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_rdzv_worker.py"), str(tmp_path),
    ]
    env = dict(os.environ)
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert proc.returncode == 0, proc.stderr[-2000:]
    ranks = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(4)]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and all(r["world"] == 4 and r["len"] == 128 for r in ranks)
    assert [r["device"] for r in ranks] == [0, 1, 2, 3]  # LOCAL_RANK -> device
    assert len({r["uid"] for r in ranks}) == 1, "ranks disagree on the unique id"
    assert len({r["ppid"] for r in ranks}) == 1  # the launch nonce relies on a common parent on one node
    assert sorted(os.listdir(tmp_path)) == sorted([f"arrived{r}" for r in range(4)] + [f"rank{r}.json" for r in range(4)]), \
        "the hand-off must not leave files behind (it goes over TCP)"


@pytest.mark.timeout(600)
def test_bench_host_fallback_when_rccl_is_unavailable_and_allowed(tmp_path):
    """bench.py --allow-gloo: if the RCCL communicator cannot be built, the ranks agree on it and
    do the moment sum / timing max on the host instead of dying (the JSON line then says so)."""
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_fallback_worker.py"), str(tmp_path),
    ]
    env = dict(os.environ)
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert proc.returncode == 0, proc.stderr[-2000:]
    for r in range(2):
        rec = json.loads((tmp_path / f"rank{r}.json").read_text())
        assert rec["sum"] == [1.0, 3.0, 5.0, 7.0] and rec["max"] == [1.0]
        assert rec["description"].startswith("host fallback") and "simulated" in rec["description"]


@pytest.mark.timeout(600)
def test_bench_refuses_to_report_without_rccl(tmp_path):
    """Without --allow-gloo a failed RCCL initialisation ends the launch with a non-zero exit code
    and no JSON line: a scaling number that did not go through RCCL can never be recorded."""
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_fallback_worker.py"), str(tmp_path), "strict",
    ]
    proc = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=540)
    assert proc.returncode != 0
    assert "could not be created on every rank" in proc.stderr and not os.listdir(tmp_path)


def test_rendezvous_store_binds_to_the_master_address_and_keys_are_write_once():
    """the store listens on MASTER_ADDR only (loopback for a one-node launch) and refuses
    a second `set` of a key, so a stray peer can neither reach it from outside nor replace an id."""
    import threading

    from bodge_amd.rendezvous import Store

    port = _free_port()
    stores = {}

    def rank(r):
        stores[r] = Store(r, 2, addr="127.0.0.1", base_port=port, nonce="test-launch", timeout=30.0)

    threads = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    server = stores[0]._server
    assert server.server_address[0] == "127.0.0.1"
    stores[0].set("id", b"abc")
    assert stores[1].get("id", 5.0) == b"abc"
    with pytest.raises(RuntimeError):
        stores[1].set("id", b"evil")
    assert stores[0].get("id", 5.0) == b"abc"
    gathered = {}
    workers = [threading.Thread(target=lambda r=r: gathered.__setitem__(r, stores[r].gather(bytes([r])))) for r in (0, 1)]
    for t in workers:
        t.start()
    for t in workers:
        t.join()
    assert gathered[0] == gathered[1] == [b"\x00", b"\x01"]
    with pytest.raises(RuntimeError):  # a client of another launch finds no server that answers its nonce
        Store(1, 2, addr="127.0.0.1", base_port=port, nonce="another-launch", timeout=0.5)
    server.shutdown()
