"""pytest configuration: markers, import path, golden fixtures."""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


# wall-clock start of this test session (the dense-ladder tests budget their wait against it)
os.environ["BODGE_AMD_TEST_SESSION_START"] = repr(__import__("time").time())


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


class Golden:
    """Values recorded from the reference by tests/golden/make_golden.py."""

    def __init__(self):
        here = os.path.join(ROOT, "tests", "golden")
        with open(os.path.join(here, "reference_values.json")) as fh:
            self.values = json.load(fh)
        self.arrays = np.load(os.path.join(here, "reference_arrays.npz"))

    def free_energy(self, name, temperature):
        return self.values[name]["free_energy"][repr(float(temperature))]

    def eigenvalues(self, name):
        return self.arrays[f"{name}/eigenvalues"]

    def ldos(self, name, n):
        return self.arrays[f"{name}/ldos{n}"]


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def api():
    import bodge_amd

    return bodge_amd


@pytest.fixture(scope="session")
def hip_library():
    """Build (if stale) and load the HIP library; used by CPU ABI tests and all GPU tests."""
    from bodge_amd import backend, build

    build.build_library()
    return backend.load()


class Knobs:
    """Run-time switches of the library for one test, through `bdg_set_option` (a process-wide
    override table inside the library) and never through `os.environ`: `setenv` next to the
    `getenv` of a library call running on another host thread is undefined behaviour."""

    def __init__(self):
        from bodge_amd import backend

        self._backend = backend
        self._before = {}  # name -> the override in force before this test touched it (None = none)

    def set(self, name, value):
        self._before.setdefault(name, self._backend.get_option(name))
        self._backend.set_option(name, value)

    def unset(self, name):
        self._before.setdefault(name, self._backend.get_option(name))
        self._backend.set_option(name, None)

    def update(self, values):
        for name, value in values.items():
            self.set(name, value)

    def clear(self):
        for name, value in self._before.items():  # back to what was there (a process-wide set_option survives a test)
            self._backend.set_option(name, value)
        self._before.clear()


@pytest.fixture
def knobs(hip_library):
    table = Knobs()
    yield table
    table.clear()


@pytest.fixture(scope="session", autouse=True)
def _dense_library_prefetch(request):
    """On a GPU box, start reading the 931 MB rocSOLVER object at session start (background file
    I/O in the library, `bdg_dense_prefetch`): the dense-ladder tests sort last, so on a fresh
    machine the cold read overlaps with the rest of the suite instead of adding to it."""
    if "not gpu" in (request.config.getoption("-m") or ""):
        return
    try:
        from bodge_amd import backend, solver

        if backend.device_count() > 0:
            solver.prefetch_rccl_library()   # (read first: the communicator tests have no other route)
            solver.prefetch_dense_library()  # rocSOLVER: an optional cross-check since the own tridiagonalisation route
    except Exception:  # library not built yet: the tests that need it say so themselves
        pass


def wait_for_library(request, ready, name: str, budget_s: float = 780.0) -> None:
    """Wait for a background library read with a progress line every 30 s (a long silent wait looks
    like a hang to whoever runs the suite); skip, loudly, if it has not arrived `budget_s` into the
    session rather than let the run's limit kill every other result with it."""
    import time

    start = float(os.environ.get("BODGE_AMD_TEST_SESSION_START", time.time()))
    capture = request.config.pluginmanager.getplugin("capturemanager")
    waited = time.time()
    while not ready(0.0):
        elapsed = time.time() - start
        if elapsed > budget_s:
            pytest.skip(f"{name} not read from cold storage after {elapsed:.0f} s of this session")
        if ready(min(30.0, budget_s - elapsed)):
            break
        with capture.global_and_fixture_disabled():
            print(f"\n[{name}] waiting for the shared object to arrive from cold storage: "
                  f"{time.time() - waited:.0f} s so far, session at {time.time() - start:.0f} s", flush=True)


@pytest.fixture(scope="session")
def rccl_library(request):
    """RCCL's 573 MB object in the page cache (tests that create a communicator)."""
    from bodge_amd import solver

    solver.prefetch_rccl_library()
    wait_for_library(request, solver.rccl_library_ready, "librccl.so")
