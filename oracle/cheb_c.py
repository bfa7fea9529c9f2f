"""ctypes wrapper of oracle/cheb_c.c (C + OpenMP restatement of the recurrence).

TEST INFRASTRUCTURE ONLY - see oracle/__init__.py.  `build()` compiles the shared object into
oracle/_build/ (git-ignored; it travels with the tree like the HIP library).
"""

from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(HERE, "cheb_c.c")
LIBRARY = os.path.join(HERE, "_build", "libcheb_c.so")
_lib = None


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIBRARY) and os.path.getmtime(LIBRARY) >= os.path.getmtime(SOURCE):
        return LIBRARY
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found")
    os.makedirs(os.path.dirname(LIBRARY), exist_ok=True)
    tmp = f"{LIBRARY}.{os.getpid()}.tmp"
    subprocess.run([gcc, "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", tmp, SOURCE], check=True)
    os.replace(tmp, LIBRARY)
    return LIBRARY


def load():
    global _lib
    if _lib is None:
        build()
        # (No OMP_PROC_BIND here: libgomp would pin the calling Python thread to one core as well, and
        # every BLAS thread and forked worker of the process inherits that mask - measured 10x slower
        # test runs.  Threads stay unpinned; the scheduler keeps busy threads where they are.)
        _lib = C.CDLL(LIBRARY)
        _lib.cheb_c_copy_rows.restype = None
        _lib.cheb_c_copy_rows.argtypes = [C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
        f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        for name in ("cheb_c_step_complex", "cheb_c_step_real"):
            fn = getattr(_lib, name)
            fn.restype = None
            fn.argtypes = [C.c_int64, i32p, i32p, f64p, C.c_int, C.c_double, f64p, f64p, f64p, f64p]
        _lib.cheb_c_threads.restype = C.c_int
        _lib.cheb_c_set_threads.argtypes = [C.c_int]
        _lib.cheb_c_pin_threads.restype = C.c_int
        _lib.cheb_c_pin_threads.argtypes = [C.POINTER(C.c_int), C.c_int]
        _lib.cheb_c_unpin_threads.restype = None
    return _lib


def threads() -> int:
    return load().cheb_c_threads()


def set_threads(n: int) -> None:
    load().cheb_c_set_threads(int(n))


def _cpu_list(text: str) -> list[int]:
    cpus = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def spread_cpus(count: int) -> list[int]:
    """`count` CPUs of this process's affinity mask for a memory-bound team: one per physical core (SMT siblings only
    once the cores are used up), evenly spaced over the cores of ONE memory node as long as that node has enough of them,
    then over the next.  (Measured on a two-socket GPU box, 16 threads of the recurrence: spread over one node 1070-1080
    steps/s, spread over both sockets 178, packed on 8 cores + their siblings 390-410; profiles/r03_numa_probe.log.
    The team's arrays are first touched by these threads, so they land on the chosen node.)"""
    import glob

    allowed = set(os.sched_getaffinity(0))
    nodes = []
    for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda p: int(p.rsplit("node", 1)[1])):
        try:
            with open(path + "/cpulist") as fh:
                cpus = [c for c in _cpu_list(fh.read()) if c in allowed]
        except OSError:
            cpus = []
        if cpus:
            nodes.append(cpus)
    if not nodes:
        nodes = [sorted(allowed)]
    pool = []
    for cpus in nodes:  # per node: physical cores first, their siblings after
        firsts, rest, seen = [], [], set()
        for cpu in cpus:
            try:
                base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
                with open(base + "physical_package_id") as a, open(base + "core_id") as b:
                    key = (a.read().strip(), b.read().strip())
            except OSError:
                key = (cpu,)
            (rest if key in seen else firsts).append(cpu)
            seen.add(key)
        pool.append((firsts, rest))
    chosen = []
    for firsts, _ in pool:  # whole nodes while more than one node's cores are wanted, an even spread over the last one
        want = count - len(chosen)
        if want <= 0:
            break
        if want >= len(firsts):
            chosen += firsts
        else:
            chosen += [firsts[(i * len(firsts)) // want] for i in range(want)]
    for _, rest in pool:  # more threads than physical cores: the siblings
        want = count - len(chosen)
        if want <= 0:
            break
        chosen += rest[:want]
    return chosen[:count]


class pinned_threads:
    """Context: the OpenMP threads bound one per CPU of `spread_cpus(threads())`, previous masks restored on exit
    (the calling thread's included)."""

    def __enter__(self):
        lib = load()
        cpus = spread_cpus(threads())
        self.cpus = cpus
        array = (C.c_int * len(cpus))(*cpus)
        self.bound = lib.cheb_c_pin_threads(array, len(cpus))
        return self

    def __exit__(self, *exc):
        load().cheb_c_unpin_threads()
        return False


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _placed_copy(lib, source: np.ndarray, row_begin: np.ndarray) -> np.ndarray:
    """Copy of `source` whose pages are first written by the OpenMP threads that will process the
    corresponding block rows (row i = bytes row_begin[i] .. row_begin[i+1]): NUMA first touch."""
    source = np.ascontiguousarray(source)
    out = np.empty_like(source)  # large: fresh, untouched pages from mmap
    begin = np.ascontiguousarray(row_begin, dtype=np.int64)
    lib.cheb_c_copy_rows(len(begin) - 1, begin.ctypes.data_as(C.POINTER(C.c_int64)), source.ctypes.data, out.ctypes.data)
    return out


class Recurrence:
    """State of one run: t_cur, t_prev as (4N, R) arrays; `step(coef)` advances in place.

    `numa` = place the pages of the matrix and the vectors by first touch from the compute threads
    (used by the CPU baseline of bench.py; the values are the same either way)."""

    def __init__(self, bsr, start: np.ndarray, real: bool = False, numa: bool = False):
        self.lib = load()
        self.nb = bsr.shape[0] // 4
        self.indptr = np.ascontiguousarray(bsr.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(bsr.indices, dtype=np.int32)
        self.real = real
        if real:
            self.blocks = np.ascontiguousarray(bsr.data.real, dtype=np.float64)
            self.cur = np.array(start.real, dtype=np.float64, order="C")
        else:
            self.blocks = np.ascontiguousarray(bsr.data, dtype=np.complex128)
            self.cur = np.array(start, dtype=np.complex128, order="C")  # private copy: steps overwrite it
        self.prev = np.zeros_like(self.cur)
        self.R = self.cur.shape[1]
        self.numa = bool(numa)
        if numa:
            block_rows = self.indptr.astype(np.int64)
            per_block = self.blocks[0].nbytes if len(self.blocks) else 0
            self.blocks = _placed_copy(self.lib, self.blocks, block_rows * per_block)
            self.indices = _placed_copy(self.lib, self.indices, block_rows * 4)
            vec_rows = np.arange(self.nb + 1, dtype=np.int64) * (4 * self.cur.strides[0])
            self.cur = _placed_copy(self.lib, self.cur, vec_rows)
            self.prev = _placed_copy(self.lib, self.prev, vec_rows)

    def step(self, coef: float):
        d, e = np.empty(self.R), np.empty(self.R)
        fn = self.lib.cheb_c_step_real if self.real else self.lib.cheb_c_step_complex
        view = (lambda a: a) if self.real else (lambda a: a.view(np.float64))
        fn(self.nb, _p(self.indptr, C.c_int32), _p(self.indices, C.c_int32), _p(view(self.blocks), C.c_double),
           self.R, float(coef), _p(view(self.cur), C.c_double), _p(view(self.prev), C.c_double),
           _p(d, C.c_double), _p(e, C.c_double))
        self.cur, self.prev = self.prev, self.cur  # t_next was written over t_prev
        return d, e


def recurrence_dots(bsr, scale: float, n_moments: int, start: np.ndarray, real: bool = False):
    """Same contract as cheb_ref.recurrence_dots, computed by the C code."""
    steps = n_moments // 2
    run = Recurrence(bsr, start, real)
    d, e = np.empty((steps, run.R)), np.empty((steps, run.R))
    for n in range(steps):
        d[n], e[n] = run.step((1.0 if n == 0 else 2.0) / scale)
    return d, e


def step_bytes(bsr, n_vectors: int, real: bool) -> float:
    """Bytes one block-step of this restatement moves at the least: every stored block and index
    once, and per (scalar row, vector) read t_n, read t_{n-1}, write t_{n+1} (SURVEY §8d's count
    for the storage actually used here: full 16-entry blocks, float64 or complex128)."""
    element = 8.0 if real else 16.0
    return (16 * element + 4) * bsr.indices.size + 4.0 * (bsr.shape[0] // 4 + 1) + 3 * element * bsr.shape[0] * n_vectors


def time_recurrence(bsr, scale, start, seconds: float = 8.0, real: bool = False, warmup: int = 2, numa: bool = False,
                    pin: bool = False):
    """(vector_steps_per_second, block_steps_timed, threads) of the OpenMP recurrence on this host.
    `pin`: threads bound one per core, spread evenly over the host, for the duration of the call."""
    if pin:
        with pinned_threads():
            return time_recurrence(bsr, scale, start, seconds, real, warmup, numa)
    run = Recurrence(bsr, start, real, numa=numa)
    run.step(1.0 / scale)
    done, t0 = 0, None
    while True:
        if done == warmup:
            t0 = time.perf_counter()
        run.step(2.0 / scale)
        done += 1
        if t0 is not None and time.perf_counter() - t0 >= seconds:
            break
    elapsed = time.perf_counter() - t0
    return (done - warmup) * run.R / elapsed, done - warmup, threads()
