"""ctypes binding of `libbodge_hip.so` (C ABI: `include/bodge_hip.h`).

There is no CPU implementation behind this module.  If the shared library has
not been built, cannot be loaded, or no GPU is visible, the calls raise
`RuntimeError` - the same exception class the reference raises when its GPU
backend is missing (reference hamiltonian.py:222-225, :296-299).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

LIBRARY_PATH = os.environ.get(  # override: A/B experiments with a second build of the same source
    "BODGE_AMD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libbodge_hip.so")
)

VEC_RADEMACHER = 0
VEC_Z4 = 1

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_handle = C.c_void_p


class Perf(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("launches", C.c_int64),
        ("vector_steps", C.c_int64),
        ("bytes_per_launch", C.c_double),
        ("lanes_per_row", C.c_int32),
        ("vectors_per_launch", C.c_int32),
        ("grid", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("pipelined", C.c_int32),
        ("real_arithmetic", C.c_int32),
        ("strip_rows", C.c_int32),
        ("ph_packed", C.c_int32),
        ("dict_blocks", C.c_int32),
        ("steps_per_launch", C.c_int32),
        ("rolling", C.c_int32),
        ("dict_skipped", C.c_int32),
        ("onsite_streamed", C.c_int32),
        ("streams", C.c_int32),
        ("bytes_moved", C.c_double),
        ("window_ms", C.c_double),
        ("sweeps", C.c_int64),
        ("persistent", C.c_int32),
        ("groups_per_launch", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol declared in include/bodge_hip.h
SIGNATURES = {
    "bdg_last_error": (C.c_char_p, []),
    "bdg_version": (C.c_char_p, []),
    "bdg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bdg_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.POINTER(_handle)]),
    "bdg_create_slab": (
        C.c_int,
        [C.c_int, C.c_int64, C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int64, C.POINTER(_handle)],
    ),
    "bdg_slab_set_exchange": (C.c_int, [_handle, _handle, C.c_int32, _i32p, _i64p, _i64p, _i64p, _i64p]),
    "bdg_group_create": (C.c_int, [C.POINTER(_handle), C.c_int32, C.POINTER(_handle)]),
    "bdg_group_destroy": (C.c_int, [_handle]),
    "bdg_group_dots_random": (
        C.c_int,
        [_handle, C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, _f64p, _f64p],
    ),
    "bdg_group_dots_unit": (C.c_int, [_handle, C.c_double, C.c_int32, C.c_int32, _i64p, _f64p, _f64p]),
    "bdg_destroy": (C.c_int, [_handle]),
    "bdg_spmv": (C.c_int, [_handle, _f64p, _f64p]),
    "bdg_cheb_dots_random": (
        C.c_int,
        [_handle, C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, _f64p, _f64p],
    ),
    "bdg_cheb_dots_unit": (C.c_int, [_handle, C.c_double, C.c_int32, C.c_int32, _i64p, _f64p, _f64p]),
    "bdg_cheb_moments": (
        C.c_int,
        [_handle, _handle, C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, _f64p],
    ),
    "bdg_cheb_diag_moments": (C.c_int, [_handle, C.c_double, C.c_int32, C.c_int32, _i64p, _f64p]),
    "bdg_lanczos_begin": (C.c_int, [_handle, C.c_int32, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]),
    "bdg_lanczos_advance": (C.c_int, [_handle, C.c_int32, _f64p, _f64p]),
    "bdg_lanczos_ritz_vectors": (C.c_int, [_handle, C.c_int32, C.c_int32, _f64p, _f64p]),
    "bdg_lanczos_ritz_pairs": (C.c_int, [_handle, C.c_int32, C.c_int32, _f64p, _f64p, C.c_double, C.c_int32,
                                        C.POINTER(C.c_int32), _f64p, _f64p]),
    "bdg_random_vector": (C.c_int, [_handle, C.c_uint64, C.c_uint64, C.c_int32, _f64p]),
    "bdg_eigh_dense": (C.c_int, [_handle, _f64p, _f64p]),
    "bdg_eigh_dense_above": (C.c_int, [_handle, C.c_double, C.c_int64, _f64p, C.POINTER(C.c_int64), _f64p]),
    "bdg_hermiticity_defect": (C.c_int, [_handle, _f64p]),
    "bdg_dense_prefetch": (C.c_int, []),
    "bdg_dense_prefetch_wait": (C.c_int, [C.c_double, C.POINTER(C.c_int32)]),
    "bdg_rccl_prefetch": (C.c_int, []),
    "bdg_rccl_prefetch_wait": (C.c_int, [C.c_double, C.POINTER(C.c_int32)]),
    "bdg_perf_query": (C.c_int, [_handle, C.POINTER(Perf)]),
    "bdg_set_lattice_shape": (C.c_int, [_handle, C.c_int32, C.c_int32, C.c_int32]),
    "bdg_set_lanes_per_row": (C.c_int, [_handle, C.c_int32]),
    "bdg_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "bdg_comm_unique_id": (C.c_int, [_u8p]),
    "bdg_host_fill_terms": (C.c_int, [_f64p, C.c_int64, _i64p, C.c_int64, _f64p, C.c_int, C.c_int, _u8p]),
    "bdg_host_scan_blocks": (C.c_int, [_f64p, _i32p, C.c_int64, _u8p, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "bdg_host_compact_blocks": (C.c_int, [_f64p, _i32p, _i32p, C.c_int64, _u8p, _f64p, _i32p, _i32p]),
    "bdg_comm_init": (C.c_int, [C.c_int, _u8p, C.c_int32, C.c_int32, C.POINTER(_handle)]),
    "bdg_comm_info": (C.c_int, [_handle, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p]),
    "bdg_comm_allreduce_sum": (C.c_int, [_handle, _f64p, C.c_int64]),
    "bdg_comm_allreduce_max": (C.c_int, [_handle, _f64p, C.c_int64]),
    "bdg_comm_destroy": (C.c_int, [_handle]),
}

_lib = None


def load():
    """Load the shared library once; raise RuntimeError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise RuntimeError(
            f"HIP library not built: {LIBRARY_PATH} is missing. Run `python3 -m bodge_amd.build`. "
            "This package has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIBRARY_PATH)
    except OSError as exc:
        raise RuntimeError(f"HIP library could not be loaded: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != 0:
        message = load().bdg_last_error().decode("utf-8", "replace")
        if status == -1:
            raise ValueError(f"bodge_hip: {message}")
        raise RuntimeError(f"bodge_hip: {message}")


def device_count() -> int:
    n = C.c_int(0)
    check(load().bdg_device_count(C.byref(n)))
    return n.value


def require_device() -> None:
    if device_count() < 1:
        raise RuntimeError(
            "No HIP device is visible: bodge_amd computes observables on the GPU only."
        )


def as_f64p(array: np.ndarray):
    return array.ctypes.data_as(_f64p)


def as_i32p(array: np.ndarray):
    return array.ctypes.data_as(_i32p)


def as_i64p(array: np.ndarray):
    return array.ctypes.data_as(_i64p)


def as_u8p(array: np.ndarray):
    return array.ctypes.data_as(_u8p)


_overrides: dict[str, str] = {}  # what set_option has set in this process (the library keeps no getter)


def set_option(name: str, value) -> None:
    """Override one BODGE_AMD_* switch of the library for this process (`None` removes the override).

    The library looks here before the environment variable of the same name; unlike changing
    `os.environ` this is safe while other host threads are inside library calls."""
    check(load().bdg_set_option(name.encode(), None if value is None else str(value).encode()))
    if value is None:
        _overrides.pop(name, None)
    else:
        _overrides[name] = str(value)


def get_option(name: str):
    """The override set for `name` in this process, or None (the environment variable, if any, then shows)."""
    return _overrides.get(name)


class options:
    """`with backend.options(BODGE_AMD_SWEEP="0", ...):` - switches set for the block; on leaving it every one goes back
    to what it was before (an enclosing block's value, a process-wide `set_option`, or no override)."""

    def __init__(self, **values):
        self._values = values
        self._before = {}

    def __enter__(self):
        self._before = {name: _overrides.get(name) for name in self._values}
        for name, value in self._values.items():
            set_option(name, value)
        return self

    def __exit__(self, *exc):
        for name, value in self._before.items():
            set_option(name, value)
