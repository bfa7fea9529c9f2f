"""The C-ABI shared library: builds for gfx950, loads, exports what the header declares.

No compute calls here (this file runs without a GPU); parity is in test_gpu_*.py.
"""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "bodge_hip.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(bdg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_library):
    from bodge_amd import backend

    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(hip_library, name), f"{name} declared in bodge_hip.h but not exported"
    assert sorted(backend.SIGNATURES) == declared, "ctypes table and header disagree"
    assert b"gfx950" in hip_library.bdg_version()


def test_perf_struct_layout_matches_header():
    from bodge_amd import backend

    with open(os.path.join(ROOT, "include", "bodge_hip.h")) as fh:
        body = re.search(r"typedef struct bdg_perf \{(.*?)\} bdg_perf;", fh.read(), flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(double|int64_t|int32_t)\s+(\w+);", body)
    ctype = {"double": ctypes.c_double, "int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32}
    assert [(n, ctype[t]) for t, n in fields] == list(backend.Perf._fields_)


def test_argument_errors_do_not_need_a_gpu(hip_library):
    from bodge_amd import backend

    count = ctypes.c_int(-1)
    assert hip_library.bdg_device_count(ctypes.byref(count)) == 0 and count.value >= 0
    handle = ctypes.c_void_p()
    indptr = np.array([0, 2, 1], dtype=np.int32)  # not monotone
    indices = np.zeros(2, dtype=np.int32)
    data = np.zeros(64)
    rc = hip_library.bdg_create(0, 2, 1, backend.as_i32p(indptr), backend.as_i32p(indices),
                                backend.as_f64p(data), ctypes.byref(handle))
    assert rc == -1 and b"indptr" in hip_library.bdg_last_error()
    assert hip_library.bdg_destroy(None) == 0
    assert hip_library.bdg_spmv(None, None, None) == -1


def test_product_path_fails_loudly_without_gpu(api, hip_library):
    """No CPU fallback: without a device the observables raise, they do not compute."""
    from bodge_amd import backend

    if backend.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device behaviour cannot be exercised here")
    import systems

    system = systems.swave_square(api, L=4)
    for call in (lambda: system.free_energy(0.1), lambda: system.diagonalize(),
                 lambda: system.ldos((1, 1, 0), [0.0, 0.1])):
        with pytest.raises(RuntimeError, match="GPU|HIP"):
            call()


def test_package_does_not_import_the_oracle():
    import subprocess
    import sys

    code = "import sys, bodge_amd, bodge_amd.observables, bodge_amd.solver; print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "False"
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bodge_amd")):
        for name in files:
            if name.endswith(".py"):
                with open(os.path.join(dirpath, name)) as fh:
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", fh.read(), flags=re.M), name


def test_c_program_links_against_the_header_and_library(tmp_path, hip_library):
    """The boundary from C: tests/c/abi_smoke.c is compiled with gcc (C99, -Wall -Werror) against
    include/bodge_hip.h, linked with the built library and run (no GPU needed for what it calls)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from bodge_amd import build

    lib_dir = os.path.dirname(build.LIBRARY)
    exe = tmp_path / "abi_smoke"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", f"-I{root}/include", os.path.join(root, "tests", "c", "abi_smoke.c"),
                    f"-L{lib_dir}", "-lbodge_hip", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bodge_hip" in out.stdout and "indptr" in out.stdout
