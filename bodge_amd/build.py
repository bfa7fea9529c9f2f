"""Compile the HIP library for gfx950 in-tree (`bodge_amd/csrc/libbodge_hip.so`).

    python3 -m bodge_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container.  The shared object is git-ignored but travels with the source tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
INCLUDE = os.path.join(os.path.dirname(ROOT), "include")
LIBRARY = os.path.join(CSRC, "libbodge_hip.so")
SOURCES = [os.path.join(CSRC, "bodge_hip.hip")]
HEADERS = [os.path.join(CSRC, name) for name in (
    "kernels.hpp", "sweep.hpp", "twostage.hpp", "host_assembly.hpp", "core.hpp", "plans.hpp", "libraries.hpp", "recurrence.hpp",
    "lanczos.hpp", "dense.hpp", "tridiag.hpp", "knobs.hpp")] + [os.path.join(INCLUDE, "bodge_hip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    found = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(found):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return found


def is_stale() -> bool:
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False, output: str | None = None,
                  defines: tuple[str, ...] = ()) -> str:
    """Compile to `output` (default: the in-tree library).  `defines` are extra -D macros,
    used only for A/B experiments that build a second library next to the product one."""
    target = output or LIBRARY
    if output is None and not force and not is_stale():
        return LIBRARY
    cmd = [
        _hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
        "-Wall", "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC}", *[f"-D{d}" for d in defines],
        "-o", target + f".{os.getpid()}.tmp", *SOURCES, "-ldl",
    ]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{proc.stdout}\n{proc.stderr}")
    if verbose and proc.stderr.strip():
        print(proc.stderr, file=sys.stderr)
    os.replace(target + f".{os.getpid()}.tmp", target)  # atomic: never expose a partial file
    return target


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
