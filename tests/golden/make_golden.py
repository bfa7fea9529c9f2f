#!/usr/bin/env python3
"""Record golden values by running the *reference* package on `tests/systems.py`.

Runs only where `/root/reference` exists (the build container).  The reference
needs `beartype`, which is not installed; an identity stub is created in a
temporary directory outside both repositories (it removes run-time type checks,
no arithmetic).  Nothing from the reference is written into this repository -
the outputs are numbers: free energies, eigenvalue arrays, LDOS arrays and, for
the small systems flagged `triple`, the BSR arrays the reference assembled.

    python3 tests/golden/make_golden.py [--only name ...]     (--only: add / refresh these systems, keep the rest)
"""

from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("BODGE_REFERENCE", "/root/reference")


def _import_reference():
    shim = tempfile.mkdtemp(prefix="beartype_shim_")
    os.makedirs(os.path.join(shim, "beartype"))
    with open(os.path.join(shim, "beartype", "__init__.py"), "w") as fh:
        fh.write("def beartype(f):\n    return f\n")
    with open(os.path.join(shim, "beartype", "typing.py"), "w") as fh:
        fh.write("from typing import Callable, Iterator\n")
    sys.dont_write_bytecode = True
    sys.path[:0] = [shim, REFERENCE]
    import bodge  # noqa: E402

    return bodge


def main():
    ref = _import_reference()
    sys.path.insert(0, os.path.dirname(HERE))
    import systems

    scalars, arrays = {}, {}
    only = sys.argv[sys.argv.index("--only") + 1:] if "--only" in sys.argv else None
    if only:  # keep what is recorded for the other systems
        with open(os.path.join(HERE, "reference_values.json")) as fh:
            scalars = json.load(fh)
        with np.load(os.path.join(HERE, "reference_arrays.npz")) as old:
            arrays = {key: old[key] for key in old.files if key.split("/")[0] not in only}
    for name, spec in systems.CATALOG.items():
        if only and name not in only:
            continue
        system = spec["build"](ref, **spec["kwargs"])
        entry = {"shape": list(system.lattice.shape)}
        skeleton = system._matrix
        entry["skeleton_nnzb"] = int(skeleton.indices.size)
        trimmed = system.matrix(format="bsr")
        entry["trimmed_nnzb"] = int(trimmed.indices.size)
        entry["row_blocks_hist"] = np.bincount(np.diff(trimmed.indptr)).tolist()
        entry["dtypes"] = [str(skeleton.indptr.dtype), str(skeleton.indices.dtype), str(skeleton.data.dtype)]
        entry["free_energy"] = {repr(float(t)): float(system.free_energy(float(t))) for t in spec["temps"]}
        if spec.get("spectrum"):
            vals, _ = system.diagonalize()
            arrays[f"{name}/eigenvalues"] = np.asarray(vals)
            entry["n_eigenvalues"] = int(vals.size)
            entry["e_min"], entry["e_max"], entry["e_sum"] = float(vals.min()), float(vals.max()), float(vals.sum())
        if spec.get("triple"):
            arrays[f"{name}/indptr"] = skeleton.indptr.copy()
            arrays[f"{name}/indices"] = skeleton.indices.copy()
            arrays[f"{name}/data"] = skeleton.data.copy()
        else:
            # structure only (cheap), data checksum for the values
            arrays[f"{name}/indptr"] = skeleton.indptr.copy()
            arrays[f"{name}/indices"] = skeleton.indices.copy()
            entry["data_abs_sum"] = float(np.abs(skeleton.data).sum())
            entry["data_sum"] = [float(skeleton.data.sum().real), float(skeleton.data.sum().imag)]
        for n, (site, energies) in enumerate(spec.get("ldos", [])):
            rho = system.ldos(tuple(site), list(energies))
            arrays[f"{name}/ldos{n}"] = np.asarray(rho, dtype=float)
        scalars[name] = entry
        print(name, entry["free_energy"], flush=True)

    with open(os.path.join(HERE, "reference_values.json"), "w") as fh:
        json.dump(scalars, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "reference_arrays.npz"), **arrays)


if __name__ == "__main__":
    main()
