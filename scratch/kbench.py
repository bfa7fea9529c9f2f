"""Interleaved A/B timing of recurrence-kernel variants selected by environment variables.

usage: python3 scratch/kbench.py "NAME=ENV1=V1,ENV2=V2" "NAME2=..." [--lattice 1000,1000,1] [--vectors 8] [--rounds 5]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bodge_amd import backend, chebyshev
from bodge_amd.solver import DeviceSolver, VEC_RADEMACHER, VEC_Z4

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(zip(sys.argv[1:], sys.argv[2:]))
shape = [int(v) for v in opts.get("--lattice", "1000,1000,1").split(",")]
vectors = int(opts.get("--vectors", "8"))
rounds = int(opts.get("--rounds", "5"))
steps = int(opts.get("--steps", "40"))
model = opts.get("--model", "swave")
kind = VEC_Z4 if opts.get("--kind", "rademacher") == "z4" else VEC_RADEMACHER
variants = []
for a in args:
    name, _, envs = a.partition("=")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e)))
system = bench.build_system(shape, model)
indptr, indices, data = system.bsr_arrays()
scale = chebyshev.spectral_bound(indptr, data)
solver = DeviceSolver(indptr, indices, data)
solver.set_lattice_shape(shape)
results = {name: [] for name, _ in variants}
for rnd in range(rounds + 1):
    for name, env in variants:
        with backend.options(**env):
            solver.dots_random(scale, steps, vectors, seed=rnd, kind=kind)
            p = solver.perf()
        if rnd:
            results[name].append((p["kernel_ms"] / p["launches"], p["bytes_moved"] / p["launches"], p, p["vector_steps"] / p["window_ms"]))
for name, _ in variants:
    ms = np.array([r[0] for r in results[name]]); b = results[name][0][1]; p = results[name][0][2]
    rate = np.median([r[3] for r in results[name]])
    print(f"{name:28s} {rate:7.2f} k vsteps/s (kernel time)  launches {p['launches']} rolling {p['rolling']}  median {np.median(ms):.4f} ms  min {ms.min():.4f}  -> {b/np.median(ms)/1e6:7.1f} GB/s (best {b/ms.min()/1e6:7.1f})"
          f"  grid {p['grid']} lds {p['lds_bytes']} rl {p['lanes_per_row']} real {p['real_arithmetic']} pipe {p['pipelined']}"
          f" steps/launch {p['steps_per_launch']} onsite {p['onsite_streamed']} -> {vectors * p['steps_per_launch'] / np.median(ms) :.1f} k vector-steps/s")
