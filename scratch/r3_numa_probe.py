"""Where does the memory of a GPU-box process live, and which cores are close to it?"""
import os, sys, glob
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for f in ("/sys/fs/cgroup/cpuset.mems.effective", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.max"):
    try: print(f, "=", open(f).read().strip())
    except OSError as e: print(f, e)
for node in sorted(glob.glob("/sys/devices/system/node/node*")):
    try:
        mem = [l for l in open(node + "/meminfo") if "MemFree" in l or "MemTotal" in l]
        print(os.path.basename(node), open(node + "/cpulist").read().strip(), " ".join(m.split(":")[1].strip() for m in mem))
    except OSError as e: print(node, e)
import numpy as np, ctypes as C
import bench
from oracle import cheb_c, cheb_ref
system = bench.build_system([1000, 1000, 1], "swave")
bsr = system.matrix("bsr")
scale = cheb_ref.spectral_bound(bsr)
start = cheb_ref.random_block(bsr.shape[0], 0, range(8), cheb_ref.VEC_RADEMACHER)
lib = cheb_c.load()
def run(cpus, label):
    cheb_c.set_threads(len(cpus))
    arr = (C.c_int * len(cpus))(*cpus)
    lib.cheb_c_pin_threads(arr, len(cpus))
    try:
        rate, steps, _ = cheb_c.time_recurrence(bsr, scale, start, seconds=1.5, real=True, numa=True)
    finally:
        lib.cheb_c_unpin_threads()
    print(f"{label:40s} {rate:8.1f} steps/s", flush=True)
nodes = []
for node in sorted(glob.glob("/sys/devices/system/node/node*")):
    lst = open(node + "/cpulist").read().strip()
    cpus = []
    for part in lst.split(","):
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    nodes.append(cpus)
for i, cpus in enumerate(nodes):
    phys = cpus[: len(cpus) // 2] if len(cpus) >= 32 else cpus
    step = max(1, len(phys) // 16)
    run(phys[::step][:16], f"16 threads spread over node {i}")
run(cheb_c.spread_cpus(16), "16 threads spread over the host")
for i, cpus in enumerate(nodes):
    run(cpus[:16], f"16 threads on the first 16 cpus of node {i}")
