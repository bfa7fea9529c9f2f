"""Why is the C/OpenMP baseline at 160-180 steps/s on the GPU box (16-CPU quota) when round 2 measured 913?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from oracle import cheb_c, cheb_ref
print("spread 16:", cheb_c.spread_cpus(16)); print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, flush=True)
def throttled():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
    except OSError:
        return {}
system = bench.build_system([1000, 1000, 1], "swave")
bsr = system.matrix("bsr")
scale = cheb_ref.spectral_bound(bsr)
start = cheb_ref.random_block(bsr.shape[0], 0, range(8), cheb_ref.VEC_RADEMACHER)
for threads in (8, 12, 14, 16, 16, 24):
    for numa, pin in ((False, False), (True, False), (True, True), (False, True)):
        cheb_c.set_threads(threads)
        s0 = throttled()
        rate, steps, t = cheb_c.time_recurrence(bsr, scale, start, seconds=2.0, real=True, numa=numa, pin=pin)
        s1 = throttled()
        print(f"threads {threads:3d} first-touch {numa!s:5s} pinned {pin!s:5s}: {rate:8.1f} steps/s ({steps} block-steps)  throttled periods +{s1.get('nr_throttled',0)-s0.get('nr_throttled',0)} "
              f"throttled time +{(s1.get('throttled_usec',0)-s0.get('throttled_usec',0))/1e6:.2f} s", flush=True)
