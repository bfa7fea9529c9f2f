"""Which parts of librocsolver.so / librocblas.so (and rocBLAS's Tensile files) does a dense
eigensolve of the ladder sizes actually touch?  Run on a FRESH box with the library's own
read-ahead switched off: solve, then ask the kernel (mincore) which pages of the files are in
the page cache.  The byte ranges go to gpurun_out/touch_map.json; tools/make_prefetch_ranges.py
turns them into bodge_amd/csrc/prefetch_ranges.inc, so that the background prefetch reads those
ranges instead of 931 MB."""
import ctypes, json, mmap, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
libc = ctypes.CDLL("libc.so.6", use_errno=True)
PAGE = os.sysconf("SC_PAGE_SIZE")
FILES = [os.path.realpath("/opt/rocm/lib/librocsolver.so"), os.path.realpath("/opt/rocm/lib/librocblas.so")]


def resident_pages(path):
    import numpy as np

    size = os.path.getsize(path)
    if size == 0:
        return size, []
    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), size, prot=mmap.PROT_READ, flags=mmap.MAP_SHARED)
        n_pages = (size + PAGE - 1) // PAGE
        vec = (ctypes.c_ubyte * n_pages)()
        view = np.frombuffer(mm, dtype=np.uint8)  # (only for the mapping's address: nothing is read)
        rc = libc.mincore(ctypes.c_void_p(view.ctypes.data), ctypes.c_size_t(size), vec)
        err = ctypes.get_errno()
        del view
        mm.close()
        if rc != 0:
            raise OSError(err, "mincore failed")
    flags = np.frombuffer(vec, dtype=np.uint8) & 1
    return size, np.flatnonzero(flags).tolist()


def to_ranges(pages, gap=16):
    """Merge pages into [offset, length] ranges, bridging gaps of up to `gap` pages."""
    out = []
    for p in pages:
        if out and p - (out[-1][0] + out[-1][1]) <= gap:
            out[-1][1] = p + 1 - out[-1][0]
        else:
            out.append([p, 1])
    return [[o * PAGE, l * PAGE] for o, l in out]


def touched_pages(path):
    """File pages of `path` that THIS process has mapped in (its page tables, /proc/self/pagemap):
    what the loader, the HIP runtime and the library itself actually touched through the mapping -
    independent of what the host's page cache happens to hold (fault-around adds up to 64 KB a fault)."""
    import struct

    pages = set()
    with open("/proc/self/maps") as maps, open("/proc/self/pagemap", "rb") as pagemap:
        for line in maps:
            parts = line.split()
            if len(parts) < 6 or os.path.realpath(parts[5]) != path:
                continue
            lo, hi = (int(v, 16) for v in parts[0].split("-"))
            file_off = int(parts[2], 16)
            n = (hi - lo) // PAGE
            pagemap.seek(lo // PAGE * 8)
            entries = struct.unpack(f"<{n}Q", pagemap.read(8 * n))
            for i, entry in enumerate(entries):
                if entry >> 63 & 1:
                    pages.add(file_off // PAGE + i)
    return os.path.getsize(path), sorted(pages)


import threading


def heartbeat():
    t = time.time()
    while True:
        time.sleep(60)
        print(f"... {time.time() - t:.0f} s (the first solve waits for the library to fault in from cold storage)", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
os.environ["BODGE_AMD_NO_PREFETCH"] = "1"
import numpy as np
import bodge_amd as ba
import systems

t0 = time.time()
for name, env in [("swave30_zeeman", "rocsolver"), ("peierls30", "rocsolver"), ("chain300", "rocsolver"),
                  ("barrier", "evj"), ("complex235", "evj"), ("complex235", "evd"), ("barrier", "evd")]:
    os.environ["BODGE_AMD_EIGH"] = env
    if name == "barrier" and env == "evj":
        os.environ["BODGE_AMD_EIGH_REAL"] = "1"
    spec = systems.CATALOG[name]
    system = spec["build"](ba, **spec["kwargs"])
    solver = system._solver()
    for vectors in (False, True):
        t1 = time.time()
        solver.eigh(vectors=vectors)
        print(f"[{time.time() - t0:6.1f}s] {name} {env} vectors={vectors}: {time.time() - t1:.2f} s", flush=True)
# a bigger one, so that the large-matrix code paths of the same routines are in the map as well
lattice = ba.CubicLattice((45, 45, 1))
big = ba.Hamiltonian(lattice)
with big as (H, D):
    H.set_sites(3.0 * ba.σ0 - 0.05 * ba.σ3); D.set_sites(-0.1 * ba.jσ2); H.set_bonds(-1.0 * ba.σ0)
os.environ["BODGE_AMD_EIGH"] = "rocsolver"
t1 = time.time(); big._solver().eigh(vectors=True); print(f"[{time.time() - t0:6.1f}s] 45x45 dsyevd vectors: {time.time() - t1:.2f} s", flush=True)

result = {}
for path in FILES:
    size, pages = touched_pages(path)
    ranges = to_ranges(pages)
    result[path] = {"size": size, "touched_bytes": len(pages) * PAGE, "ranges": ranges}
    print(f"after: {os.path.basename(path)}: {len(pages) * PAGE / 1e6:.1f} MB of {size / 1e6:.1f} touched through the mapping, "
          f"{len(ranges)} ranges covering {sum(l for _, l in ranges) / 1e6:.1f} MB", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "touch_map.json"), "w") as fh:
    json.dump(result, fh)
