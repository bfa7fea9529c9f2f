"""Cold-storage calibration on a fresh box: sequential vs multi-threaded read rate of big files
that nothing has touched yet (decides how the 931 MB rocSOLVER object is brought in)."""
import os, sys, threading, time


def read_range(path, lo, hi, out, deadline):
    fd = os.open(path, os.O_RDONLY)
    pos, n = lo, 0
    while pos < hi and time.time() < deadline:
        chunk = os.pread(fd, min(4 << 20, hi - pos), pos)
        if not chunk:
            break
        pos += len(chunk)
        n += len(chunk)
    os.close(fd)
    out.append(n)


def probe(path, threads, seconds):
    size = os.path.getsize(path)
    out, deadline, t0 = [], time.time() + seconds, time.time()
    step = (size + threads - 1) // threads
    ts = [threading.Thread(target=read_range, args=(path, k * step, min(size, (k + 1) * step), out, deadline)) for k in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    print(f"{os.path.basename(path)}: {threads:2d} threads read {sum(out) / 1e6:7.0f} MB of {size / 1e6:.0f} in {dt:5.1f} s = {sum(out) / 1e6 / dt:8.1f} MB/s", flush=True)


lib = "/opt/rocm/lib/"
probe(lib + "libMIOpen.so", 1, float(sys.argv[1]) if len(sys.argv) > 1 else 15)
probe(lib + "librocrand.so", 16, 15)
probe(lib + "librocsparse.so", 64, 15)
probe(lib + "libMIOpen.so", 1, 5)
