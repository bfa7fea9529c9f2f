import ctypes, os, time, sys
t=time.time()
for f in ("/opt/rocm/lib/librocblas.so", "/opt/rocm/lib/librocsolver.so"):
    t0=time.time(); n=0
    with open(f,"rb") as fh:
        while True:
            b=fh.read(1<<24)
            if not b: break
            n+=len(b)
    print(f, n/1e6, "MB read in", round(time.time()-t0,1), "s", flush=True)
t0=time.time(); ctypes.CDLL("/opt/rocm/lib/librocblas.so", mode=ctypes.RTLD_GLOBAL); print("dlopen rocblas", round(time.time()-t0,1), flush=True)
t0=time.time(); ctypes.CDLL("/opt/rocm/lib/librocsolver.so", mode=ctypes.RTLD_GLOBAL); print("dlopen rocsolver", round(time.time()-t0,1), flush=True)
