import ctypes, os, time, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
t0=time.time()
rb = ctypes.CDLL("/opt/rocm/lib/librocblas.so", mode=ctypes.RTLD_GLOBAL); print("dlopen rocblas", round(time.time()-t0,2), flush=True)
t0=time.time(); rs = ctypes.CDLL("/opt/rocm/lib/librocsolver.so", mode=ctypes.RTLD_GLOBAL); print("dlopen rocsolver", round(time.time()-t0,2), flush=True)
h = ctypes.c_void_p()
t0=time.time(); rc = rb.rocblas_create_handle(ctypes.byref(h)); print("rocblas_create_handle", rc, round(time.time()-t0,2), flush=True)
import numpy as np, bodge_amd as ba, systems
s = systems.swave_square(ba, L=8)
for rep in range(3):
    t0=time.time(); w,_ = s._solver().eigh(vectors=False); print("eigh values n=256", round(time.time()-t0,2), flush=True)
t0=time.time(); w,z = s._solver().eigh(vectors=True); print("eigh vectors n=256", round(time.time()-t0,2), flush=True)
s = systems.swave_square(ba, L=20)
t0=time.time(); w,z = s._solver().eigh(vectors=True); print("eigh vectors n=1600", round(time.time()-t0,2), flush=True)
