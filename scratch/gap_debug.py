import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from scipy.linalg import eigh_tridiagonal
shape=[200,200,1]
system = bench.build_system(shape)
solver = system._solver()
V=2
solver.lanczos_begin(V, seed=0, max_iter=12000)
alpha=np.zeros((0,V)); beta=np.zeros((0,V))
t=time.time()
for it in range(12):
    a,b = solver.lanczos_advance(1000); alpha=np.vstack([alpha,a]); beta=np.vstack([beta,b]); m=alpha.shape[0]
    out=[]
    for c in range(V):
        th=eigh_tridiagonal(alpha[:,c], beta[:m-1,c], select="i", select_range=(0,5), eigvals_only=True)
        out.append(np.sqrt(np.clip(th,0,None)))
    print(m, f"{time.time()-t:.1f}s", np.array2string(out[0],precision=9), np.array2string(out[1],precision=9), flush=True)
