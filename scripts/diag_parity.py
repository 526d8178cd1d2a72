import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import tfqmrgpu_amd as T
from conftest import ALL_NAMES, golden_solves, load_golden, load_problem
from oracle import pyoracle as O
for name in ALL_NAMES:
    pr, g = load_problem(name), load_golden(name)
    for prec, tol, maxit in golden_solves(g):
        t0=time.time(); st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=maxit, shadow_mode=T.SHADOW_GLIBC_RAND); t1=time.time()
        st0, X0, info0 = O.solve(pr, prec, threshold=tol, max_iterations=maxit); t2=time.time()
        h, h0 = info['bound_history'], info0['bound_history']
        n=min(len(h),len(h0))
        rel = np.abs(h[:n]-h0[:n])/np.maximum(h0[:n],1e-300)
        print(name, prec, 'st',st,st0,'it',info['iterations'],info0['iterations'],'res %.3e %.3e'%(info['residual'],info0['residual']),
              'dX %.1e'%(np.abs(X-X0).max()/np.abs(X0).max()), 'hist rel first/mid/last %.1e %.1e %.1e'%(rel[0], rel[n//2], rel[-1]), 'flops', info['flops']==info0['flops'], 'tgpu %.3f tcpu %.3f'%(t1-t0,t2-t1))
