import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests.helpers import load_golden, oracle_problem
from tests.test_engine_gpu import _engine_block_sketches
dev=torch.device('cuda',0)
for prec in ['f16','f16x2']:
    z=load_golden('rhe_1bin_nocov'); p,c=oracle_problem(z)
    XXz,yXXy,cs=_engine_block_sketches(p,c,dev,prec)
    ref=z['ref_t0_XXz_blocks'].astype(np.float64); refy=z['ref_t0_yXXy_blocks']
    for j in range(3):
        print(prec,j,'XXz relerr',np.abs(XXz[0,j]-ref[0,j]).max()/np.abs(ref[0,j]).max(),'yXXy',yXXy[0,j],refy[0,j], 'ratio', (XXz[0,j]/ref[0,j]).ravel()[:4])
