import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tools'), os.path.join(ROOT,'tests')]
os.environ['CSP_BRUTE_DEBUG']='1'
import numpy as np
from _util import product_api, emul_api, package, png_cases
from oracle import oracle as O
api=(emul_api if len(sys.argv)>1 else product_api)(); pkg=package()
cases=[c for c in png_cases() if c[0]=='RGB_flat_64x48']
p=pkg.default_parameters(png_optimize=True, png_optimization_level=3)
b=api.png_batch([c[1] for c in cases],p); b.run()
got,have=b.scores(0)
for y in range(1,4):
    for f in range(5): print('DEV row',y,'f',f,'nl,nd,extra,sub,score', got[y,f].tolist())
P=O.png_decode(cases[0][1])
sys.stderr.flush()
import ctypes
rows=P.rows()
# oracle per (row, filter) debug lines come on stderr in order y, f
want=P.scores()
print('ORACLE rows 1..3 score', want[1:4,:,4].tolist())
