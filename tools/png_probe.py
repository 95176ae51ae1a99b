import sys
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tools'), os.path.join(ROOT,'tests')]
import numpy as np
from _util import product_api, package, png_cases
from oracle import oracle as O
api=product_api(); pkg=package()
cases=png_cases(small=False)
p=pkg.default_parameters(png_optimize=True, png_optimization_level=3)
b=api.png_batch([c[1] for c in cases],p); b.run(); outs=b.fetch()
for i,(name,blob) in enumerate(cases):
    P=O.png_decode(blob)
    got,have=b.scores(i); want=P.scores()
    for k in range(5):
        if have>>k&1:
            bad=np.argwhere(got[:,:,k]!=want[:,:,k])
            if len(bad): print(name,'score',k,'nbad',len(bad),'of',got.shape[0]*5,'first',bad[:4].tolist(),[ (int(got[tuple(x)][k]),int(want[tuple(x)][k])) for x in bad[:4]])
    ref=O.png_optimize(blob,3)[0]
    print(name, 'out equal', outs[i]==ref)
