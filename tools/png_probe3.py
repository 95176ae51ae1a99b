import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[ROOT, os.path.join(ROOT,'tools'), os.path.join(ROOT,'tests')]
os.environ['CSP_DEBUG_CHUNK']='21'
import numpy as np
from _util import product_api, emul_api, package
from gen_synth import synth_png
api=product_api(); emu=emul_api(); pkg=package()
blob=synth_png(30,640,480,'RGB',texture=4.0)
p=pkg.default_parameters(png_optimize=True, png_optimization_level=3)
res=[]
for a in (api, emu):
    b=a.png_batch([blob],p); b.run(); res.append(b.chunk_bits(0,1))
g,e=np.array(res[0][-317:],dtype=np.int64),np.array(res[1][-317:],dtype=np.int64)
d=np.nonzero(g!=e)[0]
print('freq diffs (index, gpu, emul):',[(int(i),int(g[i]),int(e[i])) for i in d])
print('tokens gpu',g[:286].sum(),'emul',e[:286].sum(),'matches gpu',g[257:286].sum(),'emul',e[257:286].sum())
