import sys, os, json
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[R,R+'/tools',R+'/tests']
from _util import product_api, package, oracle_lossy
api=product_api(); pkg=package()
GOLD=R+'/tests/golden'
man=json.load(open(os.path.join(GOLD,'manifest.json')))
bad=0
for case in man['cases']:
    src=open(os.path.join(GOLD, case['name']+'.src.jpg'),'rb').read()
    for q in case['qualities']:
        got=api.compress_in_memory(src, pkg.default_parameters(jpeg_quality=q)); want=oracle_lossy(src,q)
        if got!=want:
            bad+=1
            i=next((i for i in range(min(len(got),len(want))) if got[i]!=want[i]), None)
            print(case['name'], q, len(got), len(want), 'first diff', i)
print('bad', bad, 'debug', os.environ.get('CSH_DEBUG'))
