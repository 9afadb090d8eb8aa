import sys
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R, R+'/tests', R+'/tests/emu']
import numpy as np
import harness as H
from fuzz_scenes import random_scene
from oracle_bridge import Oracle, OracleState
import newton_amd as nt
seed=int(sys.argv[1])
model = random_scene(seed)
t=model.env
em = H.EmuModel(model)
s0,ct = H.EmuState(em),H.EmuContacts(em)
o=Oracle(model); os0,oc=OracleState(model),o.contacts()
H.collide(em,s0,ct); o.collide(os0.body_q,oc)
e=ct.export(); n=int(oc.count[0])
print('count', e['count'], n)
types=np.asarray(model.shape_type)
for i in range(n):
    d=max(np.abs(e[k][i]-getattr(oc,k)[i]).max() for k in ('point0','point1','normal','offset0','offset1','margin0','margin1'))
    if d>1e-6:
        a,b=oc.shape0[i],oc.shape1[i]
        print(i, 'shapes',a,b,'types',nt.GeoType(types[a]).name, nt.GeoType(types[b]).name, 'diff %.2e'%d)
        for k in ('point0','point1','normal'):
            print('   ',k, e[k][i], getattr(oc,k)[i])
