import sys, numpy as np, torch
sys.path[:0]=['/root/repo/scene-aware-3d-multi-human_amd','/root/repo/tests','/root/repo/tests/golden']
from mhhip import engine
import golden_inputs as gi
g=np.load('/root/repo/tests/golden/reference_cpu.npz')
x=torch.tensor(gi.one_euro_inputs()).cuda()
for nm,(c,b) in {'one_euro_a':(0.01,0.02),'one_euro_b':(0.001,0.5)}.items():
    y=engine.one_euro_scan(x,c,b).cpu().numpy()
    d=np.abs(y-g[nm]); print(nm,'max',d.max(),'exact fraction',(d==0).mean(), x.shape)
x=torch.randn(200,82680,device='cuda')
import time
for _ in range(3): y=engine.one_euro_scan(x,0.001,0.5)
torch.cuda.synchronize(); t=time.time()
for _ in range(20): y=engine.one_euro_scan(x,0.001,0.5)
torch.cuda.synchronize(); print('scan 200x82680: %.1f us'%((time.time()-t)/20*1e6))
