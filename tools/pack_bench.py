import time, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from gauss_amd import api, panel, synth
d='/tmp/gauss_pack'; os.makedirs(d,exist_ok=True)
pops=synth.pop_table()
N=sum(p[1] for p in pops); S=3000
rng=np.random.default_rng(0)
G=rng.integers(0,3,size=(S,N),dtype=np.uint8)
sizes=[p[1] for p in pops]
off=np.concatenate([[0],np.cumsum(sizes)])
af=np.stack([G[:,off[k]:off[k+1]].sum(1)/(2.0*sizes[k]) for k in range(len(pops))],1)
bp=np.sort(rng.choice(np.arange(1,5_000_000),S,replace=False))
rsid=np.array([f"rs{i}" for i in range(S)]); a1=np.full(S,'A'); a2=np.full(S,'G')
panel.write_pop_desc(d+'/desc.txt',pops)
panel.write_panel(d+'/index.gz',d+'/data.gz',rsid,np.full(S,22),bp,a1,a2,G,af,sizes)
for rep in range(2):
    t0=time.perf_counter()
    n=api.pack_panel(d+'/index.gz',d+'/data.gz',d+'/desc.txt',d+'/p.gpk')
    t=time.perf_counter()-t0
    print('pack_panel',n,'SNPs x',N,'samples in %.3f s = %.3f ms/SNP'%(t,t/n*1e3))
