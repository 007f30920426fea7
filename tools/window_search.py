#!/usr/bin/env python3
"""Adversarial search against the proven stage-1 window (CPU only): a population of rows (Gaussian, aligned-residual, tent) is
mutated - low mantissa bits flipped, blocks rescaled by powers of two, signs flipped, elements aligned with a hyperplane,
prefix / suffix signs set tent-like - and a mutation is kept when it raises  max_j |y1_j - y_host_j| / window_j  (y1 from the
accumulator model the GPU suite pins to the kernel bit for bit, y_host = NumPy P_band @ x).  A ratio above 1 would be a hole in
the proof.   python tools/window_search.py <seed> <dim> <iterations>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshrs_amd.windows import window_coefficients, _bf16_rne
from oracle.build import split_stage1_model
from tests._adversary import adversarial_row, tent_row
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
dim=int(sys.argv[2]) if len(sys.argv)>2 else 256
nb,r=2,16
planes=[rng.standard_normal((r,dim)).astype(np.float32) for _ in range(nb)]
stack=np.concatenate(planes)
ca,cb,ct,info=window_coefficients(stack,1)
def score(X):
    y1=split_stage1_model(planes,X).astype(np.float64)
    yh=np.concatenate([np.stack([p@v for v in X]) for p in planes],axis=1).astype(np.float64)
    xh=_bf16_rne(X); xm=_bf16_rne(X-xh)
    nh=np.linalg.norm(xh.astype(np.float64),axis=1); nm=np.linalg.norm(xm.astype(np.float64),axis=1)
    thr=nh[:,None]*ca[None,:]+nm[:,None]*cb[None,:]
    return (np.abs(y1-yh)/thr).max(axis=1)
# population
pop=[rng.standard_normal(dim).astype(np.float32) for _ in range(16)]
pop+= [adversarial_row(stack[j],20.0,seed=j) for j in range(8)]+[tent_row(stack[j],5.0,seed=j) for j in range(8)]
pop=np.stack(pop)
sc=score(pop)
best=sc.max(); t0=time.time()
for it in range(int(sys.argv[3]) if len(sys.argv)>3 else 300):
    # mutate each member
    cand=pop.copy()
    for i in range(len(cand)):
        m=rng.integers(0,5)
        x=cand[i]
        if m==0:   # flip low mantissa bits of a few elements
            k=rng.integers(0,dim,8); u=x.view(np.uint32); u[k]^=rng.integers(1,1<<16,8).astype(np.uint32)
        elif m==1: # rescale a block by power of two
            a=rng.integers(0,dim-8); x[a:a+8]*=np.float32(2.0**rng.integers(-6,7))
        elif m==2: # sign flips
            k=rng.integers(0,dim,4); x[k]=-x[k]
        elif m==3: # align a few elements with a plane's sign & magnitude
            j=rng.integers(0,len(stack)); k=rng.integers(0,dim,16); x[k]=np.abs(x[k])*np.sign(stack[j][k])
        else:      # make prefix positive / suffix negative along a plane (tent-like)
            j=rng.integers(0,len(stack)); h=rng.integers(dim//4,3*dim//4); s=np.sign(stack[j]); x[:h]=np.abs(x[:h])*s[:h]; x[h:]=-np.abs(x[h:])*s[h:]
        cand[i]=np.where(np.isfinite(x),x,np.float32(1.0))
    sc2=score(cand)
    better=sc2>sc
    pop[better]=cand[better]; sc[better]=sc2[better]
    if sc.max()>best: best=sc.max()
print(f"dim {dim} seed {sys.argv[1] if len(sys.argv)>1 else 0}: best ratio {best:.4f} after {it+1} iterations, {time.time()-t0:.0f}s; per-member top: {np.sort(sc)[-5:]}")
