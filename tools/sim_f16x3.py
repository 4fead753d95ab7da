#!/usr/bin/env python
"""Numpy model of the two ways the convolution kernels put an fp32 dot product on the 16-bit matrix cores: three fp16
products of per-tensor-scaled two-way split operands vs six bf16 products of three-way split ones, products exact, fp32
accumulation in the kernels' order (one rounding per product per 16-deep k-group), against an fp32 FMA-like chain with exact
products -- relative L2 error of 4096 dot products of length 1152 against float64.  CPU only: python tools/sim_f16x3.py"""
import numpy as np
rng=np.random.default_rng(0)
def split16(x, k):
    xs = (x.astype(np.float64) * 2.0**k)
    h1 = xs.astype(np.float16)            # RN
    r = (xs - h1.astype(np.float64))
    h2 = r.astype(np.float16)
    return h1.astype(np.float64), h2.astype(np.float64)
def splitbf(x):
    def bf(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return u.astype(np.uint32).view(np.float32).astype(np.float64)
    x=x.astype(np.float64)
    h=bf(x.astype(np.float32)); m=bf((x-h).astype(np.float32)); l=bf((x-h-m).astype(np.float32))
    return h,m,l
def scale_k(x):
    m=np.abs(x).max(); e=np.floor(np.log2(m)); return int(14-e)
K=1152; N=4096
for name,gen in [('act~relu(N(0,1))', lambda s: np.maximum(rng.standard_normal(s),0)), ('grad~N(0,1e-5)*lognormal', lambda s: rng.standard_normal(s)*1e-5*np.exp(2*rng.standard_normal(s)))]:
    a=gen((N,K)).astype(np.float32); w=(rng.standard_normal((K,))*0.03).astype(np.float32)
    ref=(a.astype(np.float64)@w.astype(np.float64))
    # fp32 sequential-ish accumulate (chunks of 16 like MFMA then fp32 adds)
    def acc32(terms):  # terms: list of (N,K) float64 product arrays, accumulate in fp32 per k-chunk of 16
        acc=np.zeros(N,np.float32)
        for k0 in range(0,K,16):
            for t in terms:
                acc=(acc.astype(np.float64)+t[:,k0:k0+16].sum(1)).astype(np.float32)
        return acc.astype(np.float64)
    p32=acc32([a.astype(np.float64)*w.astype(np.float64)])
    ka,kw=scale_k(a),scale_k(w)
    a1,a2=split16(a,ka); w1,w2=split16(w,kw)
    x3=acc32([a1*w1, a1*w2, a2*w1])*2.0**-(ka+kw)
    x4=acc32([a2*w2, a1*w2, a2*w1, a1*w1])*2.0**-(ka+kw)
    ah,am,al=splitbf(a); wh,wm,wl=splitbf(w)
    b6=acc32([al*wh, ah*wl, am*wm, am*wh, ah*wm, ah*wh])
    def e(v): return np.linalg.norm(v-ref)/np.linalg.norm(ref)
    print(name,'ka',ka,'kw',kw,'fp32-acc exact products %.2e  fp16x3 %.2e  fp16x4 %.2e  bf16x6 %.2e'%(e(p32),e(x3),e(x4),e(b6)))
    # flush-subnormals variant
    def ftz(v): 
        v=v.copy(); v[np.abs(v)<2.0**-14]=0; return v
    x3f=acc32([a1*w1, a1*ftz(w2), ftz(a2)*w1])*2.0**-(ka+kw)
    print('   fp16x3 with subnormal pieces flushed %.2e'%e(x3f))
