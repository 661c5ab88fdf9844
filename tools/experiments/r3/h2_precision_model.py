"""numpy model of one dot product in the three arithmetics (representation error only: products and sums in float64), relative to
sum|a b|: plain sequential fp32, bf16x3 (six products), f16x2 (three / four products) without and with the power-of-two operand
scales.  Quoted in DESIGN.md section 4: with the scales the f16x2 representation error (1.5e-8 at K = 128) is below the rounding
noise of an fp32 accumulation (2.7e-8), which every mode shares.   python tools/experiments/r3/h2_precision_model.py"""
import numpy as np
rng=np.random.default_rng(0)
def bf16(x):
    u=x.astype(np.float32).view(np.uint32); u=(u+0x7FFF+((u>>16)&1))&0xFFFF0000; return u.view(np.float32)
def split_bf3(x):
    h=bf16(x); r=(x-h).astype(np.float32); m=bf16(r); l=bf16((r-m).astype(np.float32)); return [h,m,l]
def split_h2(x,scale=1.0):
    xs=(x*scale).astype(np.float32); h=xs.astype(np.float16).astype(np.float32); r=(xs-h).astype(np.float32); l=r.astype(np.float16).astype(np.float32); return [h/scale,l/scale]
for K in (32,128,1024):
    N=2000
    a=np.maximum(rng.standard_normal((N,K)),0).astype(np.float32)*rng.uniform(0.01,10,(N,1)).astype(np.float32)
    w=rng.uniform(-1,1,(N,K)).astype(np.float32)/np.sqrt(K)
    ref=(a.astype(np.float64)*w.astype(np.float64)).sum(1); den=(np.abs(a.astype(np.float64)*w)).sum(1)
    def err(v): return np.sqrt(np.mean(((v-ref)/den)**2)), np.max(np.abs((v-ref)/den))
    # fp32 sequential
    acc=np.zeros(N,np.float32)
    for k in range(K): acc=(acc+ (a[:,k].astype(np.float64)*w[:,k]).astype(np.float32)*0+np.float32(1)*0+ (a[:,k]*w[:,k])).astype(np.float32)
    print(K,'fp32 seq (mul+add)',err(acc.astype(np.float64)))
    A=split_bf3(a);W=split_bf3(w)
    v=sum((A[i].astype(np.float64)*W[j]).sum(1) for i in range(3) for j in range(3) if i+j<=2); print(K,'bf16x3 6prod repr-only',err(v))
    for sa,sw in ((1,1),(2**6,2**12)):
        A=split_h2(a,sa);W=split_h2(w,sw)
        v=(A[0].astype(np.float64)*W[0]).sum(1)+(A[0].astype(np.float64)*W[1]).sum(1)+(A[1].astype(np.float64)*W[0]).sum(1); print(K,'f16x2 3prod repr-only scale',sa,sw,err(v))
        v4=v+(A[1].astype(np.float64)*W[1]).sum(1); print(K,'f16x2 4prod repr-only',err(v4))
