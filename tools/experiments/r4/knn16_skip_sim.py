import sys, numpy as np
sys.path.insert(0,'/root/repo/point-cloud-compression_amd'); sys.path.insert(0,'/root/repo')
from pccx import synth
from scipy.spatial import cKDTree
rng=np.random.default_rng(0)
def patches(pc, S=64, K=256):
    # crude: random centres (FPS-like not needed)
    idx=rng.choice(len(pc), S, replace=False)
    t=cKDTree(pc)
    d,nn=t.query(pc[idx], K)
    return [pc[nn[s]]-pc[idx[s]] for s in range(S)]
def sim(P, order_fn, U=4):
    K=len(P); tot=0; skip=0
    D=((P[:,None,:]-P[None,:,:])**2).sum(-1)
    for w in range(K//64):
        q=np.arange(64*w,64*w+64)
        cur=np.full((64,17),np.inf)
        for ch in order_fn(w,K//16):
            for j0 in range(16*ch,16*ch+16,U):
                tot+=1
                d=D[q][:,j0:j0+U]
                if (d.min(1) < cur[:,16]).any():
                    cur=np.sort(np.concatenate([cur,d],1),1)[:,:17]
                else: skip+=1
    return skip,tot
def zig(w,nch):
    c0=4*w; up=c0; down=c0-1; out=[]
    for t in range(nch):
        if t<4 or down<0 or (up<nch and t%2==0):
            if up<nch: out.append(up); up+=1
            else: out.append(down); down-=1
        else: out.append(down); down-=1
    return out
def seq(w,nch): return list(range(nch))
for seed in (11,12,13):
    pc=synth.cad_cloud(seed,8192)
    ps=patches(pc)[:16]
    for name,fn in (('zig',zig),('seq',seq)):
        for U in (4,2,1):
            s=t=0
            for P in ps:
                a,b=sim(P,fn,U); s+=a;t+=b
            print(seed,name,U,round(s/t,3))

print("--- morton clusters")
def morton(P, bits=5):
    lo=P.min(0); hi=P.max(0); g=((P-lo)/(hi-lo+1e-12)*(2**bits-1e-6)).astype(np.int64)
    code=np.zeros(len(P),dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((g[:,a]>>b)&1) << (3*b+a)
    return np.argsort(code, kind='stable')
def sim2(P, U=4, CH=16):
    K=len(P); o=morton(P); P=P[o]
    D=((P[:,None,:]-P[None,:,:])**2).sum(-1)
    nch=K//CH
    blo=np.array([P[c*CH:(c+1)*CH].min(0) for c in range(nch)]); bhi=np.array([P[c*CH:(c+1)*CH].max(0) for c in range(nch)])
    tot=0; skipl=0; skipd=0
    for w in range(K//64):
        q=np.arange(64*w,64*w+64)
        qlo=P[q].min(0); qhi=P[q].max(0)
        # lower bound between wave bbox and chunk bbox
        gap=np.maximum(0, np.maximum(blo-qhi, qlo-bhi)); lb=(gap**2).sum(1)
        order=np.argsort(lb, kind='stable')
        cur=np.full((64,17),np.inf)
        for ch in order:
            if lb[ch] > cur[:,16].max():
                n=CH//U; tot+=n; skipd+=n; skipl+=n; continue
            for j0 in range(CH*ch,CH*ch+CH,U):
                tot+=1
                d=D[q][:,j0:j0+U]
                if (d.min(1) < cur[:,16]).any():
                    cur=np.sort(np.concatenate([cur,d],1),1)[:,:17]
                else: skipl+=1
    return skipl,skipd,tot
for seed in (11,12,13):
    pc=synth.cad_cloud(seed,8192)
    ps=patches(pc)[:16]
    for U in (4,2):
        s=d=t=0
        for P in ps:
            a,b,c=sim2(P,U); s+=a;d+=b;t+=c
        print(seed,U,'ladder skip',round(s/t,3),'dist skip',round(d/t,3))
