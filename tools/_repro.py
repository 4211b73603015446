import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import sjpeg_amd as sj
from oracle import orc
import test_gpu_parity as T
o=orc.oracle(); eng=sj.Engine(0)
cases=[]
for fmt in (4,5,1):
    rng = np.random.RandomState(100 + fmt)
    for (w, h) in ((97, 61), (250, 130), (1920, 1080)):
        planes = T._random_planes(rng, fmt, w, h)
        planes = [(p // 4 + np.arange(p.shape[1])[None, :] // 3).astype(np.uint8) for p in planes]
        dev_planes = [torch.from_numpy(p).cuda().unsqueeze(0) for p in planes]
        for mode in ((1,3,4) if fmt in (1,2) else (1,)):
            for q, method in ((75.0, 0), (40.0, 4), (92.0, 3), (60.0, 1)):
                want = o.encode_src(fmt, planes, w, h, o.quality_matrices(q), yuv_mode=mode, method=method)
                cases.append((fmt,w,h,mode,q,method,dev_planes,want))
bad={}
for it in range(40):
    for (fmt,w,h,mode,q,method,dp,want) in cases:
        got = sj.encode_source_method(fmt, dp, w, h, q, mode, method, engine=eng)
        if got != want:
            g=np.frombuffer(got,np.uint8); wv=np.frombuffer(want,np.uint8)
            n=min(len(g),len(wv)); d=np.nonzero(g[:n]!=wv[:n])[0]
            key=(fmt,w,h,mode,q,method)
            bad.setdefault(key,[]).append((it,len(got),len(want),int(d[0]) if len(d) else -1,len(d)))
for k,v in bad.items(): print("MISMATCH",k,len(v),v[:4])
print("done", len(cases))
