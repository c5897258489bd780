# experiment: encode of the bench batch on one stream vs its two halves on two streams (two contexts) at once
import os, sys, time, json
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from comprox_amd import CrGpu, CODEC_ROP, bound, corpus
import bench
dev=torch.device('cuda',0)
BLOCK=65536
host=corpus.enwik_like(100_000_000, 8)
n=host.size; nb=(n+BLOCK-1)//BLOCK
d_in=torch.from_numpy(host).to(dev)
off=torch.arange(nb,dtype=torch.int64,device=dev)*BLOCK
size=torch.from_numpy(np.minimum(BLOCK, n-np.arange(nb,dtype=np.int64)*BLOCK).astype(np.int32)).to(dev)
s_a=torch.cuda.Stream(dev); s_b=torch.cuda.Stream(dev)
ga=CrGpu(0); gb=CrGpu(0)
ga.set_stream(s_a.cuda_stream); gb.set_stream(s_b.cuda_stream)
dic=bench.host_dicpick(ga.lib, host)
da=ga.dict_create(dic); db=gb.dict_create(dic)
s1=(BLOCK+1+63)//64*64; s2=(bound(CODEC_ROP,BLOCK+1)+63)//64*64
def bufs():
    return dict(st1=torch.zeros(nb*s1,dtype=torch.uint8,device=dev), st1_off=torch.arange(nb,dtype=torch.int64,device=dev)*s1, len1=torch.zeros(nb,dtype=torch.int32,device=dev),
                enc=torch.zeros(nb*s2,dtype=torch.uint8,device=dev), enc_off=torch.arange(nb,dtype=torch.int64,device=dev)*s2, esz=torch.zeros(nb,dtype=torch.int32,device=dev))
A=bufs(); Bf=bufs()
def enc(g,d,U,b0,k):
    g.lib.crgpu_dict_encode_blocks_dev(g.h, d.h, d_in.data_ptr(), off[b0:].data_ptr(), size[b0:].data_ptr(), k, BLOCK, U['st1'].data_ptr(), U['st1_off'].data_ptr(), U['len1'][b0:].data_ptr(), 0)
    g.encode_blocks_dev(CODEC_ROP, U['st1'].data_ptr(), U['st1_off'].data_ptr(), U['len1'][b0:].data_ptr(), k, BLOCK+1, U['enc'].data_ptr(), U['enc_off'].data_ptr(), U['esz'][b0:].data_ptr())
kA=(nb+1)//2; kB=nb-kA
def whole():
    enc(ga,da,A,0,nb)
def halves():
    enc(ga,da,A,0,kA); enc(gb,db,Bf,kA,kB)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/reps*1e3
tw=timeit(whole); th=timeit(halves)
okA=torch.equal(A['esz'][:kA], A['esz'][:kA])
whole(); torch.cuda.synchronize(); ref=A['esz'].clone()
halves(); torch.cuda.synchronize()
same=bool(torch.equal(ref[:kA], A['esz'][:kA])) and bool(torch.equal(ref[kA:], Bf['esz'][kA:]))
print(json.dumps({"encode_whole_batch_ms": round(tw,3), "encode_two_halves_two_streams_ms": round(th,3), "sizes_equal": same}))
