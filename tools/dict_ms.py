#!/usr/bin/env python3
"""Kernel times of the dictionary stage's encode on the bench shard (k_dict_match, k_dict_encode): python tools/dict_ms.py
(round 4: without its three byte stores per step k_dict_encode takes 0.96 instead of 1.02 ms: the stores are not its bound)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
from comprox_amd import CrGpu, corpus
import bench
dev=torch.device('cuda',0); BLOCK=65536
host=corpus.enwik_like(100_000_000, 8); n=host.size; nb=(n+BLOCK-1)//BLOCK
d_in=torch.from_numpy(host).to(dev)
off=torch.arange(nb,dtype=torch.int64,device=dev)*BLOCK
size=torch.from_numpy(np.minimum(BLOCK, n-np.arange(nb,dtype=np.int64)*BLOCK).astype(np.int32)).to(dev)
g=CrGpu(0); g.set_stream(torch.cuda.current_stream().cuda_stream)
d=g.dict_create(bench.host_dicpick(g.lib, host))
s1=(BLOCK+1+63)//64*64
st1=torch.zeros(nb*s1,dtype=torch.uint8,device=dev); st1_off=torch.arange(nb,dtype=torch.int64,device=dev)*s1; len1=torch.zeros(nb,dtype=torch.int32,device=dev)
for r in range(3):
    g.lib.crgpu_dict_encode_blocks_dev(g.h, d.h, d_in.data_ptr(), off.data_ptr(), size.data_ptr(), nb, BLOCK, st1.data_ptr(), st1_off.data_ptr(), len1.data_ptr(), 1)
print({k: round(v,3) for k,v in g.last_stage_ms().items()})
