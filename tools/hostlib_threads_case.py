"""A small multi-threaded use of the host library for sanitizer runs over the EMULATED host library (tests/emu.py build_hostlib, compiled with
-fsanitize=thread): 12 device-resident bins over 4 submitting stream threads (kmc_hip_process_bins_device), twice, then three caller threads on the
host-buffer entry; results against the oracle. Run: LD_PRELOAD=$(gcc -print-file-name=libtsan.so) KMC_HIP_LIB=<tsan build of the emulated
library> python tools/hostlib_threads_case.py   (clean at the end of round 2: 0 reports)."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from kmc_amd import capi
import oracle_py as O
ctx=capi.Context((0,))
k=27
bins=capi.synth_bins(seed=3, genome_len=20_000, n_reads=1500, k=k, n_bins=12, n_threads=1)
p=capi.make_params(k,lut_prefix_len=3)
rec=ctx.out_rec_bytes(p); nl=ctx.lut_entries(p)
descs=(capi.BinDesc*len(bins))(); allocs=[]
for i,(img,nrec,packs,_) in enumerate(bins):
    ps=np.concatenate([[0],np.cumsum(packs)]).astype(np.uint64)
    cap=((nrec+1)//2)*rec
    d_in=ctx.malloc(img.size+256); d_ps=ctx.malloc(ps.nbytes); d_out=ctx.malloc(cap+256); d_lut=ctx.malloc(max(nl,1)*8); d_small=ctx.malloc(64)
    ctx.h2d(d_in,np.concatenate([img,np.zeros(256,dtype=np.uint8)])); ctx.h2d(d_ps,ps)
    allocs.append((d_in,d_ps,d_out,d_lut,d_small,cap))
    descs[i]=capi.BinDesc(d_in,img.size,nrec,d_ps,packs.size,d_out,cap,d_small+32,d_lut,d_small)
for rep in range(2):
    ctx.process_bins_device(p,descs,4)
    ctx.synchronize()
po=O.make_params(k,lut_prefix_len=3)
ok=True
for i,(img,nrec,packs,_) in enumerate(bins):
    small=np.zeros(8,dtype=np.uint64); ctx.d2h(small,allocs[i][4])
    out=np.zeros(int(small[4]),dtype=np.uint8)
    if out.size: ctx.d2h(out,allocs[i][2])
    w=O.process_bin(po,img,nrec)
    ok &= np.array_equal(out,w[0]) and np.array_equal(small[:4],w[2])
# host-buffer path from several threads too
import threading
def worker(slot):
    for img,nrec,packs,_ in bins[slot::3]:
        out,lut,st=ctx.process_bin(p,img,nrec,packs)
ths=[threading.Thread(target=worker,args=(s,)) for s in range(3)]
[t.start() for t in ths]; [t.join() for t in ths]
print('result ok' if ok else 'MISMATCH')
ctx.close()
