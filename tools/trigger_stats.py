"""How often and how long the limiter re-triggers on bench.py's synthetic programmes (CPU model of the
reference's recurrence, audio_effect_peak_limiter.c:237-265, on the bench's own signal recipe).
A "run" = maximal stretch of consecutive trigger samples = what the kernels' chain wave walks serially.
Round-2 finding (NOTEBOOK.md 4): runs are LONG (~240 samples = the look-ahead) but RARE - hot headline 0.2
per 1024-sample chunk (3.4 % of samples), sparse 0.7, cfg2 hot 0.5 - so the chain is about 7 % / 20 % /
12 % of a stream's time, not what bounds it.      python tools/trigger_stats.py"""
import sys, numpy as np
import os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import torch, bench, oracle_lib as O
def stats(signal, S=6, F=16, fs=1024, out="bin"):
    dev=torch.device('cpu')
    x=bench.synth_hot_device(S,16,F,fs,1000,dev)
    if signal=='quiet': x=torch.randn_like(x)*0.05
    if signal=='sparse':
        x=torch.randn_like(x)*0.05
        tt=torch.arange(F*fs).view(1,F,1,fs); ph=(torch.arange(S)*389%1531).view(S,1,1,1)
        x+=(((tt-ph)%1531)<8)*torch.where(tt%2==0,1.0,-1.0)*1.5
    x=x.numpy()  # [S][F][16][fs]
    mx=O.get_h2m(3,O.SS['BINAURAL']) if out=='bin' else O.get_m2m(O.SS['L714'],O.SS['J'])
    W=mx.array().reshape(mx.n,mx.m) if out=='bin' else mx.array().reshape(mx.m,mx.n).T
    thr=np.float32(10**(-1/20)); inc=np.float32(1)/np.float32(48000); atk=np.float32(0.001); rel=np.float32(0.2)
    runs=[]; trig_total=0; n_total=0; events_per_chunk=[]
    for s in range(S):
        xs=x[s].transpose(1,0,2).reshape(16,-1)[:W.shape[1]]
        y=(W.astype(np.float64)@xs.astype(np.float64)).astype(np.float32)
        pm=np.abs(y).max(axis=0)
        n=len(pm); D=240
        pk=np.zeros(n,np.float32)
        ring=np.zeros(D,np.float32)
        g=np.float32(1); gs=np.float32(-1); ge=np.float32(-1); tc=np.float32(-1)
        trig=np.zeros(n,bool)
        for k in range(n):
            peak=ring.max()
            if tc!=-1 and tc<atk:
                tc=np.float32(tc+inc); xx=tc/atk; e=np.float32(1) if xx>1 else np.float32(1-(xx-1)*(xx-1)); g=np.float32(gs-e*(gs-ge))
            elif tc!=-1 and tc<rel+atk:
                tc=np.float32(tc+inc); xx=(tc-atk)/rel; e=np.float32(1) if xx>1 else np.float32(1-(xx-1)*(xx-1)); g=np.float32(ge+e*(1-ge))
            else: g=np.float32(1)
            if peak*g>thr:
                gs=g; ge=np.float32(thr/peak); tc=np.float32(0); trig[k]=True
            ring[k%D]=pm[k]
        trig_total+=trig.sum(); n_total+=n
        # runs
        d=np.diff(np.concatenate([[0],trig.astype(int),[0]])); st=np.nonzero(d==1)[0]; en=np.nonzero(d==-1)[0]
        runs+=list(en-st)
        for c in range(0,n,1024):
            events_per_chunk.append(int(((st>=c)&(st<c+1024)).sum()))
    runs=np.array(runs)
    print(signal,out,"trigger fraction %.3f"%(trig_total/n_total),"runs:",len(runs),"per chunk mean %.1f max %d"%(np.mean(events_per_chunk),max(events_per_chunk)),
          "run length mean %.1f median %d p90 %d max %d"%(runs.mean(),np.median(runs),np.percentile(runs,90),runs.max()) if len(runs) else "")
    if len(runs): print("   hist run len [1,2,3-4,5-8,9-16,17-64,65-256,>256]:",np.histogram(runs,bins=[1,2,3,5,9,17,65,257,10**6])[0])
stats('hot'); stats('sparse'); stats('hot',out='J')
