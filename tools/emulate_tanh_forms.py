"""Numpy float32 emulation of the LSTM kernels cell formulation (gate columns pre-scaled, sigmoid as 1/(1+2^z), tanh as
2*sigmoid(2x)-1) against variants with an accurate tanh in chosen places: where does the 3x float32 noise of Q come from,
and would an accurate tanh(c_T) in the LAST step alone remove it?  Output: profiles/r3/tanh_forms.txt"""
import numpy as np, sys
sys.path[:0]=['/root/repo','/root/repo/tests']
from oracle import ref_path as orc
from test_gpu_parity import synth_lstm
f32=np.float32
L2E=f32(1.4426950408889634)
def sig_k(z):   # z pre-scaled by -log2e: 1/(1+2^z)
    return f32(1)/(f32(1)+np.exp2(z).astype(f32))
def tanh_k(x):  # 2*sigma(2x)-1
    return f32(2)*sig_k((f32(-2.8853900817779268)*x).astype(f32))-f32(1)
def tanh_acc(x):
    return np.tanh(x.astype(np.float64)).astype(f32)     # stand-in for "a relatively accurate f32 tanh"
def run(states, l, mode):
    K,b,wo,bo=l["kernel"],l["bias"],l["w_out"],l["b_out"]
    B,W,I=states.shape; H=K.shape[1]//4
    # pre-scaled weights as the kernels hold them
    sc=np.ones(4*H,f32)*(-L2E); sc[H:2*H]=f32(-2)*L2E
    Ks=(K.astype(np.float64)*sc).astype(f32); bs=((b.astype(np.float64)+np.r_[np.zeros(2*H),np.ones(H),np.zeros(H)])*sc).astype(f32)
    h=np.zeros((B,H),f32); c=np.zeros((B,H),f32)
    for t in range(W):
        z=(np.concatenate([states[:,t],h],1).astype(f32)@Ks+bs).astype(f32)
        i,j,f,o=z[:,:H],z[:,H:2*H],z[:,2*H:3*H],z[:,3*H:]
        si,sf,so=sig_k(i),sig_k(f),sig_k(o)
        tj=f32(2)*sig_k(j)-f32(1)
        if mode in ("j_acc","all_acc"): tj=tanh_acc((j/(f32(-2)*L2E)).astype(f32))
        c=(sf*c+si*tj).astype(f32)
        last=t==W-1
        if mode=="all_c_acc" or mode=="all_acc" or (mode=="last_c_acc" and last): tc=tanh_acc(c)
        else: tc=tanh_k(c)
        h=(so*tc).astype(f32)
    return (h@wo+bo).astype(f32)
rng=np.random.default_rng(5)
res={m:[] for m in ("numpy32","kernel","last_c_acc","all_c_acc","all_acc")}
for trial in range(40):
    H=int(rng.choice([64,128])); I=2; W=25
    l=synth_lstm(rng,I,H)
    st=rng.random((300,W,I)).astype(f32); st[:,:,1]=(st[:,:,1]>0.7)
    q64=orc.lstm_q(st,l["kernel"],l["bias"],l["w_out"],l["b_out"],1.0,dtype=np.float64)
    q32=orc.lstm_q(st,l["kernel"],l["bias"],l["w_out"],l["b_out"],1.0,dtype=np.float32)
    res["numpy32"].append(np.abs(q32-q64).mean())
    for m in ("kernel","last_c_acc","all_c_acc","all_acc"):
        res[m].append(np.abs(run(st,l,m).astype(np.float64)-q64).mean())
for m,v in res.items(): print(f"{m:12s} mean abs err of Q {np.mean(v):.3e}  (x numpy32: {np.mean(v)/np.mean(res['numpy32']):.2f})")
