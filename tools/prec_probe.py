"""Per-GEMM precision plan of the clustering stage, measured on the CPU oracle: the operands of ONE kind of GEMM (token
convolution / kv projection / q + proj) rounded to bf16 (= a one-pass bf16 product), everything else fp32; prints the deviation of
the five losses, of the global tokens and of G at c1_b16 and c2_b128.  CPU only: python tools/prec_probe.py"""
import sys, math, numpy as np, torch, torch.nn.functional as F
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle')); sys.path.insert(0,os.path.join(ROOT,'tests'))
import nr_oracle as O
from neighborretr_amd import synth
torch.set_num_threads(8)
def bf(x): return x.to(torch.bfloat16).float()
MODE=set()
_conv, _lin = F.conv1d, F.linear
def conv1d(x,w,**kw):
    if 'conv' in MODE: return _conv(bf(x),bf(w),**kw)
    return _conv(x,w,**kw)
CUR=[None]
def linear(x,w,b=None):
    tag=None
    if w.shape[0]==1024 and w.shape[1]==512 and CUR[0]=='ctm': tag='kv'
    elif w.shape==(512,512) and CUR[0]=='ctm': tag='qp'
    if tag and tag in MODE: return _lin(bf(x),bf(w),b)
    return _lin(x,w,b)
import types
orig_stage=O.ctm_stage
def stage(*a,**k):
    CUR[0]='ctm'
    F.conv1d, F.linear = conv1d, linear
    try: return orig_stage(*a,**k)
    finally:
        F.conv1d, F.linear = _conv,_lin; CUR[0]=None
O.ctm_stage=stage
def run(seed,B,Nt,Nv,M,K):
    hp=dict(synth.DEFAULT_HP,num_neighbors=K)
    P={k:torch.from_numpy(v) for k,v in synth.make_params(7).items()}
    x={k:torch.from_numpy(v) for k,v in synth.make_problem(seed,B,Nt,Nv,M).items()}
    nz={k:torch.from_numpy(v) for k,v in synth.make_noise(seed,B,Nt,Nv).items()}
    with torch.no_grad():
        out,parts=O.compute_losses(x['text_feat'],x['video_feat'],x['text_mask'],x['video_mask'],x['mb_feat_t'],x['mb_feat_v'],x['mb_mask_t'],x['mb_mask_v'],P,hp,torch.tensor(100.0),nz,return_parts=True)
    return np.array([float(o) for o in out]),parts
for case in [(1001,16,24,12,128,8),(1002,128,24,12,512,20)]:
    MODE.clear()
    base,pb=run(*case)
    print(case, 'base',base, '|G|max',float(pb['G'].abs().max()), '|gt| rms', float(pb['gt'].pow(2).mean().sqrt()))
    for mode in (['conv'],['kv'],['qp'],['conv','kv','qp']):
        MODE.clear(); MODE.update(mode)
        l,p=run(*case)
        print('  ',mode,'dL',np.abs(l-base), 'rel d gt', float((p['gt']-pb['gt']).norm()/pb['gt'].norm()), 'dG max', float((p['G']-pb['G']).abs().max()))
