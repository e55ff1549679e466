import sys, numpy as np, torch, ctypes as C
sys.path.insert(0,'.')
from muzero_amd.engine import SearchEngine
from muzero_amd import _lib
w=np.load('tests/golden/weights_tag123.npz')
dims=tuple(int(w[k]) for k in ("obs_dim","A","S_dim","H","L"))
eng=SearchEngine(dims,1,0); eng.set_weights(w["blob"])
K=5
def ok(B):
    dev=eng.device
    obs=torch.zeros((B,4),device=dev); act=torch.zeros((K,B,2),device=dev); tv=torch.zeros((K,B),device=dev); tp=torch.full((K,B,2),0.5,device=dev)
    g=torch.zeros(eng.weight_count(),device=dev); l=torch.zeros(3*K,device=dev); pr=torch.zeros((K,B),device=dev)
    p=lambda t: C.c_void_p(t.data_ptr())
    rc=eng.lib.mzs_train_grad_static(eng.h,B,K,p(obs),p(act),p(tv),p(tv),p(tp),None,C.c_float(1.0/B),p(g),p(l),p(pr),eng._stream())
    torch.cuda.synchronize()
    return rc==0
lo,hi=1,5000
while lo<hi:
    m=(lo+hi+1)//2
    if ok(m): lo=m
    else: hi=m-1
print("largest batch accepted by mzs_train_grad_static (CartPole MLP, unroll 5):",lo)
