import torch, numpy as np, ctypes
from margipose_amd import _lib
from margipose_amd._lib import BnAddOperands
L=_lib.lib()
import sys
B,F,C,J=(int(sys.argv[1]) if len(sys.argv) > 1 else 32),32,32,17
a=[torch.randn(B,F,F,C,device='cuda') for _ in range(3)]; b=[torch.randn(B,F,F,C,device='cuda') for _ in range(3)]
v=[torch.randn(C,device='cuda') for _ in range(4)]
lg=[torch.empty(B,J,F,F,device='cuda') for _ in range(3)]; h=[torch.empty(B,J,F,F,device='cuda') for _ in range(3)]
pc=torch.empty(3,B*J,2,device='cuda'); xyz=torch.empty(B,J,3,device='cuda')
ops=[]
for c in range(3):
    ao=BnAddOperands(); ao.a,ao.a_scale,ao.a_shift=a[c].data_ptr(),v[0].data_ptr(),v[1].data_ptr(); ao.b,ao.b_scale,ao.b_shift=b[c].data_ptr(),v[2].data_ptr(),v[3].data_ptr(); ao.out=lg[c].data_ptr(); ops.append(ao)
ops=(BnAddOperands*3)(*ops); st=_lib.stream_ptr()
def two():
    L.mpose_bn_add_fwd(ops,3,F*F,B,C,1,J,st); L.mpose_softmax_dsnt_fwd(_lib.ptr_array(lg),_lib.ptr_array(h),None,_lib.ptr(xyz),3,B*J,F,F,0,st)
def one():
    L.mpose_bn_add_softmax_fwd(ops,_lib.ptr_array(h),_lib.ptr(pc),3,B,F,F,C,J,0,st); L.mpose_coords_merge(_lib.ptr(pc),_lib.ptr(xyz),B*J,st)
def one_only():
    L.mpose_bn_add_softmax_fwd(ops,_lib.ptr_array(h),_lib.ptr(pc),3,B,F,F,C,J,0,st)
for name,f in (('two',two),('fused+merge',one),('fused',one_only)):
    N = 200 if B <= 256 else 20
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): f()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1)/N*1000
    print(name, us, 'us', '%.3f of 8 TB/s on 12 B per heatmap element' % (3*B*J*F*F*12/us/1e6/8000))
