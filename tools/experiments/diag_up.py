import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as TF, ctypes as C
from torchsr_amd.layers import Conv2d
from torchsr_amd import _lib
dev=torch.device('cuda:0')
for (n,h,w) in ((16,32,32),(4,32,32),(8,32,32),(16,16,16)):
    torch.manual_seed(1)
    conv=Conv2d(64,64,3,1,1,bias=False,up=2).to(dev)
    x=torch.rand(n,64,h,w)-0.5
    want=TF.conv2d(TF.interpolate(x,scale_factor=2,mode='nearest'),conv.weight.detach().cpu(),None,1,1)
    xg=x.permute(0,2,3,1).contiguous().to(dev)
    with torch.no_grad(): y=conv(xg).cpu().permute(0,3,1,2)
    d=conv._st.desc(n,h,w); out=(C.c_int*6)(); _lib.lib().srx_conv2d_plan(C.byref(d),0,out)
    err=(y-want).abs().amax(1)  # [n, 2h, 2w]
    bad=(err>1e-4).nonzero()
    print((n,h,w),'plan',list(out),'max err',float(err.max()),'bad',len(bad), bad[:6].tolist(), bad[-3:].tolist())
    if len(bad):
        m=(bad[:,0]*(4*h*w)+bad[:,1]*(2*w)+bad[:,2])
        print('  bad m range',int(m.min()),int(m.max()),'distinct tiles(144)',sorted(set((m//144).tolist()))[:10])
