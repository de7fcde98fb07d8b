import torch, torch.nn.functional as F
torch.manual_seed(0)
def wino(x, w):
    # x [N,C,H,W] fp32, w [O,C,3,3]; F(2x2,3x3), pad 1
    N,C,H,W = x.shape; O = w.shape[0]
    BT = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=x.dtype)
    G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=x.dtype)
    AT = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=x.dtype)
    U = torch.einsum('ij,ocjk,lk->ocil', G, w, G)           # [O,C,4,4]
    xp = F.pad(x, (1,1,1,1))
    tiles = xp.unfold(2,4,2).unfold(3,4,2)                     # [N,C,H/2,W/2,4,4]
    V = torch.einsum('ij,nchwjk,lk->nchwil', BT, tiles, BT)
    M = torch.einsum('nchwil,ocil->nohwil', V, U)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', AT, M, AT)       # [N,O,H/2,W/2,2,2]
    return Y.permute(0,1,2,4,3,5).reshape(N,O,H,W)
for C,O,H in [(64,64,24),(256,256,12),(512,512,6)]:
    x = torch.relu(torch.randn(2,C,H,H))           # post-ReLU activations
    w = torch.randn(O,C,3,3) * (2.0/(C*9))**0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    d = F.conv2d(x, w, padding=1)
    wi = wino(x, w)
    s = ref.abs().max()
    print(C,O,H, 'direct fp32 err %.2e' % ((d.double()-ref).abs().max()/s), 'wino fp32 err %.2e' % ((wi.double()-ref).abs().max()/s),
          'rms rel: direct %.2e wino %.2e' % (((d.double()-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()), ((wi.double()-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt())))
