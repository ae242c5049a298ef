import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from oracle import deform as od
from d3ga_amd.cage_deform import cage_deform
seed=int(sys.argv[1]); DEV="cuda"
rng = np.random.default_rng(9000 + seed)
V, T = int(rng.integers(4, 400)), int(rng.integers(1, 900))
P = int(rng.choice([1, 2, 63, 64, 65, 255, 257, int(rng.integers(1, 5000))]))
g = torch.Generator().manual_seed(seed)
canon = torch.randn(V, 3, generator=g)
tetras = torch.stack([torch.randperm(V, generator=g)[:4] for _ in range(T)]).int()
tet_id = torch.randint(0, T, (P,), generator=g).int()
barys = torch.rand(P, 4, generator=g); barys = barys / barys.sum(1, keepdim=True)
cg = od.canonical_gradient(canon.double(), tetras.long(), tet_id.long())
tp0 = canon + 0.1 * torch.randn(V, 3, generator=g)
raw_s, rot = 0.3 * torch.randn(P, 3, generator=g) - 2.0, torch.randn(P, 4, generator=g)
dbary = 0.05 * torch.randn(P, 4, generator=g)
fused = bool(seed % 2)
up_m, up_c = torch.randn(P, 3, generator=g), torch.randn(P, 6, generator=g)
def run(dt):
    L = lambda t: t.detach().clone().to(dt).requires_grad_(True)
    tp,b,s,r,d = L(tp0),L(barys),L(raw_s),L(rot),L(dbary)
    m,c = od.cage_deform(tp, tetras.long(), tet_id.long(), (b+d) if fused else b, cg.to(dt), torch.exp(s), r)
    ((m*up_m.to(dt)).sum()+(c*up_c.to(dt)).sum()).backward()
    return s.grad.double().numpy(), r.grad.double().numpy(), c.detach().double().numpy()
s64,r64,c64=run(torch.float64); s32,r32,c32=run(torch.float32)
cu=lambda t: t.detach().clone().to(DEV).requires_grad_(True)
tp,b,sr,r,db=(cu(t) for t in (tp0,barys,raw_s,rot,dbary))
for variant in ("fused","plain"):
    for t in (tp,b,sr,r,db): t.grad=None
    if variant=="fused":
        m,c=cage_deform(tp,tetras.to(DEV),tet_id.to(DEV),b,cg.float().to(DEV),sr,r,delta_barys=db,scale_activation="exp")
    else:
        m,c=cage_deform(tp,tetras.to(DEV),tet_id.to(DEV),b+db,cg.float().to(DEV),torch.exp(sr),r)
    ((m*up_m.to(DEV)).sum()+(c*up_c.to(DEV)).sum()).backward()
    sh=sr.grad.cpu().double().numpy()
    allow=1e-3*np.abs(s64)+1e-6*np.abs(s64).max()
    a=np.abs(sh-s64)/allow; i=np.unravel_index(a.argmax(),a.shape)
    print(variant,"worst",i,"hip",sh[i],"f64",s64[i],"f32torch",s32[i],"excess",a[i],"max|g|",np.abs(s64).max(),"row f64",s64[i[0]],"row hip",sh[i[0]], "cov err", np.abs(c.detach().cpu().double().numpy()-c64).max()/np.abs(c64).max(), "raw_s row", raw_s[i[0]].numpy(), "cg max row", float(cg[i[0]].abs().max()))
