import sys, contextlib, io
sys.path[:0]=["/root/repo/cold-diffusion-models_amd","/root/repo"]
import torch
from oracle import cold_oracle as O
from colddiff import runtime as rt
from deblurring_diffusion_pytorch import Unet
rt.set_precision("bf16"); rt.bump_weights_epoch()
torch.manual_seed(9)
with contextlib.redirect_stdout(io.StringIO()):
    net = Unet(dim=64, dim_mults=(1, 2, 4), channels=3)
sd = {k: v.clone() for k, v in net.state_dict().items()}
x, t = torch.rand(2, 3, 64, 64) * 2 - 1, torch.tensor([1, 40])
gy = torch.randn(2, 3, 64, 64) / 1000
net = net.to("cuda:0")
y = net(x.cuda(), t.cuda()); y.backward(gy.cuda())
ps = {k: v.clone().requires_grad_() for k, v in sd.items()}
yr = O.unet_forward(ps, x, t); yr.backward(gy)
gmax = max(p.grad.abs().max().item() for p in ps.values())
rows=[]
for n,p in net.named_parameters():
    r=ps[n].grad; e=(p.grad.cpu()-r).abs().max().item(); lim=max(r.abs().max().item(),1e-2*gmax)
    rows.append((e/lim,n,r.abs().max().item()/gmax))
rows.sort(reverse=True)
for v,n,s in rows[:12]: print("%.4f %s (|g|max / global %.3g)"%(v,n,s))
