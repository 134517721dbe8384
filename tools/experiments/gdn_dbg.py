import sys, torch
sys.path.insert(0,'.')
import licos_amd
from oracle import model as om
c,hw,inv = 128,(16,16),False
sd={}; om._gdn_init(sd,"g.",c); g=torch.Generator().manual_seed(c)
sd["g.gamma"]=sd["g.gamma"]+0.05*torch.rand(c,c,generator=g); sd["g.beta"]=sd["g.beta"]*(0.5+torch.rand(c,generator=g))
x=3*torch.randn(2,c,*hw,generator=g)
ref=om.gdn(x,sd,"g.",inverse=inv).double()
m=licos_amd.GDN(c,inverse=inv); m.load_state_dict({k[2:]:v for k,v in sd.items()})
out=m.to("cuda")(x.to("cuda")).detach().cpu().double()
d=(out-ref).abs()/ref.abs().clamp_min(1e-12)
idx=torch.nonzero(d>3e-5)
print("outliers", idx.shape[0], "of", d.numel())
for (b,i,yy,xx) in idx[:12].tolist():
    print((b,i,yy,xx), "x", float(x[b,i,yy,xx]), "out", float(out[b,i,yy,xx]), "ref", float(ref[b,i,yy,xx]), "rel", float(d[b,i,yy,xx]))
ped=(2**-18)**2
beta=(torch.clamp(sd["g.beta"], min=(1e-6+ped)**0.5)**2-ped).double()
gamma=(torch.clamp(sd["g.gamma"], min=ped**0.5)**2-ped).double()
b,yy,xx=0,13,6
xv=x[b,:,yy,xx].double()
exact=beta+gamma@(xv**2)
ngpu=(xv/out[b,:,yy,xx])**2
delta=(ngpu-exact)
print("delta norm per channel (first 16):", [round(float(v),6) for v in delta[:16]])
print("gamma[:16,9]*256:", [round(float(v)*256,3) for v in gamma[:16,9]])
big=torch.nonzero(delta.abs()>2e-4).flatten().tolist()
print("channels with |delta|>2e-4:", big)
print("x small channels (|x|<0.13):", torch.nonzero(xv.abs()<0.13).flatten().tolist(), [round(float(v),4) for v in xv[xv.abs()<0.13]])
