"""numpy prototype of the S2 multigrid-PCG (design experiments only; not part of the product or the tests)."""
import sys; sys.path.insert(0,'/root/repo/tests')
import numpy as np, time, oracle_bind, synth
orc=oracle_bind.load(); orc._decl_color()
def system(H,W,lamda,seed=3,rough_frac=0.1):
    img=synth.image(seed,H,W); lab=np.ascontiguousarray(orc.bgr2lab(img).astype(np.float64)/255.0)
    rng=np.random.default_rng(1); rough=np.where(rng.random(H*W)<rough_frac,1e-6,1.0)
    d=np.empty(H*W); wx=np.empty(H*W); wy=np.empty(H*W)
    orc.l.orc_wls_system(lab.reshape(-1),H,W,lamda,1.2,rough,d,wx,wy)
    return rough.reshape(H,W), wx.reshape(H,W), wy.reshape(H,W)
def diag_of(r,wx,wy):
    d=r.copy(); d+=wx; d+=wy; d[:,1:]+=wx[:,:-1]; d[1:,:]+=wy[:-1,:]; return d
def apply(d,wx,wy,x):
    y=d*x; y[:,:-1]-=wx[:,:-1]*x[:,1:]; y[:,1:]-=wx[:,:-1]*x[:,:-1]; y[:-1,:]-=wy[:-1,:]*x[1:,:]; y[1:,:]-=wy[:-1,:]*x[:-1,:]; return y
def coarsen(r,wx,wy):
    H,W=r.shape; Hc,Wc=(H+1)//2,(W+1)//2
    def pad(a): 
        p=np.zeros((Hc*2,Wc*2)); p[:H,:W]=a; return p
    rp=pad(r); rc=rp[0::2,0::2]+rp[0::2,1::2]+rp[1::2,0::2]+rp[1::2,1::2]
    wxp=pad(wx); wxc=wxp[0::2,1::2]+wxp[1::2,1::2]   # edges from odd col to next even col
    wyp=pad(wy); wyc=wyp[1::2,0::2]+wyp[1::2,1::2]
    return rc,wxc,wyc
def restrict(v):
    H,W=v.shape; Hc,Wc=(H+1)//2,(W+1)//2; p=np.zeros((Hc*2,Wc*2)); p[:H,:W]=v
    return p[0::2,0::2]+p[0::2,1::2]+p[1::2,0::2]+p[1::2,1::2]
def prolong(vc,H,W):
    return np.repeat(np.repeat(vc,2,0),2,1)[:H,:W]
class MG:
    def __init__(s,r,wx,wy,nu=2,omega=0.8,kappa=1.0,coarse=8):
        s.lv=[]; s.nu=nu; s.om=omega; s.kap=kappa
        while True:
            wx=wx.copy(); wx[:,-1]=0; wy=wy.copy(); wy[-1,:]=0
            s.lv.append((diag_of(r,wx,wy),wx,wy))
            if min(r.shape)<=coarse: break
            r,wx,wy=coarsen(r,wx,wy)
    def vcycle(s,l,b):
        d,wx,wy=s.lv[l]
        if l==len(s.lv)-1:
            x=np.zeros_like(b)
            for _ in range(50): x=x+s.om*(b-apply(d,wx,wy,x))/d
            return x
        x=s.om*b/d
        for _ in range(s.nu-1): x=x+s.om*(b-apply(d,wx,wy,x))/d
        res=b-apply(d,wx,wy,x)
        ec=s.vcycle(l+1,restrict(res))
        x=x+s.kap*prolong(ec,*b.shape)
        for _ in range(s.nu): x=x+s.om*(b-apply(d,wx,wy,x))/d
        return x
def pcg(d,wx,wy,b,x0,prec,rtol=1e-10,maxit=20000):
    x=x0.copy(); r=b-apply(d,wx,wy,x); z=prec(r); p=z.copy(); rz=(r*z).sum(); bb=(b*b).sum(); it=0
    while (r*r).sum()>rtol**2*bb and it<maxit:
        Ap=apply(d,wx,wy,p); al=rz/(p*Ap).sum(); x+=al*p; r-=al*Ap; z=prec(r); rz2=(r*z).sum(); p=z+(rz2/rz)*p; rz=rz2; it+=1
    return x,it
for (H,W,lam) in [(256,256,0.024*256),(256,256,0.024*16),(256,256,0.024*4),(350,350,0.024*64)]:
    r,wx,wy=system(H,W,lam); wx[:,-1]=0; wy[-1,:]=0; d=diag_of(r,wx,wy)
    rng=np.random.default_rng(5); x0=rng.random((H,W)); b=r*x0
    t=time.time(); xj,itj=pcg(d,wx,wy,b,x0,lambda v:v/d); tj=time.time()-t
    for (nu,om,kap) in [(1,0.8,1.0),(2,0.8,1.0),(2,0.8,1.5),(3,0.8,1.8),(2,0.67,2.0)]:
        mg=MG(r,wx,wy,nu,om,kap); t=time.time(); xm,itm=pcg(d,wx,wy,b,x0,lambda v:mg.vcycle(0,v)); tm=time.time()-t
        print(H,W,"lam %.3f"%lam,"jacobi it",itj,"| MG nu",nu,"om",om,"kap",kap,"it",itm,"maxdiff %.2e"%np.abs(xj-xm).max())

print("---- RB-GS")
class MGRB(MG):
    def __init__(s,r,wx,wy,nu=1,kappa=1.0,coarse=8):
        MG.__init__(s,r,wx,wy,nu,1.0,kappa,coarse)
        s.masks=[]
        for (d,wx_,wy_) in s.lv:
            yy,xx=np.mgrid[0:d.shape[0],0:d.shape[1]]
            s.masks.append(((yy+xx)%2==0))
    def half(s,l,x,b,red):
        d,wx,wy=s.lv[l]; m=s.masks[l] if red else ~s.masks[l]
        off=d*x-apply(d,wx,wy,x)   # sum w x_nbr
        xn=(b+off)/d
        x=x.copy(); x[m]=xn[m]; return x
    def vcycle(s,l,b):
        d,wx,wy=s.lv[l]
        if l==len(s.lv)-1:
            x=np.zeros_like(b)
            for _ in range(30): x=s.half(l,x,b,True); x=s.half(l,x,b,False)
            for _ in range(30): x=s.half(l,x,b,False); x=s.half(l,x,b,True)
            return x
        x=np.zeros_like(b)
        for _ in range(s.nu): x=s.half(l,x,b,True); x=s.half(l,x,b,False)
        res=b-apply(d,wx,wy,x)
        ec=s.vcycle(l+1,restrict(res))
        x=x+s.kap*prolong(ec,*b.shape)
        for _ in range(s.nu): x=s.half(l,x,b,False); x=s.half(l,x,b,True)
        return x
for (H,W,lam) in [(256,256,0.024*256),(256,256,0.024*16),(256,256,0.024*4),(350,350,0.024*64)]:
    r,wx,wy=system(H,W,lam); wx[:,-1]=0; wy[-1,:]=0; d=diag_of(r,wx,wy)
    rng=np.random.default_rng(5); x0=rng.random((H,W)); b=r*x0
    for rt in (1e-6,):
        mgj=MG(r,wx,wy,2,0.8,1.0); xj,itj=pcg(d,wx,wy,b,x0,lambda v:mgj.vcycle(0,v),rtol=rt)
        for (nu,kap) in [(1,1.0),(1,1.4),(2,1.0)]:
            mg=MGRB(r,wx,wy,nu,kap); xm,itm=pcg(d,wx,wy,b,x0,lambda v:mg.vcycle(0,v),rtol=rt)
            print(H,W,"lam %.3f"%lam,"rtol",rt,"jacobiV22 it",itj,"| RBGS nu",nu,"kap",kap,"it",itm,"maxdiff %.1e"%np.abs(xj-xm).max())
