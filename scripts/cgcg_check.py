"""Chronopoulos-Gear (single-reduction) PCG vs textbook PCG on the numpy prototype: same iteration counts, same solution."""
import os
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mg_proto.py')).read().split("for (H,W,lam) in")[0])
def pcg_cg(d,wx,wy,b,x0,prec,rtol=1e-6,maxit=2000):
    x=x0.copy(); r=b-apply(d,wx,wy,x); bb=(b*b).sum(); it=0
    p=np.zeros_like(b); s=np.zeros_like(b); gam_old=0; al_old=0
    while it<maxit:
        u=prec(r); w=apply(d,wx,wy,u); gam=(r*u).sum(); delta=(w*u).sum(); rho=(r*r).sum()
        if rho<=rtol**2*bb: break
        if it==0: be=0.0; al=gam/delta
        else: be=gam/gam_old; al=gam/(delta-be*gam/al_old)
        p=u+be*p; s=w+be*s; x+=al*p; r-=al*s; gam_old=gam; al_old=al; it+=1
    return x,it
for (H,W,lam) in [(256,256,0.024*256),(256,256,0.024*4),(350,350,0.024*64)]:
    r,wx,wy=system(H,W,lam); wx[:,-1]=0; wy[-1,:]=0; d=diag_of(r,wx,wy)
    rng=np.random.default_rng(5); x0=rng.random((H,W)); b=r*x0
    mg=MG(r,wx,wy,2,0.8,1.0)
    x1,it1=pcg(d,wx,wy,b,x0,lambda v:mg.vcycle(0,v),rtol=1e-6)
    x2,it2=pcg_cg(d,wx,wy,b,x0,lambda v:mg.vcycle(0,v),rtol=1e-6)
    res2=np.sqrt(((b-apply(d,wx,wy,x2))**2).sum()/(b*b).sum())
    print(H,W,"lam %.3f"%lam,"PCG it",it1,"CG-CG it",it2,"maxdiff %.2e"%np.abs(x1-x2).max(),"true rel res %.2e"%res2)
