exec(open(__file__.replace('mg_convergence_experiments','mg_proto')).read().split("for (H,W,lam) in")[0])
import scipy.sparse as sp, scipy.sparse.linalg as spl
def mat(d,wx,wy):
    H,W=d.shape; n=H*W; idx=np.arange(n).reshape(H,W)
    A=sp.diags(d.reshape(-1)).tolil()
    rows=[];cols=[];vals=[]
    m=wx[:,:-1]; rows+=list(idx[:,:-1].reshape(-1)); cols+=list(idx[:,1:].reshape(-1)); vals+=list(-m.reshape(-1))
    m=wy[:-1,:]; rows+=list(idx[:-1,:].reshape(-1)); cols+=list(idx[1:,:].reshape(-1)); vals+=list(-m.reshape(-1))
    B=sp.coo_matrix((vals,(rows,cols)),shape=(n,n)); return (sp.diags(d.reshape(-1))+B+B.T).tocsc()
class MG2(MG):
    """exact coarse solve at level `exact_at`; gamma = cycle index"""
    def __init__(s,r,wx,wy,nu=2,omega=0.8,kappa=1.0,coarse=8,exact_at=None,gamma=1):
        MG.__init__(s,r,wx,wy,nu,omega,kappa,coarse); s.exact_at=exact_at; s.gamma=gamma
        if exact_at is not None:
            d,wx_,wy_=s.lv[exact_at]; s.lu=spl.splu(mat(d,wx_,wy_))
    def vcycle(s,l,b):
        d,wx,wy=s.lv[l]
        if s.exact_at is not None and l==s.exact_at: return s.lu.solve(b.reshape(-1)).reshape(b.shape)
        if l==len(s.lv)-1:
            x=np.zeros_like(b)
            for _ in range(50): x=x+s.om*(b-apply(d,wx,wy,x))/d
            return x
        x=s.om*b/d
        for _ in range(s.nu-1): x=x+s.om*(b-apply(d,wx,wy,x))/d
        for g in range(s.gamma if l>=1 else 1):
            res=b-apply(d,wx,wy,x)
            ec=s.vcycle(l+1,restrict(res))
            x=x+s.kap*prolong(ec,*b.shape)
        for _ in range(s.nu): x=x+s.om*(b-apply(d,wx,wy,x))/d
        return x
for (H,W,lam) in [(256,256,0.024*256),(256,256,0.024*4)]:
    r,wx,wy=system(H,W,lam); wx[:,-1]=0; wy[-1,:]=0; d=diag_of(r,wx,wy)
    rng=np.random.default_rng(5); x0=rng.random((H,W)); b=r*x0
    for name,kw in [("V22",dict()),("twogrid-exact",dict(exact_at=1)),("threegrid-exact",dict(exact_at=2)),("W22",dict(gamma=2)),("V44",dict(nu=4)),("V22 kap1.5",dict(kappa=1.5)),("twogrid-exact kap 1.5",dict(exact_at=1,kappa=1.5)),("twogrid-exact kap 2",dict(exact_at=1,kappa=2.0))]:
        mg=MG2(r,wx,wy,**kw); xm,itm=pcg(d,wx,wy,b,x0,lambda v:mg.vcycle(0,v),rtol=1e-6)
        print(H,W,"lam %.3f"%lam,name,"it",itm)
