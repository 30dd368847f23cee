import numpy as np
f32=np.float32
rng=np.random.default_rng(1)
def mt(o,d,p0,p1,p2):
    # float32, reference order (tri_intersect in rt_kernels.hip)
    e1=(p1-p0).astype(f32); e2=(p2-p0).astype(f32)
    def cross(a,b): return np.stack([a[:,1]*b[:,2]-a[:,2]*b[:,1], a[:,2]*b[:,0]-a[:,0]*b[:,2], a[:,0]*b[:,1]-a[:,1]*b[:,0]],1).astype(f32)
    def dot(a,b): return ((a[:,0]*b[:,0]+a[:,1]*b[:,1]).astype(f32)+a[:,2]*b[:,2]).astype(f32)
    h=cross(d,e2); a=dot(e1,h)
    ok=~((a>f32(-1e-7))&(a<f32(1e-7)))
    with np.errstate(all='ignore'):
        f=(f32(1)/a).astype(f32)
        s=(o-p0).astype(f32)
        u=(f*dot(s,h)).astype(f32)
        ok&=~((u<0)|(u>1))
        q=cross(s,e1)
        v=(f*dot(d,q)).astype(f32)
        ok&=~((v<0)|((u+v).astype(f32)>1))
        t=(f*dot(e2,q)).astype(f32)
        ok&=(t>=f32(1e-7))
    return ok
N=2_000_000
tot=0; acc_by={}
for logc in (-2,-3,-4,-5,-6):
  for scale in (0.13, 1.0):
    # triangle in a random plane at distance ~7 from origin
    cen=rng.normal(size=(N,3)); cen/=np.linalg.norm(cen,axis=1,keepdims=True); cen*=rng.uniform(3,12,(N,1))
    # in-plane basis
    n=rng.normal(size=(N,3)); n/=np.linalg.norm(n,axis=1,keepdims=True)
    a1=np.cross(n,rng.normal(size=(N,3))); a1/=np.linalg.norm(a1,axis=1,keepdims=True); a2=np.cross(n,a1)
    ang=rng.uniform(0,2*np.pi,(N,3)); rad=rng.uniform(0.3,1.0,(N,3))*scale
    P=[cen+a1*(rad[:,[k]]*np.cos(ang[:,[k]]))+a2*(rad[:,[k]]*np.sin(ang[:,[k]])) for k in range(3)]
    P=[p.astype(f32) for p in P]
    p0,p1,p2=[p.astype(np.float64) for p in P]
    # min corner sine >= 0.1 filter
    def sinang(a,b,c):
        u=b-a; v=c-a; return np.linalg.norm(np.cross(u,v),axis=1)/(np.linalg.norm(u,axis=1)*np.linalg.norm(v,axis=1))
    good=(np.minimum(np.minimum(sinang(p0,p1,p2),sinang(p1,p2,p0)),sinang(p2,p0,p1))>=0.1)
    nn=np.cross(p1-p0,p2-p0); nn/=np.linalg.norm(nn,axis=1,keepdims=True)
    cc=(p0+p1+p2)/3; r=np.max([np.linalg.norm(p-cc,axis=1) for p in (p0,p1,p2)],0)
    # ray origin O: at distance L from cc; direction: in-plane dir w rotated out of plane by angle c
    c=10.0**logc*rng.uniform(0.3,3,(N,1))
    th=rng.uniform(0,2*np.pi,(N,1)); b1=np.cross(nn,rng.normal(size=(N,3))); b1/=np.linalg.norm(b1,axis=1,keepdims=True); b2=np.cross(nn,b1)
    w=b1*np.cos(th)+b2*np.sin(th)              # in-plane unit
    lat=np.cross(nn,w)                          # in-plane, perpendicular to w
    d=w*np.sqrt(1-c*c)+nn*c*np.sign(rng.normal(size=(N,1)))
    L=rng.uniform(3,12,(N,1))
    pad=2e-3+1e-3*(L[:,0]+r)
    miss=(r+pad)*rng.uniform(1.0,3.0,N)        # lateral miss of the bounding sphere, beyond the padding
    Q=cc+lat*miss[:,None]*np.sign(rng.normal(size=(N,1)))
    O=Q-d*L
    ok=mt(O.astype(f32),d.astype(f32),*P)&good
    # true distance of the float ray's line from the centre, minus r (float-rounded inputs)
    Of=O.astype(f32).astype(np.float64); df=d.astype(f32).astype(np.float64)
    v=cc-Of; dist=np.linalg.norm(np.cross(v,df),axis=1)/np.linalg.norm(df,axis=1)
    far=dist>(r+pad)
    print("log10 c",logc,"scale",scale,"good",good.sum(),"accepted although the line misses the padded sphere:",(ok&far).sum(), "accepted total", ok.sum())
