"""How much of the per-visit depth-hint traffic could a coarser filter remove? Simulated on the CPU oracle's depth buffer
(statistics only; not a test): per-tile minimum of the best depth for several tile sizes, and per-pixel hints of fewer bits.
Result (2048^2, poisson-saturne, after 2.5e8 iterations): the attractor is sheet-like — at most pixels nearly every visit
lies within 1e-3 of the front surface — so even 16x16 tiles let 65-70 % of the visits through and an 8-bit per-pixel hint
57 %; only a full-precision per-pixel hint (0.3 %) filters. See DESIGN.md section 3.5."""
import sys, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, oracle_lib as O
W=H=2048
cfg=O.poisson_saturne(); cfg.width=W; cfg.height=H
jobs=16384; n=1_000_000_000//jobs//4   # 2.5e8 iterations: "late" state of the depth buffer
st=O.start_points(1,0,jobs)
rt=O.Runtime(W,H)
t=time.time(); O.render_jobs_mt(cfg,rt,st,n); print("oracle",time.time()-t)
z=rt.zbuf.copy(); cnt=rt.count.copy()
touched=cnt>0
print("touched",touched.mean())
# simulate visits with numpy (not bit-exact; statistics only)
cx=np.array([cfg.coeff_x[k] for k in range(10)]); cy=np.array([cfg.coeff_y[k] for k in range(10)]); cz=np.array([cfg.coeff_z[k] for k in range(10)])
m=O.rotation_matrix(cfg)
cc=[cfg.center_camera[k] for k in range(3)]
T=20000
p=O.start_points(99,0,T).T.copy()
def step(p):
    x,y,zz=p
    mon=np.stack([np.ones_like(x),x,x*x,x*y,x*zz,y,y*y,y*zz,zz,zz*zz])
    return np.stack([(mon*cx[:,None]).sum(0),(mon*cy[:,None]).sum(0),(mon*cz[:,None]).sum(0)])
for _ in range(1000): p=step(p)
pix=[];zs=[]
for it in range(600):
    p=step(p)
    ss=m@p
    x2=(ss[0]+cc[0])*1.0+(ss[2]+cc[1])*0.0
    z2=(ss[0]+cc[0])*0.0-(ss[2]+cc[1])*1.0
    i=(0.5-x2)*W; j=H/2-(ss[1]+cc[2])*W
    ok=(i>=0)&(i<W)&(j>=0)&(j<H)
    pix.append((j[ok].astype(np.int64))*W+i[ok].astype(np.int64)); zs.append(z2[ok].astype(np.float32))
pix=np.concatenate(pix); zs=np.concatenate(zs)
print("visits",len(pix))
zf=z.ravel()
print("exact pass (z>=best):", (zs>=zf[pix]).mean(), " strict win:", (zs>zf[pix]).mean())
for ts in (2,4,8,16,32,64):
    zt=np.where(touched, z, np.float32(3e38)).reshape(H//ts,ts,W//ts,ts).min(axis=(1,3))   # min over TOUCHED pixels only (untouched handled separately)
    anyun=(~touched).reshape(H//ts,ts,W//ts,ts).any(axis=(1,3))
    tmin_all=np.where(anyun, np.float32(-1), zt)
    tile=(pix//W//ts)*(W//ts)+(pix%W)//ts
    p_all=(zs>=tmin_all.ravel()[tile]).mean()
    p_touched=(zs>=zt.ravel()[tile]).mean()
    # 16-bit quantised, conservative (floor)
    print("tile %2dx%-2d entries %7d  pass(min over all px) %.3f  pass(min over touched px only) %.3f"%(ts,ts,(H//ts)*(W//ts),p_all,p_touched))
# per-pixel lower-precision hints: how many bits are needed? pass rate if hint quantised to b bits (floor)
for bits in (4,6,8,10,12,16):
    q=np.floor((zf+1.0)*(2**bits)/2.0)  # z in [-1,1)
    qv=np.floor((zs+1.0)*(2**bits)/2.0)
    print("per-pixel %2d-bit hint: pass %.4f"%(bits,(qv>=q[pix]).mean()))
