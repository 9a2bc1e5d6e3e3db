#!/usr/bin/env python3
"""tools/cert_study.py -- CPU study of the matrix-pipe certificate (numpy restatement, tests/test_certificate_math.py) against the oracle's exact level set on small tanks at cube
sizes 0.5 / 1.0 / 1.5 / 2.0 r: how many of the sub-blocks that lie inside the surface the f16 tiles certify, with the records relative to the BLOCK's centre (what the kernels do)
and to the SUB-BLOCK's centre, against the same bound in exact arithmetic and the polynomial bound of rounds 3-5; and how long the near lists get (DESIGN_HISTORY.md, round 6)."""
import sys, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle import oracle as O
O.build()
import test_certificate_math as T
from splashsurf_amd import workloads as W
f32=np.float32
def study(scale, radius, cube, nblocks=60, eps_scale=1.0, centre_mode="block"):
    h=f32(4*radius); cs=f32(cube*radius)
    pts=W.tank_particles(scale, particle_radius=radius).astype(f32)
    par=O.make_params_relative(radius,2.0,cube,subdomain_num_cubes_per_dim=64)
    orc=O.reconstruct_surface(pts,par)
    mass=f32(1000.0)*(f32(2*radius))**3
    V=(mass/orc.particle_densities).astype(f32)
    gmin=orc.grid["aabb_min"].astype(f32)
    ns=[int(v) for v in orc.subdomain_grid["n_cells"]]
    best,bc=None,-1
    for flat in range(ns[0]*ns[1]*ns[2]):
        c,_=O.levelset_subdomain(pts,par,flat)
        if c>bc: best,bc=flat,c
    _,G=O.levelset_subdomain(pts,par,best)
    s3=(best//(ns[1]*ns[2]),(best//ns[2])%ns[1],best%ns[2])
    rng=np.random.default_rng(3)
    blocks=[(a,b,c) for a in range(8) for b in range(8) for c in range(8)]; rng.shuffle(blocks)
    inv_h=f32(1.0)/h
    tot=inside=cert=cert0=certp=0; lens=[]
    for (bx,by,bz) in blocks[:nblocks]:
        g0=np.array([s3[0]*64+8*bx,s3[1]*64+8*by,s3[2]*64+8*bz])
        lo=(gmin+g0.astype(f32)*cs).astype(f32); hi=(gmin+(g0+7).astype(f32)*cs).astype(f32)
        centre=(lo+f32(3.5)*cs).astype(f32)
        e=np.maximum(np.maximum(lo-pts,pts-hi),0.0)
        sel=np.nonzero((e*e).sum(1)<=(T.RNEAR*float(h))**2)[0]
        if sel.size==0: continue
        for sb in range(8):
            sx,sy,sz=(sb>>2)&1,(sb>>1)&1,sb&1
            o=g0+4*np.array([sx,sy,sz])
            ii,jj,kk=np.meshgrid(np.arange(4),np.arange(4),np.arange(4),indexing="ij")
            gp=o[None,:]+np.stack([ii.ravel(),jj.ravel(),kk.ravel()],1)
            if (gp-np.array(s3)*64).max()>64: continue
            X=(gmin[None,:]+gp.astype(f32)*cs).astype(f32)
            slo,shi=X.min(0),X.max(0)
            es=np.maximum(np.maximum(slo-pts[sel],pts[sel]-shi),0.0)
            near=sel[np.nonzero((es*es).sum(1)<=(T.RNEAR*float(h))**2)[0]]
            loc=gp-np.array(s3)*64
            exact=G[loc[:,0],loc[:,1],loc[:,2]].astype(np.float64)
            tot+=1
            if np.all(exact>0.6): inside+=1
            else: continue
            lens.append(near.size)
            c0=centre if centre_mode=="block" else (0.5*(slo+shi)).astype(f32)
            p_rel=((pts[near]-c0)*inv_h).astype(f32); x_rel=((X-c0)*inv_h).astype(f32)
            rec=T._records(p_rel,V[near],float(h),float(cs) if centre_mode=="block" else float(cs)*1.5/3.5)
            D=T._tile_values(rec,x_rel)+1e-5
            b=(np.maximum(D,0)**4).sum(0)
            if np.all(b>0.6*1.0001): cert+=1
            # precision-free quartic
            d2=((x_rel.astype(np.float64)[None,:,:]-p_rel.astype(np.float64)[:,None,:])**2).sum(2)
            sig=8.0/(np.pi*float(h)**3)
            u=np.maximum(1-d2,0)
            b0=(T.C4*sig*V[near].astype(np.float64)[:,None]*u**4).sum(0)
            if np.all(b0>0.6*1.0001): cert0+=1
            bp=(sig*V[near].astype(np.float64)[:,None]*u**3*(0.150818+0.785260*u*u)).sum(0)
            if np.all(bp>0.6*1.0001): certp+=1
    print("cube %.2f centre=%s: sub-blocks %d, inside %d; certified: f16 tiles %d, exact quartic %d, exact polynomial %d; near list len mean %.1f max %d"%(cube,centre_mode,tot,inside,cert,cert0,certp,np.mean(lens) if lens else 0,max(lens) if lens else 0))
for cube,scale in ((0.5,0.085),(1.0,0.12),(1.5,0.12),(2.0,0.16)):
    study(scale,0.005,cube)
    study(scale,0.005,cube,centre_mode="sub")
