import numpy as np, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import splashsurf_amd as S
from splashsurf_amd.api import Context
from conftest import load_golden, golden_input
ctx=Context(0)
for name,r,l,c in [("global_cube_2366",0.025,2.0,0.75),("global_config1",0.025,2.0,1.1),("bunny_7705",0.025,2.0,0.5),("global_free_particles_125",0.025,2.0,1.0)]:
    pts=golden_input(load_golden(name))
    for rep in range(2):
        t=time.perf_counter(); res=S.reconstruct_surface(pts,particle_radius=r,smoothing_length=l,cube_size=c,subdomain_grid=False,context=ctx); dt=time.perf_counter()-t
    st=res.stats
    print(name,len(pts),list(res.grid.ncells_per_dim),'wall ms %.2f'%(dt*1e3),{k:round(v,3) for k,v in st.items() if k.startswith('ms_')})
    t=time.perf_counter(); res2=S.reconstruct_surface(pts,particle_radius=r,smoothing_length=l,cube_size=c,subdomain_grid=True,subdomain_grid_auto_disable=False,context=ctx); dt=time.perf_counter()-t
    print('   subdomain path wall ms %.2f total %.3f'%(dt*1e3,res2.stats['ms_total']))
