import sys, ctypes as C, json, numpy as np, torch
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
T=g.load_package(); import bench
ctx=T.Context(0)
scene,cam,desc=bench.build_workload(T,'mesh_1m',1024)
flat=scene.flatten(ctx); L=T.lib()
bnd=flat.bvh()[0][0]
n=1<<25
rays=T.scenes.incoherent_rays(n,bnd[:3],bnd[3:])
d_rays=torch.from_numpy(rays).cuda(); d_hits=torch.empty((n,4),dtype=torch.float32,device='cuda')
for budget in (0, 20000, 5000, 2000, 1000, 500):
    ctx.set_option('debug_trace_budget', budget)
    ms=C.c_double()
    ctx.check(L.trhip_trace_closest_device(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_hits.data_ptr()), 1, C.byref(ms)))
    ctx.check(L.trhip_trace_closest_device(ctx._h, flat._h, C.c_void_p(d_rays.data_ptr()), n, C.c_void_p(d_hits.data_ptr()), 2, C.byref(ms)))
    print(json.dumps({'budget':budget,'ms':round(ms.value,2),'Mray_s':round(n/ms.value/1e3,1)}),flush=True)
