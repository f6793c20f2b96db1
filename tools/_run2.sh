#!/bin/bash
timeout 300 python - <<'PY' 2>/dev/null
import os, __graft_entry__ as g, bench
T=g.load_package(); ctx=T.default_context()
scene,cam,desc=bench.build_workload(T,"caustic_sppm",1024) if "caustic_sppm" in getattr(bench,"WORKLOADS",{"caustic_sppm":1}) else (None,None,None)
PY
timeout 300 python - <<'PY' 2>/dev/null
import os, __graft_entry__ as g
T=g.load_package(); ctx=T.default_context()
ply=os.path.join("tests","golden","caustic-glass.ply")
scene=T.scenes.caustic_scene(ply); cam=T.scenes.caustic_camera(1024)
for hyb,b in ((1,-1),(0,-1),(1,0)):
    ctx.set_option("bvh_builder",b); ctx.set_option("hybrid",hyb)
    scene._flat=None
    it=T.SPPMIntegrator(cam,0.075,8,100,-1,seed=0x5EED0004)
    it.render(scene,ctx); it.render(scene,ctx)
    s=it.stats
    print("builder",b,"hybrid",hyb,"total",round(s.ms_total,1),"closest",round(s.ms_trace_closest,1),"launches",s.launches_trace_closest,"fallback ms",round(s.ms_fallback,2),"launches_fb",s.launches_fallback,"any",round(s.ms_trace_any,1),"fallback rays",s.fallback_rays,"of",s.closest_rays)
PY
