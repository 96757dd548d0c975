import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["RTMC_DEBUG"] = "1"
import helpers as H
from raytracer_amd import RayTrace, configs
def sc(job):
    rt = RayTrace.RayTrace(demo_scene=False); rt.LoadJob(job); return rt.Scene()
for name in ["C2","C3","C4","C5","C1","C2r","C4r","S3","S14","S14r","stack8","tilted"]:
    print("==", name, file=sys.stderr, flush=True)
    s = H.raytrace_for(name).Scene(); s.run(100000, seed=1); s.close()
for nm, job in (("S30",configs.strip_job(10,strips=30)),("stack16",configs.stacked_job(10,n_slabs=16)),("twin",configs.twin_job())):
    print("==", nm, file=sys.stderr, flush=True)
    s = sc(job); s.run(100000, seed=1); s.close()
