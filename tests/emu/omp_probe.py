import sys, time, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package, load_oracle
pkg = load_package(); orc = load_oracle()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup", e)
sc = pkg.scenes.terrain(n=300, width=960, height_px=540)
o = orc.OracleContext(pkg); o.init(960,540); sc.upload(o); o.set_setting("integrator","pt")
for th in (1, 4, 16, 64, 128, 256):
    o.set_setting("threads", th)
    o.render_frame(sc.camera, pkg.RESET)
    t=time.perf_counter(); o.render_frame(sc.camera, pkg.RESET); dt=time.perf_counter()-t
    print("threads", th, "Msamples/s", 960*540/dt/1e6, flush=True)
