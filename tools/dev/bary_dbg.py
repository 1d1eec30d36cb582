import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from __graft_entry__ import load_package, load_oracle
pkg = load_package(); orc = load_oracle()
rng = np.random.default_rng(11)
r0 = np.concatenate([rng.random(100000, dtype=np.float32), np.float32([0.0, 1.0, 0.5, 0.99999994, 2.3283064e-10])])
rec = np.zeros((len(r0), 24), np.float32); rec[:, 20] = r0
a = pkg.RenderContext(0).kat("random_barycentrics", rec)[:, :3]
b = orc.OracleContext(pkg).kat("random_barycentrics", rec)[:, :3]
bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(1))[0]
print(len(bad), "of", len(r0))
for i in bad[:8]:
    print(repr(r0[i]), a[i], b[i], a[i].view(np.uint32) - b[i].view(np.uint32))
