"""Development probe: light bins per sorted run of the depth-0 connection wave (setting shadow_bins_per_run) per scene."""
import os, sys
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
W, H = 1920, 1080
for name, mk in (("terrain", lambda: pkg.scenes.terrain(n=708, width=W, height_px=H)), ("atrium", lambda: pkg.scenes.atrium(W, H)),
                 ("cornell", lambda: pkg.scenes.cornell(W, H, geometric_emitter=True)), ("cards", lambda: pkg.scenes.cards(W, H))):
    s = mk()
    c = pkg.RenderContext(0); c.init(W, H); s.upload(c)
    c.set_setting("integrator", "pt"); c.set_setting("spp", 64); c.set_setting("shadow_packets", 1)
    c.render_frame(s.camera, pkg.RESET)
    print(name, "lights", len(s.area_lights) + len(s.point_lights) + len(s.spot_lights) + len(s.directional_lights), "bins per run", c.get_setting("shadow_bins_per_run"), flush=True)
    c.destroy()
