"""rfw::Camera (RFW/system/context/rfw/context/camera.h:17-60, Camera.cpp) — the data members and the helpers the
applications call before handing the camera to render_frame.  get_view itself is computed by the core
(rfwhip_camera_get_view) so that host and oracle share no arithmetic."""
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi


@dataclass
class Camera:
    position: tuple = (0.0, 0.0, 0.0)
    direction: tuple = (0.0, 0.0, 1.0)
    focalDistance: float = 5.0
    aperture: float = 0.0001  # camera.h:31 (the parity configs set 0)
    brightness: float = 0.0
    contrast: float = 0.0
    FOV: float = 40.0
    aspectRatio: float = 1.0
    clampValue: float = 10.0
    pixelCount: tuple = field(default_factory=lambda: (1, 1))

    def resize(self, w, h):  # Camera.cpp:102-106
        self.aspectRatio = float(w) / float(h)
        self.pixelCount = (int(w), int(h))

    def look_at(self, origin, target):
        o, t = np.asarray(origin, np.float64), np.asarray(target, np.float64)
        d = t - o
        self.position = tuple(float(x) for x in o)
        self.direction = tuple(float(x) for x in d / np.linalg.norm(d))

    def pod(self):
        p = abi.CameraPOD()
        p.position[:] = [np.float32(x) for x in self.position]
        p.direction[:] = [np.float32(x) for x in self.direction]
        p.focalDistance, p.aperture = self.focalDistance, self.aperture
        p.brightness, p.contrast, p.FOV = self.brightness, self.contrast, self.FOV
        p.aspectRatio, p.clampValue = self.aspectRatio, self.clampValue
        p.pixelCount[:] = self.pixelCount
        return p

    def spread_angle(self):
        return (self.FOV * math.pi / 180.0) / float(self.pixelCount[1])
