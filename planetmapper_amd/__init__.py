"""
planetmapper_amd: MI355X-native engine for PlanetMapper's per-pixel hot path
(backplane images + map reprojection) behind the reference's BodyXY / Observation API.

The compute lives in libplanetmapper_hip.so (hand-written HIP for gfx950, C ABI in
include/planetmapper_hip.h); this package is the thin ctypes layer on top. There is no
CPU fallback: without the built library or without a GPU the engine raises.
"""

from .body_xy import Backplane, BackplaneNotFoundError, BodyXY, NotFoundError  # noqa: F401
from .engine import Engine, device_count  # noqa: F401
from .geometry import GeometryBuilder, PMDisc, PMGeometry  # noqa: F401
from .observation import Observation  # noqa: F401
from .scenarios import load_scenario  # noqa: F401

__version__ = '0.1.0'
