"""
Map-grid generators for the non-rectangular projections of
`BodyXY.generate_map_coordinates` (`planetmapper/body_xy.py:2930-2969`) without pyproj.

The reference builds these grids by running PROJ's INVERSE transform on a square grid of
projected coordinates. The three built-in projections have closed forms:

* ``'orthographic'``  -> ``+proj=ortho +a=r_eq +b=r_polar +to_meter=r_eq +y_0=...``: the
  ellipsoidal orthographic projection (EPSG 9840): topocentric east / north coordinates
  of the surface point seen from infinity along the geodetic normal at (lon_0, lat_0). Its
  inverse is a ray / ellipsoid intersection.
* ``'azimuthal'``      -> ``+proj=aeqd +a=r_eq`` (sphere), unit = pi r_eq.
* ``'azimuthal equal area'`` -> ``+proj=laea +a=r_eq`` (sphere), unit = 2 r_eq.

`+axis=wnu` for bodies with west-positive longitudes: the first projected axis points west,
and the returned longitudes are the planetographic ones. Pinned by the reference's golden
maps `map_orthographic-{1,2,3}.fits`, `map_azimuthal-{1,2,3}.fits` (LON/LAT-GRAPHIC planes).
"""

from __future__ import annotations

import numpy as np


def _grid(lim: float, size: int):
    c = np.linspace(-lim, lim, size)
    return np.meshgrid(c, c)


def orthographic(r_eq: float, r_polar: float, lon0: float, lat0: float, size: int, west_positive: bool):
    """body_xy.py:2930-2942"""
    b = r_polar / r_eq
    lim = max(1, b) * 1.01
    xx, yy = _grid(lim, size)
    y0 = r_eq * (b - 1) * np.sin(np.radians(lat0 * 2))  # the false northing PM passes to PROJ
    east = (-xx if west_positive else xx) * r_eq
    north = yy * r_eq - y0
    e2 = 1.0 - (r_polar / r_eq) ** 2
    lam0, phi0 = np.radians(lon0), np.radians(lat0)
    # geodetic frame at the projection origin (x towards lon 0, z the spin axis; east longitudes)
    sl, cl, sp, cp = np.sin(lam0), np.cos(lam0), np.sin(phi0), np.cos(phi0)
    if abs(cp) < 1e-12:  # polar aspect: exact, so the pole itself gets PROJ's atan2(0, -0) = 180 deg
        cp, sp = 0.0, float(np.sign(sp))
    nu0 = r_eq / np.sqrt(1.0 - e2 * sp * sp)
    p0 = np.array([nu0 * cp * cl, nu0 * cp * sl, nu0 * (1.0 - e2) * sp])
    e_hat = np.array([-sl, cl, 0.0])
    n_hat = np.array([-sp * cl, -sp * sl, cp])
    u_hat = np.array([cp * cl, cp * sl, sp])  # outward geodetic normal = viewing direction
    # P = p0 + E e + N n + t u on the ellipsoid (x^2 + y^2)/a^2 + z^2/c^2 = 1, larger root
    q = p0[:, None, None] + east[None] * e_hat[:, None, None] + north[None] * n_hat[:, None, None]
    w = np.array([1.0 / r_eq**2, 1.0 / r_eq**2, 1.0 / r_polar**2])[:, None, None]
    A = float(np.sum(w[:, 0, 0] * u_hat * u_hat))
    B = 2.0 * np.sum(w * q * u_hat[:, None, None], axis=0)
    C = np.sum(w * q * q, axis=0) - 1.0
    disc = B * B - 4.0 * A * C
    with np.errstate(invalid='ignore'):
        t = (-B + np.sqrt(disc)) / (2.0 * A)
    P = q + t[None] * u_hat[:, None, None]
    lon = np.degrees(np.arctan2(P[1], P[0]))
    lat = np.degrees(np.arctan2(P[2] / (1.0 - e2), np.hypot(P[0], P[1])))
    bad = ~(disc >= 0)
    lon[bad] = np.nan
    lat[bad] = np.nan
    return lon, lat, xx, yy


def _azimuthal(lon0: float, lat0: float, size: int, west_positive: bool, equal_area: bool):
    lim = 1.01
    xx, yy = _grid(lim, size)
    east = -xx if west_positive else xx
    north = yy
    rho = np.hypot(east, north)
    with np.errstate(invalid='ignore'):
        if equal_area:
            c = 2.0 * np.arcsin(rho)  # unit = 2 R: rho_R / 2 = rho
            ok = rho <= 1.0
        else:
            c = np.pi * rho  # unit = pi R
            ok = rho <= 1.0
    lam0, phi0 = np.radians(lon0), np.radians(lat0)
    sc, cc = np.sin(c), np.cos(c)
    with np.errstate(invalid='ignore', divide='ignore'):
        lat = np.arcsin(np.clip(cc * np.sin(phi0) + np.where(rho > 0, north * sc * np.cos(phi0) / rho, 0.0), -1, 1))
        lon = lam0 + np.arctan2(east * sc, rho * np.cos(phi0) * cc - north * np.sin(phi0) * sc)
    lon = np.degrees(lon)
    lat = np.degrees(lat)
    lon[~ok] = np.nan
    lat[~ok] = np.nan
    return lon, lat, xx, yy


def azimuthal(lon0, lat0, size, west_positive):
    """body_xy.py:2943-2955 (`aeqd`, sphere)"""
    return _azimuthal(lon0, lat0, size, west_positive, equal_area=False)


def azimuthal_equal_area(lon0, lat0, size, west_positive):
    """body_xy.py:2956-2968 (`laea`, sphere)"""
    return _azimuthal(lon0, lat0, size, west_positive, equal_area=True)
