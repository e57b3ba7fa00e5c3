"""
Celestial WCS of a FITS header, the part `Observation.disc_from_wcs` needs
(observation.py:427-500): pixel <-> RA/Dec for the gnomonic (`RA---TAN` / `DEC--TAN`)
projection with a `CDi_j` matrix, `PCi_j` + `CDELTi`, or `CDELTi` + `CROTA2`
(Greisen & Calabretta 2002, Calabretta & Greisen 2002). The reference uses astropy.wcs here;
this is a numpy restatement of the one projection planetary imagers write - distortion
terms (SIP, lookup tables) and other projections raise `ValueError`, like a header without
WCS does, so `reset_disc_params` falls through to its next method.
Pixel coordinates are 0-based like astropy's `*_values` methods (FITS CRPIX is 1-based).
"""

from __future__ import annotations

import numpy as np


class TanWCS:
    def __init__(self, header) -> None:
        get = header.get
        ctype1, ctype2 = str(get('CTYPE1', '')).strip().upper(), str(get('CTYPE2', '')).strip().upper()
        if not ctype1 and not ctype2 or get('CRVAL1') is None or get('CRVAL2') is None:
            raise ValueError('No WCS information found in FITS header')
        if not (ctype1.startswith('RA--') and ctype2.startswith('DEC-')):
            raise ValueError('WCS axes are not RA/Dec coordinates')
        if ctype1[5:] != 'TAN' or ctype2[5:] != 'TAN':
            raise ValueError(f'WCS projection {ctype1!r} / {ctype2!r} is not supported (TAN only)')
        for i in (1, 2):
            unit = str(get(f'CUNIT{i}', 'deg')).strip().lower()
            if unit not in ('deg', 'degree', 'degrees', ''):
                raise ValueError('WCS coordinates are not in degrees')
        self.crpix = np.array([float(get('CRPIX1', 0.0)), float(get('CRPIX2', 0.0))])
        self.crval = np.array([float(get('CRVAL1')), float(get('CRVAL2'))])
        if any(get(k) is not None for k in ('CD1_1', 'CD1_2', 'CD2_1', 'CD2_2')):
            cd = np.array([[get('CD1_1', 0.0), get('CD1_2', 0.0)], [get('CD2_1', 0.0), get('CD2_2', 0.0)]], dtype=float)
        else:
            cdelt = np.array([float(get('CDELT1', 1.0)), float(get('CDELT2', 1.0))])
            if any(get(k) is not None for k in ('PC1_1', 'PC1_2', 'PC2_1', 'PC2_2')):
                pc = np.array(
                    [[get('PC1_1', 1.0), get('PC1_2', 0.0)], [get('PC2_1', 0.0), get('PC2_2', 1.0)]], dtype=float
                )
            elif get('CROTA2') is not None:
                r = np.deg2rad(float(get('CROTA2')))
                pc = np.array(
                    [[np.cos(r), -np.sin(r) * cdelt[1] / cdelt[0]], [np.sin(r) * cdelt[0] / cdelt[1], np.cos(r)]]
                )
            else:
                pc = np.eye(2)
            cd = cdelt[:, None] * pc
        if not np.all(np.isfinite(cd)) or abs(np.linalg.det(cd)) == 0.0:
            raise ValueError('WCS pixel -> world matrix is singular')
        self.cd = cd
        self.cd_inv = np.linalg.inv(cd)

    def pixel_to_world_values(self, x, y):
        """(x, y) 0-based pixels -> (ra, dec) degrees"""
        x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
        dx, dy = x + 1.0 - self.crpix[0], y + 1.0 - self.crpix[1]
        xi = np.deg2rad(self.cd[0, 0] * dx + self.cd[0, 1] * dy)
        eta = np.deg2rad(self.cd[1, 0] * dx + self.cd[1, 1] * dy)
        a0, d0 = np.deg2rad(self.crval)
        den = np.cos(d0) - eta * np.sin(d0)
        ra = a0 + np.arctan2(xi, den)
        dec = np.arctan2(np.sin(d0) + eta * np.cos(d0), np.hypot(xi, den))
        return np.rad2deg(ra) % 360.0, np.rad2deg(dec)

    def world_to_pixel_values(self, ra, dec):
        """(ra, dec) degrees -> (x, y) 0-based pixels; NaN on the far hemisphere of the tangent point"""
        a, d = np.deg2rad(np.asarray(ra, dtype=float)), np.deg2rad(np.asarray(dec, dtype=float))
        a0, d0 = np.deg2rad(self.crval)
        cosc = np.sin(d0) * np.sin(d) + np.cos(d0) * np.cos(d) * np.cos(a - a0)
        with np.errstate(divide='ignore', invalid='ignore'):
            xi = np.where(cosc > 0, np.cos(d) * np.sin(a - a0) / cosc, np.nan)
            eta = np.where(cosc > 0, (np.cos(d0) * np.sin(d) - np.sin(d0) * np.cos(d) * np.cos(a - a0)) / cosc, np.nan)
        xi, eta = np.rad2deg(xi), np.rad2deg(eta)
        dx = self.cd_inv[0, 0] * xi + self.cd_inv[0, 1] * eta
        dy = self.cd_inv[1, 0] * xi + self.cd_inv[1, 1] * eta
        return dx + self.crpix[0] - 1.0, dy + self.crpix[1] - 1.0
