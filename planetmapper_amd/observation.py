"""
`Observation`: a `BodyXY` plus a data cube, with the reference's
`Observation.get_mapped_data` (`planetmapper/observation.py:826-905`) running on the GPU,
optionally sharded by wavelength plane over the GPUs of a node.

The callers that want every plane - `save_observation` / `save_mapped_observation`
(`observation.py:1184-1474`) - and FITS loading (`observation.py:218-318`) are provided through
the numpy-only `fits_io` module; disc fitting, the GUI and wireframe overlays are not.
"""

from __future__ import annotations

import datetime
import os

import numpy as np

from . import fits_io
from ._lib import UnsupportedError
from .body_xy import BodyXY, _AltitudeContext, _freeze
from .engine import PLANE_NAMES
from .fits_io import Header


def _check_path(path: str) -> None:
    """utils.check_path: create the parent directory of an output file"""
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)


FITS_KEYWORD = 'PLANMAP'
FITS_FILE_EXTENSIONS = ('.fits', '.fits.gz', '.fit', '.fit.gz', '.fts', '.fts.gz')
__version__ = '1.12.5-hip'
__url__ = 'https://github.com/ortk95/planetmapper'


def _try_get_header_value(kw: dict, header: Header, key: str, candidates, value_fn=None) -> None:
    """observation.py:1615-1632: first header card among `candidates` fills kw[key] if absent"""
    if key in kw:
        return
    for name in candidates:
        if name in header:
            v = header[name]
            kw[key] = value_fn(v) if value_fn is not None else v
            return


class GetWavelengthsError(Exception):
    """utils.GetWavelengthsError"""


def _angular_dist(ra1: float, dec1: float, ra2: float, dec2: float) -> float:
    """SpiceBase.angular_dist base.py:690-716: great-circle separation in degrees"""
    d1, d2 = np.deg2rad(dec1), np.deg2rad(dec2)
    return float(
        np.rad2deg(np.arccos(np.clip(np.sin(d1) * np.sin(d2) + np.cos(d1) * np.cos(d2) * np.cos(np.deg2rad(ra1) - np.deg2rad(ra2)), -1.0, 1.0)))
    )


class Observation(BodyXY):
    """
    Args:
        path: FITS (or PNG/JPEG/...) file to load the cube from; target / observer / date are
            taken from its header when not given (observation.py:87-150, 254-318). The geometry
            itself still comes from `geometry=`, `scenario=` or `kernels=`.
        data: image cube of shape (P, ny, nx), or a single (ny, nx) image which is
            treated as a one-plane cube like the reference does for 2D FITS/PNG data
            (observation.py:228-238). Any dtype the reference accepts; kept as-is
            (float64 / float32 / int16 / int32 / uint8 / uint16 are read natively by
            the kernel, everything else is converted to float64 once).
        header: FITS header (`fits_io.Header` or dict) that goes with `data`.
        **kwargs: `BodyXY` arguments (`geometry=` or `scenario=`, `optimize_speed`, ...).
            `nx`, `ny` and `sz` are taken from the data.
    """

    FITS_KEYWORD = FITS_KEYWORD
    FITS_FILE_EXTENSIONS = FITS_FILE_EXTENSIONS

    def __init__(self, path=None, *args, data: np.ndarray | None = None, header=None, **kwargs) -> None:
        for k in ('nx', 'ny', 'sz'):
            if k in kwargs:  # observation.py:95-97
                raise TypeError(f'Cannot set {k} for Observation objects')
        self.path = None if path is None else os.fspath(path)
        self.header = Header()
        if self.path is not None:
            if data is not None:
                raise ValueError('`path` and `data` are mutually exclusive')  # observation.py:115-119
            data = self._load_data_from_path()
            if header is not None:
                raise ValueError('`header` cannot be given together with `path`')
        elif data is None:
            raise ValueError('Either `path` or `data` must be provided')
        elif header is not None:
            self.header = header.copy() if isinstance(header, Header) else Header(header)
        data = np.asarray(data)
        if data.ndim == 2:
            data = data[None]
        if data.ndim != 3:
            raise ValueError('data must be a 2D image or a 3D cube (P, ny, nx)')
        self.data = data
        if len(self.header):
            names = ('target', 'utc', 'observer')
            given = {n: v for n, v in zip(names, args)}
            given.update({n: kwargs[n] for n in names if n in kwargs})
            self._add_kw_from_header(given, self.header)
            args = ()
            for n in names:
                if n in given:
                    kwargs[n] = given[n]
        fill_in_header = not len(self.header)
        super().__init__(*args, nx=data.shape[2], ny=data.shape[1], **kwargs)
        if fill_in_header:  # observation.py:152-158
            self.header = Header({'OBJECT': self.target or '', 'DATE-OBS': self.utc or ''})

    # ------------------------------------------------------------------ loading
    def _load_data_from_path(self) -> np.ndarray:
        """observation.py:218-252"""
        assert self.path is not None
        if any(self.path.lower().endswith(ext) for ext in self.FITS_FILE_EXTENSIONS):
            hdus = fits_io.read(self.path)
            for idx, hdu in enumerate(hdus):
                if hdu.data is not None:
                    header = hdus[0].header.copy()
                    if idx:
                        header.update(hdu.header)
                    self.header = header
                    return hdu.data
            raise ValueError('No data found in provided FITS file')
        import PIL.Image  # same route as the reference for PNG / JPEG / ...

        image = np.flipud(np.array(PIL.Image.open(self.path)))
        return image if image.ndim == 2 else np.moveaxis(image, 2, 0)

    @classmethod
    def _make_fits_kw(cls, keyword: str) -> str:
        return f'HIERARCH {cls.FITS_KEYWORD} {keyword}'

    @classmethod
    def _add_kw_from_header(cls, kw: dict, header: Header) -> None:
        """observation.py:254-318: target / observer / utc from the usual header cards"""
        mk = cls._make_fits_kw
        _try_get_header_value(kw, header, 'target', [mk('TARGET'), 'OBJECT', 'TARGET', 'TARGNAME'])
        _try_get_header_value(
            kw, header, 'observer', [mk('OBSERVER'), 'TELESCOP'],
            value_fn=lambda v: 'EARTH' if str(v).startswith('ESO-') else v,
        )  # fmt: skip
        _try_get_header_value(kw, header, 'utc', [mk('UTC-OBS'), 'MJD-AVG', 'EXPMID', 'DATE-AVG'])
        if 'utc' not in kw:
            try:
                kw['utc'] = (float(header['MJD-BEG']) + float(header['MJD-END'])) / 2
            except (KeyError, TypeError, ValueError):
                pass
            if 'utc' not in kw:
                try:
                    kw['utc'] = header['DATE-OBS'] + ' ' + header['TIME-OBS']
                except (KeyError, TypeError):
                    pass
            _try_get_header_value(kw, header, 'utc', ['DATE-OBS', 'DATE-BEG', 'DATE-END', 'MJD-BEG', 'MJD-END'])

    # ------------------------------------------------------------------ automatic disc parameters
    def reset_disc_params(self) -> str:
        """
        observation.py:376-397: the disc parameters a previous run stored in the header, else those
        implied by the header's WCS, else `BodyXY.reset_disc_params` (centred disc).
        """
        try:
            self.disc_from_header()
        except ValueError:
            try:
                self.disc_from_wcs(suppress_warnings=True)
            except ValueError:
                return super().reset_disc_params()
        return self.get_disc_method()

    def _get_disc_params_from_wcs(self, suppress_warnings: bool = False, validate: bool = True,
                                  use_header_offsets: bool = True,
                                  distortion_warning_threshold: float | None = 0.25) -> tuple[float, float, float, float]:
        """
        observation.py:434-490 on the numpy gnomonic WCS of `wcs.py` (the reference uses astropy.wcs;
        headers with distortion terms or another projection raise ValueError here).
        """
        from .wcs import TanWCS

        wcs = TanWCS(self.header)
        x0, y0 = wcs.world_to_pixel_values(self.target_ra, self.target_dec)
        if not (np.isfinite(x0) and np.isfinite(y0)):
            raise ValueError('target is not on the hemisphere of the WCS tangent point')
        b1, b2 = wcs.pixel_to_world_values(x0, y0 + 1)
        c1, c2 = wcs.pixel_to_world_values(x0, y0)
        rotation = float(np.rad2deg(np.arctan2(b1 - c1, b2 - c2)))
        s = _angular_dist(b1, b2, c1, c2)  # degrees per pixel
        r0 = self.target_diameter_arcsec / (2 * s * 3600.0)
        x0, y0 = float(x0), float(y0)
        if use_header_offsets:
            dra = float(self.header.get('HIERARCH NAV RA_OFFSET', 0.0))
            ddec = float(self.header.get('HIERARCH NAV DEC_OFFSET', 0.0))
            if dra != 0 or ddec != 0:
                # a throwaway BodyXY applies the offsets exactly like add_arcsec_offset
                body = self.to_body_xy()
                body.set_disc_params(x0, y0, r0, rotation)
                body.add_arcsec_offset(dra_arcsec=dra, ddec_arcsec=ddec)
                x0, y0, r0, rotation = body.get_disc_params()
        return x0, y0, r0, rotation

    # which of (x0, y0, r0, rotation) each *_from_wcs method takes over, and the disc method it records
    _WCS_PARTS = {
        'wcs': (0, 1, 2, 3),
        'wcs_position': (0, 1),
        'wcs_rotation': (3,),
        'wcs_plate_scale': (2,),
    }

    def _adopt_wcs(self, method: str, *args, **kwargs) -> None:
        found = self._get_disc_params_from_wcs(*args, **kwargs)
        setters = (self.set_x0, self.set_y0, self.set_r0, self.set_rotation)
        for k in self._WCS_PARTS[method]:
            setters[k](found[k])
        self.set_disc_method(method)

    def disc_from_wcs(self, suppress_warnings: bool = False, validate: bool = True, use_header_offsets: bool = True,
                      distortion_warning_threshold: float | None = 0.25) -> None:
        """observation.py:502-558"""
        self._adopt_wcs('wcs', suppress_warnings, validate, use_header_offsets, distortion_warning_threshold)

    def position_from_wcs(self, *args, **kwargs) -> None:
        """observation.py:560-577"""
        self._adopt_wcs('wcs_position', *args, **kwargs)

    def rotation_from_wcs(self, *args, **kwargs) -> None:
        """observation.py:579-594"""
        self._adopt_wcs('wcs_rotation', *args, **kwargs)

    def plate_scale_from_wcs(self, *args, **kwargs) -> None:
        """observation.py:596-612"""
        self._adopt_wcs('wcs_plate_scale', *args, **kwargs)

    def get_wcs_offset(self, *args, **kwargs) -> tuple[float, float, float, float]:
        """(dx, dy, dr, drotation) of the current disc from the WCS one. observation.py:614-668"""
        dx, dy, dr, drot = (mine - theirs for mine, theirs in zip(self.get_disc_params(), self._get_disc_params_from_wcs(*args, **kwargs)))
        return dx, dy, dr, drot % 360

    def get_wcs_arcsec_offset(self, *args, check_is_position_offset_only: bool = True, **kwargs) -> tuple[float, float]:
        """(dra_arcsec, ddec_arcsec) of the current disc from the WCS one. observation.py:670-748"""
        dx, dy, dr, drotation = self.get_wcs_offset(*args, **kwargs)
        ra0, dec0 = self.xy2radec(0, 0)
        ra1, dec1 = self.xy2radec(dx, dy)
        if check_is_position_offset_only:
            if abs(dr) > 1e-3:
                raise ValueError(f'r0 is different between WCS and observation (dr={dr})')
            if abs((drotation + 180) % 360 - 180) > 1e-3:
                raise ValueError(f'rotation is different between WCS and observation (drotation={drotation})')
        return (ra1 - ra0) * 3600, (dec1 - dec0) * 3600

    def get_wavelengths_from_header(self, *, check_ctype: bool = True) -> np.ndarray:
        """
        Wavelengths of the cube's planes from NAXIS3 / CRVAL3 / CDELT3 (or CD3_3) / CRPIX3
        (observation.py:346-373, utils.generate_wavelengths_from_header).
        """
        h = self.header
        try:
            if check_ctype and str(h['CTYPE3']).strip() != 'WAVE':
                raise GetWavelengthsError('Header item CTYPE3 is not WAVE')
            naxis3 = int(h['NAXIS3'])
            wavl0 = float(h['CRVAL3'])
            dwavl = float(h['CDELT3']) if 'CDELT3' in h else float(h['CD3_3'])
            i0 = float(h.get('CRPIX3', 1))
        except KeyError as exc:
            raise GetWavelengthsError('Could not find required header keywords') from exc
        return wavl0 + (np.arange(1, naxis3 + 1) - i0) * dwavl

    def disc_from_header(self) -> None:
        """observation.py:399-425: take x0, y0, r0, rotation from the `HIERARCH PLANMAP DISC ...`
        cards written by a previous `save_observation`"""
        mk = self._make_fits_kw
        if mk('MAP PROJECTION') in self.header or mk('DEGREE-INTERVAL') in self.header:
            raise ValueError('FITS header refers to mapped data')
        try:
            self.set_disc_params(
                x0=self.header[mk('DISC X0')], y0=self.header[mk('DISC Y0')],
                r0=self.header[mk('DISC R0')], rotation=self.header[mk('DISC ROT')],
            )  # fmt: skip
            self.set_disc_method('header')
        except KeyError as exc:
            raise ValueError('No disc parameters found in FITS header') from exc

    def to_body_xy(self) -> BodyXY:
        """observation.py:183-195: a `BodyXY` with the same geometry, image size and disc parameters"""
        new = BodyXY(
            self.target, self.utc, self.observer, nx=self._nx, ny=self._ny, geometry=self._geometry,
            optimize_speed=self._optimize_speed, engine=self._engine,
        )  # fmt: skip
        new.set_disc_params(*self.get_disc_params())
        new.set_disc_method(self.get_disc_method())
        return new

    # ------------------------------------------------------------------ header metadata
    def append_to_header(
        self, keyword: str, value, comment: str | None = None, *, hierarch_keyword: bool = True,
        header: Header | None = None, truncate_strings: bool = True, remove_existing: bool = True,
    ) -> None:  # fmt: skip
        """observation.py:908-951"""
        if header is None:
            header = self.header
        if hierarch_keyword:
            keyword = self._make_fits_kw(keyword)
        if truncate_strings and isinstance(value, str):
            if len(keyword) + len(value) + 4 > 80:
                n = 80 - len(keyword) - 4 - 3
                value = value[:n] + '...'
        if remove_existing:
            header.remove(keyword, ignore_missing=True, remove_all=True)
        header.append(keyword, value, comment or '')

    def add_header_metadata(self, header: Header | None = None) -> None:
        """observation.py:956-1159: the `HIERARCH PLANMAP ...` cards, same keywords, order and
        comments as the reference writes"""
        put = lambda k, v, c: self.append_to_header(k, v, c, header=header)  # noqa: E731
        put('VERSION', __version__, 'PlanetMapper version.')
        put('URL', __url__, 'Webpage.')
        put('DATE', datetime.datetime.now().strftime('%Y-%m-%dT%H:%M:%S'), 'File generation datetime.')
        if self.path is not None:
            put('INFILE', os.path.split(self.path)[1], 'Input file name.')
        put('DISC X0', self.get_x0(), '[pixels] x coordinate of disc centre.')
        put('DISC Y0', self.get_y0(), '[pixels] y coordinate of disc centre.')
        put('DISC R0', self.get_r0(), '[pixels] equatorial radius of disc.')
        put('DISC ROT', self.get_rotation(), '[degrees] rotation of image.')
        put('DISC METHOD', self.get_disc_method(), 'Method used to find disc.')
        put('ALTITUDE-ADJUSTMENT', self._alt_adjustment, '[km] Adjustment to surface altitude.')
        put('UTC-OBS', self.utc, 'UTC date of observation')
        put('ET-OBS', self.et, 'J2000 ephemeris seconds of observation.')
        put('TARGET', self.target, 'Target body name used in SPICE.')
        put('TARGET-ID', self.target_body_id, 'Target body ID from SPICE.')
        put('SUBPOINT LAT', self.subpoint_lat, '[degrees] Sub-observer pgr latitude.')
        put('SUBPOINT LON', self.subpoint_lon, '[degrees] Sub-observer pgr longitude.')
        put('SUBSOL LAT', self.subsol_lat, '[degrees] Sub-solar pgr latitude.')
        put('SUBSOL LON', self.subsol_lon, '[degrees] Sub-solar pgr longitude.')
        put('LON-DIRECTION', self.positive_longitude_direction, 'Positive pgr longitude direction.')
        put('NP-ANGLE', self.north_pole_angle(), '[degrees] North pole angle.')
        put('TARGET RA', self.target_ra, '[degrees] RA of target centre.')
        put('TARGET DEC', self.target_dec, '[degrees] Dec of target centre.')
        put('TARGET DIAMETER', self.target_diameter_arcsec, '[arcsec] Equatorial angular diameter of target.')
        put('R EQ', self.r_eq, '[km] Target equatorial radius from SPICE.')
        put('R POLAR', self.r_polar, '[km] Target polar radius from SPICE.')
        put('FLATTENING', self.flattening, 'Flattening of target body.')
        put('LIGHT-TIME', self.target_light_time, '[seconds] Light time to target from SPICE.')
        put('DISTANCE', self.target_distance, '[km] Distance to target from SPICE.')
        put('OBSERVER', self.observer, 'Observer name used in SPICE.')
        put('TARGET-FRAME', self.target_frame, 'Target frame used in SPICE.')
        put('OBSERVER-FRAME', self.observer_frame, 'Observer frame used in SPICE.')
        put('ILLUMINATION', self.illumination_source, 'Illumination source used in SPICE.')
        put('ABCORR', self.aberration_correction, 'Aberration correction used in SPICE.')
        put('SUBPOINT-METHOD', self.subpoint_method, 'Subpoint method used in SPICE.')
        put('SURFACE-METHOD', self.surface_method, 'Surface intercept method used in SPICE.')
        put('OPTIMIZATION-USED', self._optimize_speed, 'Speed optimizations used.')

    def make_filename(self, extension: str = '.fits', prefix: str = '', suffix: str = '') -> str:
        """observation.py:1161-1182, e.g. 'JUPITER_2005-01-01T000000.fits'"""
        date = self.dtm.strftime('%Y-%m-%dT%H%M%S') if self.dtm is not None else str(self.utc)
        return f'{prefix}{self.target}_{date}{suffix}{extension}'

    # ------------------------------------------------------------------ saving
    def _get_backplane_names_to_save(self, backplanes_to_save, backplanes_to_skip) -> set[str]:
        if backplanes_to_save is None:
            backplanes_to_save = self.backplanes.keys()
        return {self.standardise_backplane_name(n) for n in backplanes_to_save} - {
            self.standardise_backplane_name(n) for n in backplanes_to_skip
        }

    @staticmethod
    def _check_wireframe(include_wireframe: bool) -> None:
        if include_wireframe:
            raise UnsupportedError(
                'the WIREFRAME overlay is rendered with matplotlib in the reference '
                '(body_xy.py:2155-2310) and is not part of this path: pass include_wireframe=False'
            )

    def save_observation(
        self, path, *, backplanes_to_save=None, backplanes_to_skip=frozenset(), include_wireframe: bool = False,
        wireframe_kwargs=None, show_progress: bool = False, print_info: bool = True, alt: float = 0.0,
    ) -> None:  # fmt: skip
        """
        observation.py:1184-1303: FITS file with the observed cube + metadata in the primary
        HDU and one IMAGE extension per backplane. All requested planes are generated by the
        fused GPU launches (`prefetch_backplane_imgs`) before the file is assembled.
        """
        self._check_wireframe(include_wireframe)
        path = os.fspath(path)
        names = self._get_backplane_names_to_save(backplanes_to_save, backplanes_to_skip)
        if print_info:
            print('Saving observation to', path)
        with _AltitudeContext(self, alt):
            header = self.header.copy()
            self.add_header_metadata(header)
            hdus = [fits_io.HDU(self.data, header, 'PRIMARY')]
            own = [n for n in names if n in PLANE_NAMES and n in self.backplanes]
            if own:
                self.prefetch_backplane_imgs(own, alt=alt)
            for name, backplane in self.backplanes.items():
                if name not in names:
                    continue
                if print_info:
                    print(' Creating backplane:', name)
                h = Header([('ABOUT', backplane.description)])
                h.add_comment('Backplane generated by PlanetMapper software.')
                hdus.append(fits_io.HDU(backplane.get_img(), h, name))
            if print_info:
                print(' Saving file...')
            _check_path(path)
            fits_io.write(path, hdus, overwrite=True)
        if print_info:
            print('File saved')

    def save_mapped_observation(
        self, path, *, interpolation='linear', propagate_nan: bool = True, spline_smoothing: float = 0,
        smooth_oversample_by: int = 5, smooth_max_oversampled_img_size: int = 10_000,
        include_backplanes: bool = True, backplanes_to_save=None, backplanes_to_skip=frozenset(),
        include_wireframe: bool = False, wireframe_kwargs=None, show_progress: bool = False,
        print_info: bool = True, **map_kwargs,
    ) -> None:  # fmt: skip
        """observation.py:1317-1474: mapped cube + map-space backplanes + map WCS cards"""
        self._check_wireframe(include_wireframe)
        path = os.fspath(path)
        names = self._get_backplane_names_to_save(backplanes_to_save, backplanes_to_skip)
        if print_info:
            print('Saving map to', path)
        with _AltitudeContext(self, map_kwargs.get('alt', 0.0)):
            if print_info:
                print(' Projecting mapped data...')
            interp_kw = dict(
                interpolation=interpolation, spline_smoothing=spline_smoothing, propagate_nan=propagate_nan,
                smooth_oversample_by=smooth_oversample_by,
                smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
            )  # fmt: skip
            data = self.get_mapped_data(**interp_kw, **map_kwargs)
            header = self.header.copy()
            self.add_header_metadata(header)
            self._add_map_header_metadata(header, **interp_kw, **map_kwargs)
            self._add_map_wcs_to_header(header, **map_kwargs)
            hdus = [fits_io.HDU(data, header, 'PRIMARY')]
            if include_backplanes:
                own = [n for n in names if n in PLANE_NAMES and n in self.backplanes]
                if own:
                    self._map_planes(own, dict(map_kwargs))  # one launch for all requested planes
                for name, backplane in self.backplanes.items():
                    if name not in names:
                        continue
                    if print_info:
                        print(' Creating backplane:', name)
                    h = Header([('ABOUT', backplane.description)])
                    h.add_comment('Backplane generated by PlanetMapper software.')
                    self._add_map_wcs_to_header(h, **map_kwargs)
                    hdus.append(fits_io.HDU(backplane.get_map(**map_kwargs), h, name))
            if print_info:
                print(' Saving file...')
            _check_path(path)
            fits_io.write(path, hdus, overwrite=True)
        if print_info:
            print('File saved')

    def _add_map_header_metadata(
        self, header: Header, *, interpolation, spline_smoothing, propagate_nan, smooth_oversample_by,
        smooth_max_oversampled_img_size, **map_kwargs,
    ) -> None:  # fmt: skip
        """observation.py:1476-1572"""
        info = self.generate_map_coordinates(**map_kwargs)[5]
        put = lambda k, v, c: self.append_to_header(k, v, c, header=header)  # noqa: E731
        put('MAP INTERPOLATION', str(interpolation) if isinstance(interpolation, tuple) else interpolation,
            'Interpolation method used in mapping.')  # fmt: skip
        if interpolation not in {'nearest', 'smooth'}:
            put('MAP SPLINE-SMOOTHING', spline_smoothing, 'Interpolation spline smoothing factor used in mapping.')
            put('MAP PROPAGATE-NAN', propagate_nan, 'Propagate NaN pixels to map when mapping.')
        if interpolation == 'smooth':
            put('MAP SMOOTH-OVERSAMPLE-BY', smooth_oversample_by, 'Oversampling factor used in map interpolation.')
            put('MAP SMOOTH-MAX-OVERSAMPLED-IMG-SIZE', smooth_max_oversampled_img_size,
                'Maximum oversampled image size allowed map interpolation.')  # fmt: skip
        put('MAP PROJECTION', info['projection'], 'Projection used for mapping.')
        for key, kw, comment in (
            ('degree_interval', 'MAP DEGREE-INTERVAL', '[deg] Degree interval in output map.'),
            ('lon', 'MAP LON', 'Central longitude of map projection.'),
            ('lat', 'MAP LAT', 'Central latitude of map projection.'),
            ('size', 'MAP SIZE', 'Size of output map.'),
        ):
            if key in info:
                put(kw, info[key], comment)

    def _add_map_wcs_to_header(self, header: Header, **map_kwargs) -> None:
        """observation.py:1574-1612: linear WCS cards for rectangular maps, stale ones removed"""
        lons, lats, _, _, _, info = self.generate_map_coordinates(**map_kwargs)
        if info['projection'] == 'rectangular':
            header['CTYPE1'] = f'Planetographic longitude, positive {self.positive_longitude_direction}'
            header['CUNIT1'] = 'deg'
            header['CRPIX1'] = 1
            header['CRVAL1'] = float(lons[0][0])
            header['CDELT1'] = float(lons[0][1] - lons[0][0])
            header['CTYPE2'] = 'Planetographic latitude'
            header['CUNIT2'] = 'deg'
            header['CRPIX2'] = 1
            header['CRVAL2'] = float(lats[0][0])
            header['CDELT2'] = float(lats[1][0] - lats[0][0])
        else:
            for n in '12':
                for key in (f'CTYPE{n}', f'CUNIT{n}', f'CRPIX{n}', f'CRVAL{n}', f'CDELT{n}'):
                    header.remove(key, ignore_missing=True, remove_all=True)
        for a in '12':
            for b in '123':
                for key in (f'PC{a}_{b}', f'PC{b}_{a}', f'CD{a}_{b}', f'CD{b}_{a}'):
                    header.remove(key, ignore_missing=True, remove_all=True)

    def set_img_size(self, nx: int | None = None, ny: int | None = None) -> None:
        """observation.py:341-343"""
        raise TypeError('Cannot set image size for Observation objects')

    def get_mapped_data(
        self,
        interpolation: str | int | tuple[int, int] = 'linear',
        *,
        spline_smoothing: float = 0,
        propagate_nan: bool = True,
        smooth_oversample_by: int = 5,
        smooth_max_oversampled_img_size: int = 10_000,
        device: bool = False,
        **map_kwargs,
    ) -> np.ndarray:
        """
        Project every plane of `self.data` onto the map: returns a fresh
        (P, n_lat, n_lon) float64 array (copy of the cached result, observation.py:826-872).
        `device=True` (not in the reference): the mapped cube as a read-only `DeviceArray` left in the GPU's HBM
        (`__dlpack__` / `__cuda_array_interface__`; valid until the disc or the image size changes) - the cube is uploaded
        once per Observation, the lon / lat grid once per set of map keywords, x / y maps and mapped planes never leave the GPU.
        """
        if device:
            return self._get_mapped_data_device(
                interpolation=interpolation, spline_smoothing=spline_smoothing, propagate_nan=propagate_nan,
                smooth_oversample_by=smooth_oversample_by, smooth_max_oversampled_img_size=smooth_max_oversampled_img_size, **map_kwargs,
            )  # fmt: skip
        return self._get_mapped_data(
            interpolation=interpolation,
            spline_smoothing=spline_smoothing,
            propagate_nan=propagate_nan,
            smooth_oversample_by=smooth_oversample_by,
            smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
            **map_kwargs,
        ).copy()

    def _data_device(self):
        """`self.data` in HBM (uploaded once; again if `data` is replaced), in its own dtype where the kernels read that"""
        from .engine import _DTYPE_CODES  # noqa: PLC0415

        held = self.__dict__.get('_data_dev')
        if held is None or held[0] is not self.data or not held[1].valid:
            data = np.ascontiguousarray(self.data if np.dtype(self.data.dtype) in _DTYPE_CODES else np.asarray(self.data, dtype=np.float64))
            eng = self._bind()
            arr = eng.device_array(data.shape, data.dtype)
            eng.h2d(arr.ptr, data)
            if held is not None:
                held[1].invalidate()
            held = self.__dict__['_data_dev'] = (self.data, arr)
        return held[1]

    def _get_mapped_data_device(self, **kwargs):
        """the device form of `_get_mapped_data` (same cache key with 'mapped_dev'; cleared - invalidated - with it)"""
        from .engine import interpolation_code  # noqa: PLC0415

        interpolation_code(kwargs['interpolation'])
        if kwargs['spline_smoothing'] < 0:
            raise ValueError('s should be s >= 0.0')
        key = ('mapped_dev', tuple(sorted((k, _freeze(v)) for k, v in kwargs.items())), self._alt_adjustment)
        if key not in self._cache:
            map_kwargs = {k: v for k, v in kwargs.items() if k not in ('interpolation', 'spline_smoothing', 'propagate_nan',
                                                                       'smooth_oversample_by', 'smooth_max_oversampled_img_size')}  # fmt: skip
            eng = self._bind()
            if not hasattr(eng, 'device_array'):
                raise UnsupportedError('this engine keeps no results on a device')
            xm, ym = (self._map_planes_device(['PIXEL-X', 'PIXEL-Y'], dict(map_kwargs))[n] for n in ('PIXEL-X', 'PIXEL-Y'))
            n0, n1 = xm.shape
            cube = self._data_device()
            data = np.asarray(self.data)
            planes = data.shape[0] if data.ndim == 3 else 1
            out = eng.device_array((planes, n0, n1))
            interp = kwargs['interpolation']
            if interp == 'smooth':
                eng.set_smooth_options(kwargs['smooth_oversample_by'], kwargs['smooth_max_oversampled_img_size'])
            smoothing = float(kwargs['spline_smoothing']) if interp not in ('nearest', 'smooth') else 0.0
            eng.map_cube_device(cube, cube.dtype, planes, xm, ym, n0, n1, out, interp, kwargs['propagate_nan'], spline_smoothing=smoothing)
            self._cache[key] = out
        return self._cache[key]

    def _get_mapped_data(self, **kwargs) -> np.ndarray:
        """observation.py:874-905: cached per disc parameters / altitude / arguments."""
        key = ('mapped_data', tuple(sorted((k, _freeze(v)) for k, v in kwargs.items())), self._alt_adjustment)
        if key not in self._cache:
            map_kwargs = {
                k: v
                for k, v in kwargs.items()
                if k
                not in (
                    'interpolation', 'spline_smoothing', 'propagate_nan', 'smooth_oversample_by',
                    'smooth_max_oversampled_img_size',
                )
            }  # fmt: skip
            with _AltitudeContext(self, map_kwargs.get('alt', 0.0)):
                self._cache[key] = self.map_img(
                    self.data,
                    interpolation=kwargs['interpolation'],
                    spline_smoothing=kwargs['spline_smoothing'],
                    propagate_nan=kwargs['propagate_nan'],
                    smooth_oversample_by=kwargs['smooth_oversample_by'],
                    smooth_max_oversampled_img_size=kwargs['smooth_max_oversampled_img_size'],
                    **map_kwargs,
                )
        return self._cache[key]
