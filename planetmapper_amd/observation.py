"""
`Observation`: a `BodyXY` plus a data cube, with the reference's
`Observation.get_mapped_data` (`planetmapper/observation.py:826-905`) running on the GPU,
optionally sharded by wavelength plane over the GPUs of a node.

FITS/PNG I/O, header parsing, disc fitting and saving (`observation.py:87-823,
908-1612`) are callers of this path, not part of it: pass the data array directly.
"""

from __future__ import annotations

import numpy as np

from .body_xy import BodyXY, _AltitudeContext, _freeze


class Observation(BodyXY):
    """
    Args:
        data: image cube of shape (P, ny, nx), or a single (ny, nx) image which is
            treated as a one-plane cube like the reference does for 2D FITS/PNG data
            (observation.py:228-238). Any dtype the reference accepts; kept as-is
            (float64 / float32 / int16 / int32 / uint8 / uint16 are read natively by
            the kernel, everything else is converted to float64 once).
        **kwargs: `BodyXY` arguments (`geometry=` or `scenario=`, `optimize_speed`, ...).
            `nx`, `ny` and `sz` are taken from the data.
    """

    def __init__(self, path=None, *args, data: np.ndarray | None = None, **kwargs) -> None:
        for k in ('nx', 'ny', 'sz'):
            if k in kwargs:  # observation.py:95-97
                raise TypeError(f'Cannot set {k} for Observation objects')
        if path is not None:
            raise NotImplementedError(
                'reading FITS/PNG files is outside this path: load the array yourself and pass data='
            )
        if data is None:
            raise ValueError('Either `path` or `data` must be provided')
        data = np.asarray(data)
        if data.ndim == 2:
            data = data[None]
        if data.ndim != 3:
            raise ValueError('data must be a 2D image or a 3D cube (P, ny, nx)')
        self.data = data
        super().__init__(*args, nx=data.shape[2], ny=data.shape[1], **kwargs)

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
        **map_kwargs,
    ) -> np.ndarray:
        """
        Project every plane of `self.data` onto the map: returns a fresh
        (P, n_lat, n_lon) float64 array (copy of the cached result, observation.py:826-872).
        """
        return self._get_mapped_data(
            interpolation=interpolation,
            spline_smoothing=spline_smoothing,
            propagate_nan=propagate_nan,
            smooth_oversample_by=smooth_oversample_by,
            smooth_max_oversampled_img_size=smooth_max_oversampled_img_size,
            **map_kwargs,
        ).copy()

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
