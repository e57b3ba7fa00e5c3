"""
Test double for `planetmapper_amd.engine.Engine` backed by the CPU oracle, so the
host-side Python layer (BodyXY / Observation: caches, read-only views, registry, map
kwargs, error behaviour) can be tested without a GPU. Lives in tests/ because only
tests may touch the oracle; the product never falls back to it.
"""

import numpy as np

from oracle import oracle


class OracleEngine:
    def __init__(self):
        self._g = None
        self._d = None
        self._ctx = object()
        self.calls = []  # (kind, names) log, used to assert caching behaviour

    device = 0
    _chunk_cb = None
    redo_planes = 0  # what last_redo_planes() reports (tests set it to rehearse the second gather)

    def set_chunk_callback(self, fn):
        self._chunk_cb = fn

    def last_redo_planes(self):
        return self.redo_planes

    def _report_chunks(self, n_planes):
        """like the library: planes arrive in order, in a few pieces"""
        if self._chunk_cb is None:
            return
        piece = max(1, n_planes // 3)
        for first in range(0, n_planes, piece):
            self._chunk_cb(first, min(piece, n_planes - first))

    def synchronize(self):
        self.calls.append(('sync',))

    def set_geometry(self, g):
        self._g = g

    def set_disc(self, x0, y0, r0, rotation_rad, nx, ny, optimize_speed=True):
        d = oracle.make_disc(x0, y0, r0, 0.0, nx, ny, optimize_speed)
        d.rotation_rad = rotation_rad
        self._d = d

    def backplanes_img(self, names, alt=0.0):
        names = list(names)
        self.calls.append(('img', tuple(names), alt))
        return oracle.backplanes_img(self._g, self._d, names, alt=alt)

    def backplanes_img_rows(self, names, row_begin, n_rows, alt=0.0):
        self.calls.append(('rows', tuple(names), row_begin, n_rows))
        full = oracle.backplanes_img(self._g, self._d, list(names), alt=alt)
        return {n: np.ascontiguousarray(a[row_begin : row_begin + n_rows]) for n, a in full.items()}

    def backplanes_map(self, names, lon, lat, alt=0.0):
        names = list(names)
        self.calls.append(('map', tuple(names), alt))
        return oracle.backplanes_map(self._g, self._d, names, lon, lat, alt=alt)

    def map_cube(self, cube, x_map, y_map, interpolation='linear', propagate_nan=True, **smooth):  # incl. spline_smoothing
        from planetmapper_amd.engine import interpolation_code

        interpolation_code(interpolation)
        cube = np.asarray(cube)
        if cube.ndim == 2:
            cube = cube[None]
        if cube.dtype not in oracle.DTYPES:
            cube = cube.astype(np.float64)
        self.calls.append(('cube', cube.shape, interpolation))
        return oracle.map_cube(cube, x_map, y_map, interpolation, propagate_nan, **smooth)

    def map_cube_device(self, cube, dtype, n_planes, x_map, y_map, n0, n1, out, interpolation='linear',
                        propagate_nan=True):
        """stand-in for the device-resident call: CPU torch tensors in, result written into `out`"""
        import torch

        res = self.map_cube(cube.numpy()[:n_planes], x_map.numpy(), y_map.numpy(), interpolation, propagate_nan)
        out.reshape(n_planes, n0, n1).copy_(torch.from_numpy(res))
        self._report_chunks(n_planes)

    def mapped_data_device(self, cube, dtype, n_planes, lon, lat, n0, n1, x_map, y_map, out, interpolation='linear',
                           propagate_nan=True, alt=0.0):
        """stand-in for pm_mapped_data: x/y map written into x_map / y_map, planes mapped with it"""
        import torch

        o = self.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon.numpy(), lat.numpy(), alt=alt)
        x_map.copy_(torch.from_numpy(o['PIXEL-X']))
        y_map.copy_(torch.from_numpy(o['PIXEL-Y']))
        self.map_cube_device(cube, dtype, n_planes, x_map, y_map, n0, n1, out, interpolation, propagate_nan)

    # stand-ins for the device / pinned-memory calls of bench.py's host-fed cube section (CPU tensors)
    def pinned_empty(self, shape, dtype=np.float64):
        return np.empty(shape, dtype=dtype)

    def xy_map_device(self, lon, lat, n0, n1, x_map, y_map, alt=0.0):
        import torch

        o = self.backplanes_map(['PIXEL-X', 'PIXEL-Y'], lon.numpy(), lat.numpy(), alt=alt)
        x_map.copy_(torch.from_numpy(o['PIXEL-X']))
        y_map.copy_(torch.from_numpy(o['PIXEL-Y']))

    def map_cube_host_to_device(self, cube, x_map, y_map, n0, n1, out, interpolation='linear', propagate_nan=True):
        import torch

        self.calls.append(('host_cube', cube.shape))
        out.reshape(cube.shape[0], n0, n1).copy_(
            torch.from_numpy(oracle.map_cube(cube, x_map.numpy(), y_map.numpy(), interpolation, propagate_nan))
        )
        self._report_chunks(cube.shape[0])

    def radec_query(self, ra, dec, *, alt=0.0, ring_only_visible=True):
        ra, dec = np.broadcast_arrays(np.asarray(ra, dtype=np.float64), np.asarray(dec, dtype=np.float64))
        q = oracle.radec_query(self._g, ra.ravel(), dec.ravel(), alt=alt, ring_only_visible=ring_only_visible)
        return np.ascontiguousarray(q.T).reshape((8,) + ra.shape)

    def transform(self, src, dst, a, b, *, alt=0.0, not_visible_nan=False, planetocentric=False):
        self.calls.append(('transform', src, dst))
        return oracle.transform(self._g, self._d, src, dst, a, b, alt=alt, not_visible_nan=not_visible_nan,
                                planetocentric=planetocentric)
