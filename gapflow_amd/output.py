"""sol.nc / topo.nc writer with the variable names and shapes the reference's viz tools read.

The reference writes through muGrid's FileIONetCDF (problem.py:185-205, 629); its readers expect
(GaPFlow/viz/animations.py:173-182, viz/plotting.py:331-348):

    solution        (frame, 3, 1, Nx+2, Ny+2)
    pressure        (frame, Nx+2, Ny+2)
    wall_stress_xz  (frame, 12, 1, Nx+2, Ny+2)     wall_stress_yz likewise
    topography      (frame, 4, 1, Nx+2, Ny+2)      in topo.nc

netCDF4 is not available in this environment, so frames are written as NetCDF-3 (64-bit offset)
through scipy.io.netcdf_file, which netCDF4.Dataset opens transparently.
"""
import os

import numpy as np
from scipy.io import netcdf_file


class FieldWriter:

    def __init__(self, problem):
        self._p = problem
        nx, ny = problem._shape
        # without elastic deformation the topography is written once and closed; with it a frame is appended with every
        # solution frame (problem.py:183-190, 636-637)
        topo = netcdf_file(os.path.join(problem.outdir, 'topo.nc'), 'w', version=2)
        self._dims(topo, nx, ny, {'tensor_dim__topography-0': 4})
        v = topo.createVariable('topography', 'f8', ('frame', 'tensor_dim__topography-0', 'subpt__1', 'nx', 'ny'))
        v[0] = problem.topo.full[:, None]
        self._topo_file, self._topo_var, self._ntopo = None, None, 1
        if problem.topo.elastic:
            self._topo_file, self._topo_var = topo, v
            topo.flush()
        else:
            topo.close()
        self._f = netcdf_file(os.path.join(problem.outdir, 'sol.nc'), 'w', version=2)
        self._dims(self._f, nx, ny, {'tensor_dim__solution-0': 3, 'tensor_dim__wall_stress-0': 12})
        self._sol = self._f.createVariable('solution', 'f8', ('frame', 'tensor_dim__solution-0', 'subpt__1', 'nx', 'ny'))
        self._pre = self._f.createVariable('pressure', 'f8', ('frame', 'nx', 'ny'))
        self._wxz = self._f.createVariable('wall_stress_xz', 'f8', ('frame', 'tensor_dim__wall_stress-0', 'subpt__1', 'nx', 'ny'))
        self._wyz = self._f.createVariable('wall_stress_yz', 'f8', ('frame', 'tensor_dim__wall_stress-0', 'subpt__1', 'nx', 'ny'))
        self._n = 0

    @staticmethod
    def _dims(f, nx, ny, extra):
        f.createDimension('frame', None)
        f.createDimension('nx', nx)
        f.createDimension('ny', ny)
        f.createDimension('subpt__1', 1)
        for k, v in extra.items():
            f.createDimension(k, v)

    def append_frame(self):
        p = self._p
        k = self._n
        self._sol[k] = np.asarray(p.q)[:, None]
        self._pre[k] = p.pressure.pressure
        self._wxz[k] = p.wall_stress_xz.full[:, None]
        self._wyz[k] = p.wall_stress_yz.full[:, None]
        self._n += 1
        self._f.flush()
        if self._topo_file is not None:
            self._topo_var[self._ntopo] = p.topo.full[:, None]
            self._ntopo += 1
            self._topo_file.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None
        if self._topo_file is not None:
            self._topo_file.close()
            self._topo_file = None
