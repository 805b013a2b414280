"""sol.nc / topo.nc writer with the variable names and shapes the reference's viz tools read.

The reference writes through muGrid's FileIONetCDF (problem.py:185-205, 629); its readers expect
(GaPFlow/viz/animations.py:173-182, viz/plotting.py:331-348):

    solution        (frame, 3, 1, Nx+2, Ny+2)
    pressure        (frame, Nx+2, Ny+2)
    wall_stress_xz  (frame, 12, 1, Nx+2, Ny+2)     wall_stress_yz likewise
    topography      (frame, 4, 1, Nx+2, Ny+2)      in topo.nc
    pressure_var, wall_stress_xz_var, wall_stress_yz_var   (frame, Nx+2, Ny+2)   only with the matching surrogate
                    (problem.py:196-203; viz/plotting.py:340-348 draws its uncertainty bands from them)

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
        # predictive variances of the surrogates, as last evaluated (the reference writes its stored field, which is refreshed
        # one step before every output frame: problem.py:530, stress.py:353-358, 617-620)
        self._var = {}
        for name, var in (('xz', 'wall_stress_xz_var'), ('yz', 'wall_stress_yz_var'), ('zz', 'pressure_var')):
            if name in getattr(problem, '_gp_models', {}):
                self._var[name] = self._f.createVariable(var, 'f8', ('frame', 'nx', 'ny'))
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
        for name, v in self._var.items():
            m = p._gp_models[name]
            v[k] = m.variance if m._var_computed else np.zeros(p._shape)
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
