"""`from GaPFlow.md import Mock` (md/__init__.py:26).  Only the Mock runner is on the MI355X hot path; the LAMMPS
runners (LennardJones, GoldAlkane) drive an external MD code and are out of scope."""
from ..gp import Mock  # noqa: F401


def __getattr__(name):
    if name in ('LennardJones', 'GoldAlkane', 'MolecularDynamics'):
        raise NotImplementedError(f"GaPFlow.md.{name} drives LAMMPS; outside the scope of gapflow_amd (DESIGN.md section 7)")
    raise AttributeError(name)
