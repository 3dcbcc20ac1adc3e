"""``import pyatom`` as users of the reference write it: the module lives in atom_amd/pyatom.py."""
from atom_amd.pyatom import Atmosphere, Hydrosphere, Model, parameters  # noqa: F401
