"""atom_amd — MI355X-native dynamical core for ATOM's atmosphere / hydrosphere models.

The package is a thin ctypes view of ``libatomdev.so`` (HIP kernels for gfx950
behind the C-ABI declared in ``include/atomdev.h``).  There is no CPU fallback:
importing works anywhere, but creating a device model raises if the library or
a GPU is missing.
"""
from .device import (AtmLoop, AtmLoopState, AtmSlice, AtmStart, ATM, HYD, F2, F3, AtomDevError, DeviceModel, HydSlice, HydStart, Params, Radiation, default_params,
                     atm_run_2d_loops, hyd_run_time_slices, library_path, load_library, rk2d_batch)

__all__ = ["AtmLoop", "AtmLoopState", "AtmSlice", "AtmStart", "ATM", "HYD", "F2", "F3", "AtomDevError", "DeviceModel", "HydSlice", "HydStart", "Params", "Radiation", "default_params",
           "atm_run_2d_loops", "hyd_run_time_slices", "library_path", "load_library", "rk2d_batch"]
