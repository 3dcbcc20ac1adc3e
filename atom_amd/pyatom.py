"""pyatom for Python 3: the reference's Python interface (python/pyatom.pyx.template:9-188) over libatomdev.so.

``Atmosphere`` and ``Hydrosphere`` wrap the C++ classes cAtmosphereModel / cHydrosphereModel (include/atom_models.hpp)
through their flat C interface: ``load_config``, ``run``, ``run_time_slice``, one property per configuration parameter
(the parameters of param.py:9-178, listed by the library itself), and the context-manager protocol that destroys the C++
object on exit - after which every use raises ``RuntimeError("Wrapped C++ object is deleted")``.  ``Model`` bundles one
of each as the reference's does.  Where the reference's module is compiled by Cython against the model library, this one
binds the same calls with ctypes: nothing to compile, same names and behaviour.
"""
import ctypes as C
import os

from .device import AtomDevError, load_library

ATM, HYD = 0, 1


def _lib():
    lib = load_library()
    if not getattr(lib, "_models_bound", False):
        vp = C.c_void_p
        lib.atom_model_last_error.restype = C.c_char_p
        lib.atom_model_create.restype = vp
        lib.atom_model_create.argtypes = [C.c_int]
        lib.atom_model_destroy.restype = None
        lib.atom_model_destroy.argtypes = [vp]
        lib.atom_model_load_config.argtypes = [vp, C.c_char_p]
        lib.atom_model_run.argtypes = [vp]
        lib.atom_model_run_time_slice.argtypes = [vp, C.c_int]
        lib.atom_model_param_count.argtypes = [C.c_int]
        for name in ("atom_model_param_name", "atom_model_param_type", "atom_model_param_section"):
            getattr(lib, name).restype = C.c_char_p
            getattr(lib, name).argtypes = [C.c_int, C.c_int]
        lib.atom_model_get_param.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
        lib.atom_model_set_param.argtypes = [vp, C.c_char_p, C.c_char_p]
        lib.atom_model_device_handle.restype = vp
        lib.atom_model_device_handle.argtypes = [vp]
        lib.atom_model_mean_temperature_from_curve.restype = C.c_double
        lib.atom_model_mean_temperature_from_curve.argtypes = [vp, C.c_double]
        lib.atom_model_get_tropopause.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
        lib.atom_model_calculate_mean_temperature.restype = C.c_double
        lib.atom_model_calculate_mean_temperature.argtypes = [vp]
        lib.atomdev_get_rk_variant.argtypes = [vp]
        lib._models_bound = True
    return lib


def parameters(kind):
    """[(name, type, section)] of a model's configuration parameters, in the reference's order."""
    lib = _lib()
    return [(lib.atom_model_param_name(kind, n).decode(), lib.atom_model_param_type(kind, n).decode(),
             lib.atom_model_param_section(kind, n).decode()) for n in range(lib.atom_model_param_count(kind))]


_FROM_TEXT = {"int": int, "double": float, "bool": lambda t: t == "true", "string": str}


def _to_text(ctype, value):
    if ctype == "bool":
        return "true" if value else "false"
    if ctype == "int":
        return "%d" % int(value)
    if ctype == "double":
        return repr(float(value))
    return str(value)


class _Model:
    _kind = None

    def __init__(self):
        self._lib = _lib()
        self._thisptr = self._lib.atom_model_create(self._kind)
        if not self._thisptr:
            raise MemoryError(self._lib.atom_model_last_error().decode())

    def __del__(self):
        try:
            self._delete()
        except Exception:
            pass

    def _delete(self):
        if getattr(self, "_thisptr", None):
            self._lib.atom_model_destroy(self._thisptr)
            self._thisptr = None

    def _check_alive(self):
        if not self._thisptr:
            raise RuntimeError("Wrapped C++ object is deleted")

    def _check(self, rc):
        if rc != 0:
            raise AtomDevError(self._lib.atom_model_last_error().decode())

    def load_config(self, filename):
        self._check_alive()
        self._check(self._lib.atom_model_load_config(self._thisptr, os.fsencode(filename)))

    def run(self):
        self._check_alive()
        self._check(self._lib.atom_model_run(self._thisptr))

    def run_time_slice(self, t):
        self._check_alive()
        self._check(self._lib.atom_model_run_time_slice(self._thisptr, int(t)))

    def device_handle(self):
        """The atomdev handle that holds the model's fields (None before the first time slice)."""
        self._check_alive()
        return self._lib.atom_model_device_handle(self._thisptr)

    def device_rk_variant(self):
        """Which form of the 3-D sweep the model's handle runs (atomdev_get_rk_variant; None before the first slice)."""
        h = self.device_handle()
        return None if not h else self._lib.atomdev_get_rk_variant(h)

    # the context manager protocol controls the lifetime of the wrapped C++ object
    def __enter__(self):
        self._check_alive()
        return self

    def __exit__(self, exc_tp, exc_val, exc_tb):
        self._delete()
        return False  # propagate exceptions


def _property(name, ctype):
    key = name.encode()

    def getter(self):
        self._check_alive()
        buf = C.create_string_buffer(4096)
        self._check(self._lib.atom_model_get_param(self._thisptr, key, buf, len(buf)))
        return _FROM_TEXT[ctype](buf.value.decode())

    def setter(self, value):
        self._check_alive()
        self._check(self._lib.atom_model_set_param(self._thisptr, key, _to_text(ctype, value).encode()))

    return property(getter, setter, doc="%s (%s)" % (name, ctype))


def _install_properties(cls):
    try:
        for name, ctype, _ in parameters(cls._kind):
            setattr(cls, name, _property(name, ctype))
        cls._properties_installed = True
    except AtomDevError:   # the library is not built: the classes still import, creating one raises
        cls._properties_installed = False


class Atmosphere(_Model):
    """Wrapper of the C++ class cAtmosphereModel."""
    _kind = ATM

    def __init__(self):
        if not type(self)._properties_installed:
            _install_properties(type(self))
        super().__init__()

    def mean_temperature_from_curve(self, time):
        self._check_alive()
        return self._lib.atom_model_mean_temperature_from_curve(self._thisptr, float(time))

    def get_tropopause(self):
        """cAtmosphereModel::get_tropopause: the tropopause level of each of the 181 latitude rows."""
        self._check_alive()
        levels = (C.c_int * 181)()
        n = self._lib.atom_model_get_tropopause(self._thisptr, levels, 181)
        return list(levels)[:n]

    def calculate_mean_temperature(self):
        """cAtmosphereModel::calculate_mean_temperature of the model's own temperature field, in C."""
        self._check_alive()
        return self._lib.atom_model_calculate_mean_temperature(self._thisptr)


class Hydrosphere(_Model):
    """Wrapper of the C++ class cHydrosphereModel."""
    _kind = HYD

    def __init__(self):
        if not type(self)._properties_installed:
            _install_properties(type(self))
        super().__init__()


_install_properties(Atmosphere)
_install_properties(Hydrosphere)


class Model:
    """ATOM Model Object (pyatom.pyx.template:9-41): one atmosphere and one hydrosphere."""

    def __init__(self):
        self.atmosphere = Atmosphere()
        self.hydrosphere = Hydrosphere()
        self.output_path = ""
        self.time_slice = 0
        self.config_xml = ""

    def run_atm(self, t_s, output_path, cfg_xml):
        self.time_slice, self.output_path, self.config_xml = t_s, output_path, cfg_xml
        self.atmosphere.load_config(self.config_xml)
        self.atmosphere.run_time_slice(self.time_slice)

    def run_hyd(self, t_s, output_path, cfg_xml):
        self.time_slice, self.output_path, self.config_xml = t_s, output_path, cfg_xml
        self.input_path = output_path
        self.hydrosphere.load_config(self.config_xml)
        self.hydrosphere.run_time_slice(self.time_slice)
