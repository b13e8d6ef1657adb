"""Import shim: the package directory is named ``blackhole-simulation_amd`` (not a
valid Python identifier), so this module turns itself into that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "blackhole-simulation_amd")]
__package__ = "blackhole_simulation_amd"
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
