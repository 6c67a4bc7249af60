"""Import shim: the package directory is `rapidsnark-old_amd/` (not a valid Python
identifier), so this module gives it the importable name `rapidsnark_old_amd`."""
import os as _os

__package__ = "rapidsnark_old_amd"
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rapidsnark-old_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
