"""MI355X-native implementation of the MulActSeg hot path (see DESIGN.md / INTEGRATION.md)."""
import importlib
import pkgutil
import sys

__version__ = "0.1.0"

_PLUGIN_PACKAGES = ("active_selection", "trainer", "models", "dataloader", "utils")


def install_aliases():
    """Register ``active_selection.*``, ``trainer.*``, ``models``, ``dataloader``, ``utils`` as aliases of the
    packages in here, so that the reference's drivers (``train_AL.py:29-33``:
    ``importlib.import_module("active_selection." + args.active_method)``) pick up this implementation
    without being edited.  Call it before the driver's own imports."""
    for pkg in _PLUGIN_PACKAGES:
        mod = importlib.import_module(__name__ + "." + pkg)
        sys.modules[pkg] = mod
        for info in pkgutil.iter_modules(mod.__path__):
            sub = importlib.import_module("%s.%s.%s" % (__name__, pkg, info.name))
            sys.modules["%s.%s" % (pkg, info.name)] = sub
