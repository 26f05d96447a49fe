"""Drop-in alias: `import lidargen[.x.y]` resolves to `lidarcrafter_amd.lidargen[.x.y]`
(the SAME module objects, no second copy), so scripts written against the reference package
(`from lidargen.utils import inference`, `from lidargen.utils.configs import __all__`, ...)
run on the MI355X hot path unchanged."""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "lidarcrafter_amd.lidargen"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "lidargen" or name.startswith("lidargen."):
            real = _REAL + name[len("lidargen"):]
            try:
                if importlib.util.find_spec(real) is None:
                    return None
            except ModuleNotFoundError:
                return None
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len("lidargen"):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
sys.modules[__name__] = importlib.import_module(_REAL)
