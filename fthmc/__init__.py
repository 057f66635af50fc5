"""`fthmc` -- the reference's import path, served by the MI355X-native package `fthmc_amd`.

A caller of nftqcd/fthmc keeps its imports (fthmc/main.py:73-104, fthmc/train.py:162, fthmc/ft_hmc.py:109):

    from fthmc.config import Param, TrainConfig, lfConfig
    from fthmc.ft_hmc import FieldTransformation, run_ftHMC
    from fthmc.train import train, train_step, get_model, transfer_to_new_lattice
    from fthmc.hmc import run_hmc
    import fthmc.utils.qed_helpers as qed
    import fthmc.utils.layers as layers

Every `fthmc.X` resolves to the module object `fthmc_amd.X` (one module, two names: patching an attribute
through either name is seen through both).  Nothing is re-implemented here and nothing of the reference is copied.
The reference's orchestration modules that are out of scope (main, utils.io, utils.logger, utils.plot_helpers,
utils.parse_configs; DESIGN.md, Scope) do not exist under either name and raise ModuleNotFoundError.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import fthmc_amd

__version__ = fthmc_amd.__version__
_ALIAS, _REAL = __name__, fthmc_amd.__name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`fthmc.a.b` -> the already (or now) imported module `fthmc_amd.a.b`."""

    _specs: dict = {}

    @staticmethod
    def _real(fullname):
        return _REAL + fullname[len(_ALIAS):]

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + '.'):
            return None
        try:
            spec = importlib.util.find_spec(self._real(fullname))
        except ModuleNotFoundError:
            return None
        if spec is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=spec.submodule_search_locations is not None)

    def create_module(self, spec):
        mod = importlib.import_module(self._real(spec.name))
        self._specs[spec.name] = mod.__spec__
        return mod

    def exec_module(self, module):
        # the real module is already initialised; the import machinery has just pointed its __spec__ at the alias:
        # put the real one back (importlib.reload and `python -m` read it)
        for name, real in list(self._specs.items()):
            if real is not None and real.name == module.__name__:
                module.__spec__ = real
                del self._specs[name]


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
