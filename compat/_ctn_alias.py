"""Makes the reference's top-level import names resolve to speech_amd.

The reference's recipes put its `src/` directories on PYTHONPATH (egs/wsj0-mix/conv-tasnet/path.sh:4) and import
`models.conv_tasnet`, `criterion.pit`, `utils.utils`, `dataset`, `driver`, ... as top-level modules.  With THIS directory
on PYTHONPATH in their place, the same statements load the MI355X implementations: each top-level package here registers
the speech_amd modules of the same name under its own name in sys.modules, so `models.conv_tasnet.ConvTasNet` IS
`speech_amd.models.conv_tasnet.ConvTasNet` (one class object, one set of kernels, checkpoints interchangeable).
"""
import importlib
import os
import pkgutil
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def expose(alias, target):
    """Register every module of package `target` (e.g. 'speech_amd.models') as `alias`.<name>."""
    package = importlib.import_module(target)
    shell = sys.modules[alias]
    for info in pkgutil.iter_modules(package.__path__):
        module = importlib.import_module(target + "." + info.name)
        sys.modules[alias + "." + info.name] = module
        setattr(shell, info.name, module)
    return package
