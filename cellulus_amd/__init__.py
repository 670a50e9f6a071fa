"""cellulus_amd — MI355X-native hot path of funkelab/cellulus.

Mirrors the reference's Python API (``cellulus.train``, ``cellulus.infer``,
``cellulus.models.get_model``, ``cellulus.criterions.get_loss``, the config
classes) on top of hand-written HIP kernels for gfx950 (``libclx.so``, C ABI in
``include/clx.h``).  There is no CPU compute path.
"""

__version__ = "0.2.0"


def install_as_cellulus():
    """Makes ``import cellulus`` / ``from cellulus.train import train`` / ``import
    cellulus.models.unet`` resolve to this package, so that user code written against the
    reference runs unchanged.  Opt-in (call it once, before the first ``import cellulus``): the
    name is only claimed when no other ``cellulus`` has been imported."""
    import importlib
    import importlib.abc
    import importlib.util
    import sys

    other = sys.modules.get("cellulus")
    if other is not None and other is not sys.modules[__name__]:
        raise ImportError("another package named 'cellulus' is already imported")

    class _AliasLoader(importlib.abc.Loader):
        def __init__(self, target):
            self.target = target

        def create_module(self, spec):
            module = importlib.import_module(self.target)    # the SAME module object, not a copy
            self.real_spec = module.__spec__
            return module

        def exec_module(self, module):
            module.__spec__ = self.real_spec                 # the import machinery replaced it

    class _AliasFinder(importlib.abc.MetaPathFinder):
        def find_spec(self, fullname, path=None, target=None):
            if fullname != "cellulus" and not fullname.startswith("cellulus."):
                return None
            real = __name__ + fullname[len("cellulus"):]
            try:
                real_spec = importlib.util.find_spec(real)
            except ModuleNotFoundError:
                return None
            if real_spec is None:
                return None
            return importlib.util.spec_from_loader(
                fullname, _AliasLoader(real), is_package=real_spec.submodule_search_locations is not None)

    if not any(type(f).__name__ == "_AliasFinder" for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
