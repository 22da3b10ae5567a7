"""Make the reference's import paths resolve to this package.

The reference's trainer and scripts import `sg2im.*`, `spade.*` and `scripts.*` as TOP-LEVEL packages
(scripts/train.py:18-24: `from sg2im.meta_models import MetaGeneratorModel`, ...).  The HIP-backed
modules live one level down (`canonicalsg2im_amd.sg2im...`) and use package-relative imports, so they
cannot simply be put first on `sys.path`.  `install()` registers a meta-path finder that aliases

    sg2im[.x.y]    -> canonicalsg2im_amd.sg2im[.x.y]
    spade[.x.y]    -> canonicalsg2im_amd.spade[.x.y]
    scripts[.x.y]  -> canonicalsg2im_amd.scripts[.x.y]      (optional)

to the SAME module objects (no second copy is executed), so both spellings share state:

    import canonicalsg2im_amd.dropin as dropin; dropin.install()      # before the reference's own imports
    from sg2im.meta_models import MetaGeneratorModel, MetaDiscriminatorModel
    from spade.models.networks.sync_batchnorm import DataParallelWithCallback
"""
import importlib
import importlib.abc
import importlib.util
import sys

_PKG = __name__.rsplit(".", 1)[0]


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)          # the real module object becomes sys.modules[alias]

    def exec_module(self, module):                           # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def __init__(self, roots):
        self.roots = tuple(roots)

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".", 1)[0]
        if root not in self.roots:
            return None
        real = _PKG + "." + fullname
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            return None
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=real_spec.submodule_search_locations is not None)
        return spec


def install(scripts=False):
    """Alias `sg2im` and `spade` (and `scripts` when asked: the reference's own `scripts/train.py` must stay
    importable when a maintainer runs IT against these modules) to the HIP-backed packages.  Idempotent.
    Raises if one of the names is already imported from somewhere else."""
    roots = ["sg2im", "spade"] + (["scripts"] if scripts else [])
    for r in roots:
        m = sys.modules.get(r)
        if m is not None and not getattr(m, "__name__", "").startswith(_PKG + "."):
            raise ImportError("%s is already imported from %s; call canonicalsg2im_amd.dropin.install() before the "
                              "reference's imports" % (r, getattr(m, "__file__", "?")))
    for f in sys.meta_path:
        if isinstance(f, _AliasFinder):
            f.roots = tuple(sorted(set(f.roots) | set(roots)))
            return f
    f = _AliasFinder(roots)
    sys.meta_path.insert(0, f)
    return f
