"""Import redirect: an UNMODIFIED Wheat-3DGS checkout runs on this package's fast path.

Putting `wheat-3dgs_amd/` on PYTHONPATH swaps the three rasterizer packages the reference binds by name
(`diff_gaussian_rasterization`, `flashsplat_rasterization`, `simple_knn._C`).  The rest of the training step — the model's
activations and its six-group torch.optim.Adam (reference scene/gaussian_model.py:101-121,172-186), render()'s marshalling
(gaussian_renderer/__init__.py:22-106) and the conv2d SSIM (utils/loss_utils.py:39-63) — lives in modules of the CHECKOUT,
imported by name from its scripts (train_vanilla_3dgs.py:16-18, run_3d_seg.py:21-22, render.py:17-22, scene/__init__.py:17).
`install()` redirects exactly those names and nothing else:

    scene.gaussian_model.GaussianModel        -> w3d_amd.gaussian_model.GaussianModel   (flat store, FlatAdam)
    gaussian_renderer.render                  -> w3d_amd.gaussian_renderer.render       (one autograd node over raw blocks)
    gaussian_renderer.flashsplat_render       -> w3d_amd.gaussian_renderer.flashsplat_render
    gaussian_renderer.GaussianModel           -> (the same class; render.py:22 imports it from there)
    utils.loss_utils.l1_loss / ssim           -> w3d_amd.loss.l1_loss / ssim            (fused kernel pair)

Every module is still LOADED FROM THE CHECKOUT (so `BasicPointCloud`, `l2_loss`, `network_gui`, `render_helper`, … stay what
they were); the named attributes are replaced right after the module body has run, i.e. before any `from X import name` of an
importing script binds them.  Three ways to switch it on, none of which edits a reference file:

    python -m w3d_amd.dropin train_vanilla_3dgs.py -s <scene> ...           # runs the script under the redirect
    PYTHONPATH=<repo>/wheat-3dgs_amd/dropin_site:<repo>/wheat-3dgs_amd python train_vanilla_3dgs.py ...   # sitecustomize
    import w3d_amd.dropin; w3d_amd.dropin.install()                          # before the first import of the checkout

This module imports nothing heavy (no torch): it only registers a finder; the replacements are imported when the first
redirected module is.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # .../wheat-3dgs_amd


def _patch_gaussian_model(mod):
    from .gaussian_model import GaussianModel
    mod._reference_GaussianModel = getattr(mod, "GaussianModel", None)
    mod.GaussianModel = GaussianModel


def _patch_renderer(mod):
    from . import gaussian_renderer as ours
    from .gaussian_model import GaussianModel
    for name in ("render", "flashsplat_render"):
        if hasattr(mod, name):
            setattr(mod, "_reference_" + name, getattr(mod, name))
        setattr(mod, name, getattr(ours, name))
    mod.GaussianModel = GaussianModel


def _patch_loss(mod):
    from . import loss as ours
    for name in ("l1_loss", "ssim"):
        if hasattr(mod, name):
            setattr(mod, "_reference_" + name, getattr(mod, name))
        setattr(mod, name, getattr(ours, name))


# module name -> the attributes replaced there (also the documentation of the redirect: tests compare against it)
REDIRECTS = {
    "scene.gaussian_model": (_patch_gaussian_model, ("GaussianModel",)),
    "gaussian_renderer": (_patch_renderer, ("render", "flashsplat_render", "GaussianModel")),
    "utils.loss_utils": (_patch_loss, ("l1_loss", "ssim")),
}


class _PatchingLoader(importlib.abc.Loader):
    """The checkout's own loader, plus the attribute replacement once the module body has run."""

    def __init__(self, loader, patch):
        self._loader, self._patch = loader, patch

    def create_module(self, spec):
        return self._loader.create_module(spec)

    def exec_module(self, module):
        self._loader.exec_module(module)
        self._patch(module)
        module.__w3d_dropin__ = True

    def __getattr__(self, name):                 # get_code / get_source / is_package / get_filename ... for runpy, inspect
        return getattr(self._loader, name)


class _Finder(importlib.abc.MetaPathFinder):
    def __init__(self, only=None):
        self._busy = set()
        self.names = frozenset(REDIRECTS if only is None else only)

    def find_spec(self, fullname, path=None, target=None):
        if fullname not in self.names or fullname in self._busy:
            return None
        self._busy.add(fullname)
        try:
            spec = None
            for finder in sys.meta_path:
                if finder is self or not hasattr(finder, "find_spec"):
                    continue
                spec = finder.find_spec(fullname, path, target)
                if spec is not None:
                    break
        finally:
            self._busy.discard(fullname)
        if spec is None or spec.loader is None:
            return None                          # not a Wheat-3DGS checkout on the path: nothing to redirect
        spec.loader = _PatchingLoader(spec.loader, REDIRECTS[fullname][0])
        return spec


_finder = None


def installed():
    return _finder is not None and _finder in sys.meta_path


def install(verbose=False, only=None):
    """Register the redirect (idempotent).  Call before the checkout's modules are imported; modules of REDIRECTS that are
    already in sys.modules are patched in place (names other modules have already bound with `from X import …` cannot be
    reached that way — a warning says so).  `only`: a subset of REDIRECTS' module names (bench.py's breakdown of which swap
    buys what); default all."""
    global _finder
    if only is not None:
        unknown = set(only) - set(REDIRECTS)
        if unknown:
            raise ValueError(f"w3d_amd.dropin.install(only=...): not a redirected module: {sorted(unknown)}")
    if _PKG_DIR not in sys.path:                 # the three rasterizer packages + this one, by their import names
        sys.path.insert(0, _PKG_DIR)
    if installed() and _finder.names != frozenset(REDIRECTS if only is None else only):
        uninstall()
    if not installed():
        _finder = _Finder(only)
        sys.meta_path.insert(0, _finder)
    for name, (patch, _) in REDIRECTS.items():
        mod = sys.modules.get(name)
        if name in _finder.names and mod is not None and not getattr(mod, "__w3d_dropin__", False):
            import warnings
            warnings.warn(f"w3d_amd.dropin.install(): {name} was imported before the redirect; patched in place — modules that "
                          "already did `from " + name + " import …` keep the reference's objects", stacklevel=2)
            patch(mod)
            mod.__w3d_dropin__ = True
    if verbose or os.environ.get("W3D_DROPIN_VERBOSE"):
        print("[w3d_amd.dropin] redirect installed: " + ", ".join(f"{m}.{{{','.join(a)}}}" for m, (_, a) in REDIRECTS.items()
                                                                  if m in _finder.names), file=sys.stderr)


def uninstall(purge=()):
    """Remove the finder.  `purge`: top-level package names whose modules are dropped from sys.modules (so that a later import
    loads the checkout's modules unpatched; bench.py times the same loop script both ways in one process)."""
    global _finder
    if _finder is not None and _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    _finder = None
    purge_modules(purge)


def purge_modules(tops):
    for name in list(sys.modules):
        if name.split(".")[0] in tops:
            del sys.modules[name]
    importlib.invalidate_caches()


def status():
    """name -> True / False / None: the redirected attribute currently IS this package's object / is not / module not imported."""
    out = {}
    for name, (_, attrs) in REDIRECTS.items():
        mod = sys.modules.get(name)
        for a in attrs:
            obj = None if mod is None else getattr(mod, a, None)
            out[f"{name}.{a}"] = None if mod is None else (getattr(obj, "__module__", "") or "").startswith("w3d_amd")
    return out


def main(argv=None):
    """python -m w3d_amd.dropin <script.py> [args...]: the script runs as __main__ under the redirect, with its own directory
    first on sys.path exactly as `python <script.py>` would have it."""
    import runpy
    argv = sys.argv[1:] if argv is None else list(argv)
    if not argv or argv[0] in ("-h", "--help"):
        print("usage: python -m w3d_amd.dropin <reference script.py> [its arguments]", file=sys.stderr)
        return 2
    script = os.path.abspath(argv[0])
    from w3d_amd import dropin as _d           # (not this __main__ copy: one finder, one state)
    _d.install(verbose=True)
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
