"""Opt-in start-up hook: with this directory on PYTHONPATH every `python <reference script>` runs under
w3d_amd.dropin's import redirect — no reference file is edited:

    PYTHONPATH=<repo>/wheat-3dgs_amd/dropin_site:<repo>/wheat-3dgs_amd python train_vanilla_3dgs.py -s <scene> ...

Python imports the first `sitecustomize` it finds; one that this file shadows is chained below."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.dirname(_here)
if _pkg not in sys.path:
    sys.path.insert(0, _pkg)

import w3d_amd.dropin as _dropin  # noqa: E402

_dropin.install()

# chain to a sitecustomize this one shadows (distribution / virtualenv hooks)
for _p in sys.path:
    _f = os.path.join(_p or ".", "sitecustomize.py")
    if os.path.abspath(_p or ".") != _here and os.path.isfile(_f):
        import importlib.util as _u
        _spec = _u.spec_from_file_location("_w3d_chained_sitecustomize", _f)
        _mod = _u.module_from_spec(_spec)
        try:
            _spec.loader.exec_module(_mod)
        except Exception as _e:                 # as site.py does: report, carry on
            print(f"[w3d_amd.dropin] chained sitecustomize {_f} failed: {_e!r}", file=sys.stderr)
        break
