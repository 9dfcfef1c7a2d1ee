"""Run the reference's own scripts on this package: `keymorph_amd.dropin.install()` BEFORE the first `import keymorph`.

The reference has no plugin layer; its callers (scripts/run.py:13-17, train.py:8-11, register.py:10-12,
pairwise_register_eval.py:6-9, groupwise_register_eval.py:8-10) import `keymorph.<module>` by name.  install() registers
the modules this package implements under those names in `sys.modules` -- and ONLY those: `keymorph.viz_tools`,
`keymorph.baselines.*` and anything else the maintainer's checkout has keep resolving to the maintainer's own files,
because the parent package stays theirs when it is importable (its `from . import model` lines then pick up the aliases).
Without a `keymorph` checkout on the path an empty namespace package stands in as the parent."""
import importlib
import importlib.util
import sys
import types

SUBMODULES = ("utils", "transformations", "layers", "loss_ops", "keypoint_aligners", "augmentation", "net", "unet3d",
              "unet3d.model", "model")


def install(name: str = "keymorph"):
    """Alias the implemented submodules as `<name>.<sub>`; returns the parent package module."""
    parent = sys.modules.get(name)
    if parent is not None and not getattr(parent, "__keymorph_amd_dropin__", False):
        raise RuntimeError(f"keymorph_amd.dropin.install() must run before the first `import {name}` "
                           f"({name} is already imported from {getattr(parent, '__file__', '?')})")
    mods = {sub: importlib.import_module("keymorph_amd." + sub) for sub in SUBMODULES}
    for sub, mod in mods.items():
        sys.modules[f"{name}.{sub}"] = mod
    if parent is None:
        try:
            spec = importlib.util.find_spec(name)
        except (ImportError, ValueError):
            spec = None
        if spec is None:            # no checkout of the reference on the path: an empty parent
            parent = types.ModuleType(name)
            parent.__path__ = []
            parent.__doc__ = "namespace created by keymorph_amd.dropin.install()"
            sys.modules[name] = parent
        else:                       # the maintainer's package: its __init__ imports resolve to the aliases above
            parent = importlib.import_module(name)
    for sub, mod in mods.items():
        if "." not in sub:
            setattr(parent, sub, mod)
    parent.__keymorph_amd_dropin__ = True
    return parent


def uninstall(name: str = "keymorph"):
    """Remove every alias install() made (tests)."""
    for key in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
        del sys.modules[key]
