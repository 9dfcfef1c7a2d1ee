"""Loud placeholders for reference symbols that are OFF the registration hot path.

The reference's scripts import some of these by name next to the symbols this package implements
(`from keymorph.unet3d.model import UNet2D, UNet3D, TruncatedUNet3D`, scripts/run.py:13), so the names must exist for
the import lines to succeed; USING one raises NotImplementedError naming the reference definition."""


def absent_function(name, where):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name} ({where}) is outside the hot path keymorph_amd rebuilds; use the reference's own")
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = f"Not provided: {where}.  Raises NotImplementedError."
    return fn


def absent_class(name, where, base=object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f"{name} ({where}) is outside the hot path keymorph_amd rebuilds; use the reference's own")
    return type(name, (base,), {"__init__": __init__, "__doc__": f"Not provided: {where}.  Constructing it raises "
                                                                  f"NotImplementedError.", "__module__": __name__})
