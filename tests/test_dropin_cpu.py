"""CPU: the drop-in boundary.  keymorph_amd.dropin.install() must let the reference's
scripts import everything they import from `keymorph.*`, with the modules this package does not rebuild (viz_tools,
baselines) still resolving to the maintainer's own files.  Only the scripts' IMPORT STATEMENTS are restated here."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# scripts/run.py:13-17, train.py:8-11, register.py:10-12, pairwise_register_eval.py:6-9, groupwise_register_eval.py:8-10,
# pretrain.py:8-10, hyperparameters.py:2 -- verbatim import lines
SCRIPT_IMPORTS = """
from keymorph.unet3d.model import UNet2D, UNet3D, TruncatedUNet3D
from keymorph.net import ConvNet
from keymorph.model import KeyMorph
from keymorph import utils as keymorph_utils
from keymorph.viz_tools import imshow_img_and_points_3d
from keymorph.utils import align_img, one_hot, one_hot_subsampled_pair
from keymorph.viz_tools import imshow_registration_2d, imshow_registration_3d
from keymorph.augmentation import random_affine_augment
import keymorph.loss_ops as loss_ops
from keymorph.utils import align_img, one_hot
from keymorph.augmentation import affine_augment
from keymorph.utils import convert_points_norm2real, convert_points_real2norm
from keymorph.utils import rescale_intensity
"""

CHECKS = """
import keymorph_amd.model, keymorph_amd.loss_ops, keymorph_amd.utils
assert KeyMorph is keymorph_amd.model.KeyMorph and loss_ops is keymorph_amd.loss_ops and keymorph_utils is keymorph_amd.utils
assert TruncatedUNet3D.__module__ == "keymorph_amd.unet3d.model" and ConvNet.__module__ == "keymorph_amd.net"
assert imshow_registration_3d.__module__ == "keymorph.viz_tools"          # NOT shadowed: the maintainer's own module
for ctor in (UNet2D,):
    try:
        ctor()
    except NotImplementedError as e:
        assert "unet3d/model.py:266" in str(e)
    else:
        raise AssertionError("UNet2D() must raise")
try:
    loss_ops.hausdorff_distance(None, None)
except NotImplementedError:
    pass
else:
    raise AssertionError("hausdorff_distance must raise")
# the classes the scripts go on to build (scripts/run.py:339-404), constructed on the CPU (no kernel is launched)
net = TruncatedUNet3D(1, 8, 1, final_sigmoid=False, f_maps=8, layer_order="gcr", num_groups=8, num_levels=3,
                      is_segmentation=False, conv_padding=1)
km = KeyMorph(net, 8, 3, use_amp=False, use_checkpoint=False, max_train_keypoints=None, weight_keypoints=None,
              align_keypoints_in_real_world_coords=False)
assert hasattr(km, "backbone") and len(list(km.parameters())) > 0
print("DROPIN_OK")
"""


def _run(code, extra_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=os.pathsep.join([ROOT] + extra_path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stdout + r.stderr


def test_install_with_a_standin_checkout(tmp_path):
    """The maintainer's checkout is represented by a stand-in package holding ONLY a viz_tools module (what the scripts import
    and this package does not provide); every other keymorph.* module is absent from it, so an import that succeeds for
    them can only have come from the aliases."""
    pkg = tmp_path / "keymorph"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "viz_tools.py").write_text(textwrap.dedent("""
        def imshow_img_and_points_3d(*a, **k): pass
        def imshow_registration_2d(*a, **k): pass
        def imshow_registration_3d(*a, **k): pass
    """))
    _run("import keymorph_amd.dropin as d\nd.install()\n" + SCRIPT_IMPORTS + CHECKS, [str(tmp_path)])


def test_install_without_any_checkout():
    """No `keymorph` on the path at all: the implemented modules import, viz_tools is a plain ModuleNotFoundError (nothing
    of the maintainer's is shadowed or faked)."""
    code = ("import keymorph_amd.dropin as d\np = d.install()\nassert p.__path__ == []\n"
            "from keymorph.model import KeyMorph\nfrom keymorph.unet3d.model import UNet2D, UNet3D, TruncatedUNet3D\n"
            "import keymorph.loss_ops as loss_ops\n"
            "try:\n    import keymorph.viz_tools\nexcept ModuleNotFoundError:\n    print('DROPIN_OK')\n")
    _run(code, [])


def test_install_refuses_after_the_reference_was_imported(tmp_path):
    pkg = tmp_path / "keymorph"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    code = ("import keymorph\nimport keymorph_amd.dropin as d\n"
            "try:\n    d.install()\nexcept RuntimeError as e:\n    assert 'before' in str(e)\n    print('DROPIN_OK')\n")
    _run(code, [str(tmp_path)])


@pytest.mark.skipif(not os.path.isdir("/root/reference/keymorph"), reason="the reference checkout exists only in the build container")
def test_install_over_the_real_reference_checkout():
    """In the build container: the REAL package is the parent.  Its __init__ (`from . import model`, ...) picks up the
    aliases, its viz_tools is its own file, and none of the third-party imports of the replaced modules (nibabel, skimage,
    h5py) is needed any more."""
    _run("import keymorph_amd.dropin as d\np = d.install()\nassert p.__file__.startswith('/root/reference')\n"
         + SCRIPT_IMPORTS + CHECKS + "import keymorph\nassert keymorph.__version__ == '2.0.1'\n", ["/root/reference"])
