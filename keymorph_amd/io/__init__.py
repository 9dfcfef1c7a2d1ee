"""On-disk formats and loader shims either side of the registration path: host-side
only -- no GPU code and no third-party readers (torchio / nibabel are not needed)."""
from .checkpoint import load_checkpoint, save_checkpoint          # noqa: F401
from .nifti import read_nifti, write_nifti                        # noqa: F401
from .pairs import DATA, AFFINE, PairLoader, make_subject         # noqa: F401
from .groupwise import evaluate_group, save_dict_as_json         # noqa: F401
