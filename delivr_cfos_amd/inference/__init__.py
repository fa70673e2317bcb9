from .inference import create_nifti_seg, run_inference  # noqa: F401
from .sliding_window_inferer import SlidingWindowInferer, sliding_window_inference  # noqa: F401
