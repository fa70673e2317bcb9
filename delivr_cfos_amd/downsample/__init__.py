from .downsample_and_mask import (downsample_mask, downsample_volume, get_real_size, mask_and_pad,  # noqa: F401
                                  upsample_mask, write_masked_nifti_npy)
