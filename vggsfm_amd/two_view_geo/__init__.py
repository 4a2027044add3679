"""Two-view stage in front of the hot path (vggsfm/two_view_geo/): fundamental matrices for all (query frame, other
frame) pairs at once on the device.  SURVEY.md section 8(f).3; PARITY UNPINNED against the reference."""
from .estimate_preliminary import estimate_preliminary_cameras  # noqa: F401
from .fundamental import estimate_fundamental  # noqa: F401
from .utils import (calculate_residual_indicator, generate_samples, inlier_by_fundamental,  # noqa: F401
                    sampson_epipolar_distance_batched)
