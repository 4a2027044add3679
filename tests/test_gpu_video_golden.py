"""``vggsfm_amd.video.VideoGeometry.run`` on the device against the snapshots the reference's UNMODIFIED
``VideoRunner.run`` loop left behind (vggsfm/runners/video_runner.py:156-191, 640-905 incl. the shrink and step-back
branches :712-751; oracle/gen_golden_video.py, tests/golden/video_radial_t60.npz): the same call sequence, window bounds
and success flags; observation tables bit for bit after every call; poses and points within 1e-4."""
import pytest

from tests.video_golden_driver import run_against_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["radial_t60", "pinhole_t160"])
def test_video_geometry_reproduces_the_reference_loop(case):
    run_against_golden(case, "cuda")
