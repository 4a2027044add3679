"""The video DRIVER against the reference's own loop, on the CPU: ``vggsfm_amd.video.VideoGeometry.run`` (window bounds,
the shrink / step-back rule of video_runner.py:712-751, carried-over point selection, table updates, joint BA read-back)
with the oracle's arithmetic under it (tests/cpu_backend.py) must reproduce every snapshot of
tests/golden/video_radial_t60.npz -- written by the reference's UNMODIFIED ``VideoRunner.run`` in the build container
(oracle/gen_golden_video.py).  The -m gpu twin (tests/test_gpu_video_golden.py) runs the device kernels."""
import pytest

from tests import cpu_backend
from tests.video_golden_driver import run_against_golden


# radial_t60: 60 frames, init 16 / window 8 / joint BA every 3 windows, SIMPLE_RADIAL, one shrink, one step-back.
# pinhole_t160 (round 6): the reference's own defaults (cfgs/video_demo.yaml:8-13: init 32 / window 16 / joint BA every 6
# windows / 1024 query points) on its other camera branch (SIMPLE_PINHOLE shared, video_runner.py:1019-1039), 160 frames, a
# shrink 16 -> 8 and a step-back + retry; 13 snapshots, 201 k observations at the end.
@pytest.mark.parametrize("case", ["radial_t60", "pinhole_t160"])
def test_video_geometry_reproduces_the_reference_loop_cpu(case, monkeypatch):
    cpu_backend.patch_video_geometry(monkeypatch)
    run_against_golden(case, "cpu")
