"""The video DRIVER against the reference's own loop, on the CPU: ``vggsfm_amd.video.VideoGeometry.run`` (window bounds,
the shrink / step-back rule of video_runner.py:712-751, carried-over point selection, table updates, joint BA read-back)
with the oracle's arithmetic under it (tests/cpu_backend.py) must reproduce every snapshot of
tests/golden/video_radial_t60.npz -- written by the reference's UNMODIFIED ``VideoRunner.run`` in the build container
(oracle/gen_golden_video.py).  The -m gpu twin (tests/test_gpu_video_golden.py) runs the device kernels."""
from tests import cpu_backend
from tests.video_golden_driver import run_against_golden


def test_video_geometry_reproduces_the_reference_loop_cpu(monkeypatch):
    cpu_backend.patch_video_geometry(monkeypatch)
    run_against_golden("radial_t60", "cpu")
