"""End-to-end geometry stage at BASELINE configs[1] / configs[2]: vggsfm_amd.models.Triangulator.forward (triangulation + pose
refinement + iterative bundle adjustment) on synthetic tracks -- wall time and stage breakdown.  The measurement itself lives in
bench.py (`pipeline_leg`, the `pipeline` key of the bench line); this prints it alone.
usage: python scripts/prof/bench_pipeline.py [c2] [c3]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

names = tuple(a for a in sys.argv[1:] if a in ("c2", "c3")) or ("c2", "c3")
print(json.dumps(bench.pipeline_leg(torch.device("cuda", 0), names), indent=1))
