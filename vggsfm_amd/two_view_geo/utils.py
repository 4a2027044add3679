"""Host helpers of the two-view stage (vggsfm/two_view_geo/utils.py)."""
import numpy as np


def generate_samples(N, target_num, sample_num, expand_ratio=2):
    """vggsfm/two_view_geo/utils.py:39-60: `target_num` index tuples of `sample_num` distinct match indices in [0, N),
    drawn like the reference does (numpy's global RNG: target_num * expand_ratio tuples, the ones with a repeated
    index dropped, the first target_num kept -- fewer when too many are dropped)."""
    draw = np.random.randint(0, N, size=(target_num * expand_ratio, sample_num))
    srt = np.sort(draw, axis=1)
    distinct = (srt[:, 1:] != srt[:, :-1]).all(axis=1)
    return draw[distinct][:target_num]
