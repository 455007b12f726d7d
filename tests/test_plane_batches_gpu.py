"""The y/z passes of every transform run x-plane batch by x-plane batch (fb_fft_launch.inc yz_passes).  Batching
only regroups the same per-line transforms, so every result must be bit-identical whatever the batch size, ragged
last batch and second stream included.  One worker process per setting (the environment is read once)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(N, precision, batch, streams):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("FB_PLANE_BATCH", None)
    env.pop("FB_PLANE_STREAMS", None)
    if batch is not None:
        env["FB_PLANE_BATCH"] = str(batch)
    if streams is not None:
        env["FB_PLANE_STREAMS"] = str(streams)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_plane_batch_worker.py"), str(N), precision],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln.split() for ln in out.stdout.splitlines() if len(ln.split()) == 2]
    assert len(lines) == 11, out.stdout
    return dict(lines)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_batched_passes_are_bit_identical(precision):
    whole = _run(128, precision, 0, None)            # one batch = the path without batching
    for batch, streams in ((24, 1), (24, 2), (1, 2), (127, 1)):     # ragged last batch; 128 batches; 127 + 1
        got = _run(128, precision, batch, streams)
        assert got == whole, (batch, streams, [k for k in whole if got.get(k) != whole[k]])


def test_default_batching_at_the_headline_size():
    """512^3 is batched by default (4 x 128 planes): same bits as the whole box in one go."""
    assert _run(512, "f32", None, None) == _run(512, "f32", 0, None)
