"""GPU: a consumer of libfastbox_hip.so written in plain C (examples/slab_box_c_abi.c: no Python, no PyTorch) runs
realise_density -> binned_power_spectrum of a slab-decomposed box through the C ABI alone -- host geometry, shell
amplitudes and bin thresholds formed in C with the reference's expressions, the library's own communicator for the
all-to-alls and the all-reduce -- and gets the numbers fastbox_amd.SlabBox gets for the same seed and spectrum."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pk(k):
    with np.errstate(all="ignore"):
        x = np.asarray(k, dtype=np.float64) / 0.02
        p = 2.0e4 * x ** 0.96 / (1. + x * x) ** 1.8
        return np.where(np.asarray(k) > 0., p, np.nan)


@pytest.mark.parametrize("N,nb", [(64, 20), (256, 12)])
def test_c_consumer_matches_the_python_slab_box(tmp_path, N, nb):
    exe = str(tmp_path / "slab_box")
    lib = os.path.join(ROOT, "fastbox_amd", "lib")
    r = subprocess.run(["gcc", "-O2", "-std=c99", "-D_DEFAULT_SOURCE", os.path.join(ROOT, "examples", "slab_box_c_abi.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lfastbox_hip", "-lm", "-Wl,-rpath," + lib,
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L, seed = 1000., 5
    r = subprocess.run([exe, str(N), repr(L), str(nb), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    rows = np.array([[float(x) for x in ln.split()] for ln in r.stdout.strip().splitlines()])
    assert rows.shape == (nb - 1, 3)
    from fastbox_amd import default_cosmo
    from fastbox_amd.distributed import SlabBox
    box = SlabBox(default_cosmo, box_scale=L, nsamp=N, precision="f32", seed=seed, rank=0, world=1, device=0, pk_fn=_pk)
    kc, pk, err = box.realise_and_power(nbins=nb, lognormal=False)
    assert np.allclose(rows[:, 0], kc, rtol=1e-14, atol=0)                  # centres: libm's pow against numpy's own (last bit)
    assert np.array_equal(np.isnan(rows[:, 1]), np.isnan(pk))
    m = ~np.isnan(pk)
    assert np.allclose(rows[m, 1], pk[m], rtol=1e-12, atol=0), np.max(np.abs(rows[m, 1] / pk[m] - 1))
    assert np.array_equal(rows[:, 2], box.ops.bin_counts()[1:])
