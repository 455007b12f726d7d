"""
TEST INFRASTRUCTURE ONLY.  Captures golden vectors from the reference itself
(`/root/reference/fastbox/box.py`, loaded by oracle/ref_loader.py with the
stand-in cosmology provider) into tests/golden/*.npz.

Run in the build container only (the reference does not exist on the GPU box):

    python -m oracle.make_golden

Fixtures hold inputs-by-seed and expected outputs, never reference source.
The legacy numpy stream (np.random.seed + np.random.normal) is frozen by numpy
policy, so the noise is stored as a seed, not as cubes.  For N >= 32 the 3-D
fields are stored as a strided sub-lattice [::s, ::s, ::s] plus full-array sums.
numpy/scipy versions are recorded in each file.
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import standin                      # noqa: E402
from oracle.ref_loader import load_reference_box  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, nsamp, box_scale, redshift, seed, probe stride, what)
CASES = [
    ("n16_cube", 16, (1e2, 1e2, 1e2), 0.0, 11, 1, "all"),
    ("n16_cuboid", 16, (1e2, 2e2, 1e3), 1.0, 11, 1, "all"),
    ("n32_l1000", 32, 1e3, 0.0, 14, 2, "all"),
    ("n64_l1000", 64, (1e3, 1e3, 1e3), 0.0, 14, 4, "all"),
    ("n48_l1000", 48, 1e3, 0.0, 14, 4, "all"),          # a grid that is not a power of two (numpy's FFT takes any nsamp)
    ("n64_l100", 64, (1e2, 1e2, 1e2), 0.0, 11, 4, "pk"),
    ("n64_l4000", 64, 4e3, 0.8, 10, 4, "pk"),
    ("n128_l1000", 128, 1e3, 0.0, 14, 8, "pk"),
    ("n256_l1000", 256, 1e3, 0.0, 14, 16, "all"),       # BASELINE configs[2]'s chain at the largest size that is cheap here
    # BASELINE.json configs[1] (the size the metric is quoted on): ~25 GB and ~10 min of host work
    ("n512_l1000", 512, 1e3, 0.0, 14, 32, "pkln"),
]


ONLY = set(sys.argv[1:])          # python -m oracle.make_golden [case ...]: default all


def probe(a, s):
    return np.ascontiguousarray(a[::s, ::s, ::s])


def capture(ref, name, N, box_scale, redshift, seed, s, what):
    np.random.seed(seed)
    box = ref.CosmoBox(cosmo=standin.DEFAULT_COSMO, box_scale=box_scale, nsamp=N, redshift=redshift,
                       realise_now=False)
    import warnings
    warnings.simplefilter("ignore")
    out = dict(numpy_version=np.__version__, scipy_version=scipy.__version__,
               N=N, box_scale=np.atleast_1d(np.asarray(box_scale, dtype=np.float64)), redshift=redshift,
               seed=seed, stride=s,
               Lx=box.Lx, Ly=box.Ly, Lz=box.Lz, boxfactor=box.boxfactor, kmin=box.kmin, kmax=box.kmax,
               x=box.x, z=box.z)
    box.realise_density()
    out["delta_x"] = probe(box.delta_x, s)
    out["delta_x_sum"] = np.sum(box.delta_x)
    out["delta_x_sumsq"] = np.sum(box.delta_x ** 2.)
    out["delta_k"] = probe(box.delta_k, s)
    for nb in (20, 50):
        kc, pk, err = box.binned_power_spectrum(nbins=nb)
        out["pk%d_k" % nb], out["pk%d_p" % nb], out["pk%d_e" % nb] = kc, pk, err
    kb = np.linspace(0.5 * box.kmin, 0.4 * box.kmax, 12)
    out["kbins"] = kb
    out["pkkb_k"], out["pkkb_p"], out["pkkb_e"] = box.binned_power_spectrum(kbins=kb)
    s1, s2 = box.test_parseval()
    out["parseval"] = np.array([s1, s2])
    # sigma_R from the realisation's binned spectrum, the top-hat windows, the theory curve, and the numbers
    # test_sampling_error() prints (box.py:595-694, 770-782, 871-928)
    out["sigma8"] = box.sigma8()
    out["sigmaR20"] = box.sigmaR(20.)
    kw = np.logspace(-2.5, 0.5, 32)
    out["window_k"] = kw
    out["window8"] = box.window(kw, 8.0 / standin.DEFAULT_COSMO['h'])
    out["window1_8"] = box.window1(kw, 8.0 / standin.DEFAULT_COSMO['h'])
    tk, tp = box.theoretical_power_spectrum()
    out["theory_k"], out["theory_pk"] = tk[::25], tp[::25]
    if N <= 256:
        import contextlib
        import io
        import re
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            box.test_sampling_error()
        nums = [float(x) for x in re.findall(r"(?:\t|= )\s*([-+0-9.eE]+|nan)\s*$", buf.getvalue(), flags=re.M)]
        assert len(nums) == 9, buf.getvalue()
        # s8_real, s8_th_win, s8_th_full, s8_realspace, ratio, s20_real, s20_realspace, ratio, std(delta)
        out["sampling_report"] = np.array(nums)
    out["freq_array"] = box.freq_array()
    ax, ay = box.pixel_array(redshift=max(redshift, 0.5))
    out["pixel_x"], out["pixel_y"] = ax, ay
    if what == "pkln":
        ln = box.lognormal(box.delta_x)
        out["lognormal"] = probe(ln, s)
        out["lognormal_mean_exp"] = np.mean(np.exp(box.delta_x))
        out["pkln_k"], out["pkln_p"], out["pkln_e"] = box.binned_power_spectrum(delta_x=ln)
    if what == "all":
        ln = box.lognormal(box.delta_x)
        out["lognormal"] = probe(ln, s)
        out["lognormal_mean_exp"] = np.mean(np.exp(box.delta_x))
        out["pkln_k"], out["pkln_p"], out["pkln_e"] = box.binned_power_spectrum(delta_x=ln)
        out["tf_beam"] = probe(box.apply_transfer_fn(box.delta_k, standin.beam_highpass), s)
        out["tf_hp3"] = probe(box.apply_transfer_fn(box.delta_k, standin.highpass3), s)
        out["tf_wedge"] = probe(box.apply_transfer_fn(box.delta_k, standin.wedge03), s)
        out["smooth8"] = probe(box.smooth_field(box.delta_k, 8.0), s)
        vel = box.realise_velocity()
        for c in range(3):
            out["vel%d_k" % c] = probe(vel[c], s)
        vz = np.fft.ifftn(vel[2]).real
        out["vel_z"] = probe(vz, s)
        out["phi_k"] = probe(box.realise_potential(), s)
        out["rsd0"] = probe(box.redshift_space_density(delta_x=box.delta_x, velocity_z=vz, sigma_nl=0.), s)
        # stream position: 2 N^3 normals consumed so far; the next N^3 are the LOS noise
        out["rsd200"] = probe(box.redshift_space_density(delta_x=box.delta_x, velocity_z=vz, sigma_nl=200.), s)
        rs = box.redshift_space_density(delta_x=box.delta_x, velocity_z=vz, sigma_nl=0.)
        out["pkrsd_k"], out["pkrsd_p"], out["pkrsd_e"] = box.binned_power_spectrum(delta_x=rs)
        # BASELINE configs[2]: wedge-filtered redshift-space field and its P(k)
        fw = box.apply_transfer_fn(np.fft.fftn(rs), standin.wedge03)
        out["rsd_wedge"] = probe(fw, s)
        out["pkrsdw_k"], out["pkrsdw_p"], out["pkrsdw_e"] = box.binned_power_spectrum(delta_x=fw.real)
        # the other regridding rule box.py:433-437 accepts (sigma_nl = 0: no draw, the stream stays where it is)
        out["rsd0_nearest"] = probe(box.redshift_space_density(delta_x=box.delta_x, velocity_z=vz, sigma_nl=0.,
                                                               method='nearest'), s)
        if N <= 64:      # griddata's third one-dimensional rule (round 4): N^2 spline fits by the reference, small cases only
            out["rsd0_cubic"] = probe(box.redshift_space_density(delta_x=box.delta_x, velocity_z=vz, sigma_nl=0.,
                                                                 method='cubic'), s)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "(%d arrays)" % len(out))


def main():
    ref = load_reference_box()
    os.makedirs(OUT, exist_ok=True)
    for case in CASES:
        if not ONLY or case[0] in ONLY:
            capture(ref, *case)


if __name__ == "__main__":
    main()
