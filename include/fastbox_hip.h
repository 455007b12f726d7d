/*
 * libfastbox_hip -- C ABI of the MI355X (gfx950) density-field hot path.
 *
 * The reference (philbull/FastBox, fastbox/box.py) has no FFI layer: its hot
 * path is numpy calls inside the methods of `CosmoBox`.  This header is the
 * boundary a ctypes/cffi binding of those methods binds instead; every entry
 * point names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a
 *     negative FB_ERR_* code, and fb_last_error() (thread local) says why;
 *   - all field pointers are DEVICE pointers owned by the caller (hipMalloc,
 *     a torch tensor's data_ptr(), ...).  `stream` is a hipStream_t (NULL =
 *     default stream); calls are asynchronous unless stated otherwise;
 *   - precision is fixed per plan: 4 = float / complex64, 8 = double /
 *     complex128 (the reference is float64 throughout);
 *   - layouts are C order with z fastest, like the reference's ndarrays:
 *       real : T          [N][N][N]
 *       full : complex<T> [N][N][N]
 *       half : complex<T> [N][rows][pitch], k_z = 0..N/2 stored, pitch = fb_half_pitch(),
 *              rows = fb_half_rows() = N + 1 of which the first N are used (the spare row keeps
 *              the x stride off a multiple of 64 KiB; size the buffer with fb_half_bytes())
 *     A half spectrum represents the Hermitian array fftn(real field).
 *   - one plan per host thread at a time (thread compatible, not thread safe).
 */
#ifndef FASTBOX_HIP_H
#define FASTBOX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FB_OK 0
#define FB_ERR_INVALID -1      /* bad argument                        */
#define FB_ERR_HIP -2          /* HIP runtime / launch failure        */
#define FB_ERR_UNSUPPORTED -3  /* grid size not a power of two 16..2048 */
#define FB_ERR_NOMEM -4
#define FB_ERR_STATE -5        /* tables not set before use           */
#define FB_ERR_RCCL -6         /* RCCL missing or a collective failed (fb_comm_*, fb_slab_exchange*, fb_allreduce_f64) */

typedef struct fb_plan fb_plan;

int fb_version(void);
const char* fb_last_error(void);

/* ---- plan: geometry of one CosmoBox (box.py:25-127) ---------------------------------- */
/* axis2[3*N]: (m_i/L_a)**2; ksc[3*N]: m_i*(2 pi/L_a); kpar[N]: 2 pi m_i/Lz; zgrid[N]: self.z.
 * They are computed by the host binding with the reference's numpy expressions
 * (box.py:119-127, 254-256, 375, 79-88) so that |k| is bit-identical.            */
/* N: a power of two in 16 .. 2048 (the tuned kernels, every entry point), or any other EVEN size with prime factors 2, 3, 5 in
 * 16 .. 1024 (round 4): there fb_fft_c2c / _r2c / _c2r run plain Stockham passes (radices 4, 2, 3, 5) and every kernel that is not
 * an FFT pass works as usual (the transverse and 2-D transforms behind fb_fft_transverse, fb_beam_convolve, fb_sky_realise_map take
 * the same plain passes), while the entry points that fuse something into an FFT pass (fb_realise_density_*,
 * fb_power_spectrum_*, fb_slab_*) return FB_ERR_UNSUPPORTED: compose
 * fb_colour_device / fb_colour_noise + fb_fft_c2r, fb_fft_r2c + fb_bin_power, fb_lognormal ... instead, as
 * fastbox_amd.CosmoBox does on such a grid.  Anything else (odd, other prime factors): FB_ERR_UNSUPPORTED. */
int fb_plan_create(fb_plan** plan, int N, double Lx, double Ly, double Lz, int precision, int device,
                   const double* axis2, const double* ksc, const double* kpar, const double* zgrid);
int fb_plan_destroy(fb_plan* plan);
int fb_half_pitch(const fb_plan* plan);          /* complex elements per (x,y) row of a half spectrum */
int fb_half_rows(const fb_plan* plan);           /* stored rows per x-plane of a half spectrum (N + 1)   */
int64_t fb_real_bytes(const fb_plan* plan);
int64_t fb_half_bytes(const fb_plan* plan);
int64_t fb_full_bytes(const fb_plan* plan);

/* ---- 3-D FFTs (numpy.fft.fftn / ifftn, box.py:187,193,246,337,380,654,736) ------------- */
/* in place on a full complex array; direction -1 = fftn, +1 = ifftn (scale applied: pass 1/N^3) */
int fb_fft_c2c(fb_plan* plan, void* full_inout, int direction, double scale, void* stream);
/* fftn of a real field -> half spectrum.  pre_exp != 0 transforms exp(field) (log-normal fusion) */
int fb_fft_r2c(fb_plan* plan, const void* real_in, void* half_out, int pre_exp, void* stream);
/* Re ifftn of the Hermitian array a half spectrum stands for; `half_inout` is destroyed */
int fb_fft_c2r(fb_plan* plan, void* half_inout, void* real_out, double scale, void* stream);

/* ---- Gaussian realisation (realise_density, box.py:161-187) --------------------------- */
/* sqrt(nan_to_num(P(k)) * boxfactor): per integer shell n^2=i^2+j^2+l^2 (cubic boxes) ...   */
int fb_set_amplitude_shells(fb_plan* plan, const double* amp, int64_t nshell);
/* ... or, for any box shape, per (|m_x|, |m_y|, |m_z|): HOST double[M][M][M], M = N/2 + 1 (|k| depends on the mode
 * numbers only through their magnitudes: an eighth of the modes to evaluate P(k) on) ...                     */
int fb_set_amplitude_sym(fb_plan* plan, const double* amp, int64_t n);
/* ... or per stored mode, a DEVICE array T[N][rows][pitch] (any box shape)                 */
int fb_set_amplitude_dense(fb_plan* plan, const void* amp_dev);
/* parity mode: re, im are the reference's np.random.normal draws, T[N][N][N] on the device  */
int fb_colour_noise(fb_plan* plan, const void* re, const void* im, void* half_out, void* stream);
/* throughput mode: Philox4x32-10, key = seed, counter = (mode index, stream, realisation); fb_rng.h */
int fb_colour_device(fb_plan* plan, uint64_t seed, uint64_t realisation, void* half_out, void* stream);

/* fused throughput path: generator inside the first inverse pass, then y and z (c2r) passes.
 * real_out = Re ifftn(X)/1 with numpy's 1/N^3; work_half is scratch (half-spectrum sized).  */
int fb_realise_density_device(fb_plan* plan, uint64_t seed, uint64_t realisation, void* work_half,
                              void* real_out, void* stream);

/* velocity component `comp` (0 x, 1 y, 2 z) of the SAME realisation in real space:
 * real_out = Re ifftn(i fac delta_k k_c / k^2) (realise_velocity, box.py:251-284, then the caller's
 * np.fft.ifftn(velocity_k[c]).real, e.g. examples/example_redshift_space.py:17).  The counter-based
 * generator reproduces delta_k inside the first pass; no spectrum is stored.  work_half is scratch. */
int fb_realise_velocity_device(fb_plan* plan, uint64_t seed, uint64_t realisation, int comp, double fac,
                               void* work_half, void* real_out, void* stream);

/* deferred form of the above: _begin runs the generator + x pass into `pending_half` (half-spectrum
 * sized); the y and z passes are done later by _finish (real_out = delta_x), or by
 * fb_power_spectrum_pending, which fuses the z pass with the first pass of the power spectrum:
 * real_out is written (delta_x is still produced) but never read back.  The y and z passes of every
 * transform run x-plane batch by x-plane batch, so that a batch stays in the Infinity Cache between
 * them (env FB_PLANE_BATCH = planes per batch, 0 = off; FB_PLANE_STREAMS = 1 | 2: with 2, the default
 * from 1024^3 up, alternate batches run on a plan-owned auxiliary stream that is joined back into
 * `stream` before the call returns).                                                            */
int fb_realise_density_begin(fb_plan* plan, uint64_t seed, uint64_t realisation, void* pending_half, void* stream);
int fb_realise_density_finish(fb_plan* plan, void* pending_half, void* real_out, void* stream);

/* ---- binned power spectrum (binned_power_spectrum, box.py:741-764) ------------------------- */
/* edges[nbins] as in np.digitize(k, edges).  thr/amb describe, for cubic boxes, the bin as a
 * step function of the integer shell (thr[b] = first n^2 with |k| >= edges[b]; amb = shells
 * to decide with the exact fp64 |k|); pass thr = NULL to always use the exact expression.   */
int fb_set_bins(fb_plan* plan, const double* edges, int nbins, const int32_t* thr, const int32_t* amb, int namb);
/* spec: layout 1 = half spectrum (mirrored modes counted twice), 0 = full complex array.
 * host outputs: count[nbins] (modes of the full grid per bin), sum[nbins] = sum |dk|^2,
 * sumsq[nbins] = sum |dk|^4.  Synchronises the stream.                                       */
int fb_bin_power(fb_plan* plan, const void* spec, int layout, double* count, double* sum, double* sumsq,
                 void* stream);
/* the same sums of spec * T(k_perp, k_par) (filter kinds and params as fb_apply_filter): the power
 * spectrum of apply_transfer_fn's result (box.py:356-381 then :741-764) for a real, k_par-even filter,
 * without storing the filtered spectrum or transforming back and forth.                      */
int fb_bin_power_filtered(fb_plan* plan, const void* spec, int layout, int kind, const double* params,
                          const void* table_dev, double* count, double* sum, double* sumsq, void* stream);

/* fused "filter + estimate" for cubic boxes (needs fb_set_bins with thr): the power spectrum of
 * ifftn(fftn(real_in) * T(k_perp, k_par)) for a real, k_par-even filter -- apply_transfer_fn followed by
 * binned_power_spectrum (box.py:356-381, :696-768).  r2c and y pass, then the x pass multiplies by T,
 * writes the FILTERED spectrum to filtered_half (fb_fft_c2r of it delivers the filtered field) and bins
 * it.  kind: FB_FILT_TABLE, FB_FILT_BEAM_HIGHPASS or FB_FILT_WEDGE (FB_FILT_TOPHAT, smooth_field's window, is
 * applied with fb_apply_filter: FB_ERR_UNSUPPORTED here).  Asynchronous; results_dev as for fb_power_spectrum_device. */
int fb_power_spectrum_filtered(fb_plan* plan, const void* real_in, void* filtered_half, int kind,
                               const double* params, const void* table_dev, void* results_dev, void* stream);
/* the same when what the caller goes on to read is the filtered FIELD (apply_transfer_fn's return value,
 * box.py:381): the x pass that filters and bins a line also takes the line's inverse transform, so work_half
 * receives the filtered spectrum already transformed back along x and fb_fft_c2r_yz (the y and z passes of
 * fb_fft_c2r; work_half is destroyed) delivers the field -- one read and one write of the spectrum fewer.   */
int fb_power_spectrum_filtered_field(fb_plan* plan, const void* real_in, void* work_half, int kind,
                                     const double* params, const void* table_dev, void* results_dev, void* stream);
int fb_fft_c2r_yz(fb_plan* plan, void* half_x_done, void* real_out, double scale, void* stream);

/* BASELINE configs[2] as one chain (examples/example_redshift_space.py: realise_density, realise_velocity, box.py:384-438
 * redshift_space_density, then np.fft.fftn of the result and box.py:356-381 / :696-768 on it), for a device-generator
 * realisation.  fb_realise_velocity_begin: generator + first inverse pass of velocity component comp (fac = box.py:280-281),
 * deferred like fb_realise_density_begin (fb_realise_density_finish delivers the component in real space).
 * fb_power_spectrum_redshift_space: pending_delta / pending_vz (both destroyed) -> delta_x_out (the density in real space;
 * NULL: not written), the P(k) sums of the redshift-space field in results_dev (as fb_power_spectrum_device), and, when
 * filter_kind >= 0 (FB_FILT_TABLE, _BEAM_HIGHPASS, _WEDGE as fb_power_spectrum_filtered; -1: no filter, work_half is
 * scratch), the filtered spectrum in work_half -- want_field: already transformed back along x, for fb_fft_c2r_yz.
 * The two inverse z passes, the line-of-sight remap and the forward z pass are ONE kernel per plane batch: v_z in real
 * space and the redshift-space field itself never reach memory (4 half-sweeps of traffic instead of 9).  The numbers are
 * those of fb_realise_density_finish + fb_realise_velocity_device + fb_redshift_space + fb_power_spectrum_filtered[_field],
 * bit for bit.  Hz, sigma_nl, seed, method: as fb_redshift_space (sigma_nl > 0 draws the device noise of that call).
 * Single-precision plans with 64 <= N <= 512 (FB_ERR_UNSUPPORTED otherwise: use the separate calls).                  */
int fb_realise_velocity_begin(fb_plan* plan, uint64_t seed, uint64_t realisation, int comp, double fac,
                              void* pending_half, void* stream);
int fb_power_spectrum_redshift_space(fb_plan* plan, void* pending_delta, void* pending_vz, void* delta_x_out,
                                     void* work_half, double Hz, double sigma_nl, uint64_t seed, int method,
                                     int filter_kind, const double* params, const void* table_dev, int want_field,
                                     void* results_dev, void* stream);

/* fused path for cubic boxes (needs fb_set_bins with thr): r2c of real_in (of exp(real_in) when
 * pre_exp) with the binning inside the last pass.  Asynchronous: results_dev[2*nbins+1] (DEVICE)
 * receives (sum |dk|^2, sum |dk|^4) per bin, then sum(exp(real_in)) (0 unless pre_exp).
 * work_half is scratch and holds fftn(real_in) afterwards only if keep_spectrum.              */
int fb_power_spectrum_device(fb_plan* plan, const void* real_in, void* work_half, int pre_exp,
                             int keep_spectrum, double* results_dev, void* stream);
/* same results for a field still pending from fb_realise_density_begin (see there); consumes pending_half.
 * real_out = NULL: delta_x itself is not written (a Monte-Carlo loop that only wants the spectrum; the realisation can
 * be regenerated from (seed, realisation) at any time) -- a quarter of the fused z pass's traffic less.            */
int fb_power_spectrum_pending(fb_plan* plan, void* pending_half, void* real_out, int pre_exp, double* results_dev,
                              void* stream);
/* A Monte-Carlo loop in one call: for i < count, realisation (seed, first + i stride) is drawn (fb_realise_density_begin into
 * work_half) and its P(k) estimated (fb_power_spectrum_pending; real_out, if given, receives every field in turn and holds
 * the last one), results_dev[i * results_stride ...] = the 2 nbins + 1 sums of realisation i.  The same launches in the same
 * order as the two calls per realisation -- identical numbers -- without an interpreter between them: what a step of a box
 * below 256^3 is bound by.  Needs the amplitude table and fb_set_bins with thr, like the calls it stands for.               */
int fb_montecarlo_power(fb_plan* plan, uint64_t seed, uint64_t first, uint64_t stride, int count, void* work_half, void* real_out,
                        int pre_exp, double* results_dev, int64_t results_stride, void* stream);
/* number of full-grid modes per bin for the current bin set (host array, nbins doubles) */
int fb_bin_counts(fb_plan* plan, double* count);

/* ---- transfer functions (apply_transfer_fn box.py:374-379, smooth_field :651-653) ---------- */
#define FB_FILT_TABLE 0          /* table: real multiplier, same layout as the field */
#define FB_FILT_BEAM_HIGHPASS 1  /* (1-exp(-.5(|kpar|/p0)^p2)) [p0>0] * exp(-.5(kperp/p1)^2) [p1>0] */
#define FB_FILT_WEDGE 2          /* 0 where |kpar| < p0*kperp + p1, else 1 */
#define FB_FILT_TOPHAT 3         /* 3(sin x - x cos x)/x^3, x = |k| p0 */
/* out = nan_to_num(in * T(kperp, kpar)); layout 0 = full, 1 = half; in == out allowed */
int fb_apply_filter(fb_plan* plan, const void* in, void* out, int layout, int kind, const double* params,
                    const void* table_dev, void* stream);

/* ---- velocity / potential (box.py:251-284, 347-348) ------------------------------------------ */
int fb_velocity_k(fb_plan* plan, const void* dk, void* out, int layout, int component, double fac, void* stream);
int fb_potential_k(fb_plan* plan, const void* dk, void* out, int layout, void* stream);

/* ---- real-space operators ---------------------------------------------------------------------- */
/* out = exp(in)/mean(exp(in)) - 1 (box.py:457-460), formed as exp(in - max in)/mean(exp(in - max in)) - 1 so that
 * no intermediate leaves the plan's floating-point range; *mean_out receives mean(exp(in)). Synchronises. */
int fb_lognormal(fb_plan* plan, const void* real_in, void* real_out, double* mean_out, void* stream);
/* largest finite value of a real field (the exact shift of a fused log-normal transform, fb_set_exp_shift).
 * Synchronises. */
int fb_max_real(fb_plan* plan, const void* real, double* out, void* stream);
/* redshift_space_density (box.py:405-437). noise: T[N][N][N] standard normals in LOS order
 * (parity) or NULL -> Philox stream 1 of `seed` when sigma_nl > 0.  method: what box.py:433-437 hands to
 * scipy's griddata -- 'linear' (the default; points outside the shifted samples get the end-point average),
 * 'nearest' (no fill: scipy extrapolates with the end samples) or 'cubic' (the not-a-knot cubic spline through the
 * sorted shifted samples, the end-point average outside them; a line with two EQUAL shifted coordinates -- scipy raises
 * there -- comes back non-finite).                                                                          */
#define FB_RSD_LINEAR 0
#define FB_RSD_NEAREST 1
#define FB_RSD_CUBIC 2        /* griddata(method='cubic') in 1-D: the not-a-knot cubic spline through the sorted shifted samples */
int fb_redshift_space(fb_plan* plan, const void* delta, const void* vz, const void* noise, void* out,
                      double Hz, double sigma_nl, uint64_t seed, int method, void* stream);
/* sum(x) / sum(x^2) over a real field; sum |dk|^2 over the FULL grid from a half spectrum
 * (test_parseval, box.py:944-946).  Synchronise.                                                    */
int fb_sum_real(fb_plan* plan, const void* real, int squared, double* out, void* stream);
int fb_sumsq_half(fb_plan* plan, const void* half, double* out, void* stream);

/* ---- layout conversion -------------------------------------------------------------------------- */
int fb_expand_half(fb_plan* plan, const void* half, void* full, void* stream);  /* Hermitian extension */
int fb_crop_full(fb_plan* plan, const void* full, void* half, void* stream);    /* keep k_z <= N/2      */

/* ---- per-kernel timing with HIP events on the launch stream ---------------------------------------- */
#define FB_PROF_FFT_STRIDED 0   /* x / y passes of the 3-D FFT            */
#define FB_PROF_FFT_CONTIG 1    /* z pass (c2c, r2c, c2r)                 */
#define FB_PROF_COLOUR 2
#define FB_PROF_BIN 3
#define FB_PROF_FILTER 4
#define FB_PROF_VELPOT 5
#define FB_PROF_REALOP 6
#define FB_PROF_RSD 7
#define FB_PROF_LAYOUT 8
#define FB_PROF_FFT_GEN 9       /* x pass with the fused Gaussian generator */
#define FB_PROF_FFT_BIN 10      /* x pass with the fused shell binning      */
#define FB_PROF_PCA 11          /* channel covariance and projection of the PCA cleaning */
#define FB_PROF_NCAT 12
/* between start and stop every kernel launch of this plan is bracketed by an event pair;
 * stop synchronises and returns summed milliseconds and launch counts per class above.  */
int fb_profile_select(fb_plan* plan, unsigned mask);   /* bit i = bracket class i; default all */
/* bracket only every stride-th selected launch (an event pair costs ~3 us of stream time; 1 = all, the default);
 * stride = 0 leaves the stride as it is; *seen (may be NULL) receives the number of selected launches since the last
 * fb_profile_start */
int fb_profile_sample(fb_plan* plan, int stride, int64_t* seen);
int fb_profile_start(fb_plan* plan);
int fb_profile_stop(fb_plan* plan, void* stream, double* ms, int64_t* launches, int ncat);

/* element-wise arithmetic on real cubes T[N][N][N], the callers' own numpy expressions between the steps
 * (examples/example_endtoend.py:47, :75, :86): out = a x + b y + c (y may be NULL); out = x * y.  out may alias. */
int fb_real_axpby(fb_plan* plan, const void* x, const void* y, void* out, double a, double b, double c, void* stream);
int fb_real_multiply(fb_plan* plan, const void* x, const void* y, void* out, void* stream);

/* ---- per-channel 2-D operations on a complex cube complex<T>[N][N][N] (frequency = last axis), as
 * filters.angular_bandpass_filter (fastbox/filters.py:58-90) needs them ----
 * fb_fft_transverse: np.fft.fftn(cube, axes=[0,1]) (direction -1) / np.fft.ifftn(cube, axes=[0,1]) (+1), in place.
 * fb_mask_transverse: cube[kx, ky, :] *= mask2d[kx][ky] (T[N][N] on the device; :89).                             */
int fb_real_to_complex(fb_plan* plan, const void* real_cube, void* full_cube, void* stream);
int fb_fft_transverse(fb_plan* plan, void* full_cube, int direction, void* stream);
int fb_mask_transverse(fb_plan* plan, void* full_cube, const void* mask2d, void* stream);

/* ---- beam convolution, channel by channel (BeamModel.convolve_fft / convolve_real, fastbox/beams.py:63-137) ----
 * out[n][n][n] = (field (*) beam)[mode='same'] / sum_xy beam, for real cubes field, beam of T[n][n][n] (frequency =
 * last axis).  `plan` is the plan of the TRANSFORM size M = plan N: periodic = 0: M = 2 n, zero-padded linear
 * convolution = scipy.signal.fftconvolve(beam, field, mode='same', axes=[0, 1]) (:85-87); periodic = 1: M = n,
 * circular convolution = scipy.signal.convolve2d(beam[:, :, i], field[:, :, i], mode='same', boundary='wrap')
 * (:134-136).  work_a, work_b: distinct device buffers of complex<T>[M][M][n].  work_b holds the beam's transform
 * afterwards: beam_ready = 1 on a later call with the same plan, mode and work_b reuses it (beam is then ignored). */
int fb_beam_convolve(fb_plan* plan, const void* field, const void* beam, void* work_a, void* work_b, void* out,
                     int periodic, int beam_ready, void* stream);

/* ---- PCA foreground cleaning of a data cube T[N][N][N] (frequency = last axis), fastbox/filters.py:93-183 ----
 * mean_dev[N]: per-channel mean over the N^2 pixels (:142), fp64 on the DEVICE.                                   */
int fb_channel_means(fb_plan* plan, const void* cube, double* mean_dev, void* stream);
/* cov_dev[N][N] = np.cov of the mean-subtracted channels (:157-158; divisor N^2 - 1), fp64 on the DEVICE, formed on
 * the fp64 matrix cores (v_mfma_f64_16x16x4_f64) with a fixed summation order.  Its leading eigenvectors
 * (:161-169; an N x N problem): fb_leading_eigenvectors below, or the caller's own solver -- modes_dev[N][nmodes].   */
int fb_channel_covariance(fb_plan* plan, const void* cube, const double* mean_dev, double* cov_dev, void* stream);
/* The nmodes leading eigenpairs of cov_dev[N][N] (symmetric, fp64, DEVICE; not modified) -- np.linalg.eig + the sort by
 * decreasing eigenvalue of :161-169 -- by cyclic two-sided Jacobi in fp64 on the device (round-robin ordering, N/2
 * rotations per launch; 6-10 sweeps): modes_dev[N][nmodes] (orthonormal columns, the sign of a column is not defined,
 * as with LAPACK), vals_dev[nmodes] (descending; may be NULL), *sweeps_out (host; may be NULL).  Equal eigenvalues keep
 * the order of their diagonal positions.  Synchronises the stream (one convergence check per sweep).  FB_ERR_INVALID
 * for a matrix with non-finite entries, FB_ERR_STATE if 60 sweeps do not converge.  0 <= nmodes <= N.                 */
int fb_leading_eigenvectors(fb_plan* plan, const double* cov_dev, int nmodes, double* modes_dev, double* vals_dev,
                            int* sweeps_out, void* stream);
/* cube_out = cube - (U (U^T x) + mean), x = cube - mean (:172-176); amps_dev[nmodes][N^2] = U^T x (:172) or NULL   */
int fb_pca_clean(fb_plan* plan, const void* cube, const double* mean_dev, const double* modes_dev, int nmodes,
                 void* cube_out, double* amps_dev, void* stream);

/* ---- the steps after the density-field path: foregrounds (fastbox/foregrounds.py:48-175) and radiometer
 * noise (fastbox/noise.py:25-75).  2-D maps are T[N][N] over (x, y); cubes T[N][N][N], frequency axis last. ---- */
/* realise_foreground_amp (:99-107): map_out = Re ifft2((re + i im) amp2d) + monopole.  amp2d = sqrt(C_ell) per
 * 2-D mode with the k_perp = 0 entry 0 (:83-103, evaluated on the host); re, im = the reference's np.random.normal
 * draws as device maps, or both NULL for the counter generator (stream 2).  work_cplx: complex T[N][N] scratch. */
int fb_sky_realise_map(fb_plan* plan, const void* amp2d, const void* re, const void* im, uint64_t seed,
                       double monopole, void* work_cplx, void* map_out, void* stream);
/* realise_spectral_index (:137-138): map_out = mean + std n; n = `unit` (device map of N(0,1) draws) or generator */
int fb_sky_normal_map(fb_plan* plan, const void* unit, uint64_t seed, double mean, double std, void* map_out,
                      void* stream);
/* scipy.ndimage.gaussian_filter(map, sigma, mode='wrap') (:113, :144) with the 2*radius+1 HOST weights scipy
 * forms (exp(-x^2 / 2 sigma^2) normalised, radius = int(4 sigma + 0.5)); in place, tmp = T[N][N] scratch           */
int fb_sky_gaussian_filter(fb_plan* plan, void* map_inout, void* tmp, const double* weights, int radius, void* stream);
/* construct_cube (:165-175): cube = amps[x,y] * ratio[z] ** alpha[x,y] (alpha NULL: ** alpha_scalar);
 * ratio = freqs / freq_ref, HOST double[N]                                                                      */
int fb_sky_foreground_cube(fb_plan* plan, const void* amps, const void* alpha, double alpha_scalar,
                           const double* ratio, void* cube_out, void* stream);
/* realise_radiometer_noise (noise.py:72-74): cube = n[x,y,z] * sigma[z]; sigma = HOST double[N] (radiometer rms per
 * channel, :53-69 on the host); n = `unit` (device cube of the reference's draws) or the generator (stream 4)      */
int fb_sky_noise_cube(fb_plan* plan, const double* sigma, const void* unit, uint64_t seed, void* cube_out,
                      void* stream);

/* ---- slab-decomposed 3-D FFT for one box spread over `nparts` GPUs (one process per GPU) ---------
 * Rank `part` owns x-planes [part*N/nparts, ...) of real fields (T[N/nparts][N][N]) and, in k space,
 * k_y rows [part*N/nparts, ...) of every x-plane: kslab = complex<T>[N][N/nparts][pitch].
 * A forward transform is: fb_slab_forward_local (z r2c + y pass on the x-slab,
 * complex<T>[N/nparts][rows][pitch]) -> fb_slab_pack -> ONE all-to-all of equal blocks (done by the
 * caller: RCCL / torch.distributed) -> the receive buffer IS the kslab -> fb_slab_x_pass or the
 * fused fb_slab_x_bin.  The inverse mirrors it: fb_slab_x_generate (or x_pass) -> all-to-all of the
 * kslab's contiguous x-blocks -> fb_slab_unpack -> fb_slab_inverse_local (y pass + z c2r, 1/N^3).
 * The noise of fb_slab_x_generate depends on global mode indices only: the field is the same for
 * every nparts.  Cubic boxes only (shell amplitude / threshold tables).                          */
int64_t fb_slab_half_bytes(const fb_plan* plan, int nparts);     /* x-slab half spectrum           */
int64_t fb_slab_kspace_bytes(const fb_plan* plan, int nparts);   /* kslab = all-to-all buffer size  */
int fb_slab_forward_local(fb_plan* plan, const void* real_local, void* half_local, int nparts, int pre_exp,
                          double* expsum_dev, void* stream);
int fb_slab_inverse_local(fb_plan* plan, void* half_local, void* real_local, int nparts, void* stream);
/* the same two with fb_slab_pack / fb_slab_unpack folded into the y pass (it writes / reads the exchange buffer
 * directly); needs nparts to divide the points each thread holds of a y line: 1, 2, 4, 8 ranks (16 at N = 2048) */
int fb_slab_forward_packed(fb_plan* plan, const void* real_local, void* half_local, void* sendbuf, int nparts,
                           int pre_exp, double* expsum_dev, void* stream);
int fb_slab_inverse_packed(fb_plan* plan, const void* recvbuf, void* half_local, void* real_local, int nparts,
                           void* stream);
/* fb_slab_inverse_packed followed by fb_slab_forward_packed of the field just made, with the two z passes as one:
 * real_local (this rank's x-slab of delta_x) is written once and never read back.  recvbuf and sendbuf may be
 * the same buffer only if it is not half_local (the y passes are out of place).                                   */
int fb_slab_turnaround(fb_plan* plan, const void* recvbuf, void* half_local, void* real_local, void* sendbuf,
                       int nparts, int pre_exp, double* expsum_dev, void* stream);
int fb_slab_pack(fb_plan* plan, const void* half_local, void* sendbuf, int nparts, void* stream);
int fb_slab_unpack(fb_plan* plan, const void* recvbuf, void* half_local, int nparts, void* stream);
int fb_slab_x_pass(fb_plan* plan, void* kslab, int nparts, int direction, void* stream);
int fb_slab_x_generate(fb_plan* plan, void* kslab, int nparts, int part, uint64_t seed, uint64_t realisation,
                       void* stream);
/* results_dev[2*nbins]: this rank's (sum |dk|^2, sum |dk|^4) per bin; all-reduce (sum) over ranks */
int fb_slab_x_bin(fb_plan* plan, void* kslab, int nparts, int part, double* results_dev, void* stream);

/* ---- the same transform a range of k_z columns at a time, so that the all-to-all of one chunk of the half spectrum runs
 * beside the passes of the next (the x and y passes never mix k_z columns; only the z pass needs whole rows).
 * The strided passes work on tiles of `tile_columns` adjacent k_z columns; a row of the half spectrum has `tiles_per_row`
 * of them (fb_slab_tile_geometry); a chunk is the tile range [tile0, tile0 + ntile).  A chunk of the k-space slab and of
 * an exchange buffer is an ARRAY OF ITS OWN with row pitch W = ntile * tile_columns:
 *     k-chunk        complex[N][N/nparts][W]                 (x-major; block q = x-planes of rank q: contiguous)
 *     exchange chunk complex[nparts][N/nparts][N/nparts][W]  ([rank][x][k_y in rank][k_z in chunk])
 * -- equal contiguous blocks per rank, which is what the collective moves.  The x-slab's half spectrum `half_local`
 * ([N/nparts][N+1][pitch]) keeps all columns in one array.
 *   realise_density : fb_slab_x_generate_chunk(c) -> all-to-all(c) -> fb_slab_y_inverse_chunk(c), all c; fb_slab_z_pass(0)
 *   P(k)            : fb_slab_z_pass(1) -> fb_slab_y_forward_chunk(c) -> all-to-all(c) -> fb_slab_x_bin_chunk(c), all c
 *   both in one     : ... fb_slab_y_inverse_chunk(c) all c; fb_slab_z_pass(2); fb_slab_y_forward_chunk(c) ...
 * Fields are bit-identical to the unchunked calls for any chunking.                                                  */
int fb_slab_tile_geometry(const fb_plan* plan, int* tile_columns, int* tiles_per_row);
int fb_slab_x_generate_chunk(fb_plan* plan, void* kchunk, int nparts, int part, uint64_t seed, uint64_t realisation,
                             int tile0, int ntile, void* stream);
int fb_slab_y_inverse_chunk(fb_plan* plan, const void* recv_chunk, void* half_local, int nparts, int tile0, int ntile,
                            void* stream);
int fb_slab_y_forward_chunk(fb_plan* plan, const void* half_local, void* send_chunk, int nparts, int tile0, int ntile,
                            void* stream);
/* which: 0 = half_local -> real_local (c2r, numpy's 1/N^3), 1 = real_local -> half_local (r2c; of exp(real - shift) when
 * pre_exp, expsum_dev[0] = this rank's sum of them), 2 = c2r, store the field, r2c of (exp of) it from registers        */
int fb_slab_z_pass(fb_plan* plan, void* half_local, void* real_local, int nparts, int which, int pre_exp,
                   double* expsum_dev, void* stream);
/* first / last: the first and the last chunk of a spectrum (any order in between); results_dev as fb_slab_x_bin, written by
 * the last call */
int fb_slab_x_bin_chunk(fb_plan* plan, void* kchunk, int nparts, int part, int tile0, int ntile, int first, int last,
                        double* results_dev, void* stream);

/* ---- one box over several GPUs: communicator (RCCL over xGMI) --------------------------------------------------
 * The collectives of the slab-decomposed transform -- the all-to-all of equal blocks between the x-slab and the k_y-slab
 * layout (ONE per transform: fastbox/box.py:187 ifftn, :193 / :736 fftn), the all-reduce of the 2 nbins + 1 bin sums
 * (box.py:741-764) and of a field maximum -- behind the C ABI, so that a consumer of libfastbox_hip.so needs neither
 * PyTorch nor an MPI to run one box over the GPUs of a node: one process (or thread) per GPU, each with its own plan.
 *     rank 0:      fb_comm_unique_id(id);  hand the 128 bytes to the other ranks (file, socket, MPI_Bcast, ...)
 *     every rank:  fb_comm_create(plan, world, rank, id);
 *     per transform (or per k_z chunk):  fb_slab_x_generate[_chunk] -> fb_slab_exchange_begin -> ... the passes of the next
 *                  chunk on the caller's stream ... -> fb_slab_exchange_wait -> fb_slab_y_inverse_chunk / _inverse_packed
 *     fb_allreduce_f64(plan, results_dev, 2 * nbins + 1, 0, stream);   fb_comm_destroy(plan)   (fb_plan_destroy does it too)
 * The exchange runs on a stream the communicator owns (event hand-off from and to the caller's stream: the host never
 * blocks, and the caller's stream keeps computing between begin and wait); at most 16 exchanges in flight.  librccl is
 * opened on the first of these calls (FASTBOX_RCCL_LIB overrides the name); failures are FB_ERR_RCCL with RCCL's own message.
 * world = 1 with id = NULL needs no RCCL at all (the block moves by a device copy); world = 1 WITH an id makes a real
 * one-rank RCCL communicator.  fastbox_amd.distributed.RcclComm / SlabBox(comm="rccl") drive it from Python (ctypes). */
int fb_comm_unique_id(void* id128);                                   /* 128 bytes out                              */
int fb_comm_create(fb_plan* plan, int world, int rank, const void* id128);
int fb_comm_destroy(fb_plan* plan);
int fb_comm_info(const fb_plan* plan, int* world, int* rank, int* rccl_version);   /* world 0: no communicator     */
/* block q of send (bytes_per_peer bytes) -> rank q; block q of recv <- rank q; send != recv; ordered after `stream` */
int fb_slab_exchange_begin(fb_plan* plan, const void* send, void* recv, int64_t bytes_per_peer, void* stream, int* ticket);
int fb_slab_exchange_wait(fb_plan* plan, int ticket, void* stream);   /* `stream` waits for that exchange           */
int fb_slab_exchange(fb_plan* plan, const void* send, void* recv, int64_t bytes_per_peer, void* stream);   /* begin + wait */
/* in place on the device, in stream order of `stream` (issued, like the exchanges, to the communicator's own stream between two
 * event hand-overs: every operation of a communicator goes to one stream, in the same order on every rank); op 0 = sum, 1 = max */
int fb_allreduce_f64(fb_plan* plan, double* data_dev, int count, int op, void* stream);

/* The y and z passes of a transform run x-plane batch by x-plane batch so that a batch stays in the 256 MiB Infinity
 * Cache between them.  planes = -1: sized for ONE box using the GPU (default); when several boxes run concurrently on
 * their own streams, give each its share (e.g. 64 planes of a 512^3 box for two).  0 = whole box in one go.
 * streams = 2 sends alternate batches to a second stream of the plan; 0 = by grid size. */
int fb_set_plane_batching(fb_plan* plan, int planes, int streams);
/* How the strided FFT passes (x, y) of this plan are scheduled, per class (plain pass, fused generator pass, fused
 * binning pass): 0 = one workgroup per tile, 1 = resident workgroups that walk the tiles and load their next tile
 * while finishing the current one, -1 (default) = by grid size (resident where a CU holds one workgroup of the pass: N = 2048, and the generator pass from N = 1024).  The transforms are bit-identical in both forms; the binning pass groups its fp64
 * partial sums by resident workgroup instead of by tile (differences at the 1e-16 level).  Grids below 256^3 always use 0. */
int fb_set_pass_schedule(fb_plan* plan, int plain, int generator, int binning);
/* Row segment of the strided passes' tiles where a plan has a choice -- single precision at N = 2048: 128 bytes = 2048 rows x 16
 * columns exchanged through LDS as real and imaginary halves, 32 points per thread; 64 bytes = 2048 rows x 8 columns, 16 points
 * per thread (the form of rounds 1-3, which the slab-decomposed path's k_z-chunk launches keep).  0 = the library's choice per
 * pass class (the plain passes wide, the fused generator and binning passes narrow: they need their registers for the parked
 * stores / the second tile).  Transforms are bit-identical in both forms (the same radix-8/8/8/4 stages); the binning pass groups
 * its single-precision partial sums differently (1e-9 relative on a bin). */
int fb_set_tile_rows(fb_plan* plan, int bytes);
/* Fused log-normal transforms (pre_exp of fb_fft_r2c / fb_power_spectrum_device / _pending) form exp(x - shift).  The
 * estimate exp(d)/mean(exp(d)) - 1 does not depend on the shift (results[2 nbins] is the sum of the SHIFTED exponentials,
 * which is what the caller normalises with); the right shift keeps a single-precision plan's sum of exponentials (the
 * k = 0 mode), its |delta_k|^2 and |delta_k|^4 sums inside the float range whatever the field's variance: about
 * ln(sum exp(d)) - 7, see fastbox_amd/hostgeom.py lognormal_shift (from the field's variance) and lognormal_shift_exact
 * (from fb_max_real). */
int fb_set_exp_shift(fb_plan* plan, double shift);
/* tuning aid: a single strided FFT pass over a half spectrum (axis 0 = x, 1 = y;
 * mode 0 plain in place, 1 fused generator, 2 fused binning without store) */
int fb_debug_strided_pass(fb_plan* plan, void* half, int axis, int mode, void* stream);
/* diagnostic builds (-DFB_STAMPS) only: per-workgroup phase time stamps of the last plain pass */
int fb_debug_read_stamps(fb_plan* plan, long long* host, int64_t count);

/* ---- device memory helpers for bindings without their own allocator ---------------------------- */
int fb_malloc(void** dev_ptr, size_t bytes);
int fb_free(void* dev_ptr);
int fb_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int fb_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int fb_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int fb_stream_create(void** stream);     /* a non-blocking hipStream_t, for running independent boxes concurrently */
int fb_stream_create_priority(void** stream, int priority);   /* priority < 0: the device's highest, 0: middle, > 0: lowest */
int fb_stream_destroy(void* stream);
int fb_stream_sync(void* stream);
/* work queued on `waiter` after this call starts only when everything queued on `signaller` before it has finished
 * (both streams on the current device); the host does not wait */
int fb_stream_wait_stream(void* waiter, void* signaller);
int fb_device_count(int* count);
/* The calling thread's current HIP device.  Every entry point that takes a plan makes the plan's device current for the
 * duration of the call and restores the caller's before it returns, so the library never moves the current device under
 * other users of the runtime (torch, RCCL).  The plan-less helpers above (fb_malloc, fb_stream_create,
 * fb_stream_wait_stream) act on the CURRENT device: a binding that serves a box on another device brackets them with
 * fb_device_get / fb_device_set / fb_device_set(previous), as fastbox_amd/_lib.py on_device() does. */
int fb_device_get(int* device);
int fb_device_set(int device);

#ifdef __cplusplus
}
#endif
#endif
