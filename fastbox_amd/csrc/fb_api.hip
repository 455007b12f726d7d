// extern "C" surface of libfastbox_hip.so (see include/fastbox_hip.h).
#include "../../include/fastbox_hip.h"
#include "fb_plan.h"
#include <dlfcn.h>
#include <rccl/rccl.h>           // types and prototypes only: the functions are looked up in a dlopen'ed librccl (fb_comm.inc)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static thread_local std::string g_last_error;

void fb_set_error(const std::string& msg) { g_last_error = msg; }

int fb_hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return FB_OK;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    // the runtime also keeps the error as its "last error": read it away, or the next launch check
    // (hipGetLastError after a kernel launch) would report this failure a second time
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FB_ERR_NOMEM : FB_ERR_HIP;
}

// (every entry point starts with an argument check: it also reads away a stale "last error" some other user of the
// HIP runtime may have left in this thread, so that the launch checks below report this call's errors only)
#define FB_REQUIRE(cond, msg) do { (void)hipGetLastError(); if (!(cond)) { fb_set_error(msg); return FB_ERR_INVALID; } } while (0)
#define FB_DISPATCH(p, call32, call64) ((p)->prec == 4 ? (call32) : (call64))
// A plan belongs to one device.  Every entry point that takes a plan makes that device current for the duration of the
// call (allocations, NULL-stream launches and the plan's own auxiliary stream / events all follow the current device)
// and puts the caller's device back when it returns -- other users of the HIP runtime in this thread (torch, RCCL)
// keep the current device they had -- so plans on different GPUs can be used from one thread in any order.
namespace {
struct FbDeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
        if (prev != dev) {
            const int r = fb_hip_check(hipSetDevice(dev), "hipSetDevice");
            if (r) return r;
            changed = prev >= 0;
        }
        return FB_OK;
    }
    ~FbDeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};
}  // namespace
#define FB_USE_DEVICE(p) FbDeviceGuard _fb_devguard; do { const int _r = _fb_devguard.enter((p)->device); if (_r) return _r; } while (0)

namespace {
template <typename T> int upload(T** dst, const T* src, size_t n) {
    FB_HIP(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(T)));
    FB_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return FB_OK;
}
template <typename T> int make_twiddles(void** out, int N) {
    std::vector<T> h((size_t)2 * N);
    for (int j = 0; j < N; ++j) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)N;
        h[2 * (size_t)j] = (T)cosl(a);
        h[2 * (size_t)j + 1] = (T)sinl(a);
    }
    FB_HIP(hipMalloc(out, h.size() * sizeof(T)));
    FB_HIP(hipMemcpy(*out, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return FB_OK;
}
}  // namespace

extern "C" {

int fb_version(void) { return 100; }
const char* fb_last_error(void) { return g_last_error.c_str(); }

int fb_device_count(int* count) {
    FB_REQUIRE(count, "null pointer");
    FB_HIP(hipGetDeviceCount(count));
    return FB_OK;
}

int fb_device_get(int* device) {
    FB_REQUIRE(device, "null pointer");
    FB_HIP(hipGetDevice(device));
    return FB_OK;
}

int fb_device_set(int device) {
    (void)hipGetLastError();
    FB_HIP(hipSetDevice(device));
    return FB_OK;
}

int fb_plan_create(fb_plan** plan, int N, double Lx, double Ly, double Lz, int precision, int device,
                   const double* axis2, const double* ksc, const double* kpar, const double* zgrid) {
    FB_REQUIRE(plan && axis2 && ksc && kpar && zgrid, "null pointer");
    FB_REQUIRE(precision == 4 || precision == 8, "precision must be 4 or 8");
    FB_REQUIRE(Lx > 0 && Ly > 0 && Lz > 0, "box sides must be positive");
    // powers of two 16 .. 2048: the tuned kernels.  Other EVEN sizes whose prime factors are 2, 3, 5, up to 1024 (round 4): the
    // generic FFT passes (fb_fft_generic.h) behind fb_fft_c2c / _r2c / _c2r and every kernel that is not an FFT pass; the entry
    // points that fuse something into an FFT pass return FB_ERR_UNSUPPORTED on such a plan.
    bool pow2 = N >= 16 && N <= 2048 && !(N & (N - 1)), smooth = false;
    if (!pow2 && N >= 16 && N <= 1024 && N % 2 == 0) {
        int q = N;
        for (int f : {2, 3, 5}) while (q % f == 0) q /= f;
        smooth = q == 1;
    }
    if (!pow2 && !smooth) {
        fb_set_error("unsupported grid size: nsamp must be a power of two in 16..2048, or even with prime factors 2, 3, 5 in 16..1024");
        return FB_ERR_UNSUPPORTED;
    }
    FB_HIP(hipSetDevice(device));
    fb_plan* p = new fb_plan();
    p->N = N; p->prec = precision; p->device = device;
    p->generic = pow2 ? 0 : 1;
    p->L[0] = Lx; p->L[1] = Ly; p->L[2] = Lz;
    p->cubic = (Lx == Ly && Ly == Lz);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            p->num_cu = prop.multiProcessorCount;
    }
    p->NZV = N / 2 + 1;
    p->NZP = (p->NZV + 15) & ~15;
    p->NR = N + 1;
    int r = precision == 4 ? make_twiddles<float>(&p->tw, N) : make_twiddles<double>(&p->tw, N);
    if (!r) r = upload(&p->axis2, axis2, (size_t)3 * N);
    if (!r) r = upload(&p->ksc, ksc, (size_t)3 * N);
    if (!r) r = upload(&p->kpar, kpar, (size_t)N);
    if (!r) r = upload(&p->zgrid, zgrid, (size_t)N);
    if (!r) {          // k_perp per (k_x, k_y), the value kperp_exact() forms on the device (IEEE sqrt, same order)
        std::vector<double> kp((size_t)N * N);
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) kp[(size_t)i * N + j] = 6.283185307179586476925286766559 * std::sqrt(axis2[i] + axis2[N + j]);
        r = upload(&p->kperp_tab, kp.data(), kp.size());
    }
    p->prow = 2048;      // 8 four-wave workgroups per CU
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->partials, (size_t)p->prow * 2 * FB_MAX_BINS * sizeof(double)), "hipMalloc");
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->scratch, FB_SCRATCH * sizeof(double)), "hipMalloc");
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->counts, FB_MAX_BINS * sizeof(unsigned long long)), "hipMalloc");
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->bins, FB_MAX_BINS * sizeof(double)), "hipMalloc");
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->thr, FB_MAX_BINS * sizeof(int)), "hipMalloc");
    if (!r) r = fb_hip_check(hipMalloc((void**)&p->plane_buf, (size_t)N * N * 2 * precision), "hipMalloc");
    if (r) { fb_plan_destroy(p); return r; }
    p->nbins = 0;
    *plan = p;
    return FB_OK;
}

int fb_comm_destroy(fb_plan* p);
int fb_plan_destroy(fb_plan* p) {
    if (!p) return FB_OK;
    (void)fb_comm_destroy(p);
    (void)hipSetDevice(p->device);
    void* ptrs[] = {p->tw, p->axis2, p->ksc, p->kpar, p->zgrid, p->amp_shell, p->amp_sym, p->kperp_tab, p->pca_work, p->bins, p->thr, p->counts,
                    p->partials, p->scratch, p->bin_partials, p->exp_partials, p->plane_buf};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (p->aux_stream) { (void)hipStreamSynchronize(p->aux_stream); (void)hipStreamDestroy(p->aux_stream); }
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
    delete p;
    return FB_OK;
}

int fb_half_pitch(const fb_plan* p) { return p ? p->NZP : 0; }
int fb_half_rows(const fb_plan* p) { return p ? p->NR : 0; }
int64_t fb_real_bytes(const fb_plan* p) { return p ? (int64_t)p->N * p->N * p->N * p->prec : 0; }
int64_t fb_half_bytes(const fb_plan* p) { return p ? (int64_t)p->N * p->NR * p->NZP * 2 * p->prec : 0; }
int64_t fb_full_bytes(const fb_plan* p) { return p ? (int64_t)p->N * p->N * p->N * 2 * p->prec : 0; }

int fb_fft_c2c(fb_plan* p, void* d, int direction, double scale, void* stream) {
    FB_REQUIRE(p && d, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(direction == 1 || direction == -1, "direction must be +1 or -1");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_fft_c2c_f32(p, d, direction, scale, s), fbi_fft_c2c_f64(p, d, direction, scale, s));
}
int fb_fft_r2c(fb_plan* p, const void* in, void* out, int pre_exp, void* stream) {
    FB_REQUIRE(p && in && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_fft_r2c_f32(p, in, out, pre_exp, s), fbi_fft_r2c_f64(p, in, out, pre_exp, s));
}
int fb_fft_c2r(fb_plan* p, void* half, void* out, double scale, void* stream) {
    FB_REQUIRE(p && half && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_fft_c2r_f32(p, half, out, scale, s), fbi_fft_c2r_f64(p, half, out, scale, s));
}

// The plan's lookup tables (amplitudes, bin edges / thresholds) are read by kernels on whatever stream the caller
// launches them on, so a table is only rewritten once everything queued on the device has finished: table changes are
// rare (a new P(k), a new bin set), a device-wide wait is the one form that covers every caller stream.
#define FB_QUIESCE() FB_HIP(hipDeviceSynchronize())

int fb_set_amplitude_shells(fb_plan* p, const double* amp, int64_t nshell) {
    FB_REQUIRE(p && amp, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(p->cubic, "shell amplitudes need a cubic box (use fb_set_amplitude_dense)");
    const int64_t need = 3LL * (p->N / 2) * (p->N / 2) + 1;
    FB_REQUIRE(nshell == need, "nshell must be 3 (N/2)^2 + 1");
    FB_QUIESCE();
    p->amp_dense = nullptr;
    return FB_DISPATCH(p, fbi_set_amp_shells_f32(p, amp, nshell), fbi_set_amp_shells_f64(p, amp, nshell));
}
int fb_set_amplitude_sym(fb_plan* p, const double* amp, int64_t n) {
    FB_REQUIRE(p && amp, "null pointer");
    FB_USE_DEVICE(p);
    FB_QUIESCE();
    return FB_DISPATCH(p, fbi_set_amp_sym_f32(p, amp, n), fbi_set_amp_sym_f64(p, amp, n));
}
int fb_set_amplitude_dense(fb_plan* p, const void* amp_dev) {
    FB_REQUIRE(p && amp_dev, "null pointer");
    FB_USE_DEVICE(p);
    FB_QUIESCE();
    p->amp_dense = amp_dev;
    return FB_OK;
}
int fb_colour_noise(fb_plan* p, const void* re, const void* im, void* out, void* stream) {
    FB_REQUIRE(p && re && im && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_colour_noise_f32(p, re, im, out, s), fbi_colour_noise_f64(p, re, im, out, s));
}
int fb_colour_device(fb_plan* p, uint64_t seed, uint64_t realisation, void* out, void* stream) {
    FB_REQUIRE(p && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_colour_device_f32(p, seed, realisation, out, s),
                       fbi_colour_device_f64(p, seed, realisation, out, s));
}

int fb_set_bins(fb_plan* p, const double* edges, int nbins, const int32_t* thr, const int32_t* amb, int namb) {
    FB_REQUIRE(p && edges, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(nbins >= 1 && nbins <= FB_MAX_BINS, "nbins must be in 1..256");
    FB_REQUIRE(namb >= 0 && namb <= 8, "at most 8 ambiguous shells");
    for (int q = 1; q < nbins; ++q) FB_REQUIRE(edges[q] >= edges[q - 1], "bin edges must be ascending");
    FB_REQUIRE(!thr || p->cubic, "shell thresholds need a cubic box");
    FB_QUIESCE();
    FB_HIP(hipMemcpy(p->bins, edges, (size_t)nbins * sizeof(double), hipMemcpyHostToDevice));
    p->nbins = nbins;
    static_assert(sizeof(int) == sizeof(int32_t), "int");
    p->namb = 0;
    p->use_thr = thr ? 1 : 0;
    if (thr) {
        FB_HIP(hipMemcpy(p->thr, thr, (size_t)nbins * sizeof(int), hipMemcpyHostToDevice));
        p->namb = namb;
        for (int q = 0; q < namb; ++q) p->amb[q] = amb[q];
    }
    int r = fbi_bin_count(p, 0);
    if (r) return r;
    FB_HIP(hipStreamSynchronize(0));      // caller streams may be non-blocking: the counts / tables are complete on return
    return FB_OK;
}

namespace {
int bin_power_common(fb_plan* p, const void* spec, int layout, int kind, const double* params, const void* table_dev,
                     double* count, double* sum, double* sumsq, void* stream) {
    FB_REQUIRE(p && spec && count && sum && sumsq, "null pointer");
    FB_REQUIRE(layout == 0 || layout == 1, "layout must be 0 (full) or 1 (half)");
    hipStream_t s = (hipStream_t)stream;
    int r = FB_DISPATCH(p, fbi_bin_power_f32(p, spec, layout, kind, params, table_dev, p->scratch, s),
                        fbi_bin_power_f64(p, spec, layout, kind, params, table_dev, p->scratch, s));
    if (r) return r;
    std::vector<double> h((size_t)2 * p->nbins);
    FB_HIP(hipMemcpyAsync(h.data(), p->scratch, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    FB_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < p->nbins; ++q) {
        count[q] = p->counts_host[q];
        sum[q] = h[(size_t)2 * q];
        sumsq[q] = h[(size_t)2 * q + 1];
    }
    return FB_OK;
}
}  // namespace

int fb_bin_power(fb_plan* p, const void* spec, int layout, double* count, double* sum, double* sumsq, void* stream) {
    return bin_power_common(p, spec, layout, -1, nullptr, nullptr, count, sum, sumsq, stream);
}
int fb_bin_power_filtered(fb_plan* p, const void* spec, int layout, int kind, const double* params,
                          const void* table_dev, double* count, double* sum, double* sumsq, void* stream) {
    FB_REQUIRE(kind >= 0, "unknown filter kind");
    FB_USE_DEVICE(p);
    return bin_power_common(p, spec, layout, kind, params, table_dev, count, sum, sumsq, stream);
}

int fb_apply_filter(fb_plan* p, const void* in, void* out, int layout, int kind, const double* params,
                    const void* table_dev, void* stream) {
    FB_REQUIRE(p && in && out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(layout == 0 || layout == 1, "layout must be 0 (full) or 1 (half)");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_apply_filter_f32(p, in, out, layout, kind, params, table_dev, s),
                       fbi_apply_filter_f64(p, in, out, layout, kind, params, table_dev, s));
}
int fb_velocity_k(fb_plan* p, const void* dk, void* out, int layout, int component, double fac, void* stream) {
    FB_REQUIRE(p && dk && out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(layout == 0 || layout == 1, "layout must be 0 (full) or 1 (half)");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_velocity_f32(p, dk, out, layout, component, fac, s),
                       fbi_velocity_f64(p, dk, out, layout, component, fac, s));
}
int fb_potential_k(fb_plan* p, const void* dk, void* out, int layout, void* stream) {
    FB_REQUIRE(p && dk && out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(layout == 0 || layout == 1, "layout must be 0 (full) or 1 (half)");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_potential_f32(p, dk, out, layout, s), fbi_potential_f64(p, dk, out, layout, s));
}
int fb_lognormal(fb_plan* p, const void* in, void* out, double* mean_out, void* stream) {
    FB_REQUIRE(p && in && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_lognormal_f32(p, in, out, mean_out, s), fbi_lognormal_f64(p, in, out, mean_out, s));
}
int fb_redshift_space(fb_plan* p, const void* delta, const void* vz, const void* noise, void* out, double Hz,
                      double sigma_nl, uint64_t seed, int method, void* stream) {
    FB_REQUIRE(p && delta && vz && out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(Hz > 0, "Hz must be positive");
    FB_REQUIRE(out != delta && out != vz, "redshift_space is out of place");
    FB_REQUIRE(method == FB_RSD_LINEAR || method == FB_RSD_NEAREST || method == FB_RSD_CUBIC, "method: FB_RSD_LINEAR, FB_RSD_NEAREST or FB_RSD_CUBIC");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_rsd_f32(p, delta, vz, noise, out, Hz, sigma_nl, seed, method, s),
                       fbi_rsd_f64(p, delta, vz, noise, out, Hz, sigma_nl, seed, method, s));
}
int fb_sum_real(fb_plan* p, const void* x, int squared, double* out, void* stream) {
    FB_REQUIRE(p && x && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sum_real_f32(p, x, squared, out, s), fbi_sum_real_f64(p, x, squared, out, s));
}
int fb_max_real(fb_plan* p, const void* x, double* out, void* stream) {
    FB_REQUIRE(p && x && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_max_real_f32(p, x, out, s), fbi_max_real_f64(p, x, out, s));
}
int fb_sumsq_half(fb_plan* p, const void* h, double* out, void* stream) {
    FB_REQUIRE(p && h && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sumsq_half_f32(p, h, out, s), fbi_sumsq_half_f64(p, h, out, s));
}
int fb_expand_half(fb_plan* p, const void* half, void* full, void* stream) {
    FB_REQUIRE(p && half && full, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_expand_half_f32(p, half, full, s), fbi_expand_half_f64(p, half, full, s));
}
int fb_crop_full(fb_plan* p, const void* full, void* half, void* stream) {
    FB_REQUIRE(p && half && full, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_crop_full_f32(p, full, half, s), fbi_crop_full_f64(p, full, half, s));
}

int fb_realise_density_device(fb_plan* p, uint64_t seed, uint64_t realisation, void* work_half, void* real_out,
                              void* stream) {
    FB_REQUIRE(p && work_half && real_out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_realise_fused_f32(p, seed, realisation, work_half, real_out, scale, s),
                       fbi_realise_fused_f64(p, seed, realisation, work_half, real_out, scale, s));
}
int fb_realise_velocity_device(fb_plan* p, uint64_t seed, uint64_t realisation, int comp, double fac,
                               void* work_half, void* real_out, void* stream) {
    FB_REQUIRE(p && work_half && real_out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(comp >= 0 && comp <= 2, "component must be 0, 1 or 2");
    FB_REQUIRE(p->N % 2 == 0, "velocity needs an even grid size");
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_realise_velocity_fused_f32(p, seed, realisation, comp, fac, work_half, real_out, scale, s),
                       fbi_realise_velocity_fused_f64(p, seed, realisation, comp, fac, work_half, real_out, scale, s));
}
int fb_realise_density_begin(fb_plan* p, uint64_t seed, uint64_t realisation, void* pending_half, void* stream) {
    FB_REQUIRE(p && pending_half, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_realise_begin_f32(p, seed, realisation, pending_half, s),
                       fbi_realise_begin_f64(p, seed, realisation, pending_half, s));
}
int fb_realise_density_finish(fb_plan* p, void* pending_half, void* real_out, void* stream) {
    FB_REQUIRE(p && pending_half && real_out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_realise_finish_f32(p, pending_half, real_out, scale, s),
                       fbi_realise_finish_f64(p, pending_half, real_out, scale, s));
}
int fb_power_spectrum_filtered(fb_plan* p, const void* real_in, void* filtered_half, int kind, const double* params,
                               const void* table_dev, void* results_dev, void* stream) {
    FB_REQUIRE(p && real_in && filtered_half && results_dev, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_power_filtered_f32(p, real_in, filtered_half, kind, params, table_dev, (double*)results_dev, 0, s),
                       fbi_power_filtered_f64(p, real_in, filtered_half, kind, params, table_dev, (double*)results_dev, 0, s));
}
int fb_power_spectrum_filtered_field(fb_plan* p, const void* real_in, void* work_half, int kind, const double* params,
                                     const void* table_dev, void* results_dev, void* stream) {
    FB_REQUIRE(p && real_in && work_half && results_dev, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_power_filtered_f32(p, real_in, work_half, kind, params, table_dev, (double*)results_dev, 1, s),
                       fbi_power_filtered_f64(p, real_in, work_half, kind, params, table_dev, (double*)results_dev, 1, s));
}
int fb_realise_velocity_begin(fb_plan* p, uint64_t seed, uint64_t realisation, int comp, double fac, void* pending_half,
                              void* stream) {
    FB_REQUIRE(p && pending_half, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(comp >= 0 && comp <= 2, "component must be 0, 1 or 2");
    FB_REQUIRE(p->N % 2 == 0, "velocity needs an even grid size");
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_realise_velocity_begin_f32(p, seed, realisation, comp, fac, pending_half, s),
                       fbi_realise_velocity_begin_f64(p, seed, realisation, comp, fac, pending_half, s));
}
int fb_power_spectrum_redshift_space(fb_plan* p, void* pending_delta, void* pending_vz, void* delta_x_out, void* work_half,
                                     double Hz, double sigma_nl, uint64_t seed, int method, int filter_kind,
                                     const double* params, const void* table_dev, int want_field, void* results_dev,
                                     void* stream) {
    FB_REQUIRE(p && pending_delta && pending_vz && work_half && results_dev, "null pointer");   // delta_x_out may be NULL
    FB_USE_DEVICE(p);
    FB_REQUIRE(Hz > 0, "Hz must be positive");
    FB_REQUIRE(method == FB_RSD_LINEAR || method == FB_RSD_NEAREST, "method: FB_RSD_LINEAR or FB_RSD_NEAREST");
    FB_REQUIRE(pending_delta != pending_vz && work_half != pending_delta && work_half != pending_vz, "three distinct half spectra");
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_power_redshift_space_f32(p, pending_delta, pending_vz, delta_x_out, work_half, scale, Hz, sigma_nl, seed,
                                                       method, filter_kind, params, table_dev, (double*)results_dev, want_field, s),
                       fbi_power_redshift_space_f64(p, pending_delta, pending_vz, delta_x_out, work_half, scale, Hz, sigma_nl, seed,
                                                    method, filter_kind, params, table_dev, (double*)results_dev, want_field, s));
}
int fb_fft_c2r_yz(fb_plan* p, void* half, void* out, double scale, void* stream) {
    FB_REQUIRE(p && half && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_fft_c2r_yz_f32(p, half, out, scale, s), fbi_fft_c2r_yz_f64(p, half, out, scale, s));
}
int fb_power_spectrum_pending(fb_plan* p, void* pending_half, void* real_out, int pre_exp, double* results_dev,
                              void* stream) {
    FB_REQUIRE(p && pending_half && results_dev, "null pointer");       // real_out may be NULL: delta_x is not written
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_power_from_pending_f32(p, pending_half, real_out, scale, pre_exp, results_dev, s),
                       fbi_power_from_pending_f64(p, pending_half, real_out, scale, pre_exp, results_dev, s));
}
// The Monte-Carlo loop "draw realisation r, estimate its P(k)" for count realisations in ONE call: per realisation the launches
// of fb_realise_density_begin + fb_power_spectrum_pending, queued back to back on `stream` (the host's share of a step of a
// small box -- two library calls from an interpreter -- is the step's limit below 256^3).
int fb_montecarlo_power(fb_plan* p, uint64_t seed, uint64_t first, uint64_t stride, int count, void* work_half, void* real_out,
                        int pre_exp, double* results_dev, int64_t results_stride, void* stream) {
    FB_REQUIRE(p && work_half && results_dev, "null pointer");          // real_out may be NULL: the fields are not written
    FB_REQUIRE(count >= 0 && stride >= 1, "count >= 0, stride >= 1");
    FB_USE_DEVICE(p);
    FB_REQUIRE(results_stride >= 2 * (int64_t)p->nbins + 1, "results_stride: at least 2 nbins + 1 doubles per realisation");
    hipStream_t s = (hipStream_t)stream;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    for (int i = 0; i < count; ++i) {
        const uint64_t real = first + (uint64_t)i * stride;
        int r = FB_DISPATCH(p, fbi_realise_begin_f32(p, seed, real, work_half, s), fbi_realise_begin_f64(p, seed, real, work_half, s));
        if (r) return r;
        double* res = results_dev + (size_t)i * results_stride;
        r = FB_DISPATCH(p, fbi_power_from_pending_f32(p, work_half, real_out, scale, pre_exp, res, s),
                        fbi_power_from_pending_f64(p, work_half, real_out, scale, pre_exp, res, s));
        if (r) return r;
    }
    return FB_OK;
}
int fb_power_spectrum_device(fb_plan* p, const void* real_in, void* work_half, int pre_exp, int keep_spectrum,
                             double* results_dev, void* stream) {
    FB_REQUIRE(p && real_in && work_half && results_dev, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_power_fused_f32(p, real_in, work_half, pre_exp, keep_spectrum, results_dev, s),
                       fbi_power_fused_f64(p, real_in, work_half, pre_exp, keep_spectrum, results_dev, s));
}
int fb_bin_counts(fb_plan* p, double* count) {
    FB_REQUIRE(p && count, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(p->nbins > 0, "bin edges not set");
    for (int q = 0; q < p->nbins; ++q) count[q] = p->counts_host[q];
    return FB_OK;
}

int fb_real_axpby(fb_plan* p, const void* x, const void* y, void* out, double a, double b, double c, void* stream) {
    FB_REQUIRE(p && x && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_real_axpby_f32(p, x, y, out, a, b, c, 0, s), fbi_real_axpby_f64(p, x, y, out, a, b, c, 0, s));
}
int fb_real_multiply(fb_plan* p, const void* x, const void* y, void* out, void* stream) {
    FB_REQUIRE(p && x && y && out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_real_axpby_f32(p, x, y, out, 0, 0, 0, 1, s), fbi_real_axpby_f64(p, x, y, out, 0, 0, 0, 1, s));
}

// ---- transverse (per-channel 2-D) transforms and masks (fastbox/filters.py:58-90) ---------------------------------
int fb_fft_transverse(fb_plan* p, void* full_cube, int direction, void* stream) {
    FB_REQUIRE(p && full_cube, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(direction == 1 || direction == -1, "direction must be -1 (fftn) or +1 (ifftn)");
    hipStream_t s = (hipStream_t)stream;
    const double scale = direction > 0 ? 1.0 / ((double)p->N * p->N) : 1.0;
    return FB_DISPATCH(p, fbi_fft_axes01_f32(p, full_cube, direction, scale, s), fbi_fft_axes01_f64(p, full_cube, direction, scale, s));
}
int fb_real_to_complex(fb_plan* p, const void* real_cube, void* full_cube, void* stream) {
    FB_REQUIRE(p && real_cube && full_cube, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_real_to_complex_f32(p, real_cube, full_cube, s), fbi_real_to_complex_f64(p, real_cube, full_cube, s));
}
int fb_mask_transverse(fb_plan* p, void* full_cube, const void* mask2d, void* stream) {
    FB_REQUIRE(p && full_cube && mask2d, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_mask_xy_f32(p, full_cube, mask2d, s), fbi_mask_xy_f64(p, full_cube, mask2d, s));
}

// ---- beam convolution (fastbox/beams.py:63-137) ---------------------------------------------------------------
int fb_beam_convolve(fb_plan* p, const void* field, const void* beam, void* work_a, void* work_b, void* out, int periodic,
                     int beam_ready, void* stream) {
    FB_REQUIRE(p && field && work_a && work_b && out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(beam || beam_ready, "null beam cube");
    FB_REQUIRE(work_a != work_b, "the two work cubes must be distinct");
    FB_REQUIRE((periodic == 0 || periodic == 1) && (beam_ready == 0 || beam_ready == 1), "periodic, beam_ready must be 0 or 1");
    FB_REQUIRE(periodic || p->N >= 32, "zero-padded convolution: the plan is that of the transform size 2n >= 32");
    hipStream_t s = (hipStream_t)stream;
    const int flags = periodic | (beam_ready << 1);
    return FB_DISPATCH(p, fbi_beam_convolve_f32(p, field, beam, work_a, work_b, out, flags, s),
                       fbi_beam_convolve_f64(p, field, beam, work_a, work_b, out, flags, s));
}

// ---- PCA foreground cleaning (fastbox/filters.py:93-183) -----------------------------------------------------
int fb_channel_means(fb_plan* p, const void* cube, double* mean_dev, void* stream) {
    FB_REQUIRE(p && cube && mean_dev, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_channel_means_f32(p, cube, mean_dev, s), fbi_channel_means_f64(p, cube, mean_dev, s));
}
int fb_channel_covariance(fb_plan* p, const void* cube, const double* mean_dev, double* cov_dev, void* stream) {
    FB_REQUIRE(p && cube && mean_dev && cov_dev, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_channel_cov_f32(p, cube, mean_dev, cov_dev, s), fbi_channel_cov_f64(p, cube, mean_dev, cov_dev, s));
}
int fb_pca_clean(fb_plan* p, const void* cube, const double* mean_dev, const double* modes_dev, int nmodes, void* cube_out,
                 double* amps_dev, void* stream) {
    FB_REQUIRE(p && cube && mean_dev && cube_out && (modes_dev || nmodes == 0), "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_pca_clean_f32(p, cube, mean_dev, modes_dev, nmodes, cube_out, amps_dev, s),
                       fbi_pca_clean_f64(p, cube, mean_dev, modes_dev, nmodes, cube_out, amps_dev, s));
}

// ---- foreground maps / cube, radiometer noise (the steps after the density-field path) -------------------------
int fb_sky_realise_map(fb_plan* p, const void* amp2d, const void* re, const void* im, uint64_t seed, double monopole,
                       void* work_cplx, void* map_out, void* stream) {
    FB_REQUIRE(p && amp2d && work_cplx && map_out, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE((re == nullptr) == (im == nullptr), "give both re and im, or neither (device generator)");
    hipStream_t s = (hipStream_t)stream;
    int r = FB_DISPATCH(p, fbi_sky_colour_map_f32(p, amp2d, re, im, seed, work_cplx, s),
                        fbi_sky_colour_map_f64(p, amp2d, re, im, seed, work_cplx, s));
    if (r) return r;
    const double scale = 1.0 / ((double)p->N * p->N);
    r = FB_DISPATCH(p, fbi_fft2d_c2c_f32(p, work_cplx, +1, scale, s), fbi_fft2d_c2c_f64(p, work_cplx, +1, scale, s));
    if (r) return r;
    return FB_DISPATCH(p, fbi_sky_real_plus_f32(p, work_cplx, map_out, monopole, s),
                       fbi_sky_real_plus_f64(p, work_cplx, map_out, monopole, s));
}
int fb_sky_normal_map(fb_plan* p, const void* unit, uint64_t seed, double mean, double std, void* map_out, void* stream) {
    FB_REQUIRE(p && map_out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sky_normal_map_f32(p, unit, seed, mean, std, map_out, s),
                       fbi_sky_normal_map_f64(p, unit, seed, mean, std, map_out, s));
}
int fb_sky_gaussian_filter(fb_plan* p, void* map_inout, void* tmp, const double* weights, int radius, void* stream) {
    FB_REQUIRE(p && map_inout && tmp && weights, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sky_gaussian_f32(p, map_inout, tmp, weights, radius, s),
                       fbi_sky_gaussian_f64(p, map_inout, tmp, weights, radius, s));
}
int fb_sky_foreground_cube(fb_plan* p, const void* amps, const void* alpha, double alpha_scalar, const double* ratio,
                           void* cube_out, void* stream) {
    FB_REQUIRE(p && amps && ratio && cube_out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sky_fg_cube_f32(p, amps, alpha, alpha_scalar, ratio, cube_out, s),
                       fbi_sky_fg_cube_f64(p, amps, alpha, alpha_scalar, ratio, cube_out, s));
}
int fb_sky_noise_cube(fb_plan* p, const double* sigma, const void* unit, uint64_t seed, void* cube_out, void* stream) {
    FB_REQUIRE(p && sigma && cube_out, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_sky_noise_cube_f32(p, sigma, unit, seed, cube_out, s),
                       fbi_sky_noise_cube_f64(p, sigma, unit, seed, cube_out, s));
}

// ---- slab-decomposed transforms -------------------------------------------------------------------
#define FB_SLAB_CHECK(p, nparts) \
    FB_REQUIRE((p), "null pointer"); \
    FB_REQUIRE((nparts) >= 1 && (p)->N % (nparts) == 0, "the number of slabs must divide N"); \
    FB_USE_DEVICE(p)

int fb_slab_forward_local(fb_plan* p, const void* real_local, void* half_local, int nparts, int pre_exp,
                          double* expsum_dev, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(real_local && half_local, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_forward_local_f32(p, real_local, half_local, nxl, pre_exp, expsum_dev, s),
                       fbi_slab_forward_local_f64(p, real_local, half_local, nxl, pre_exp, expsum_dev, s));
}
int fb_slab_inverse_local(fb_plan* p, void* half_local, void* real_local, int nparts, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(real_local && half_local, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_slab_inverse_local_f32(p, half_local, real_local, nxl, scale, s),
                       fbi_slab_inverse_local_f64(p, half_local, real_local, nxl, scale, s));
}
int fb_slab_forward_packed(fb_plan* p, const void* real_local, void* half_local, void* sendbuf, int nparts, int pre_exp,
                           double* expsum_dev, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(real_local && half_local && sendbuf && half_local != sendbuf, "bad buffers");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_forward_packed_f32(p, real_local, half_local, sendbuf, nxl, nparts, pre_exp, expsum_dev, s),
                       fbi_slab_forward_packed_f64(p, real_local, half_local, sendbuf, nxl, nparts, pre_exp, expsum_dev, s));
}
int fb_slab_inverse_packed(fb_plan* p, const void* recvbuf, void* half_local, void* real_local, int nparts, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(recvbuf && real_local && half_local && half_local != recvbuf, "bad buffers");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_slab_inverse_packed_f32(p, recvbuf, half_local, real_local, nxl, nparts, scale, s),
                       fbi_slab_inverse_packed_f64(p, recvbuf, half_local, real_local, nxl, nparts, scale, s));
}
int fb_slab_turnaround(fb_plan* p, const void* recvbuf, void* half_local, void* real_local, void* sendbuf, int nparts,
                       int pre_exp, double* expsum_dev, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(recvbuf && half_local && real_local && sendbuf && half_local != recvbuf && half_local != sendbuf,
               "bad buffers");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_slab_turnaround_f32(p, recvbuf, half_local, real_local, sendbuf, nxl, nparts, scale, pre_exp, expsum_dev, s),
                       fbi_slab_turnaround_f64(p, recvbuf, half_local, real_local, sendbuf, nxl, nparts, scale, pre_exp, expsum_dev, s));
}
int fb_slab_pack(fb_plan* p, const void* half_local, void* sendbuf, int nparts, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(half_local && sendbuf && half_local != sendbuf, "bad buffers");
    return fbi_slab_permute(p, half_local, sendbuf, p->N / nparts, nparts, 1, (hipStream_t)stream);
}
int fb_slab_unpack(fb_plan* p, const void* recvbuf, void* half_local, int nparts, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(half_local && recvbuf && half_local != recvbuf, "bad buffers");
    return fbi_slab_permute(p, recvbuf, half_local, p->N / nparts, nparts, 0, (hipStream_t)stream);
}
int fb_slab_x_pass(fb_plan* p, void* kslab, int nparts, int direction, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(kslab && (direction == 1 || direction == -1), "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nyl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_x_pass_f32(p, kslab, nyl, direction, s), fbi_slab_x_pass_f64(p, kslab, nyl, direction, s));
}
int fb_slab_x_generate(fb_plan* p, void* kslab, int nparts, int part, uint64_t seed, uint64_t realisation, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(kslab && part >= 0 && part < nparts, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nyl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_x_generate_f32(p, kslab, nyl, part * nyl, seed, realisation, s),
                       fbi_slab_x_generate_f64(p, kslab, nyl, part * nyl, seed, realisation, s));
}
int fb_slab_x_bin(fb_plan* p, void* kslab, int nparts, int part, double* results_dev, void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(kslab && results_dev && part >= 0 && part < nparts, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int nyl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_x_bin_f32(p, kslab, nyl, part * nyl, results_dev, s),
                       fbi_slab_x_bin_f64(p, kslab, nyl, part * nyl, results_dev, s));
}
// ---- chunked form: a range of k_z tile columns at a time (see fb_fft_launch.inc) ----
int fb_slab_tile_geometry(const fb_plan* p, int* tile_columns, int* tiles_per_row) {
    FB_REQUIRE(p && tile_columns && tiles_per_row, "null pointer");
    const int tz = FB_DISPATCH(p, fbi_slab_tile_cols_f32(p), fbi_slab_tile_cols_f64(p));
    *tile_columns = tz;
    *tiles_per_row = (p->NZV + tz - 1) / tz;
    return FB_OK;
}
static bool fb_chunk_ok(const fb_plan* p, int tile0, int ntile) {
    const int tz = FB_DISPATCH(p, fbi_slab_tile_cols_f32(p), fbi_slab_tile_cols_f64(p));
    return tile0 >= 0 && ntile > 0 && tile0 + ntile <= (p->NZV + tz - 1) / tz;
}
// (FB_SLAB_CHECK declares the device guard: it must stay in the function's own scope)
#define FB_CHUNK_CHECK(p, nparts, tile0, ntile) \
    FB_SLAB_CHECK(p, nparts);                   \
    FB_REQUIRE(fb_chunk_ok(p, tile0, ntile), "tile range outside the half spectrum")
int fb_slab_x_generate_chunk(fb_plan* p, void* kchunk, int nparts, int part, uint64_t seed, uint64_t realisation, int tile0,
                             int ntile, void* stream) {
    FB_CHUNK_CHECK(p, nparts, tile0, ntile);
    FB_REQUIRE(kchunk && part >= 0 && part < nparts, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nyl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_x_generate_chunk_f32(p, kchunk, nyl, part * nyl, seed, realisation, tile0, ntile, s),
                       fbi_slab_x_generate_chunk_f64(p, kchunk, nyl, part * nyl, seed, realisation, tile0, ntile, s));
}
int fb_slab_y_inverse_chunk(fb_plan* p, const void* recv_chunk, void* half_local, int nparts, int tile0, int ntile, void* stream) {
    FB_CHUNK_CHECK(p, nparts, tile0, ntile);
    FB_REQUIRE(recv_chunk && half_local && recv_chunk != half_local, "bad buffers");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_y_inverse_chunk_f32(p, recv_chunk, half_local, nxl, nparts, tile0, ntile, s),
                       fbi_slab_y_inverse_chunk_f64(p, recv_chunk, half_local, nxl, nparts, tile0, ntile, s));
}
int fb_slab_y_forward_chunk(fb_plan* p, const void* half_local, void* send_chunk, int nparts, int tile0, int ntile, void* stream) {
    FB_CHUNK_CHECK(p, nparts, tile0, ntile);
    FB_REQUIRE(send_chunk && half_local && send_chunk != half_local, "bad buffers");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_y_forward_chunk_f32(p, half_local, send_chunk, nxl, nparts, tile0, ntile, s),
                       fbi_slab_y_forward_chunk_f64(p, half_local, send_chunk, nxl, nparts, tile0, ntile, s));
}
int fb_slab_z_pass(fb_plan* p, void* half_local, void* real_local, int nparts, int which, int pre_exp, double* expsum_dev,
                   void* stream) {
    FB_SLAB_CHECK(p, nparts);
    FB_REQUIRE(half_local && (real_local || which == 2) && which >= 0 && which <= 2, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nxl = p->N / nparts;
    const double scale = 1.0 / ((double)p->N * p->N * p->N);
    return FB_DISPATCH(p, fbi_slab_z_pass_f32(p, half_local, real_local, nxl, which, scale, pre_exp, expsum_dev, s),
                       fbi_slab_z_pass_f64(p, half_local, real_local, nxl, which, scale, pre_exp, expsum_dev, s));
}
int fb_slab_x_bin_chunk(fb_plan* p, void* kchunk, int nparts, int part, int tile0, int ntile, int first, int last,
                        double* results_dev, void* stream) {
    FB_CHUNK_CHECK(p, nparts, tile0, ntile);
    FB_REQUIRE(kchunk && part >= 0 && part < nparts && (results_dev || !last), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nyl = p->N / nparts;
    return FB_DISPATCH(p, fbi_slab_x_bin_chunk_f32(p, kchunk, nyl, part * nyl, tile0, ntile, first, last, results_dev, s),
                       fbi_slab_x_bin_chunk_f64(p, kchunk, nyl, part * nyl, tile0, ntile, first, last, results_dev, s));
}

int64_t fb_slab_half_bytes(const fb_plan* p, int nparts) {
    return (p && nparts > 0) ? (int64_t)(p->N / nparts) * p->NR * p->NZP * 2 * p->prec : 0;
}
int64_t fb_slab_kspace_bytes(const fb_plan* p, int nparts) {
    return (p && nparts > 0) ? (int64_t)p->N * (p->N / nparts) * p->NZP * 2 * p->prec : 0;
}

int fb_debug_strided_pass(fb_plan* p, void* half, int axis, int mode, void* stream) {
    FB_REQUIRE(p && half, "null pointer");
    FB_USE_DEVICE(p);
    hipStream_t s = (hipStream_t)stream;
    return FB_DISPATCH(p, fbi_debug_pass_f32(p, half, axis, mode, s), fbi_debug_pass_f64(p, half, axis, mode, s));
}

int fb_set_pass_schedule(fb_plan* p, int plain, int generator, int binning) {
    FB_REQUIRE(p, "null pointer");
    FB_REQUIRE(plain >= -1 && plain <= 1 && generator >= -1 && generator <= 1 && binning >= -1 && binning <= 1,
               "schedule: 0 (one workgroup per tile), 1 (resident workgroups) or -1 (by grid size)");
    p->pass_schedule[0] = plain; p->pass_schedule[1] = generator; p->pass_schedule[2] = binning;
    return FB_OK;
}
int fb_set_tile_rows(fb_plan* p, int bytes) {
    FB_REQUIRE(p, "null pointer");
    FB_REQUIRE(bytes == 0 || bytes == 64 || bytes == 128, "row segment of the strided passes' tiles: 0 (default), 64 or 128 bytes");
    p->wide_rows = bytes == 0 ? -1 : (bytes == 64 ? 0 : 7);
    return FB_OK;
}
int fb_set_exp_shift(fb_plan* p, double shift) {
    FB_REQUIRE(p, "null pointer");
    FB_REQUIRE(shift == shift && shift > -1e4 && shift < 1e4, "shift out of range");
    p->exp_shift = shift;
    return FB_OK;
}
int fb_set_plane_batching(fb_plan* p, int planes, int streams) {
    FB_REQUIRE(p, "null pointer");
    FB_REQUIRE(planes >= -1 && streams >= 0 && streams <= 2, "planes >= -1 (auto), streams 0 (auto), 1 or 2");
    p->plane_batch = planes; p->plane_streams = streams;
    return FB_OK;
}
int fb_debug_read_stamps(fb_plan* p, long long* host, int64_t count) {
    FB_REQUIRE(p && host && p->bin_partials, "no stamps");
    FB_USE_DEVICE(p);
    FB_HIP(hipDeviceSynchronize());
    FB_HIP(hipMemcpy(host, p->bin_partials, (size_t)count * sizeof(long long), hipMemcpyDeviceToHost));
    return FB_OK;
}

int fb_profile_select(fb_plan* p, unsigned mask) {
    FB_REQUIRE(p, "null pointer");
    FB_USE_DEVICE(p);
    p->prof_mask = mask;
    return FB_OK;
}
int fb_profile_sample(fb_plan* p, int stride, int64_t* seen) {
    FB_REQUIRE(p, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(stride >= 0, "stride must be >= 1, or 0 to read the count only");
    if (seen) *seen = (int64_t)p->prof_seen;
    if (stride > 0) p->prof_stride = stride;
    return FB_OK;
}
int fb_profile_start(fb_plan* p) {
    FB_REQUIRE(p, "null pointer");
    FB_USE_DEVICE(p);
    p->prof_seen = 0;
    p->prof_used = 0;
    p->prof_cat.clear();
    p->prof_on = true;
    return FB_OK;
}
int fb_profile_stop(fb_plan* p, void* stream, double* ms, int64_t* launches, int ncat) {
    FB_REQUIRE(p && ms && launches, "null pointer");
    FB_USE_DEVICE(p);
    FB_REQUIRE(ncat >= FBK_NCAT, "ncat must be >= FB_PROF_NCAT");
    p->prof_on = false;
    FB_HIP(hipStreamSynchronize((hipStream_t)stream));
    for (int q = 0; q < ncat; ++q) { ms[q] = 0.0; launches[q] = 0; }
    for (size_t i = 0; i + 1 < p->prof_used; i += 2) {
        float t = 0.f;
        FB_HIP(hipEventElapsedTime(&t, p->prof_ev[i], p->prof_ev[i + 1]));
        const int c = p->prof_cat[i / 2];
        ms[c] += (double)t;
        launches[c] += 1;
    }
    p->prof_used = 0;
    return FB_OK;
}

int fb_malloc(void** dev_ptr, size_t bytes) {
    FB_REQUIRE(dev_ptr, "null pointer");
    FB_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return FB_OK;
}
int fb_free(void* dev_ptr) {
    if (dev_ptr) FB_HIP(hipFree(dev_ptr));
    return FB_OK;
}
int fb_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    FB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    FB_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FB_OK;
}
int fb_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    FB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    FB_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FB_OK;
}
int fb_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    FB_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FB_OK;
}
int fb_stream_create(void** stream) {
    FB_REQUIRE(stream, "null pointer");
    hipStream_t s;
    FB_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return FB_OK;
}
int fb_stream_create_priority(void** stream, int priority) {
    FB_REQUIRE(stream, "null pointer");
    int lo = 0, hi = 0;                                  // numerically lower = higher priority
    FB_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int pr = priority < 0 ? hi : (priority > 0 ? lo : (lo + hi) / 2);
    hipStream_t s;
    FB_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr));
    *stream = (void*)s;
    return FB_OK;
}
int fb_stream_destroy(void* stream) {
    if (stream) FB_HIP(hipStreamDestroy((hipStream_t)stream));
    return FB_OK;
}
int fb_stream_sync(void* stream) {
    FB_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FB_OK;
}
int fb_stream_wait_stream(void* waiter, void* signaller) {
    hipEvent_t ev;
    FB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);      // released once the wait has been satisfied
    FB_HIP(e);
    return FB_OK;
}

#include "fb_comm.inc"
#include "fb_eigen.inc"

}  // extern "C"
