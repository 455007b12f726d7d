// Internal plan object behind the opaque `fb_plan*` of include/fastbox_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#define FB_OK 0
#define FB_ERR_INVALID -1
#define FB_ERR_HIP -2
#define FB_ERR_UNSUPPORTED -3
#define FB_ERR_NOMEM -4
#define FB_ERR_STATE -5
#define FB_ERR_RCCL -6

struct fb_comm;                  // fb_comm.inc: RCCL communicator + its stream and events (fb_comm_create)

struct fb_plan {
    fb_comm* comm = nullptr;     // one box over several GPUs: this rank's communicator, or null
    int N = 0;
    int generic = 0;         // 1: N is not a power of two (even, factors 2, 3, 5, <= 1024): the plain passes of fb_fft_generic.h, no fused entry points
    int prec = 4;            // bytes per real: 4 (float) or 8 (double)
    double L[3] = {0, 0, 0};
    int device = 0;
    int debug_no_mem = 0;    // tuning aid (fb_debug_strided_pass mode >= 10)
    int pass_schedule[3] = {-1, -1, -1};   // per strided-pass class (plain, generator, binning): 0 one workgroup per tile,
                                           // 1 resident workgroups walking the tiles, -1 by grid size (fb_set_pass_schedule)
    int wide_rows = -1;      // strided passes at N = 2048, single precision: bit mask of the pass classes (1 plain, 2 generator, 4 binning)
                             // that run in 128-byte rows; -1: the library's choice (fb_set_tile_rows; FB_WIDE_ROWS=<mask> overrides it)
    int plane_batch = -1;    // x-planes per batch of the y/z passes: -1 sized to the Infinity Cache, 0 whole box (fb_set_plane_batching)
    int plane_streams = 0;   // 1 | 2 streams for alternate batches; 0: by grid size
    double exp_shift = 0.0;  // fused log-normal transforms use exp(x - exp_shift) (fb_set_exp_shift)
    int num_cu = 256;        // compute units of the device (persistent-grid sizing)
    int NZV = 0;             // stored k_z modes of a half spectrum: N/2+1
    int NZP = 0;             // row pitch of a half spectrum (complex elements)
    int NR = 0;              // stored rows per x-plane of a half spectrum (N + 1: see KGeom::NR)
    int cubic = 0;           // Lx == Ly == Lz (integer shells usable)

    void* tw = nullptr;      // forward twiddles W_N^j, j < N, in plan precision
    double* axis2 = nullptr; // [3][N] (m_i / L_a)^2 exactly as numpy computes it (box.py:125-127)
    double* kpar = nullptr;  // [N]  2 pi m / Lz
    double* ksc = nullptr;   // [3][N] m_i * (2 pi / L_a)  (velocity numerators, box.py:254-256)
    double* zgrid = nullptr; // [N]  self.z

    // sqrt(P(k) boxfactor) lookup (box.py:161-176)
    void* amp_shell = nullptr;   // [nshell] plan precision, index n^2 = i^2+j^2+l^2 (cubic only)
    int64_t nshell = 0;
    const void* amp_dense = nullptr;  // caller-owned [N][N][NZP]
    void* pca_work = nullptr;    // channel-sum / covariance partials (grown on demand)
    size_t pca_work_cap = 0;
    double* kperp_tab = nullptr; // [N][N] 2 pi sqrt((m_x/L_x)^2 + (m_y/L_y)^2), box.py:374
    int amp_sym_only = 0;        // amp_sym was given directly (any box shape); no shell table behind it
    void* amp_sym = nullptr;     // [N/2+1][N/2+1][NZP] plan precision: amp_shell spread over (|m_x|, |m_y|, k_z)

    // P(k) binning (box.py:745-764)
    double* bins = nullptr;      // [nbins] edges
    int nbins = 0;
    int* thr = nullptr;          // [FB_MAX_BINS] shell thresholds (cubic boxes)
    int use_thr = 0;             // 0: decide every mode with the exact fp64 |k|
    int namb = 0;
    int amb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long* counts = nullptr;  // [FB_MAX_BINS] device
    double counts_host[256];
    double* partials = nullptr;  // [prow][3*FB_MAX_BINS] per-workgroup partial sums
    int prow = 0;
    double* scratch = nullptr;   // small reduction scratch [FB_SCRATCH]
    double* bin_partials = nullptr;   // fused binning: [workgroups][2*nbins]
    size_t bin_partials_cap = 0;
    long long bin_rows = 0;
    long long bin_rows_main = 0;      // of which the fused binning pass itself wrote (the rest: k_bin_packed_plane)
    long long bin_append_cap = 0;     // > 0: binning launches append their columns to a shared table of this many (k_z chunks)
    void* plane_buf = nullptr;        // packed work spectra: the shared plane column after the last forward pass, [N][N] complex
    double* exp_partials = nullptr;   // r2c with exp(): [workgroups]
    size_t exp_partials_cap = 0;
    long long exp_rows = 0;
    long long exp_base = 0;           // block offset of the plane batch being launched (fb_fft_launch.inc yz_passes)

    // second stream for alternate plane batches of the y/z passes (created on first use)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // per-kernel HIP-event timing (fb_profile_start / fb_profile_stop)
    bool prof_on = false;
    unsigned prof_mask = 0xFFFFFFFFu;   // kernel classes to bracket (bit = FBK_* index)
    int prof_stride = 1;                // bracket every prof_stride-th selected launch (fb_profile_sample)
    long long prof_seen = 0;            // selected launches since fb_profile_start
    std::vector<hipEvent_t> prof_ev;   // pairs
    std::vector<int> prof_cat;
    size_t prof_used = 0;
};

// kernel classes reported by fb_profile_stop (keep in sync with FB_PROF_* in fastbox_hip.h)
enum { FBK_FFT_STRIDED = 0, FBK_FFT_CONTIG, FBK_COLOUR, FBK_BIN, FBK_FILTER, FBK_VELPOT, FBK_REALOP, FBK_RSD,
       FBK_LAYOUT, FBK_FFT_GEN, FBK_FFT_BIN, FBK_PCA, FBK_NCAT };

// RAII: records an event pair around one launch while profiling is on
struct FbProfScope {
    fb_plan* p; hipStream_t s; size_t slot; bool on;
    FbProfScope(fb_plan* plan, int cat, hipStream_t stream)
        : p(plan), s(stream), slot(0), on(plan->prof_on && ((plan->prof_mask >> cat) & 1u)) {
        if (!on) return;
        if ((p->prof_seen++ % p->prof_stride) != 0) { on = false; return; }
        if (p->prof_used + 2 > p->prof_ev.size()) {
            for (int q = 0; q < 64; ++q) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
                                           p->prof_ev.push_back(e); }
        }
        slot = p->prof_used; p->prof_used += 2;
        p->prof_cat.resize(p->prof_used / 2); p->prof_cat[slot / 2] = cat;
        (void)hipEventRecord(p->prof_ev[slot], s);
    }
    ~FbProfScope() { if (on) (void)hipEventRecord(p->prof_ev[slot + 1], s); }
};

#define FB_MAX_BINS 256
#define FB_SCRATCH 8192

void fb_set_error(const std::string& msg);
int fb_hip_check(hipError_t e, const char* what);
#define FB_HIP(x) do { int _r = fb_hip_check((x), #x); if (_r) return _r; } while (0)
#define FB_LAUNCH_CHECK(name) do { int _r = fb_hip_check(hipGetLastError(), name); if (_r) return _r; } while (0)

// ---- per-precision launchers (fb_fft_launch.inc, fb_field_launch.inc) ------------
#define FB_DECL(sfx) \
    int fbi_fft_c2c_##sfx(fb_plan* p, void* data, int sign, double scale, hipStream_t s); \
    int fbi_fft_r2c_##sfx(fb_plan* p, const void* real_in, void* half_out, int pre_exp, hipStream_t s); \
    int fbi_fft_c2r_##sfx(fb_plan* p, void* half_inout, void* real_out, double scale, hipStream_t s); \
    int fbi_set_amp_shells_##sfx(fb_plan* p, const double* amp, int64_t n); \
    int fbi_set_amp_sym_##sfx(fb_plan* p, const double* amp, int64_t n); \
    int fbi_colour_noise_##sfx(fb_plan* p, const void* re, const void* im, void* out, hipStream_t s); \
    int fbi_colour_device_##sfx(fb_plan* p, uint64_t seed, uint64_t real, void* out, hipStream_t s); \
    int fbi_power_filtered_##sfx(fb_plan* p, const void* real_in, void* filtered_half, int kind, const double* prm, \
                                 const void* table, double* results, int inverse_x, hipStream_t s); \
    int fbi_fft_c2r_yz_##sfx(fb_plan* p, void* half_inout, void* real_out, double scale, hipStream_t s); \
    int fbi_realise_velocity_begin_##sfx(fb_plan* p, uint64_t seed, uint64_t real, int comp, double fac, void* work_half, \
                                         hipStream_t s); \
    int fbi_power_redshift_space_##sfx(fb_plan* p, void* work_d, void* work_v, void* real_d_out, void* out_half, double scale, \
                                       double Hz, double sigma_nl, uint64_t seed, int nearest, int kind, const double* prm, \
                                       const void* table, double* results, int inverse_x, hipStream_t s); \
    int fbi_slab_forward_packed_##sfx(fb_plan* p, const void* real_local, void* half_local, void* xbuf, int nxl, \
                                      int nparts, int pre_exp, double* expsum, hipStream_t s); \
    int fbi_slab_inverse_packed_##sfx(fb_plan* p, const void* xbuf, void* half_local, void* real_local, int nxl, \
                                      int nparts, double scale, hipStream_t s); \
    int fbi_slab_turnaround_##sfx(fb_plan* p, const void* recvbuf, void* half_local, void* real_local, void* sendbuf, \
                                  int nxl, int nparts, double scale, int pre_exp, double* expsum, hipStream_t s); \
    int fbi_channel_means_##sfx(fb_plan* p, const void* cube, double* mean_dev, hipStream_t s); \
    int fbi_channel_cov_##sfx(fb_plan* p, const void* cube, const double* mean_dev, double* cov_dev, hipStream_t s); \
    int fbi_pca_clean_##sfx(fb_plan* p, const void* cube, const double* mean_dev, const double* U_dev, int nm, \
                            void* out, double* amps_dev, hipStream_t s); \
    int fbi_real_axpby_##sfx(fb_plan* p, const void* x, const void* y, void* out, double a, double b, double c, \
                             int mul, hipStream_t s); \
    int fbi_fft_axes01_##sfx(fb_plan* p, void* data, int sign, double scale, hipStream_t s); \
    int fbi_beam_convolve_##sfx(fb_plan* p, const void* field, const void* beam, void* work_a, void* work_b, void* out, \
                                int flags, hipStream_t s); \
    int fbi_real_to_complex_##sfx(fb_plan* p, const void* in, void* out, hipStream_t s); \
    int fbi_mask_xy_##sfx(fb_plan* p, void* cube, const void* mask2d, hipStream_t s); \
    int fbi_fft2d_c2c_##sfx(fb_plan* p, void* data, int sign, double scale, hipStream_t s); \
    int fbi_sky_colour_map_##sfx(fb_plan* p, const void* amp2d, const void* re, const void* im, uint64_t seed, \
                                 void* out, hipStream_t s); \
    int fbi_sky_real_plus_##sfx(fb_plan* p, const void* in, void* out, double add, hipStream_t s); \
    int fbi_sky_normal_map_##sfx(fb_plan* p, const void* unit, uint64_t seed, double mean, double std, void* out, \
                                 hipStream_t s); \
    int fbi_sky_gaussian_##sfx(fb_plan* p, void* map, void* tmp, const double* weights, int radius, hipStream_t s); \
    int fbi_sky_fg_cube_##sfx(fb_plan* p, const void* amps, const void* alpha, double alpha_scalar, \
                              const double* ratio, void* out, hipStream_t s); \
    int fbi_sky_noise_cube_##sfx(fb_plan* p, const double* sigma, const void* unit, uint64_t seed, void* out, \
                                 hipStream_t s); \
    int fbi_bin_power_##sfx(fb_plan* p, const void* spec, int layout, int filter_kind, const double* prm, \
                            const void* table, double* sums_dev, hipStream_t s); \
    int fbi_apply_filter_##sfx(fb_plan* p, const void* in, void* out, int layout, int kind, const double* prm, \
                               const void* table, hipStream_t s); \
    int fbi_velocity_##sfx(fb_plan* p, const void* dk, void* out, int layout, int comp, double fac, hipStream_t s); \
    int fbi_potential_##sfx(fb_plan* p, const void* dk, void* out, int layout, hipStream_t s); \
    int fbi_lognormal_##sfx(fb_plan* p, const void* in, void* out, double* mean_out, hipStream_t s); \
    int fbi_rsd_##sfx(fb_plan* p, const void* d, const void* vz, const void* noise, void* out, double Hz, \
                      double sigma, uint64_t seed, int nearest, hipStream_t s); \
    int fbi_sum_real_##sfx(fb_plan* p, const void* x, int squared, double* out, hipStream_t s); \
    int fbi_max_real_##sfx(fb_plan* p, const void* x, double* out, hipStream_t s); \
    int fbi_sumsq_half_##sfx(fb_plan* p, const void* h, double* out, hipStream_t s); \
    int fbi_expand_half_##sfx(fb_plan* p, const void* h, void* f, hipStream_t s); \
    int fbi_crop_full_##sfx(fb_plan* p, const void* f, void* h, hipStream_t s); \
    int fbi_realise_velocity_fused_##sfx(fb_plan* p, uint64_t seed, uint64_t real, int comp, double fac, \
                                         void* work_half, void* real_out, double scale, hipStream_t s); \
    int fbi_realise_fused_##sfx(fb_plan* p, uint64_t seed, uint64_t real, void* work_half, void* real_out, \
                                double scale, hipStream_t s); \
    int fbi_power_fused_##sfx(fb_plan* p, const void* real_in, void* work_half, int pre_exp, int store, \
                              double* results, hipStream_t s); \
    int fbi_debug_pass_##sfx(fb_plan* p, void* half, int axis, int mode, hipStream_t s); \
    int fbi_realise_begin_##sfx(fb_plan* p, uint64_t seed, uint64_t real, void* work_half, hipStream_t s); \
    int fbi_realise_finish_##sfx(fb_plan* p, void* work_half, void* real_out, double scale, hipStream_t s); \
    int fbi_power_from_pending_##sfx(fb_plan* p, void* work_half, void* real_out, double scale, int pre_exp, \
                                     double* results, hipStream_t s); \
    int fbi_slab_forward_local_##sfx(fb_plan* p, const void* real_local, void* half_local, int nxl, int pre_exp, \
                                     double* expsum, hipStream_t s); \
    int fbi_slab_inverse_local_##sfx(fb_plan* p, void* half_local, void* real_local, int nxl, double scale, hipStream_t s); \
    int fbi_slab_x_pass_##sfx(fb_plan* p, void* kslab, int nyl, int sign, hipStream_t s); \
    int fbi_slab_x_generate_##sfx(fb_plan* p, void* kslab, int nyl, int ky0, uint64_t seed, uint64_t real, hipStream_t s); \
    int fbi_slab_x_bin_##sfx(fb_plan* p, void* kslab, int nyl, int ky0, double* results, hipStream_t s); \
    int fbi_slab_tile_cols_##sfx(const fb_plan* p); \
    int fbi_slab_x_generate_chunk_##sfx(fb_plan* p, void* kchunk, int nyl, int ky0, uint64_t seed, uint64_t real, int tile0, \
                                        int ntile, hipStream_t s); \
    int fbi_slab_y_inverse_chunk_##sfx(fb_plan* p, const void* recv_chunk, void* half_local, int nxl, int nparts, int tile0, \
                                       int ntile, hipStream_t s); \
    int fbi_slab_y_forward_chunk_##sfx(fb_plan* p, const void* half_local, void* send_chunk, int nxl, int nparts, int tile0, \
                                       int ntile, hipStream_t s); \
    int fbi_slab_z_pass_##sfx(fb_plan* p, void* half_local, void* real_local, int nxl, int which, double scale, int pre_exp, \
                              double* expsum, hipStream_t s); \
    int fbi_slab_x_bin_chunk_##sfx(fb_plan* p, void* kchunk, int nyl, int ky0, int tile0, int ntile, int first, int last, \
                                   double* results, hipStream_t s);
FB_DECL(f32)
FB_DECL(f64)
int fbi_bin_count(fb_plan* p, hipStream_t s);
int fbi_slab_permute(fb_plan* p, const void* in, void* out, int nxl, int nparts, int pack, hipStream_t s);
