/* One box through the C ABI alone -- no Python, no PyTorch, no MPI: the consumer INTEGRATION.md's route B describes, including
 * the "one box over several GPUs" collectives (fb_comm_create, fb_slab_exchange, fb_allreduce_f64).
 *
 *   realise_density (fastbox/box.py:130-194, device generator) -> binned_power_spectrum (box.py:696-768)
 *
 * of a cubic box of N^3 voxels and side L Mpc with a closed-form P(k), slab-decomposed over `world` ranks (one process per
 * GPU; rank r uses device r): generator fused into the x pass of the k-space slab, ONE all-to-all per transform, the two z
 * passes fused (fb_slab_turnaround), the shell binning fused into the last x pass, the bin sums all-reduced.
 *
 *   gcc -O2 -std=c99 examples/slab_box_c_abi.c -Iinclude -Lfastbox_amd/lib -lfastbox_hip -lm \
 *       -Wl,-rpath,$PWD/fastbox_amd/lib -o /tmp/slab_box
 *   /tmp/slab_box 64 1000 20 5                       one rank
 *   /tmp/slab_box 1024 1000 20 5 8 $RANK /tmp/id     eight ranks: rank 0 writes the RCCL id to the file, the others read it
 *
 * Prints one line per bin: centre, P(k), modes.  tests/test_c_example_gpu.py compares them with fastbox_amd.SlabBox. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fastbox_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, fb_last_error()); return 2; } } while (0)

/* the stand-in spectrum of the tests: P(k) = A (k/k0)^n / (1 + (k/k0)^2)^m, not a number at k = 0 (as pyccl gives) */
static double pk_model(double k) {
    if (!(k > 0.0)) return NAN;
    const double x = k / 0.02;
    return 2.0e4 * pow(x, 0.96) / pow(1.0 + x * x, 1.8);
}

/* np.digitize(x, edges) for increasing edges: number of edges <= x */
static int digitize(double x, const double* edges, int nb) {
    int lo = 0, hi = nb;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (edges[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s N L nbins seed [world rank idfile]\n", argv[0]); return 1; }
    const int N = atoi(argv[1]);
    const double box_scale = atof(argv[2]);
    const int nb = atoi(argv[3]);
    const uint64_t seed = strtoull(argv[4], NULL, 10);
    const int world = argc > 5 ? atoi(argv[5]) : 1, rank = argc > 6 ? atoi(argv[6]) : 0;
    const char* idfile = argc > 7 ? argv[7] : NULL;
    const double PI = 3.14159265358979323846;

    /* ---- geometry, with the reference's numpy expressions (box.py:76-101, 119-127, 254-256, 375) ---- */
    double* z = malloc(sizeof(double) * N);
    const double start = -0.5 * box_scale, stop = 0.5 * box_scale, step = (stop - start) / (N - 1);
    for (int i = 0; i < N; ++i) z[i] = start + i * step;
    z[N - 1] = stop;                                              /* np.linspace: the end point exactly */
    const double L = z[N - 1] - z[0];
    const double boxfactor = pow((double)N, 6.0) / (L * L * L);
    double* m = malloc(sizeof(double) * N);                       /* (N * fftfreq(N)).astype(int) */
    for (int i = 0; i < N; ++i) m[i] = (double)(i < (N + 1) / 2 ? i : i - N);
    double *axis2 = malloc(sizeof(double) * 3 * N), *ksc = malloc(sizeof(double) * 3 * N), *kpar = malloc(sizeof(double) * N);
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < N; ++i) { const double q = m[i] / L; axis2[a * N + i] = q * q; ksc[a * N + i] = m[i] * (2.0 * PI / L); }
    for (int i = 0; i < N; ++i) kpar[i] = 2.0 * PI * m[i] / L;

    /* ---- sqrt(nan_to_num(P(k)) * boxfactor) per integer shell n^2 = i^2 + j^2 + l^2 (box.py:161-171) ---- */
    const int64_t nshell = 3 * (int64_t)(N / 2) * (N / 2) + 1;
    double* amp = malloc(sizeof(double) * nshell);
    double* kshell = malloc(sizeof(double) * nshell);
    for (int64_t s = 0; s < nshell; ++s) {
        kshell[s] = 2.0 * PI * sqrt((double)s) / L;
        double p = pk_model(kshell[s]);
        if (p != p) p = 0.0;
        amp[s] = sqrt(p * boxfactor);
    }

    /* ---- np.logspace(log10 kmin, log10 kmax, nbins) and the bin as a step function of the shell (box.py:745-751) ---- */
    const double kmin = 2.0 * PI / L, kmax = 2.0 * PI * sqrt(3.0) * N / L;
    double* edges = malloc(sizeof(double) * nb);
    {
        const double a = log10(kmin), b = log10(kmax), st = (b - a) / (nb - 1);
        for (int i = 0; i < nb; ++i) edges[i] = pow(10.0, i == nb - 1 ? b : a + i * st);
    }
    int32_t* thr = malloc(sizeof(int32_t) * nb);
    int32_t amb[9];
    int namb = 0;
    {
        const double eps = 64.0 * 2.220446049250313e-16;
        for (int b = 0; b < nb; ++b) thr[b] = (int32_t)nshell;
        for (int64_t s = nshell - 1; s >= 0; --s) {               /* thr[b] = first shell whose (upper) bin index reaches b + 1 */
            const int lo = digitize(kshell[s] * (1.0 - eps), edges, nb), hi = digitize(kshell[s] * (1.0 + eps), edges, nb);
            if (lo != hi) { if (namb < 8) amb[namb] = (int32_t)s; ++namb; }
            for (int b = 0; b < hi && b < nb; ++b) thr[b] = (int32_t)s;
        }
        if (namb > 8) { fprintf(stderr, "more than 8 shells within rounding of a bin edge: use thr = NULL\n"); return 1; }
        /* (amb was filled from the top shell down: the library takes any order) */
    }

    /* ---- the plan of this rank, its communicator, its buffers ---- */
    fb_plan* plan = NULL;
    CHECK(fb_device_set(rank));
    CHECK(fb_plan_create(&plan, N, L, L, L, 4, rank, axis2, ksc, kpar, z));
    CHECK(fb_set_amplitude_shells(plan, amp, nshell));
    CHECK(fb_set_bins(plan, edges, nb, thr, amb, namb));
    unsigned char id[128];
    if (world > 1) {
        if (!idfile) { fprintf(stderr, "several ranks need an id file\n"); return 1; }
        if (rank == 0) {
            CHECK(fb_comm_unique_id(id));
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, 128, f) != 128) { fprintf(stderr, "cannot write %s\n", tmp); return 1; }
            fclose(f);
            rename(tmp, idfile);                                  /* appears atomically */
        } else {
            FILE* f = NULL;
            for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); ++tries) usleep(100000);
            if (!f || fread(id, 1, 128, f) != 128) { fprintf(stderr, "cannot read %s\n", idfile); return 1; }
            fclose(f);
        }
    }
    CHECK(fb_comm_create(plan, world, rank, world > 1 ? id : NULL));
    void *stream = NULL, *kslab = NULL, *xbuf = NULL, *half = NULL, *real = NULL, *res = NULL;
    CHECK(fb_stream_create(&stream));
    const int64_t kbytes = fb_slab_kspace_bytes(plan, world);
    CHECK(fb_malloc(&kslab, (size_t)kbytes));
    CHECK(fb_malloc(&xbuf, (size_t)kbytes));
    CHECK(fb_malloc(&half, (size_t)fb_slab_half_bytes(plan, world)));
    CHECK(fb_malloc(&real, (size_t)(fb_real_bytes(plan) / world)));
    CHECK(fb_malloc(&res, sizeof(double) * (2 * nb + 1)));

    /* ---- one realisation and its power spectrum ---- */
    CHECK(fb_slab_x_generate(plan, kslab, world, rank, seed, 0, stream));                 /* noise, sqrt(P), inverse x pass */
    CHECK(fb_slab_exchange(plan, kslab, xbuf, kbytes / world, stream));                   /* k_y slabs -> x slabs          */
    CHECK(fb_slab_turnaround(plan, xbuf, half, real, kslab, world, 0, NULL, stream));     /* y, z (delta_x written), z, y   */
    CHECK(fb_slab_exchange(plan, kslab, xbuf, kbytes / world, stream));                   /* x slabs -> k_y slabs          */
    CHECK(fb_slab_x_bin(plan, xbuf, world, rank, (double*)res, stream));                  /* forward x pass + shell binning */
    CHECK(fb_allreduce_f64(plan, (double*)res, 2 * nb, 0, stream));
    double* h = malloc(sizeof(double) * (2 * nb + 1));
    double* cnt = malloc(sizeof(double) * nb);
    CHECK(fb_memcpy_d2h(h, res, sizeof(double) * 2 * nb, stream));                        /* synchronises the stream */
    CHECK(fb_bin_counts(plan, cnt));
    if (rank == 0)
        for (int b = 1; b < nb; ++b) {                                                    /* bin 0 is dropped (box.py:761-764) */
            const double centre = 0.5 * (edges[b] + edges[b - 1]);
            if (cnt[b] > 0.0) printf("%.17g %.17g %.0f\n", centre, h[2 * b] / (cnt[b] * boxfactor), cnt[b]);
            else printf("%.17g nan 0\n", centre);
        }
    CHECK(fb_comm_destroy(plan));
    fb_free(kslab); fb_free(xbuf); fb_free(half); fb_free(real); fb_free(res);
    fb_stream_destroy(stream);
    CHECK(fb_plan_destroy(plan));
    return 0;
}
