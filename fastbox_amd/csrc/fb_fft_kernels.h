// 3-D FFT passes built from fb_fft.h.  Two kernels:
//   k_fft_strided : c2c along an axis whose lines are strided in memory (x, y).
//                   One workgroup owns a tile of TZ adjacent k_z columns x the
//                   whole line; global traffic is 128-byte row segments.
//                   MODE GEN  fuses the Gaussian-field generator (box.py:161-176) into
//                             the loads of the first inverse pass (nothing is read);
//                   MODE BIN  fuses |delta_k|^2 shell binning (box.py:741-764) into the
//                             stores of the last forward pass (nothing need be written).
//   k_fft_contig  : along the contiguous z axis; c2c, r2c (forward, optional exp() on
//                   load for the log-normal transform) and c2r (inverse) through the
//                   packed half-length complex transform.
#pragma once
#include <type_traits>
#include "fb_fft.h"
#include "fb_field_kernels.h"

namespace fb {

#ifndef FB_E16_FROM
#define FB_E16_FROM 2048      // lines this long use 16 points per thread and a 128 KiB tile (one workgroup per CU): 2048^3 5.0 -> 7.8 boxes/s; at 1024 it loses (83 -> 77)
#endif
// elements per thread of the strided passes (16 = two radix-8 butterflies per stage was
// measured slower on MI355X: 8 waves per CU cannot hide the LDS-exchange latency)
constexpr int strided_elems(int n) { return n >= FB_E16_FROM ? 16 : fb_min(8, n); }

// columns per tile of the strided pass: one 128-byte row segment, shrunk so the
// tile stays within 64 KiB of LDS (two workgroups per CU), and widened for tiny grids so
// that a workgroup is at least one full wave.
#ifndef FB_TILE_LINE
#define FB_TILE_LINE 128      // bytes of a tile's row segment (tuning: 64 = half-line tiles, four workgroups per CU at 512^3)
#endif
template <typename T> constexpr int tile_cols(int n) {
    return fb_max(fb_max(2, fb_min(FB_TILE_LINE / (2 * (int)sizeof(T)),
                                   // 16 points per thread: 128 KB tile (fp32), 64 KB for fp64 so that the tile, the
                                   // fp64 twiddles (32 KB at 2048) and the binning rows stay within the 160 KB of a CU
                                   (n >= FB_E16_FROM ? (sizeof(T) == 8 ? 65536 : 131072) : 65536) / (n * 2 * (int)sizeof(T)))),
                  64 / (n / strided_elems(n)));
}

// workgroups of the strided pass that share a CU: bounded by the tile's LDS (160 KB per CU) and by 2048 threads
template <typename T> constexpr int strided_wg_per_cu(int n) {
    const int tile = n * tile_cols<T>(n) * 2 * (int)sizeof(T), nt = tile_cols<T>(n) * (n / strided_elems(n));
    return fb_max(1, fb_min(2048 / nt, tile > 81920 ? 1 : tile > 36864 ? 2 : 4));
}

#ifndef FB_GEN_STORE_AUX
#define FB_GEN_STORE_AUX 0     // cache policy of the generator pass's stores (tuning: 2 = nt)
#endif
#ifndef FB_BIN_LOAD_AUX
#define FB_BIN_LOAD_AUX 0      // cache policy of the binning pass's loads (tuning: 2 = nt)
#endif
#ifndef FB_OCC
#define FB_OCC(x) 1      // let the allocator use what the prefetching loop needs (no spills)
#endif
enum { SMODE_PLAIN = 0, SMODE_GEN = 1, SMODE_BIN = 2, SMODE_BINF = 3 };
// BINF: as BIN, but the spectrum is multiplied by a k_perp/k_par filter first and written out: the last
// forward pass of "P(k) of apply_transfer_fn's result", leaving the filtered spectrum for the inverse.
constexpr bool smode_bins(int mode) { return mode == SMODE_BIN || mode == SMODE_BINF; }

// Half-LDS form of a strided pass (SplitTileLayout): 16 points per thread, 512-thread workgroups on the same 16-column
// tile, real and imaginary parts exchanged one after the other through 32 KiB -- three or four workgroups per CU instead
// of two 1024-thread ones, so that one workgroup's barrier-bound exchange phases run beside another's arithmetic.
// FB_SPLIT_MODES: bit 0 plain, bit 1 generator, bit 2 binning / filtering passes (single precision, N = 512).
#ifndef FB_SPLIT_MODES
#define FB_SPLIT_MODES 0
#endif
#ifndef FB_SPLIT_WGS
#define FB_SPLIT_WGS 3            // workgroups per CU the half-LDS form is compiled for (3: 80 registers per thread, 4: 64)
#endif
// N = 2048, single precision: the full-complex tile holds 2048 rows x 8 columns (64-byte row segments), and 64-byte segments cap
// the far-strided passes at 3.4-4.3 TB/s where 128-byte rows read at 6.2 and copy at 5.2-5.3 (tools/stride_copy_2048.hip,
// profiles/r04_stride_copy_2048.txt).  2048 rows x 16 columns fit the LDS as REAL values (128 KiB): the half-LDS exchange with 32
// points per thread, 1024 threads, one resident workgroup per CU (4 waves per SIMD: 128 registers).
// FB_ROW128_MODES: bit 0 plain, bit 1 generator, bit 2 binning pass.
#ifndef FB_ROW128_MODES
#define FB_ROW128_MODES 7
#endif
#ifndef FB_ROW128_SWIZZLE
#define FB_ROW128_SWIZZLE 0     // 1: SplitTileLayout's bank swizzle (removes the first stage's two-way write conflicts; its XOR keeps the
                                // compiler from folding the exchange addresses into immediates: more registers)
#endif
template <typename T> constexpr bool strided_row128(int n, int mode, bool blk) {
    return sizeof(T) == 4 && n == 2048 && !blk && mode != SMODE_BINF &&
           ((FB_ROW128_MODES >> (mode == SMODE_PLAIN ? 0 : (mode == SMODE_GEN ? 1 : 2))) & 1);
}
template <typename T> constexpr bool strided_split(int n, int mode, bool blk) {
    return strided_row128<T>(n, mode, blk) ||
           (sizeof(T) == 4 && n == 512 && !blk &&
            ((FB_SPLIT_MODES >> (mode == SMODE_PLAIN ? 0 : (mode == SMODE_GEN ? 1 : 2))) & 1));
}
// The generator pass of a 1024-point line in single precision takes the 2048 shape -- 16 points per thread, a 16-column
// (128-byte row) tile of 128 KiB, one resident workgroup per CU -- because only that shape has the registers to keep the
// finished tile while the next one is drawn (FB_GEN_PARK below): the pass is its arithmetic PLUS its store time otherwise
// (profiles/r03_gen_knockout_1024_2048.txt).  FB_GEN_WIDE_FROM: smallest such N (tuning; 1 << 30: off).
#ifndef FB_GEN_WIDE_FROM
#define FB_GEN_WIDE_FROM 1024       // (1024^3: generator pass 2.17 -> 1.90 ms, 114.6 -> 117 boxes/s)
#endif
template <typename T> constexpr bool strided_wide(int n, int mode, bool blk) {
    return sizeof(T) == 4 && mode == SMODE_GEN && !blk && n >= (FB_GEN_WIDE_FROM) && n >= 1024 && n < FB_E16_FROM;
}
template <typename T> constexpr int strided_elems_of(int n, int mode, bool blk) {
    return strided_row128<T>(n, mode, blk) ? 32 : ((strided_split<T>(n, mode, blk) || strided_wide<T>(n, mode, blk)) ? 16 : strided_elems(n));
}
template <typename T> constexpr int tile_cols_of(int n, int mode, bool blk) {
#ifndef FB_GEN_WIDE_COLS
#define FB_GEN_WIDE_COLS 16        // (tuning: 8 = a 64 KiB tile, 512 threads, two workgroups per CU)
#endif
    return strided_row128<T>(n, mode, blk) ? 16 : (strided_wide<T>(n, mode, blk) ? fb_min(FB_GEN_WIDE_COLS, 131072 / (n * 2 * (int)sizeof(T))) : tile_cols<T>(n));
}
template <typename T> constexpr int strided_wgs_of(int n, int mode, bool blk) {
    return strided_row128<T>(n, mode, blk) ? 1 : (strided_split<T>(n, mode, blk) ? FB_SPLIT_WGS : (strided_wide<T>(n, mode, blk) ? (FB_GEN_WIDE_COLS <= 8 ? 2 : 1) : strided_wg_per_cu<T>(n)));
}

template <typename T> struct StridedArgs {
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;         // W_N^j
    long long stride;        // elements between consecutive points of a line
    long long outer_stride;  // elements between tiles along the outer index
    int ncols;               // valid contiguous columns
    T scale;
    int ntx;                 // tiles per outer index = ceil(ncols / TZ)      (set by the launcher)
    int ntiles;              // ntx * (number of outer indices)               (set by the launcher)
    int drop_io;             // tuning aid: do all the arithmetic but no global loads/stores
    int wide;                // a line's last point lies 4 GiB or more behind its first: every point gets its own 64-bit base
                             // (else one base per tile and a 32-bit scalar offset per point)          (set by the launcher)
    int ntx_shift;           // log2(ntx) if ntx is a power of two, else -1                              (set by the launcher)
    // Exchange-buffer addressing of the slab-decomposed transform (all zero = the plain layout on both sides).
    // A line of the blocked side is cut into 2^blk_shift ... pieces: point k lives at
    // (k / B) * blk_stride + (k % B) * stride with B = (N / elements-per-thread) << blk_shift points per piece,
    // i.e. the [rank][x][k_y in rank][k_z] buffer that goes into / comes out of the all-to-all, so that the y pass
    // itself does the pack / unpack.  out_outer_stride = 0 means "as outer_stride".
    long long out_outer_stride;
    long long blk_stride;
    int blk_in, blk_out, blk_shift;
    int packed;              // half spectrum with the k_z = N/2 plane in the imaginary direction of the k_z = 0 plane (ncols = N/2)
    // A launch over a RANGE of tile columns (the k_z chunks of the slab-decomposed transform, whose all-to-all of one
    // chunk runs beside the passes of the next): tiles tile0 .. tile0 + ntx - 1 of a row.  Columns stay absolute
    // (ncols, the generator's counters, the binning's k_z), so a chunk that lives in an array of its own is handed
    // over as  base - (first column of the chunk).
    int tile0;
    long long out_stride;    // slab addressing only: elements between consecutive points of a line on the OUTPUT side when that
                             // differs from `stride` (a chunk array has its own row pitch); 0 = as `stride`
};

// operands of the fused modes (x pass of a half spectrum: line index = k_x,
// outer index = k_y, column = k_z)
template <typename T> struct StridedOp {
    KGeom g;
    AmpSrc<T> amp;       // GEN
    RngKey key;          // GEN
    const int* thr;      // BIN: shell thresholds (cubic boxes only)
    const double* bins;  // BIN: edges, for the shells listed in amb[]
    double* partial;     // BIN: [2 * nbins][partial_stride], this launch fills columns 0 .. gridDim.x - 1
    long long partial_stride;
    cx<T>* plane_out;    // BIN of a packed half spectrum: column 0 (= X(k_z=0) + i X(k_z=N/2)) goes to plane_out[k_y][k_x]
                         // and is binned by k_bin_packed_plane, which has both members of every mirror pair
    int nbins, namb, store;
    int outer0;          // global index of this launch's first outer (k_y) row
    int vel_on, vel_comp;   // GEN: emit i fac delta_k k_c / k^2 (velocity component) instead of delta_k
    double vel_fac;
    FilterSpec filt;        // BINF
    const double* kperp_tab;   // BINF: 2 pi sqrt((m_x/L_x)^2 + (m_y/L_y)^2) per (k_x, k_y)
    int amb[8];
};

// PERSIST = 0: one tile per workgroup (the grid has as many workgroups as tiles).
// PERSIST = 1: resident workgroups (as many as fit on the chip) that each walk the tiles blockIdx.x, blockIdx.x +
// gridDim.x, ... and issue the global loads of their NEXT tile as early as the registers allow -- into the very
// registers the current tile has just left: a plain pass right behind its stores, the binning pass as soon as |X|^2
// has been staged in LDS (the whole binning epilogue then runs under the next tile's memory latency).  No register is
// added (still <= 64 VGPRs, two 1024-thread workgroups per CU), and the workgroup's hand-over -- store drain, exit,
// dispatch of the successor: 3-6 k of a tile's 16-23 k cycles, tools/phase_timeline.py -- is gone.
// (Round 1's persistent form prefetched into a SECOND register set, which costs the second resident workgroup, and
// walked the tiles so that one workgroup met all the heavy generator tiles; both measured slower and are removed.)
// BLK: the exchange-buffer addressing of the slab-decomposed transform (StridedArgs::blk_*); plain passes only.
//
// Scalar instructions: a CU issues ONE scalar-ALU instruction per cycle for all its waves (tools/salu_rate.hip:
// 1.02-1.08 cycles per instruction whatever the number of waves; vector instructions 0.65), and a strided pass is
// mostly wave-uniform bookkeeping -- tile coordinates, 64-bit bases, buffer descriptors.  Round 2's form spent ~300
// scalar instructions per wave and tile on it (16 waves: 4.8 k of a plain tile's 5.3 k cycles without memory traffic:
// the pass was bound by its scalar unit, not by its butterflies).  Hence: one descriptor per tile and a 32-bit scalar
// offset per point where the line fits 4 GiB, shifts instead of divisions for power-of-two tile counts, the slab
// addressing compiled only into the kernels that use it.
// CSIGN (plain passes): +1 / -1 = the direction is a compile-time constant, 0 = the `sign` argument decides.  With both
// directions in one kernel the compiler keeps state of both instantiations live across the branch: the 32-point form needs
// 95 registers more than its 128 and spills every tile's loads to scratch; with one direction it takes 113 and spills nothing.
template <typename T, int N, int MODE, int PERSIST, bool BLK = false, int CSIGN = 0>
__global__ __launch_bounds__((tile_cols_of<T>(N, MODE, BLK) * (N / strided_elems_of<T>(N, MODE, BLK))),
                             fb_min(8, fb_max(1, strided_wgs_of<T>(N, MODE, BLK) * tile_cols_of<T>(N, MODE, BLK) * (N / strided_elems_of<T>(N, MODE, BLK)) / 256)))
void k_fft_strided(StridedArgs<T> a, int sign, StridedOp<T> op) {
    static_assert(!BLK || MODE == SMODE_PLAIN || MODE == SMODE_GEN || MODE == SMODE_BIN, "slab addressing: plain passes (and the generator / binning\n"
                  "pass of a k_z chunk, which keep the narrow tile the chunk bounds are counted in)");
    constexpr bool SPLIT = strided_split<T>(N, MODE, BLK);
    constexpr int E = strided_elems_of<T>(N, MODE, BLK);
    constexpr int TPL = N / E;
    constexpr int TZ = tile_cols_of<T>(N, MODE, BLK);
    constexpr int NT = TZ * TPL;
    constexpr int NW = (NT + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* twl = reinterpret_cast<cx<T>*>(smem + (size_t)N * TZ * (SPLIT ? sizeof(T) : sizeof(cx<T>)));
    double* acc = reinterpret_cast<double*>(twl + N);          // BIN: [NW][2 nbins], whole launch
    int* lthr = reinterpret_cast<int*>(acc + (size_t)NW * 2 * (smode_bins(MODE) ? op.nbins : 0));

    const int tid = threadIdx.x;
    const int c = tid % TZ;
    const int t = tid / TZ;
    // The tables (twiddles; bin thresholds) are first needed after the first LDS exchange, whose
    // barrier also publishes them: in the one-tile-per-workgroup form they are fetched AFTER the
    // tile's own loads have been issued, and no barrier stands between a wave's loads and its
    // first butterflies.
    auto load_tables = [&]() {
        for (int i = tid; i < N; i += NT) twl[i] = a.tw[i];
        if constexpr (smode_bins(MODE)) {
            for (int i = tid; i < NW * 2 * op.nbins; i += NT) acc[i] = 0.0;
            for (int i = tid; i < ((op.nbins + 63) & ~63); i += NT) lthr[i] = i < op.nbins ? op.thr[i] : 0x7fffffff;
        }
    };
    if constexpr (PERSIST) load_tables();
#ifdef FB_STAMPS      // diagnostic build only: phase time stamps of every workgroup (tools/stamps.py)
    long long* stamp = (smode_bins(MODE))
        ? reinterpret_cast<long long*>(op.partial + (size_t)2 * op.nbins * op.partial_stride) + (size_t)blockIdx.x * 32
        : reinterpret_cast<long long*>(const_cast<double*>(op.bins)) + (size_t)blockIdx.x * 32;
    const bool stamp_ok = (smode_bins(MODE)) ? (op.partial != nullptr) : (op.bins != nullptr);
#define FB_STAMP(k) do { if (tid == 0 && stamp_ok) stamp[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    FB_STAMP(0);
    if (tid == 0 && stamp_ok) {                // where the workgroup ran: HW_ID (CU / SE) and XCC_ID, raw
        stamp[24] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        stamp[25] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#else
#define FB_STAMP(k) do {} while (0)
#endif

    // addressing: wave-uniform 64-bit tile/row base (scalar registers) + one 32-bit per-lane
    // element offset shared by all E rows, so the E loads/stores cost no address VGPRs
    // (the wave's first line index goes into the scalar base: at 2048^3 the x stride is 17-34 MB
    // and t * stride would overflow the 32-bit lane offset)
    const int t0 = __builtin_amdgcn_readfirstlane(t);
    const long long tbase = (long long)t0 * a.stride;
    const unsigned loff = (unsigned)(((long long)(t - t0) * a.stride + c) * (long long)sizeof(cx<T>));
    const long long estep = (long long)TPL * a.stride;
    const unsigned estep_b = (unsigned)(estep * (long long)sizeof(cx<T>));       // meaningful when !a.wide
    // offset of a thread's e-th point in the slab addressing (e is a compile-time constant after unrolling, the rest is
    // wave-uniform)
    [[maybe_unused]] auto eoff = [&](int e, int blocked, long long es) -> long long {
        return blocked ? (long long)(e >> a.blk_shift) * a.blk_stride + (long long)(e & ((1 << a.blk_shift) - 1)) * es
                       : (long long)e * es;
    };
    // output side of the slab addressing: its own row pitch when the two sides differ (chunk arrays)
    [[maybe_unused]] const long long ostride = (BLK && a.out_stride) ? a.out_stride : a.stride;
    [[maybe_unused]] const long long tbase_o = BLK ? (long long)t0 * ostride : tbase;
    [[maybe_unused]] const unsigned loff_o = BLK ? (unsigned)(((long long)(t - t0) * ostride + c) * (long long)sizeof(cx<T>)) : loff;
    [[maybe_unused]] const long long estep_o = BLK ? (long long)TPL * ostride : estep;
    // the E points of this thread from / to the tile whose first row (this wave's) starts at src / dst
    auto load_rows = [&](cx<T> (&r)[E], const cx<T>* src, const unsigned voff) {
        constexpr int AUX = smode_bins(MODE) ? FB_BIN_LOAD_AUX : 0;
        if constexpr (BLK) {
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = buf_load<AUX>(make_rsrc(src + eoff(e, a.blk_in, estep)), voff, 0u, src);
        } else if (!a.wide) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(src);
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = buf_load<AUX>(rs, voff, (unsigned)e * estep_b, src);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = buf_load<AUX>(make_rsrc(src + (long long)e * estep), voff, 0u, src);
        }
    };
    auto store_rows = [&](cx<T>* dst, const unsigned voff, const cx<T> (&r)[E], const T scale) {
        // (generator pass at N = 1024 -- 64-byte row segments written by one workgroup per tile: streaming stores, 2.30 ->
        // 2.17 ms; no gain at 512, a loss under the resident schedule of 2048: profiles/r03_gen_store_variants.txt)
        constexpr int AUX = MODE == SMODE_GEN ? ((N == 1024 && !PERSIST && sizeof(T) == 4) ? 2 : FB_GEN_STORE_AUX) : 0;
        if constexpr (BLK) {
#pragma unroll
            for (int e = 0; e < E; ++e) buf_store<AUX>(make_rsrc(dst + eoff(e, a.blk_out, estep_o)), voff, 0u, cscale(r[e], scale));
        } else if (!a.wide) {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(dst);
#pragma unroll
            for (int e = 0; e < E; ++e) buf_store<AUX>(rs, voff, (unsigned)e * estep_b, cscale(r[e], scale));
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) buf_store<AUX>(make_rsrc(dst + (long long)e * estep), voff, 0u, cscale(r[e], scale));
        }
    };
    const long long out_outer = a.out_outer_stride ? a.out_outer_stride : a.outer_stride;

    cx<T> v[E];
    [[maybe_unused]] int wb_lo = 0, wb_hi = 0, wb_edge = 0x7fffffff;   // BIN: this wave's bins (see epilogue)
    [[maybe_unused]] bool wb_ok = false, wb_rng = false;
    // BIN: lane l keeps thresholds l, l + 64, ... (FB_MAX_BINS = 256: four registers), fetched before the tile so that
    // one memory latency covers both; "number of thresholds <= n^2" is then a compare and a ballot per 64 thresholds
    // instead of a binary search of dependent scalar loads (5.8 k of a workgroup's 20 k cycles, tools/stamps.py)
    // (resident form: the thresholds are read from their LDS copy when a tile needs them -- the table was published once
    // before the tile loop, and four registers held across it would spill)
    [[maybe_unused]] int thrv[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if constexpr (smode_bins(MODE) && !PERSIST) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r * 64 < op.nbins) { const int q = r * 64 + (tid & 63); thrv[r] = q < op.nbins ? op.thr[q] : 0x7fffffff; }
    }
    [[maybe_unused]] auto wave_bin = [&](int n2) -> int {          // n2 wave-uniform
        int b = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r * 64 < op.nbins) {
                int th;
                if constexpr (PERSIST) th = lthr[r * 64 + (tid & 63)];      // padded with INT_MAX to a multiple of 64
                else th = thrv[r];
                b += __builtin_popcountll(__ballot(th <= n2));
            }
        return b;
    };
    // tile number -> (tile column bx, outer index by).  Speed only: nothing depends on the placement.
    [[maybe_unused]] const int rot_rows = (int)gridDim.x >= a.ntx ? (int)gridDim.x / a.ntx : 1;   // rows per sweep of a resident grid
    auto place = [&](int id, int& bx, int& by) {
        const bool p2 = a.ntx_shift >= 0;                     // (a division costs ~45 scalar instructions)
        bx = p2 ? (id & (a.ntx - 1)) : id % a.ntx;
        by = p2 ? (id >> a.ntx_shift) : id / a.ntx;
#ifndef FB_NO_TILE_ROTATE
        // Workgroup b runs on XCD b % 8 and ntx is a multiple of 8 at every power-of-two size, so without this every
        // row's tile 0 -- the self-mirrored planes of the generator, twice the draws of any other tile -- would land on
        // XCD 0, which then finishes 60 % after the other seven (tools/phase_timeline.py).  Rotating the tile index by
        // the row number deals the heavy tiles round all eight XCDs.
        // (resident workgroups: + the number of the sweep, so that a workgroup's own tiles -- ids gridDim.x apart, which
        // would otherwise all have the same column -- also take every column in turn.  A function of `by` alone, so the
        // map tile id -> tile stays one to one whatever the grid size.)
        int rot = bx + by;
        if constexpr (PERSIST) rot += by / rot_rows;
        bx = p2 ? (rot & (a.ntx - 1)) : rot % a.ntx;
#endif
#ifndef FB_NO_XCD_PAIR
        if constexpr (TZ * sizeof(cx<T>) < 128) {
            // Tiles narrower than a 128-byte line (N >= 1024: 64-byte row segments, 32 for fp64 at 2048): the LPT tiles
            // that share every line of a row go to workgroups b, b + 8, b + 16, ... -- the same XCD under round-robin
            // placement, started within the same dispatch wave -- so that a line is brought into ONE L2 once instead of
            // into LPT different ones.
            constexpr int LPT = 128 / (TZ * (int)sizeof(cx<T>)), G = 8 * LPT;
            const int lg = bx % G;
            if (bx - lg + G <= a.ntx) bx = (bx - lg) + LPT * (lg % 8) + lg / 8;
        }
#endif
        bx += a.tile0;            // (0 unless the launch covers a range of tile columns)
    };
    int tile_id = blockIdx.x;
    // Resident generator pass with room in the register file (16 points per thread: one workgroup per CU, 128 registers): the
    // finished tile is not stored in one burst -- a CU's store path takes a tile's 128 KB only as fast as HBM drains the
    // far-strided rows, all waves reach their stores together and wait there, and the store time ADDS to the arithmetic
    // (profiles/r03_gen_knockout_1024_2048.txt).  The last FB_GEN_PARK rows a thread holds stay in registers instead and
    // go out two at a time between the batches of the next tile's random numbers and between its transform's stages.
#ifndef FB_GEN_PARK
#define FB_GEN_PARK 16     // (2048^3: generator pass 20.4 -> 18.8 ms with all 16 rows parked, 19.5 with 8; 119 of 128 registers)
#endif
    // (single precision only: the register budgets above are float's; 16 parked cx<double> would be 64 more registers)
    constexpr int PARK = (PERSIST && MODE == SMODE_GEN && !SPLIT && E == 16 && sizeof(T) == 4) ? FB_GEN_PARK : 0;     // rows parked (0, 8 or 16)
    constexpr int PARK0 = E - PARK;                    // first parked row
    [[maybe_unused]] cx<T> wpark[PARK > 0 ? PARK : 1];
    [[maybe_unused]] bool staged = false;              // wave-uniform
    [[maybe_unused]] cx<T>* dst_prev = nullptr;
    [[maybe_unused]] unsigned voff_prev = FB_BUF_OOB;
    [[maybe_unused]] auto store_point = [&](cx<T>* dst, const unsigned voff, const int e, const cx<T> val) {
        if (!a.wide) buf_store<0>(make_rsrc(dst), voff, (unsigned)e * estep_b, val);
        else buf_store<0>(make_rsrc(dst + (long long)e * estep), voff, 0u, val);
    };
    // slot S of 8: parked rows [S PARK/8, (S+1) PARK/8)
    [[maybe_unused]] auto drain = [&](auto slot_tag) {
        constexpr int S = decltype(slot_tag)::value;
        if constexpr (PARK > 0 && S >= 0 && S < 8) {
            constexpr int PER = PARK / 8;
            if (staged) {
                asm volatile("");                      // (wave-uniform: keep it a scalar branch)
#pragma unroll
                for (int q = 0; q < PER; ++q) store_point(dst_prev, voff_prev, PARK0 + S * PER + q, wpark[S * PER + q]);
            }
        }
    };
    // loads of tile `id` into v[] (ids past the last tile: every lane gets the out-of-range offset, which the buffer
    // range check turns into "no access" -- no branch around the loads, so the compiler's in-order vmcnt bookkeeping
    // sees the same queue on every path)
    // Resident binning pass with room in the register file (16 points per thread: one workgroup per CU, 128 registers): a
    // second register set takes the NEXT tile's loads at the very start of a tile, so that they have the whole tile time to
    // arrive instead of the last 40 % of it (FB_BIN_DBUF).
#ifndef FB_BIN_DBUF
#define FB_BIN_DBUF 1        // (2048^3: binning pass 12.67 -> 12.40 ms; 122 of 128 registers)
#endif
    constexpr bool DBUF = PERSIST && MODE == SMODE_BIN && !SPLIT && E == 16 && sizeof(T) == 4 && FB_BIN_DBUF;
    [[maybe_unused]] cx<T> vnext[DBUF ? E : 1];
    [[maybe_unused]] auto load_tile = [&](int id) {
        int lbx, lby;
        place(id, lbx, lby);
        const cx<T>* src = a.in + ((long long)lby * a.outer_stride + lbx * TZ + tbase);
        const unsigned voff = (id < a.ntiles && lbx * TZ + c < a.ncols && !a.drop_io) ? loff : FB_BUF_OOB;
        if constexpr (DBUF) load_rows(vnext, src, voff);
        else load_rows(v, src, voff);
    };
    if constexpr (MODE != SMODE_GEN && PERSIST) load_tile(tile_id);     // the launcher guarantees gridDim.x <= ntiles
    do {    // PERSIST: tiles blockIdx.x, blockIdx.x + gridDim.x, ...; otherwise exactly one tile
        if constexpr (DBUF) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = vnext[e];
            load_tile(tile_id + (int)gridDim.x);
        }
        // Resident form: everything a tile derives from the lane's coordinates (LDS addresses of every stage, mode
        // numbers, row pointers: dozens of values) is loop-invariant, and hoisted out of the tile loop it would stay
        // live across it -- 10 to 90 registers spilled, and reloads from scratch that queue behind the next tile's
        // loads.  An opaque per-tile copy of the thread index makes all of it per-tile again, recomputed where it is
        // used, as in the one-tile form.  (Shadows the kernel-scope tid, c, t on purpose.)
        int tid_tile = tid;
        if constexpr (PERSIST) asm volatile("" : "+v"(tid_tile));
        const int tid = tid_tile, c = tid % TZ, t = tid / TZ;
        int bx, by;
        place(tile_id, bx, by);
        const int col = bx * TZ + c;
        const bool valid = col < a.ncols;
        const long long ubase = (long long)by * a.outer_stride + bx * TZ;
        typedef typename std::conditional<SPLIT, SplitTileLayout<T, TZ, (FB_ROW128_SWIZZLE && strided_row128<T>(N, MODE, BLK))>, TileLayout<T, TZ>>::type Lay;
        const Lay layf{reinterpret_cast<decltype(Lay::base)>(smem), c};

        if constexpr (MODE == SMODE_GEN) {
            // generator: thread pair j holds the modes k_x = t + j TPL (< N/2) and k_x + N/2, which share one
            // Philox call (fb_rng.h).  `column(kz, ...)` draws and colours this thread's E modes of column k_z.
            constexpr int H = N >> 1, E2 = E / 2;
            const int ky = by + op.outer0;             // global k_y (slab-decomposed runs own a k_y range)
            const int my = mode_of(ky, N);
            const int amy = my < 0 ? -my : my;
            const bool rowself = (ky == 0 || ky == H);                          // the row is its own mirror image
            const bool pk0 = a.packed && bx == 0;      // this tile's column 0 carries the k_z = 0 AND the k_z = N/2 plane
            const bool has_plane = bx == 0 || (bx * TZ <= H && H < bx * TZ + TZ);
            // Thread pairs J0 .. J0 + NJ - 1 of column kz.  ACC = false: v[j], v[j + E2] = the coloured modes;
            // ACC = true: v += i * (those), in the lanes of column 0 only (the shared plane column).  The pass holds 64
            // VGPRs (two 1024-thread workgroups per CU): the main call forms all E modes in place in v[] with the E/2
            // Philox calls interleaved, the ACC calls go pair by pair (a second set of E temporaries would spill).
            auto column = [&](const int kz, auto j0_tag, auto nj_tag, auto acc_tag, auto slot_tag) {
                constexpr int J0 = decltype(j0_tag)::value, NJ = decltype(nj_tag)::value;
                constexpr bool ACC = decltype(acc_tag)::value;
                constexpr int SLOT = decltype(slot_tag)::value;            // >= 0: this call drains parked rows, slots SLOT and SLOT + 1
                // (has_plane is wave-uniform: 14 of 16 tiles hold neither self-mirrored plane, and the per-lane selects of the
                // mirror-image draws -- 30 to 40 vector instructions per thread -- are skipped there by scalar branches;
                // the asm keeps them branches)
                bool plane = false, flip = false;          // flip: drawn as the conjugate of the mirror image's draw
                int kyd = ky;
                if (has_plane) {
                    asm volatile("");
                    plane = (kz == 0) || (kz == H);
                    flip = plane && ky > H;
                    kyd = flip ? N - ky : ky;
                }
                // lo(j), hi(j): where pair j's two modes are formed (v[] itself, or temporaries of the ACC call)
                cx<T> wl[ACC ? NJ : 1], wh[ACC ? NJ : 1];
                auto lo = [&](int j) -> cx<T>& { if constexpr (ACC) return wl[j - J0]; else return v[j]; };
                auto hi = [&](int j) -> cx<T>& { if constexpr (ACC) return wh[j - J0]; else return v[j + E2]; };
                // amplitudes sqrt(P boxfactor / 2) (E |z|^2 = 1, fb_rng.h), fetched first so that the loads are in flight
                // while the random numbers are computed: spread table rows |m_x| = k_x and N/2 - k_x of plane |m_y|,
                // column k_z.  The source is wave-uniform: one loop per source, so that its loads are issued together.
                T A0[NJ], A1[NJ];
                auto A0v = [&](int j) -> T { return A0[j - J0]; };
                auto A1v = [&](int j) -> T { return A1[j - J0]; };
                {
                    const T pf = (T)0.70710678118654752440;
                    if (op.amp.sym) {
                        const long long sym_step = (long long)TPL * (H + 1) * op.g.NZP;
                        const T* sym0 = op.amp.sym + ((long long)t * (H + 1) + amy) * op.g.NZP + kz;
                        const T* sym1 = op.amp.sym + ((long long)(H - t) * (H + 1) + amy) * op.g.NZP + kz;
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
#ifdef FB_EXPERIMENT_NOAMP
                            A0[j - J0] = pf; A1[j - J0] = pf;
#else
                            A0[j - J0] = sym0[j * sym_step] * pf;
                            A1[j - J0] = sym1[-j * sym_step] * pf;
#endif
                        }
                    } else if (op.amp.shell) {
                        const int c2 = my * my + kz * kz;            // k_z <= N/2 is its own mode number
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
                            const int kx = t + j * TPL, mh = kx - H;
                            A0[j - J0] = op.amp.shell[kx * kx + c2] * pf;
                            A1[j - J0] = op.amp.shell[mh * mh + c2] * pf;
                        }
                    } else {
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
                            const int kx = t + j * TPL;
                            A0[j - J0] = op.amp.dense[((long long)kx * op.g.NR + ky) * op.g.NZP + kz] * pf;
                            A1[j - J0] = op.amp.dense[((long long)(kx + H) * op.g.NR + ky) * op.g.NZP + kz] * pf;
                        }
                    }
                }
                uint32_t X[NJ][4];
#pragma unroll
                for (int j = J0; j < J0 + NJ; ++j) {
                    const int g = t + j * TPL;
                    const int gd = (flip && g > 0) ? H - g : g;
                    philox_counter(((unsigned long long)gd * N + kyd) * op.g.NZV + kz, 0u, op.key, X[j - J0]);
                }
                drain(std::integral_constant<int, SLOT>{});
                philox4x32_batch<NJ>(X, op.key.k[0], op.key.k[1]);
                drain(std::integral_constant<int, (SLOT >= 0 ? SLOT + 1 : -1)>{});
#pragma unroll
                for (int j = J0; j < J0 + NJ; ++j) {
                    T a0, a1, b0, b1;
                    box_muller(X[j - J0][0], X[j - J0][1], a0, a1);
                    box_muller(X[j - J0][2], X[j - J0][3], b0, b1);
                    lo(j) = cx<T>{a0, a1};
                    hi(j) = cx<T>{b0, b1};
                    if (has_plane) {
                        asm volatile("");
                        const bool sw = flip && (t + j * TPL) > 0;      // the mirror images of (g, g + N/2) are (N/2 - g) + N/2, N/2 - g
                        lo(j) = cx<T>{sw ? b0 : a0, sw ? b1 : a1};
                        hi(j) = cx<T>{sw ? a0 : b0, sw ? a1 : b1};
                        if (flip) { lo(j).y = -lo(j).y; hi(j).y = -hi(j).y; }
                    }
                }
                if (has_plane && rowself) {
                    // rows k_y = 0, N/2 of a plane mirror onto themselves: k_x in (0, N/2) is drawn, k_x + N/2 is the
                    // conjugate of the draw of N/2 - k_x (a second call), k_x = 0 and N/2 are real
#pragma unroll
                    for (int j = J0; j < J0 + NJ; ++j) {
                        const int g2 = (H - (t + j * TPL)) & (H - 1);
                        philox_counter(((unsigned long long)g2 * N + ky) * op.g.NZV + kz, 0u, op.key, X[j - J0]);
                    }
                    philox4x32_batch<NJ>(X, op.key.k[0], op.key.k[1]);
#pragma unroll
                    for (int j = J0; j < J0 + NJ; ++j) {
                        T c0, c1;
                        box_muller(X[j - J0][0], X[j - J0][1], c0, c1);
                        if (plane) {
                            if (t + j * TPL > 0) hi(j) = cx<T>{c0, -c1};
                            else {
                                lo(j) = cx<T>{(T)1.41421356237309504880 * lo(j).x, 0};
                                hi(j) = cx<T>{(T)1.41421356237309504880 * hi(j).x, 0};
                            }
                        }
                    }
                }
                if (!op.vel_on) {
#pragma unroll
                    for (int j = J0; j < J0 + NJ; ++j) {
                        lo(j) = cscale(lo(j), A0[j - J0]);
                        hi(j) = cscale(hi(j), A1[j - J0]);
                    }
                } else {      // wave-uniform: the velocity field of the same realisation, i fac delta_k k_c / k^2
                    const int cmp = op.vel_comp;
                    const int kzc = kz < op.g.NZV ? kz : 0;                      // padding columns: any valid entry
                    if constexpr (sizeof(T) == 4) {
                        // single precision: k_c fac / k^2, k^2 = 4 pi^2 (s_x + (s_y + s_z)), hardware reciprocal
                        const float syz = (float)(op.g.axis2[N + ky] + op.g.axis2[2 * N + kzc]);
                        const int icf = cmp == 1 ? ky : kzc;
                        const float kcf = (float)(op.g.ksc[cmp * N + icf] * op.vel_fac);
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
                            const int kx = t + j * TPL, kh = kx + H;
                            const T A0 = A0v(j), A1 = A1v(j);
                            const float s0 = (float)op.g.axis2[kx] + syz, s1 = (float)op.g.axis2[kh] + syz;
                            float k0 = cmp == 0 ? (float)(op.g.ksc[kx] * op.vel_fac) : kcf;
                            float k1 = cmp == 0 ? (float)(op.g.ksc[kh] * op.vel_fac) : kcf;
                            if ((cmp == 0 ? kx : icf) == H) k0 = 0.f;       // Nyquist plane of the component
                            if ((cmp == 0 ? kh : icf) == H) k1 = 0.f;
                            const float m0 = s0 > 0.f ? A0 * k0 * __builtin_amdgcn_rcpf(39.47841760435743f * s0) : 0.f;
                            const float m1 = s1 > 0.f ? A1 * k1 * __builtin_amdgcn_rcpf(39.47841760435743f * s1) : 0.f;
                            lo(j) = cx<T>{-lo(j).y * m0, lo(j).x * m0};
                            hi(j) = cx<T>{-hi(j).y * m1, hi(j).x * m1};
                        }
                    } else {
                        const double ay = op.g.axis2[N + ky], az = op.g.axis2[2 * N + kzc];   // kmag_exact's order
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
                            const int kx = t + j * TPL, kh = kx + H;
                            const T A0 = A0v(j), A1 = A1v(j);
                            lo(j) = velocity_of<T>(op.g, cmp, op.vel_fac, cmp == 0 ? kx : (cmp == 1 ? ky : kzc),
                                                   (op.g.axis2[kx] + ay) + az, cscale(lo(j), A0));
                            hi(j) = velocity_of<T>(op.g, cmp, op.vel_fac, cmp == 0 ? kh : (cmp == 1 ? ky : kzc),
                                                   (op.g.axis2[kh] + ay) + az, cscale(hi(j), A1));
                        }
                    }
                }
                if constexpr (ACC) {
                    if (c == 0) {
#pragma unroll
                        for (int j = J0; j < J0 + NJ; ++j) {
                            v[j] = cx<T>{v[j].x - lo(j).y, v[j].y + lo(j).x};
                            v[j + E2] = cx<T>{v[j + E2].x - hi(j).y, v[j + E2].y + hi(j).x};
                        }
                    }
                }
            };
            // (two half batches: Philox's 64-bit products are live for a whole batch, and with all E/2 calls interleaved
            // the pass does not fit its 64 VGPRs)
            if constexpr (E2 >= 16) {
                // 32 points per thread (128-byte rows at N = 2048): batches of FB_ROW128_GEN_BATCH calls, so that a batch's
                // counters, products and amplitudes fit beside the 64 registers of v[]
#ifndef FB_ROW128_GEN_BATCH
#define FB_ROW128_GEN_BATCH 2
#endif
                constexpr int GB = FB_ROW128_GEN_BATCH;
                auto batches = [&](auto self, auto j_tag) {
                    constexpr int J = decltype(j_tag)::value;
                    if constexpr (J < E2) {
                        column(col, std::integral_constant<int, J>{}, std::integral_constant<int, GB>{}, std::false_type{}, std::integral_constant<int, -1>{});
                        self(self, std::integral_constant<int, J + GB>{});
                    }
                };
                batches(batches, std::integral_constant<int, 0>{});
            } else if constexpr (E2 >= 4) {
                column(col, std::integral_constant<int, 0>{}, std::integral_constant<int, E2 / 2>{}, std::false_type{}, std::integral_constant<int, 0>{});
                column(col, std::integral_constant<int, E2 / 2>{}, std::integral_constant<int, E2 / 2>{}, std::false_type{}, std::integral_constant<int, 2>{});
            } else {
                column(col, std::integral_constant<int, 0>{}, std::integral_constant<int, E2>{}, std::false_type{}, std::integral_constant<int, -1>{});
            }
            // packed, tile 0 (wave-uniform): column 0 = (k_z = 0 plane) + i (k_z = N/2 plane), both Hermitian planes
            if (pk0) {
                column(H, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                if constexpr (E2 > 1) column(H, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                if constexpr (E2 > 2) column(H, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                if constexpr (E2 > 3) column(H, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                if constexpr (E2 > 4) {
                    column(H, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                    column(H, std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                    column(H, std::integral_constant<int, 6>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                    column(H, std::integral_constant<int, 7>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                }
                if constexpr (E2 > 8) {      // pairs 8 .. E2 - 1, one by one as above
                    auto rest = [&](auto self, auto j_tag) {
                        constexpr int J = decltype(j_tag)::value;
                        if constexpr (J < E2) {
                            column(H, std::integral_constant<int, J>{}, std::integral_constant<int, 1>{}, std::true_type{}, std::integral_constant<int, -1>{});
                            self(self, std::integral_constant<int, J + 1>{});
                        }
                    };
                    rest(rest, std::integral_constant<int, 8>{});
                }
            }
            if (!valid) {
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = cx<T>{0, 0};
            }
        } else if constexpr (!PERSIST) {
            const cx<T>* src = a.in + ubase + tbase;
            const unsigned voff = (valid && !a.drop_io) ? loff : FB_BUF_OOB;
            load_rows(v, src, voff);
        }       // (PERSIST: v[] was loaded ahead -- before the loop, or while the previous tile was being finished)
#ifdef FB_STAMPS
        FB_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FB_STAMP(2);
#endif
        if constexpr (!PERSIST) load_tables();
        else if constexpr (smode_bins(MODE)) __syncthreads();   // the previous tile's binning has read its |X|^2 out of LDS
        // (plain and generator passes: a tile's last LDS access is the read that ends its last exchange, behind a barrier)
        if constexpr (smode_bins(MODE) && (64 * E) % TZ == 0 && (N / 2) % ((64 * E) / TZ) == 0 && (TZ % E == 0 || E % TZ == 0)) {
            // The bins of this wave's block of the tile depend on the tile's coordinates only: look
            // them up now, while the tile's data is still on its way from HBM.
            constexpr int RW = (64 * E) / TZ;
            const int r0 = __builtin_amdgcn_readfirstlane((tid >> 6) * RW);
            const int ma = mode_of(r0, N), mb = mode_of(r0 + RW - 1, N);
            const int a2 = ma * ma, b2 = mb * mb;
            const int myy = mode_of(by + op.outer0, N);
            const int kzhi = (bx * TZ + TZ - 1 < a.ncols - 1) ? bx * TZ + TZ - 1 : a.ncols - 1;
            const int wlo = (a2 < b2 ? a2 : b2) + myy * myy + bx * TZ * bx * TZ;
            const int whi = (a2 > b2 ? a2 : b2) + myy * myy + kzhi * kzhi;
            wb_lo = wave_bin(wlo);
            wb_hi = wave_bin(whi);
            bool hit = false;
            for (int z = 0; z < op.namb; ++z) hit |= (op.amb[z] >= wlo && op.amb[z] <= whi);
            wb_rng = hit;                          // an edge-shell lies in the wave's n^2 RANGE: look closer below
            wb_ok = wb_hi - wb_lo <= 1;
            wb_edge = 0x7fffffff;                   // first n^2 of bin wb_hi = thr[wb_lo], from the lane that holds it
            if (wb_ok && wb_hi > wb_lo) {
                if constexpr (PERSIST) wb_edge = lthr[wb_lo];
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((wb_lo >> 6) == r) wb_edge = __builtin_amdgcn_readlane(thrv[r], wb_lo & 63);
                }
            }
        }
        FB_STAMP(3);
#ifdef FB_EXPERIMENT_NOFFT     // knock-out build (tools/knockout.sh): the generator pass without its transform
        if constexpr (MODE == SMODE_GEN) { __syncthreads(); }
#else
        if constexpr (MODE == SMODE_GEN && PARK > 0) {
            typedef StageHook<decltype(drain), 4> StageDrain;          // slots 4, 5, 6 between the transform's stages
            fft_stages<T, N, E, +1, 1, 1, Lay, false, StageDrain>(v, t, twl, layf, StageDrain{drain});
            drain(std::integral_constant<int, 7>{});
        } else if constexpr (MODE == SMODE_GEN) fft_stages<T, N, E, +1, 1, 1>(v, t, twl, layf);
#endif
        else if constexpr (smode_bins(MODE)) fft_stages<T, N, E, -1, 1, 1>(v, t, twl, layf);
        else if constexpr (CSIGN != 0) fft_stages<T, N, E, CSIGN, 1, 1>(v, t, twl, layf);
        else {
            if (sign < 0) fft_stages<T, N, E, -1, 1, 1>(v, t, twl, layf);
            else          fft_stages<T, N, E, +1, 1, 1>(v, t, twl, layf);
        }
        FB_STAMP(4);
        if constexpr (MODE == SMODE_BINF) {
            // multiply by T(k_perp, k_par) (box.py:374-379; k_perp from the plan's [k_x][k_y] table, the
            // same fp64 value k_apply_filter computes), nan_to_num, keep the filtered spectrum
            const int kyf = by + op.outer0;
            if (valid) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int kx = t + e * TPL;
                    const long long idx = ((long long)kx * op.g.NR + kyf) * op.g.NZP + col;
                    const T m = filter_value<T, false>(op.filt, op.g, kx, kyf, col, idx, op.kperp_tab[(long long)kx * N + kyf]);
                    v[e] = cx<T>{nan_to_num(v[e].x * m), nan_to_num(v[e].y * m)};
                }
            }
        }
        [[maybe_unused]] T pq[E];                       // binning modes: |X|^2 of this thread's modes
        if constexpr (smode_bins(MODE)) {
#pragma unroll
            for (int e = 0; e < E; ++e) pq[e] = v[e].x * v[e].x + v[e].y * v[e].y;
        }
        if constexpr (MODE == SMODE_BINF) {
            // store == 2: the caller wants the filtered FIELD (apply_transfer_fn's product, box.py:381): its inverse
            // transform starts with this very x line, so it is taken here, on the registers that hold the filtered
            // line, and the pass that would re-read and re-write the whole spectrum for it never runs
            if (op.store == 2) fft_stages<T, N, E, +1, 1, 1>(v, t, twl, layf);
        }
        if constexpr (!smode_bins(MODE) || MODE == SMODE_BINF) {
            cx<T>* dst = a.out + ((long long)by * out_outer + bx * TZ) + tbase_o;
            const unsigned voff = (valid && !a.drop_io) ? loff_o : FB_BUF_OOB;
            if constexpr (PARK > 0) {
#pragma unroll
                for (int e = 0; e < PARK0; ++e) store_point(dst, voff, e, cscale(v[e], a.scale));
#pragma unroll
                for (int e = 0; e < PARK; ++e) wpark[e] = cscale(v[PARK0 + e], a.scale);
                dst_prev = dst; voff_prev = voff; staged = true;
            } else
            store_rows(dst, voff, v, a.scale);
            // resident workgroup, plain pass: the next tile's loads go out right behind the stores (a store has read its
            // registers when it issues), so its memory latency runs beside this tile's store drain
            if constexpr (PERSIST && MODE == SMODE_PLAIN) load_tile(tile_id + (int)gridDim.x);
        }
        if constexpr (smode_bins(MODE)) {
            // Re-stage p = |X|^2 through LDS so that each lane bins E consecutive elements of
            // one k_x row neighbourhood and each wave 64*E/TZ consecutive rows: a lane's modes
            // then span a narrow range of |k| (almost always one bin) and a wave needs one
            // reduction per distinct bin.  tile[] is free: the last exchange ended in a barrier.
            const int nb = op.nbins;
            T* ptile = reinterpret_cast<T*>(smem);                                     // [N][TZ]
            const bool pk0 = a.packed && bx == 0;          // column 0 of this tile is the shared plane column: not binned here
            if (pk0 && c == 0) {
                cx<T>* po = op.plane_out + (long long)(by + op.outer0) * N;
#pragma unroll
                for (int e = 0; e < E; ++e) po[t + e * TPL] = v[e];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) ptile[(t + e * TPL) * TZ + c] = pq[e];
            // resident workgroup: v[] is free from here on -- the next tile's loads are issued now and the binning below
            // runs under their latency
            if constexpr (PERSIST && !DBUF) load_tile(tile_id + (int)gridDim.x);
            __syncthreads();
            FB_STAMP(5);
            double* row = acc + (size_t)(tid >> 6) * 2 * nb;
            const int ky = by + op.outer0;             // global k_y (slab-decomposed runs own a k_y range)
            const int my = mode_of(ky, N);
            const int my2 = my * my;
            const int col0 = bx * TZ;
            const int off0 = tid * E;                      // lane's E consecutive elements
            // Wave-level path.  A wave owns RW consecutive k_x rows x the tile's TZ columns, so the
            // bounds of n^2 over its block are wave-uniform and the two threshold searches are done once
            // per wave.  Almost always the block lies in one bin or straddles a single edge: each lane
            // then splits its E modes at that edge, four wave reductions finish the job.  Weights: a
            // stored mode counts twice except on the k_z = 0, N/2 planes; padding columns count 0.
            bool done = false;
            if constexpr ((64 * E) % TZ == 0 && (N / 2) % ((64 * E) / TZ) == 0 && (TZ % E == 0 || E % TZ == 0)) {
                const int blo = wb_lo, bhi = wb_hi, edge = wb_edge;
                const int mx = mode_of(off0 / TZ, N);
                const int n2row = mx * mx + my2;
                const int kz0 = col0 + off0 % TZ;
                // a lane's E consecutive elements: part of one k_x row (E <= TZ), or E / TZ whole rows (16 points per
                // thread at N = 2048: two rows of 8 columns) -- row and column of element q are then compile-time offsets
                auto n2row_of = [&](int q) -> int {
                    if constexpr (E <= TZ) return n2row;
                    else { const int m = mode_of(off0 / TZ + q / TZ, N); return m * m + my2; }
                };
                auto kz_of = [&](int q) -> int {
                    if constexpr (E <= TZ) return kz0 + q;
                    else return col0 + q % TZ;
                };
                bool exact = false;                // does a mode of this wave sit ON one of the edge-shells?
                if (wb_ok && wb_rng) {
                    bool mine = false;
#pragma unroll
                    for (int q = 0; q < E; ++q) {
                        const int kzq = kz_of(q);
                        const int n2 = n2row_of(q) + kzq * kzq;
                        for (int z = 0; z < op.namb; ++z) mine |= (op.amb[z] == n2) && (kzq < a.ncols) && !(pk0 && kzq == 0);
                    }
                    exact = __any(mine);
                }
                if (wb_ok && !exact) {
                    // tiles away from the k_z = 0 and k_z >= N/2 columns: every mode is stored once
                    // for itself and once for its mirror image (weight 2, applied after the sums)
                    const bool inner = col0 > 0 && col0 + TZ <= (N >> 1);
                    T s1 = 0, s2 = 0, u1 = 0, u2 = 0;
                    if (inner && bhi == blo) {                  // one bin, uniform weight: the common case
#pragma unroll
                        for (int q = 0; q < E; ++q) { const T p = ptile[off0 + q]; s1 += p; s2 += p * p; }
                        s1 *= (T)2; s2 *= (T)2;
                    } else if (inner) {
#pragma unroll
                        for (int q = 0; q < E; ++q) {
                            const int kz = kz_of(q);
                            const T p = ptile[off0 + q], p2 = p * p;
                            const bool up = n2row_of(q) + kz * kz >= edge;
                            s1 += up ? (T)0 : p; s2 += up ? (T)0 : p2;
                            u1 += up ? p : (T)0; u2 += up ? p2 : (T)0;
                        }
                        s1 *= (T)2; s2 *= (T)2; u1 *= (T)2; u2 *= (T)2;
                    } else {
#pragma unroll
                        for (int q = 0; q < E; ++q) {
                            const int kz = kz_of(q);
                            T p = ptile[off0 + q];
                            p = (kz >= a.ncols || (pk0 && kz == 0)) ? (T)0 : ((kz == 0 || kz == (N >> 1)) ? p : (T)2 * p);   // weight in p
                            const T p2 = (kz == 0 || kz == (N >> 1)) ? p * p : (T)0.5 * p * p;            // w p^2
                            const bool up = n2row_of(q) + kz * kz >= edge;
                            s1 += up ? (T)0 : p; s2 += up ? (T)0 : p2;
                            u1 += up ? p : (T)0; u2 += up ? p2 : (T)0;
                        }
                    }
                    s1 = wave_sum(s1); s2 = wave_sum(s2);
                    if (bhi > blo) { u1 = wave_sum(u1); u2 = wave_sum(u2); }
                    if ((tid & 63) == 0) {
                        if (blo < nb) { row[2 * blo] += (double)s1; row[2 * blo + 1] += (double)s2; }
                        if (bhi > blo && bhi < nb) { row[2 * bhi] += (double)u1; row[2 * bhi + 1] += (double)u2; }
                    }
                    done = true;
                }
            }
            if (!done) {
            // (this path is rare; it keeps nothing per element in registers -- a mode's n^2 is recomputed and its |X|^2
            // re-read from LDS where needed -- because in the resident form the next tile's loads are in flight in v[])
            auto mode_n2 = [&](int q) -> int {                  // n^2 of the lane's q-th element, -1: not a stored mode
                const int kx = (off0 + q) / TZ, kz = col0 + (off0 + q) % TZ;
                const int mx = mode_of(kx, N);
                return (kz < a.ncols && !(pk0 && kz == 0)) ? mx * mx + my2 + kz * kz : -1;
            };
            int n2lo = 0x7fffffff, n2hi = -1;
#pragma unroll 1
            for (int q = 0; q < E; ++q) {
                const int n2 = mode_n2(q);
                if (n2 >= 0) { n2lo = n2 < n2lo ? n2 : n2lo; n2hi = n2 > n2hi ? n2 : n2hi; }
            }
            const bool any_ok = n2hi >= 0;
            const int blo = any_ok ? shell_bin(lthr, nb, n2lo) : 0, bhi = any_ok ? shell_bin(lthr, nb, n2hi) : 0;
            bool hit = false;
            for (int z = 0; z < op.namb; ++z) hit |= (op.amb[z] >= n2lo && op.amb[z] <= n2hi);
            if (__all(!any_ok || (bhi - blo <= 1 && !hit))) {
                // a lane's modes fall into bin blo or blo+1: one compare against the edge
                const int edge = (any_ok && blo < nb) ? lthr[blo] : 0x7fffffff;   // first n^2 of bin blo+1
                double s1 = 0.0, s2 = 0.0, u1 = 0.0, u2 = 0.0;
#pragma unroll 1
                for (int q = 0; q < E; ++q) {
                    const int kz = col0 + (off0 + q) % TZ;
                    const int n2 = mode_n2(q);
                    const double w = (n2 < 0) ? 0.0 : ((kz == 0 || kz == (N >> 1)) ? 1.0 : 2.0);
                    const double p = (double)ptile[off0 + q];
                    const bool up = n2 >= edge;
                    s1 += up ? 0.0 : w * p; s2 += up ? 0.0 : w * p * p;
                    u1 += up ? w * p : 0.0; u2 += up ? w * p * p : 0.0;
                }
                wave_flush(blo, s1, s2, any_ok && blo < nb, row);
                if (__any(any_ok && bhi > blo)) wave_flush(bhi, u1, u2, any_ok && bhi > blo && bhi < nb, row);
            } else {
                // rare: a lane's own modes straddle an edge (or touch a shell that needs the
                // exact |k|): bin element by element
#pragma unroll 1
                for (int q = 0; q < E; ++q) {
                    const int kx = (off0 + q) / TZ, kz = col0 + (off0 + q) % TZ;
                    const int n2 = mode_n2(q);
                    int bb = n2 >= 0 ? shell_bin(lthr, nb, n2) : nb;
                    for (int z = 0; z < op.namb; ++z)
                        if (n2 >= 0 && op.amb[z] == n2) bb = bin_exact(op.bins, nb, kmag_exact(op.g, kx, ky, kz));
                    const double w = (kz == 0 || kz == (N >> 1)) ? 1.0 : 2.0;
                    const double p = (double)ptile[off0 + q];
                    wave_flush(bb, w * p, w * p * p, n2 >= 0 && bb < nb, row);
                }
            }
            }   // !done
        }
#ifdef FB_STAMPS
        if constexpr (!smode_bins(MODE)) FB_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FB_STAMP(6);
#endif
        if constexpr (!PERSIST) break;
        tile_id += gridDim.x;
    } while (tile_id < a.ntiles);
    if constexpr (PARK > 0) {
        if (staged) {
#pragma unroll
            for (int e = 0; e < PARK; ++e) store_point(dst_prev, voff_prev, PARK0 + e, wpark[e]);
        }
    }
    if constexpr (smode_bins(MODE)) {
#ifdef FB_STAMPS
        if ((tid & 63) == 0 && stamp_ok) stamp[8 + (tid >> 6)] = (long long)__builtin_amdgcn_s_memtime();   // this wave's binning is done
#endif
        __syncthreads();
        const int nb = op.nbins;
        for (int i = tid; i < 2 * nb; i += NT) {          // partial[value][workgroup]
            double sum = 0.0;
            for (int w2 = 0; w2 < NW; ++w2) sum += acc[(size_t)w2 * 2 * nb + i];
            op.partial[(size_t)i * op.partial_stride + blockIdx.x] = sum;
        }
        FB_STAMP(7);
    }
}

// out[q] = sum_r partial[q][r], fixed order: one workgroup per value
// (workgroup nvals, if launched: out[nvals] = sum of the single column partial2[nrows2], 0 if that is null --
// the log-normal mean's block sums ride along with the bin sums in one launch)
static __global__ __launch_bounds__(1024) void k_sum_columns(const double* __restrict__ partial, long long nrows,
                                                              int nvals, double* __restrict__ out,
                                                              const double* __restrict__ partial2, long long nrows2,
                                                              long long stride = 0) {      // rows a value's partials are apart (0: nrows)
    __shared__ double sh[1024];
    const int q = blockIdx.x, nt = blockDim.x;       // nt: a power of two, 64 .. 1024
    const double* src = partial + (size_t)q * (stride ? stride : nrows);
    if (q == nvals) {
        if (!partial2) { if (threadIdx.x == 0) out[q] = 0.0; return; }
        src = partial2; nrows = nrows2;
    }
    // fixed summation order (thread-strided, 8 independent chains so that the loads pipeline)
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long r = threadIdx.x;
    for (; r + 7 * nt < nrows; r += 8 * nt) {
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] += src[r + u * nt];
    }
    for (int u = 0; r < nrows; r += nt, ++u) c[u & 7] += src[r];
    const double s = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = nt >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[q] = sh[0];
}

// The shared plane column of a packed half spectrum: G(k_x, k_y) = X0 + i X1 with X0 = X(k_x, k_y, 0) and
// X1 = X(k_x, k_y, N/2), each Hermitian in (k_x, k_y), so  X0 = (G(k) + conj G(-k)) / 2,  X1 = (G(k) - conj G(-k)) / (2 i).
// One workgroup per k_y row bins its 2 N modes (weight 1 each, as the k_z = 0, N/2 planes count in a half spectrum)
// into columns col0 + k_y of partial[value][stride]; a mode's bin as in k_bin_rows' per-mode path (shell thresholds,
// the exact fp64 |k| for shells within rounding of an edge).  Fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void k_bin_packed_plane(const cx<T>* __restrict__ plane, KGeom g, BinGeom bg,
                                                           double* __restrict__ partial, long long stride, long long col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lbins = reinterpret_cast<double*>(smem);           // [nbins]
    double* acc = lbins + bg.nbins;                            // [4][2 nbins]
    int* lthr = reinterpret_cast<int*>(acc + 4 * 2 * bg.nbins);
    const int N = g.N, nb = bg.nbins, tid = threadIdx.x;
    for (int q = tid; q < nb; q += 256) { lbins[q] = bg.bins[q]; lthr[q] = bg.thr ? bg.thr[q] : 0; }
    for (int q = tid; q < 8 * nb; q += 256) acc[q] = 0.0;
    __syncthreads();
    const int ky = blockIdx.x, kym = (N - ky) & (N - 1);
    double* row = acc + (size_t)(tid >> 6) * 2 * nb;
    for (int kx0 = 0; kx0 < N; kx0 += 256) {                   // wave-uniform trip count
        const int kx = kx0 + tid;
        const bool have = kx < N;
        cx<T> G{0, 0}, M{0, 0};
        if (have) { G = plane[(long long)ky * N + kx]; M = plane[(long long)kym * N + ((N - kx) & (N - 1))]; }
        const cx<T> x0{(T)0.5 * (G.x + M.x), (T)0.5 * (G.y - M.y)};
        const cx<T> x1{(T)0.5 * (G.y + M.y), (T)0.5 * (M.x - G.x)};
        const T p0 = x0.x * x0.x + x0.y * x0.y, p1 = x1.x * x1.x + x1.y * x1.y;
        const int b0 = have ? bin_of_mode(bg, g, lbins, lthr, kx, ky, 0) : nb;
        const int b1 = have ? bin_of_mode(bg, g, lbins, lthr, kx, ky, N >> 1) : nb;
        wave_flush(b0, p0, p0 * p0, have && b0 < nb, row);       // sums of a wave in T (DPP), rows in fp64, as the main pass
        wave_flush(b1, p1, p1 * p1, have && b1 < nb, row);
    }
    __syncthreads();
    for (int i = tid; i < 2 * nb; i += 256)
        partial[(size_t)i * stride + col0 + blockIdx.x] = (acc[i] + acc[2 * nb + i]) + (acc[4 * nb + i] + acc[6 * nb + i]);
}

// exp() of the log-normal transform (box.py:457).  -DFB_FAST_EXP: hardware 2^x on x log2(e) for single precision
// (relative error ~ |x| 2^-24 from the rounded product plus 1 ulp) instead of libm's expf.
__device__ __forceinline__ double fb_exp(double x) { return exp(x); }
__device__ __forceinline__ float fb_exp(float x) {
#ifdef FB_LIBM_EXP
    return expf(x);
#else
    // 2^(x log2 e) on the hardware exponential (1 ulp), with the product's rounding error and the constant's own
    // truncation carried into a first-order correction: t + e = x log2(e) to ~2^-45 |x|, exp(x) = 2^t (1 + e ln 2).
    // Same accuracy class as libm's expf (2 ulp over |x| < 80 given a correctly rounded 2^t: tests/test_host_math.py), 6
    // instructions instead of 10 and no range-reduction branches; the fused z pass is issue-bound.
    const float t = x * 1.44269502162933349609375f;                        // float(log2 e)
    float e = __builtin_fmaf(x, 1.44269502162933349609375f, -t);           // exact rounding error of the product
    e = __builtin_fmaf(x, 1.925963033500011e-8f, e);                       // log2 e - float(log2 e)
    const float r = __builtin_amdgcn_exp2f(t);
    // (overflow: r = inf and, where the correction is exactly 0, inf * 0 + inf would be NaN; libm's expf gives inf)
    return r > 3.4028234663852886e38f ? r : __builtin_fmaf(r, 0.693147182464599609375f * e, r);
#endif
}
enum { ZMODE_C2C = 0, ZMODE_R2C = 1, ZMODE_C2R = 2, ZMODE_C2R2C = 3 };
// C2R2C: inverse z pass, write the real field, then (optionally exp() and) forward z pass of the
// same line from registers: realise_density's last pass fused with the power spectrum's first,
// so the real field is written once and never read back.

template <typename T> struct ContigArgs {
    const void* in;
    void* out;
    const cx<T>* tw;        // W_M^j with M = n (c2c) or 2n (r2c / c2r)
    long long in_pitch;     // elements of the input type between lines
    long long out_pitch;    // elements of the output type between lines
    int in_skip, out_skip;  // > 0: that side is a half spectrum with one spare row after every `skip` lines
    long long nlines;
    T scale;
    int pre_exp;            // r2c: transform exp(x) instead of x (log-normal fusion)
    double* exp_partial;    // r2c + pre_exp: [gridDim.x] block sums of exp(x)
    void* out2;             // C2R2C: half spectrum out (may alias `in`), pitch/skip as `in`
    int packed;             // half spectrum rows hold k_z = 0 .. N/2-1, element 0 = X[0] + i X[N/2] (both are real for a real line)
    T exp_shift;            // pre_exp: transform exp(x - exp_shift) (fb_set_exp_shift: keeps a high-variance field's sums in range)
};

#ifndef FB_CONTIG_THREADS
#define FB_CONTIG_THREADS 256      // threads per workgroup of the contiguous-axis passes (tuning: 128, 512)
#endif
template <typename T, int NF> constexpr int contig_lines() {   // lines per workgroup
    return fb_max(1, FB_CONTIG_THREADS / (NF / elems_per_thread<T>(NF)));
}

// ---- one line of the contiguous (z) axis through the packed half-length complex transform ----------------------
// (shared by k_fft_contig and the fused redshift-space z pass k_rsd_turn: the same instructions, so the same bits)
// c2r: the half-spectrum row `in` (N/2 + 1 entries, or N/2 with entry 0 = X[0] + i X[N/2] when packed) -> the real line,
// thread t of the line's NF / E threads ending with v[e] = (x[2j], x[2j+1]) * scale, j = t + e NF / E.
//   Z[k] = (X[k] + conj X[n-k]) + i e^{+2 pi i k/N} (X[k] - conj X[n-k]); the imaginary parts of X[0], X[n] are dropped
//   (Hermitian projection).  twl = W_{2 NF}^j in LDS, published before the call.
template <typename T, int NF, bool WAVE>
__device__ __forceinline__ void c2r_line(const cx<T>* __restrict__ in, const int packed, const int t, const cx<T>* twl,
                                         const LineLayout<T>& lay, const T scale, cx<T> (&v)[elems_per_thread<T>(NF)]) {
    constexpr int E = elems_per_thread<T>(NF), TPL = NF / E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int k = t + e * TPL;
        cx<T> xk = in[k], xn = in[packed ? ((NF - k) & (NF - 1)) : NF - k];
        if (k == 0) { if (packed) xn.x = xk.y; xk.y = 0; xn.y = 0; }
        if constexpr (sizeof(T) == 4) {          // packed forms: s + i conj(w) d in five instructions
            const cx<T> s = pk_add_conj(xk, xn), d = pk_sub_conj(xk, xn);
            v[e] = pk_add_i<+1>(s, pk_cmul<+1>(d, twl[k]));
        } else {
            cx<T> s = xk + cconj(xn), d = xk - cconj(xn);
            cx<T> w = cconj(twl[k]);
            cx<T> wd = cmul(w, d);
            v[e] = cx<T>{s.x - wd.y, s.y + wd.x};
        }
    }
    fft_stages<T, NF, E, +1, 2, 1, LineLayout<T>, WAVE>(v, t, twl, lay);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = cscale(v[e], scale);
}
// r2c: v[e] = (x[2j], x[2j+1]) of a real line -> its half-spectrum row at `out` (nullptr: nothing is stored), times sc.
//   X[k] = (Z[k] + conj Z[n-k])/2 - (i/2) W_N^k (Z[k] - conj Z[n-k])
template <typename T, int NF, bool WAVE>
__device__ __forceinline__ void r2c_line(cx<T> (&v)[elems_per_thread<T>(NF)], const int t, const cx<T>* twl, const LineLayout<T>& lay,
                                         const int packed, cx<T>* __restrict__ out, const T sc) {
    constexpr int E = elems_per_thread<T>(NF), TPL = NF / E;
    fft_stages<T, NF, E, -1, 2, 1, LineLayout<T>, WAVE>(v, t, twl, lay);
#pragma unroll
    for (int e = 0; e < E; ++e) lay.at(t + e * TPL) = v[e];
    exchange_sync<WAVE>();
    if (out) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = t + e * TPL;
            cx<T> zk = v[e];
            cx<T> res;
            if constexpr (sizeof(T) == 4) {      // (s - i w d) / 2 with s = z_k + conj z_{n-k}, d = z_k - conj z_{n-k}
                const cx<T> zr = lay.at((NF - k) & (NF - 1));
                const cx<T> s = pk_add_conj(zk, zr), d = pk_sub_conj(zk, zr);
                res = cscale(pk_add_i<-1>(s, pk_cmul<-1>(d, twl[k])), (T)0.5 * sc);
            } else {
                cx<T> zn = cconj(lay.at((NF - k) & (NF - 1)));
                cx<T> s = zk + zn, d = zk - zn;
                cx<T> wd = cmul(twl[k], d);
                res = cx<T>{(T)0.5 * (s.x + wd.y) * sc, (T)0.5 * (s.y - wd.x) * sc};
            }
            if (k == 0 && packed) out[0] = cx<T>{(zk.x + zk.y) * sc, (zk.x - zk.y) * sc};
            else out[k] = res;
            if (k == 0 && !packed) out[NF] = cx<T>{(zk.x - zk.y) * sc, (T)0};
        }
    }
}

// NF = complex transform length (N for c2c, N/2 for r2c/c2r)
template <typename T, int NF, int MODE>
__global__ __launch_bounds__((contig_lines<T, NF>() * (NF / elems_per_thread<T>(NF))))
void k_fft_contig(ContigArgs<T> a, int sign_c2c) {
    constexpr int E = elems_per_thread<T>(NF);
    constexpr int TPL = NF / E;
    constexpr int LPW = contig_lines<T, NF>();
    constexpr int NT = LPW * TPL;
    constexpr int TWS = (MODE == ZMODE_C2C) ? 1 : 2;
    constexpr int M = NF * TWS;
    constexpr int LP = LineLayout<T>::padded(NF);
#ifdef FB_CONTIG_BARRIERS
    constexpr bool WAVE = false;
#else
    constexpr bool WAVE = TPL <= 64;       // a line's threads share a wavefront: exchanges without workgroup barriers
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* lines = reinterpret_cast<cx<T>*>(smem);
    cx<T>* twl = lines + LPW * LP;

    const int tid = threadIdx.x;
    const int t = tid % TPL;
    const int l = tid / TPL;
    for (int i = tid; i < M; i += NT) twl[i] = a.tw[i];
    const long long line = (long long)blockIdx.x * LPW + l;
    const bool valid = line < a.nlines;
    const long long line_ld = valid ? line : 0;      // lines past the end (last workgroup only) load line 0 and store nothing:
                                                     // no per-element branches in the load / exp loops
    LineLayout<T> lay{lines + l * LP};

    cx<T> v[E];
    if constexpr (MODE == ZMODE_C2R || MODE == ZMODE_C2R2C) {
        // Z[k] = (X[k] + conj X[n-k]) + i e^{+2 pi i k/N} (X[k] - conj X[n-k]); the
        // imaginary parts of X[0], X[n] are dropped (Hermitian projection).
        const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + (line_ld + (a.in_skip ? line_ld / a.in_skip : 0)) * a.in_pitch;
        // (taking the untangle twiddles straight from the global table, so that the line's own loads are not held behind
        // this barrier, was measured slower: 0.3245 against 0.3062 ms per step for the fused z pass)
        __syncthreads();
        c2r_line<T, NF, WAVE>(in, a.packed, t, twl, lay, a.scale, v);
        if (valid && a.out) {                          // C2R2C with no real output: the field is only passed on
            cx<T>* out = reinterpret_cast<cx<T>*>(reinterpret_cast<T*>(a.out) + line * a.out_pitch);
#pragma unroll
            for (int e = 0; e < E; ++e) {
#ifndef FB_REAL_STORE_PLAIN
                // the field of a fused chain is written for the caller and not read again by the chain: streaming
                // stores leave the Infinity Cache to the plane batch (512^3: +1.5 % one box, +0.5 .. 2.5 % two boxes)
                if constexpr (MODE == ZMODE_C2R2C) {
                    __builtin_nontemporal_store(v[e].x, &out[t + e * TPL].x);
                    __builtin_nontemporal_store(v[e].y, &out[t + e * TPL].y);
                } else
#endif
                out[t + e * TPL] = v[e];
            }
        }
    }
    if constexpr (MODE == ZMODE_R2C || MODE == ZMODE_C2R2C) {
        double esum = 0.0;
        if constexpr (MODE == ZMODE_R2C) {
            const cx<T>* in = reinterpret_cast<const cx<T>*>(reinterpret_cast<const T*>(a.in) + line_ld * a.in_pitch);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = in[t + e * TPL];
        }
        if (a.pre_exp) {
            T part = 0;                      // a thread's 2 E exponentials in T, then one double add
#pragma unroll
            for (int e = 0; e < E; ++e) {
                v[e].x = fb_exp(v[e].x - a.exp_shift); v[e].y = fb_exp(v[e].y - a.exp_shift);
                part += v[e].x + v[e].y;
            }
            esum = valid ? (double)part : 0.0;
        }
        if constexpr (MODE == ZMODE_R2C || !WAVE) __syncthreads();      // R2C: the twiddles; C2R2C: lines[] is this wave's own
        else exchange_sync<true>();
        {
            cx<T>* out = (MODE == ZMODE_C2R2C)
                ? reinterpret_cast<cx<T>*>(a.out2) + (line + (a.in_skip ? line / a.in_skip : 0)) * a.in_pitch
                : reinterpret_cast<cx<T>*>(a.out) + (line + (a.out_skip ? line / a.out_skip : 0)) * a.out_pitch;
            const T sc = (MODE == ZMODE_C2R2C) ? (T)1 : a.scale;     // C2R2C: `scale` belongs to the inverse half
            r2c_line<T, NF, WAVE>(v, t, twl, lay, a.packed, valid ? out : nullptr, sc);
        }
        if (a.pre_exp) {                       // wave-uniform
            __syncthreads();                   // lines[] no longer needed
            double* red = reinterpret_cast<double*>(smem);
            const double ws = wave_sum(esum);
            if ((tid & 63) == 0) red[tid >> 6] = ws;
            __syncthreads();
            if (tid == 0) {
                double s = 0.0;
                for (int q = 0; q < (NT + 63) / 64; ++q) s += red[q];
                a.exp_partial[blockIdx.x] = s;
            }
        }
    } else if constexpr (MODE == ZMODE_C2C) {
        const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + (line + (a.in_skip ? line / a.in_skip : 0)) * a.in_pitch;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = valid ? in[t + e * TPL] : cx<T>{0, 0};
        __syncthreads();
        if (sign_c2c < 0) fft_stages<T, NF, E, -1, TWS, 1, LineLayout<T>, WAVE>(v, t, twl, lay);
        else              fft_stages<T, NF, E, +1, TWS, 1, LineLayout<T>, WAVE>(v, t, twl, lay);
        if (valid) {
            cx<T>* out = reinterpret_cast<cx<T>*>(a.out) + (line + (a.out_skip ? line / a.out_skip : 0)) * a.out_pitch;
#pragma unroll
            for (int e = 0; e < E; ++e) out[t + e * TPL] = cscale(v[e], a.scale);
        }
    }
}

// ---- redshift-space z pass: c2r(delta) + c2r(v_z) + line-of-sight remap + r2c in one kernel -------------------------
// The chain  realise_density -> realise_velocity[2] -> redshift_space_density -> fftn  (BASELINE configs[2],
// examples/example_redshift_space.py + box.py:384-438) turns round on the z axis: both inverse transforms end with a z
// pass, the remap works along z, the forward transform starts with one.  One wave per line of sight does all four on
// the line while it is on the chip: lanes 0 .. NF/8-1 take the inverse z transform of delta's row and the next NF/8
// lanes that of v_z's row (the instructions of k_fft_contig, c2r_line), the two real lines are handed over through LDS
// so that lane l owns cells l E .. l E + E - 1 of both (k_rsd_cells' ownership), the remap runs as in k_rsd_cells
// (rsd_remap_line), and the first lanes take the forward z transform of the result (r2c_line).  delta_x is written for
// the caller; v_z in real space, the redshift-space field and two reads of delta_x never touch memory: 4 half-sweeps
// of traffic instead of 9 (the three z passes 2 each, the remap 3).  Bit-identical to the separate kernels.
// Single-precision plans, 64 <= N <= 512 (both rows' transforms fit one wave; per-wave LDS as k_rsd_cells).
#ifndef FB_RSDT_WAVES
#define FB_RSDT_WAVES 4           // lines of sight (waves) per workgroup
#endif
#ifndef FB_RSDT_ZG_LDS
#define FB_RSDT_ZG_LDS 0          // 1: the grid in LDS as in k_rsd_cells (4 KiB more per workgroup at N = 512: one resident
                                  // workgroup fewer per CU); 0: each lane reads its own E grid points from the (cached) table
#endif
template <typename T> struct RsdTurnArgs {
    const cx<T>* in_d;        // delta's work spectrum after the inverse x and y passes: rows `pitch` apart, one spare row per N
    const cx<T>* in_v;        // v_z's, same layout
    T* dx_out;                // [nlines][N] delta_x of these lines, or null
    cx<T>* out;               // forward z spectrum of the remapped lines: N/2 + 1 entries per row, same row layout
    const cx<T>* tw;          // W_N^j, N entries
    const double* zgrid;      // [N]
    long long pitch;
    long long nlines, line0;  // lines of this launch (a plane batch); global index of the first (the noise's counters)
    T scale_d, scale_v;
    int packed_in;            // in_d / in_v rows: entry 0 = X[0] + i X[N/2], N/2 entries
    double Hz, sigma_nl;
    RngKey rkey;
    int nearest;
};
template <int E> constexpr size_t rsd_turn_region() {      // per wave: 32-bit keys + values, or the two transforms' line buffers
    const size_t kv = (size_t)4 * 64 * E + 4 * (64 * E + 16), lb = (size_t)2 * LineLayout<float>::padded(32 * E) * 8;
    return ((kv > lb ? kv : lb) + 15) / 16 * 16;
}
template <int E> constexpr size_t rsd_turn_lds() {
    return (size_t)(FB_RSDT_ZG_LDS ? 16 : 8) * 64 * E + FB_RSDT_WAVES * rsd_turn_region<E>();
}
template <typename T, int E>
__global__ __launch_bounds__(64 * FB_RSDT_WAVES, FB_RSD_OCC) void k_rsd_turn(RsdTurnArgs<T> a) {
    static_assert(sizeof(T) == 4 && E >= 1 && E <= 8, "single precision, 64 <= N <= 512");
    constexpr int N = 64 * E, NF = N / 2, EF = elems_per_thread<T>(NF), TPL = NF / EF;
    constexpr int LP = LineLayout<T>::padded(NF);
    static_assert(2 * TPL <= 64 && 2 * LP * sizeof(cx<T>) <= rsd_turn_region<E>() && 2 * N * sizeof(T) <= rsd_turn_region<E>(), "layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    cx<T>* twl = reinterpret_cast<cx<T>*>(smem);                                     // [N]   shared by the block
    [[maybe_unused]] double* zgl = reinterpret_cast<double*>(smem + (size_t)8 * N);   // [N]   (FB_RSDT_ZG_LDS)
    char* mine = smem + (size_t)(FB_RSDT_ZG_LDS ? 16 : 8) * N + w * rsd_turn_region<E>();
    typedef typename rsd_key<T>::type key_t;
    key_t* kex = reinterpret_cast<key_t*>(mine);                                     // [N]      this wave's
    T* vex = reinterpret_cast<T*>(mine + sizeof(key_t) * N);                         // [N + 16] this wave's
    cx<T>* fl = reinterpret_cast<cx<T>*>(mine);        // the transforms' line buffers ([2][LP]) and
    T* rl = reinterpret_cast<T*>(mine);                // the two real lines ([2][N]) live in the same bytes, before the remap
    const double zmin = a.zgrid[0], zmax = a.zgrid[N - 1];
    const double len = zmax - zmin;
    for (int i = threadIdx.x; i < N; i += 64 * FB_RSDT_WAVES) twl[i] = a.tw[i];
    if constexpr (FB_RSDT_ZG_LDS) {
        auto sw = [](int c) { return (c % E) * 64 + c / E; };
        for (int m = threadIdx.x; m < N; m += 64 * FB_RSDT_WAVES) zgl[sw(m)] = (a.zgrid[m] - zmin) + len;
    }
    __syncthreads();                                   // the only workgroup barrier: from here on a wave is on its own
    const long long ll = (long long)blockIdx.x * FB_RSDT_WAVES + w;
    if (ll >= a.nlines) return;
    const long long row = ll + ll / N;
    // inverse z transforms: lanes [0, TPL) delta, [TPL, 2 TPL) v_z
    {
        const int l = lane / TPL, t = lane % TPL;
        cx<T> v[EF];
        if (2 * TPL == 64 || l < 2) {
            const cx<T>* in = (l ? a.in_v : a.in_d) + row * a.pitch;
            const LineLayout<T> lay{fl + (l & 1) * LP};
#if defined(FB_RSDT_KNOCK) && (FB_RSDT_KNOCK & 1)      // tuning aid: no inverse transforms
#pragma unroll
            for (int e = 0; e < EF; ++e) v[e] = cscale(in[t + e * TPL], l ? a.scale_v : a.scale_d);
#else
            c2r_line<T, NF, true>(in, a.packed_in, t, twl, lay, l ? a.scale_v : a.scale_d, v);
#endif
        }
        exchange_sync<true>();
        if (2 * TPL == 64 || l < 2) {
#pragma unroll
            for (int e = 0; e < EF; ++e) reinterpret_cast<cx<T>*>(rl + (l & 1) * N)[t + e * TPL] = v[e];
        }
        exchange_sync<true>();
    }
    T val[E], vin[E];
    if constexpr (E % 4 == 0) {
        typedef T vec4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const vec4 b = reinterpret_cast<const vec4*>(rl + lane * E)[q], c = reinterpret_cast<const vec4*>(rl + N + lane * E)[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) { val[4 * q + u] = b[u]; vin[4 * q + u] = c[u]; }
            // (streaming store: the chain does not read delta_x again)
            if (a.dx_out) __builtin_nontemporal_store(b, reinterpret_cast<vec4*>(a.dx_out + ll * N + lane * E) + q);
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) { val[e] = rl[lane * E + e]; vin[e] = rl[N + lane * E + e]; }
        if (a.dx_out) {
#pragma unroll
            for (int e = 0; e < E; ++e) __builtin_nontemporal_store(val[e], a.dx_out + ll * N + lane * E + e);
        }
    }
    rsd_wave_sync();
#pragma unroll
    for (int e = 0; e < E; ++e) kex[lane + 64 * e] = 0;
    rsd_wave_sync();
    // (delta[0] + delta[N-1]) / 2: lane 0's first and lane 63's last cell
    const T d0 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val[0]), 0));
    const T dn = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val[E - 1]), 63));
    const double fill = 0.5 * ((double)d0 + (double)dn);
    T y_out[E];
#if defined(FB_RSDT_KNOCK) && (FB_RSDT_KNOCK & 2)      // tuning aid: no remap
#pragma unroll
    for (int e = 0; e < E; ++e) y_out[e] = val[e] + vin[e] * (T)fill;
#else
    rsd_remap_line<T, E, !FB_RSDT_ZG_LDS>(vin, val, nullptr, a.line0 + ll, fill, FB_RSDT_ZG_LDS ? zgl : a.zgrid, kex, vex, zmin, len, a.Hz,
                                          a.sigma_nl, a.rkey, a.nearest, lane, y_out);
#endif
    rsd_wave_sync();
    // forward z transform of the remapped line: cells (2 j, 2 j + 1) are point j of the half-length complex sequence
    {
        const LineLayout<T> lay{fl};
        if constexpr (E % 2 == 0) {
#pragma unroll
            for (int q = 0; q < E / 2; ++q) lay.at(lane * (E / 2) + q) = cx<T>{y_out[2 * q], y_out[2 * q + 1]};
        } else {
            reinterpret_cast<T*>(&lay.at(lane >> 1))[lane & 1] = y_out[0];
        }
        exchange_sync<true>();
        if (lane < TPL) {
            cx<T> v[EF];
#pragma unroll
            for (int e = 0; e < EF; ++e) v[e] = lay.at(lane + e * TPL);
            exchange_sync<true>();
#if defined(FB_RSDT_KNOCK) && (FB_RSDT_KNOCK & 4)      // tuning aid: no forward transform
#pragma unroll
            for (int e = 0; e < EF; ++e) (a.out + row * a.pitch)[lane + e * TPL] = v[e];
#else
            r2c_line<T, NF, true>(v, lane, twl, lay, 0, a.out + row * a.pitch, (T)1);
#endif
        }
    }
}

}  // namespace fb
