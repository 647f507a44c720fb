// Columns of the oversampled grid cut into tasks along dimension 3: the per-point-set work decomposition of the MFMA-patch
// spreading engine (patch columns) and of the z-marching interpolation ring (grid columns); built by balance.hip.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define NUFFT_HD __host__ __device__
#else
#define NUFFT_HD
#endif

namespace nufft {

// ncolx x ncoly columns of bxw x byw bins, nseg segments of segl layers when cut evenly (ntasks = columns x
// nseg), boundaries at multiples of zq layers, a task's work counted over clo .. chi layers beyond its own, no segment
// longer than maxlen layers (0: any length)
struct ColumnTasks {
    int ncolx, ncoly, bxw, byw, nseg, segl, ntasks, zq, clo, chi, maxlen;
};

// Entries of a task table: column c gets max(min_seg, round(T points(c) / Np)) segments, where min_seg covers the column
// with segments of at most maxlen layers — at most T + columns / 2 + min_seg x columns in all.
inline int column_task_table_entries(const ColumnTasks& ct, int nz) {
    const int min_seg = ct.maxlen > 0 ? (nz + ct.maxlen - 1) / ct.maxlen : 1;
    return ct.ntasks + (min_seg + 1) * ct.ncolx * ct.ncoly;
}

// Decomposition of a grid for the z-marching spreading ring (smarch_kernels.h), chosen at plan creation (smarch_plan)
struct SMarchPlan {
    bool eligible;
    int n1, n2;                 // column interior chosen for this grid (multiples of the bin edge, <= the kernel's compile-time column)
    int hlo, hhi;               // layers of points a segment visits below / above its own
    int halo;                   // 0: columns clipped in x and y; 2: halo variant (every point spread once by its own column, the stencil reach
                                // into a side buffer that the consumer of the grid adds: smarch_kernels.h)
    int64_t halo_reals;         // halo = 2: reals of the side buffer per component (parts = 2: per part of a component)
    int parts;                  // 2: complex data through the REAL kernel, real and imaginary parts as two launch rows per component (smarch_kernels.h); else 1
    int lds_bytes, threads;
    ColumnTasks ct;             // columns and evenly cut tasks
    double visits, efficiency;  // model: point visits per point, and the share of the chip the launch keeps busy
};

// Side buffer of the halo variant: one record of reals per (plane z, column ty, tx) —
//   [n2 strips of SW reals: the x reach of the column's own rows, XLO cells below then XHI above]
//   [YLO + YHI rows of RW reals: the y reach (rows below, then above), each the full width of the window]
// SW and RW rounded up to even so that aligned pairs of reals stay 2-real aligned.  (XLO = M - 1 rounded up to even: the first
// cell of the window row is padding for odd M - 1; it is stored as the zero it holds.)
struct HaloLayout {
    int n1, n2, xlo, xhi, ylo, yhi, nc;     // column, reach in cells, components per cell
    int ntx, nty;
    int sw, rw, rec;                        // strip width, row width, record size (reals)
    int64_t plane;                          // reals per plane of the grid's side buffer
};
NUFFT_HD inline HaloLayout make_halo_layout(int n1, int n2, int M, int nc, int ntx, int nty) {
    HaloLayout h;
    h.n1 = n1; h.n2 = n2; h.nc = nc; h.ntx = ntx; h.nty = nty;
    h.xlo = (M - 1) + ((M - 1) & 1); h.xhi = M; h.ylo = M - 1; h.yhi = M;
    h.sw = ((h.xlo + h.xhi) * nc + 1) & ~1;
    h.rw = ((n1 + h.xlo + h.xhi) * nc + 1) & ~1;
    h.rec = n2 * h.sw + (h.ylo + h.yhi) * h.rw;
    h.plane = (int64_t)h.rec * ntx * nty;
    return h;
}
// offset (reals) inside a record of window element (real index lxw of the window row, row ly relative to the column's first row);
// the element must lie outside the column
NUFFT_HD inline int halo_record_offset(const HaloLayout& h, int lxw, int ly) {
    if (ly < 0) return h.n2 * h.sw + (ly + h.ylo) * h.rw + lxw;
    if (ly >= h.n2) return h.n2 * h.sw + (h.ylo + ly - h.n2) * h.rw + lxw;
    return ly * h.sw + (lxw < h.xlo * h.nc ? lxw : lxw - h.n1 * h.nc);
}

// Column-layer sort (binsort.hip): plans whose spreading window (halo variant) and interpolation ring own the same columns of
// cbx x cby bins only need the points grouped by (column, layer of bins) — ncx x ncy x nb[2] keys instead of nbins fine bins: few
// enough for a histogram per workgroup in LDS, so neither pass issues a global atomic.  The fake fine histogram it leaves in
// `counts` (a column layer's total in its first bin, zero elsewhere) scans to offsets that present a column layer as ONE run of the
// sorted array to every consumer that walks rows of bins.  Point sets that either ring hands to the tile kernels (device flags
// flag_a, flag_b of set_points' task kernels) are sorted by fine bins as before.
struct CoarseSort {
    int enabled;
    int cbx, cby, ncx, ncy;    // bins per column, columns
    int nkeys;                 // ncx * ncy * nb[2] (<= kCoarseMaxKeys)
    int groups;                // workgroups = contiguous slices of the point set (one per compute unit)
    uint32_t* table;           // [groups][nkeys]: per-slice counts, then exclusive prefixes over the slices
    const uint32_t* flag_a;    // both nonzero: this point set is column-layer sorted
    const uint32_t* flag_b;
    // two-level sort by slabs (mode 2, below): level-1 records, what a level-2 workgroup holds, its LDS, {largest slab, flag}
    int mode;                  // 1: column-layer sort; 2: slab sort
    void* temp;
    int cap, lds2;
    uint32_t* flagmem;
    // feedback of the device-side decisions to the host, without a synchronisation: workgroup 0 of the scatter pass writes {ring flag a, ring flag b,
    // sequence number} into host-mapped memory; the NEXT set_points reads whatever has arrived (plan.cpp: adaptive sort choice)
    const uint32_t* fb_a;
    const uint32_t* fb_b;
    uint32_t* feedback;
    uint32_t seq;
};
// Two-level fine sort (mode 2; plans without a column-layer sort, D = 3): the same two passes with a SLAB of bins as the key — cbx = nb[0]
// (one column along x), cby rows of bins, one layer: a contiguous range of fine bins — into a temporary array; then one workgroup per slab
// sorts its records by fine bin in LDS and writes them, and the slab's share of the fine offsets, in order.  The result is the array and
// the offsets of the fine sort (up to the order inside a bin) without a global atomic and with every store of level 2 coalesced; the slab
// height is chosen per point set (set_points) so that a slab's points fit a workgroup's LDS on average with room to spare; fuller slabs (denser
// regions) are sorted by the same workgroup in two passes over global memory, and point sets whose fullest slab exceeds kSlabOverfill
// capacities (clusters) take the fine sort with global atomics instead (device flag, flagmem[4]).
constexpr int kSlabMaxBins = 4096;         // fine bins of a slab (16 KiB of LDS counters in level 2)
constexpr int kHaloStateWord = 12;         // word of the spreading ring's 16-word device record: 1 = side buffer written, not yet added to the grid
constexpr int kSlabOverfill = 8;           // a slab may hold this many LDS capacities (level 2 then sorts it through global memory)
constexpr int kCoarseMaxKeys = 36864;      // 144 KiB of LDS counters

}  // namespace nufft
