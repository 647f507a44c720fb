// Columns of the oversampled grid cut into tasks along dimension 3: the per-point-set work decomposition of the MFMA-patch
// spreading engine (patch columns) and of the z-marching interpolation ring (grid columns); built by balance.hip.
#pragma once

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
    int halo;                   // 0: columns clipped in x and y; 1: input-driven in x (halo in LDS, atomics bands); 2: in x and y
    int lds_bytes, threads;
    ColumnTasks ct;             // columns and evenly cut tasks
    double visits, efficiency;  // model: point visits per point, and the share of the chip the launch keeps busy
};

}  // namespace nufft
