// Host-side dispatch of the tile kernels over (precision, complex?, D, M).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "tile_kernels.h"
#include "patch_kernels.h"
#include <algorithm>
#include <vector>

#include "march_kernels.h"
#include "patch32_kernels.h"

namespace nufft {

const void* march_kernel_f32r(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f32c(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f64r(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f64c(int M, bool poly, int* lds_bytes, int* n);
static const void* march_kernel(int dtype, int is_complex, int M, bool poly, int* lds_bytes, int* n) {
    if (dtype == NUFFT_F32) return is_complex ? march_kernel_f32c(M, poly, lds_bytes, n) : march_kernel_f32r(M, poly, lds_bytes, n);
    return is_complex ? march_kernel_f64c(M, poly, lds_bytes, n) : march_kernel_f64r(M, poly, lds_bytes, n);
}
// interp_march_staged_kernel: the ring for column-layer sorted point sets (null: none for this configuration)
const void* march_kernel_f32r_staged(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f32c_staged(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f64r_staged(int M, bool poly, int* lds_bytes, int* n);
const void* march_kernel_f64c_staged(int M, bool poly, int* lds_bytes, int* n);
static const void* march_staged_kernel(int dtype, int is_complex, int M, bool poly, int* lds_bytes, int* n) {
    if (dtype == NUFFT_F32) return is_complex ? march_kernel_f32c_staged(M, poly, lds_bytes, n) : march_kernel_f32r_staged(M, poly, lds_bytes, n);
    return is_complex ? march_kernel_f64c_staged(M, poly, lds_bytes, n) : march_kernel_f64r_staged(M, poly, lds_bytes, n);
}
// is there a staged instantiation, and does its compile-time column hold a column of n1 x n2 cells?
bool interp_march_staged_available(int dtype, int is_complex, int M, bool poly, int n1, int n2) {
    int lds = 0, n[4];
    return march_staged_kernel(dtype, is_complex, M, poly, &lds, n) != nullptr && n1 <= n[0] && n2 <= n[1];
}
hipError_t prepare_interp_march_staged(int dtype, int is_complex, int M, bool poly) {
    int lds = 0, n[4];
    const void* fn = march_staged_kernel(dtype, is_complex, M, poly, &lds, n);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}
bool interp_march_available(int dtype, int is_complex, int D, int M, bool poly, const Geom& g, bool other) {
    int lds = 0, n[4];
    if (D != 3 || other || !march_kernel(dtype, is_complex, M, poly, &lds, n)) return false;
    for (int d = 0; d < 3; ++d)
        if (g.blog[d] != 2 || g.Nover[d] % 4 != 0) return false;
    // columns shorter than the axis (a point's stencil then never reaches a column from both sides)
    return n[0] + 2 * M - 1 <= g.Nover[0] && n[1] + 2 * M - 1 <= g.Nover[1] && 4 + 2 * M - 1 <= g.Nover[2];
}
hipError_t prepare_interp_march(int dtype, int is_complex, int M, bool poly) {
    int lds = 0, n[4];
    const void* fn = march_kernel(dtype, is_complex, M, poly, &lds, n);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// n1, n2 > 0: the column the plan gives the ring instead of the kernel's own (<= the compile-time columns of the kernels that will run)
ColumnTasks march_column_tasks(int dtype, int is_complex, int M, bool poly, const Geom& g, int n1, int n2) {
    int lds = 0, n[4];
    ColumnTasks ct{};
    if (!march_kernel(dtype, is_complex, M, poly, &lds, n)) return ct;
    if (n1 > 0 && n2 > 0) {
        if (n1 > n[0] || n2 > n[1] || n1 % 4 || n2 % 4) return ct;
        n[0] = n1; n[1] = n2;
    }
    ct.ncolx = (g.Nover[0] + n[0] - 1) / n[0];
    ct.ncoly = (g.Nover[1] + n[1] - 1) / n[1];
    ct.bxw = n[0] / 4;
    ct.byw = n[1] / 4;
    // segments along z: about four workgroups per CU in flight over the run, whole bin layers, at most n[2] each
    const int cols = ct.ncolx * ct.ncoly;
    int nseg = (1024 + cols - 1) / cols;
    const int min_seg = (g.nb[2] + n[2] - 1) / n[2];
    if (nseg < min_seg) nseg = min_seg;
    if (nseg > g.nb[2] / 8 && g.nb[2] / 8 >= min_seg) nseg = g.nb[2] / 8;
    if (nseg < 1) nseg = 1;
    ct.segl = (g.nb[2] + nseg - 1) / nseg;
    ct.nseg = (g.nb[2] + ct.segl - 1) / ct.segl;
    ct.ntasks = cols * ct.nseg;
    ct.zq = 1;
    ct.clo = 0;                 // a point is gathered once, by the task of its own column and layer
    ct.chi = 0;
    ct.maxlen = n[2];           // layers the kernel's run tables hold (MarchCfg::kSegMax)
    return ct;
}

// ---- spreading on the z-marching LDS ring (smarch_kernels.h) -----------------------------------------------------------
const void* smarch_kernel_f32r(int M, int halo, bool poly, int* lds_bytes, int* n);
const void* smarch_kernel_f32c(int M, int halo, bool poly, int* lds_bytes, int* n);
const void* smarch_kernel_f64r(int M, int halo, bool poly, int* lds_bytes, int* n);
const void* smarch_kernel_f64c(int M, int halo, bool poly, int* lds_bytes, int* n);
hipError_t smarch_halo_add_f32(void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, const Geom& g, int nc, int C,
                               int n1, int n2, int M, int ntx, int nty, const uint32_t* flag, hipStream_t stream);
hipError_t smarch_halo_add_f64(void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, const Geom& g, int nc, int C,
                               int n1, int n2, int M, int ntx, int nty, const uint32_t* flag, hipStream_t stream);
// halo: 0 / 2 = output-driven columns / the halo variant; poly: the instantiation with the piecewise-polynomial window
// (FastApproximation) or the direct one
static const void* smarch_kernel(int dtype, int is_complex, int M, int halo, bool poly, int* lds_bytes, int* n) {
    if (M < 2 || M > 10) return nullptr;
    if (dtype == NUFFT_F32) return is_complex ? smarch_kernel_f32c(M, halo, poly, lds_bytes, n) : smarch_kernel_f32r(M, halo, poly, lds_bytes, n);
    return is_complex ? smarch_kernel_f64c(M, halo, poly, lds_bytes, n) : smarch_kernel_f64r(M, halo, poly, lds_bytes, n);
}

// Launch model of the ring: blocks go to the 8 XCDs round-robin and to the first free CU there, in launch order (task table
// order through xcd_remap_chunked, component by component).  Returns the time of the last block in units of work.
static double smarch_makespan(const std::vector<double>& task_work, int C, int cus, int xcd_chunk) {
    const int nt = (int)task_work.size(), per = std::max(1, cus / 8);
    double mk = 0.0;
    std::vector<double> heap;
    for (int x = 0; x < 8; ++x) {
        heap.assign((size_t)per, 0.0);
        auto cmp = [](double u, double v) { return u > v; };
        for (int c = 0; c < C; ++c)
            for (int b = x; b < nt; b += 8) {
                int t = b;
                if (xcd_chunk > 0) {
                    const int group = 8 * xcd_chunk, full = nt / group * group;
                    if (b < full) { const int k = b >> 3; t = ((k / xcd_chunk) * 8 + (b & 7)) * xcd_chunk + k % xcd_chunk; }
                } else {
                    const int q = nt >> 3, r = nt & 7, k = b >> 3, xc = b & 7;
                    t = ((xc < r) ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + k;
                }
                std::pop_heap(heap.begin(), heap.end(), cmp);
                heap.back() += task_work[(size_t)t];
                std::push_heap(heap.begin(), heap.end(), cmp);
            }
        for (double h : heap) mk = std::max(mk, h);
    }
    return mk;
}

// Column and segments for this grid: the compile-time column bounds the LDS ring; within it the plan takes the column
// (multiples of the bin edge) and the number of segments along z that minimise  point visits / chip utilisation,
// where visits = prod (n + 2M - 1) / n over x, y (partial last columns counted) x (segl + c_z) / segl for the layers a
// segment visits beyond its own, and the utilisation comes from the launch model above.
SMarchPlan smarch_plan(int dtype, int is_complex, int D, int M, const Geom& g, bool other, int cus, int C, int halo, int parts) {
    SMarchPlan sp{};
    int lds = 0, n[5];
    if (halo != 2) halo = 0;
    sp.halo = halo;
    sp.parts = (parts == 2 && is_complex) ? 2 : 1;
    if (sp.parts == 2) {            // complex data as two real transforms of the same points: the real kernel, twice the launch rows
        is_complex = 0;
        C *= 2;
    }
    if (D != 3 || other || !smarch_kernel(dtype, is_complex, M, halo, true, &lds, n)) return sp;
    const bool hx = halo == 2, hy = halo == 2;
    const int xreach = hx ? (M - 1) + ((M - 1) & 1) + M : 0, yreach = hy ? 2 * M - 1 : 0;
    for (int d = 0; d < 3; ++d)
        if (g.blog[d] != 2 || g.Nover[d] % 4 != 0) return sp;
    const int L = 2 * M, hlo = n[2], hhi = n[3];
    // columns shorter than the axis (a stencil then never reaches a column from both sides); enough layers for the halo
    if (8 + L - 1 > g.Nover[0] || 8 + L - 1 > g.Nover[1] || g.nb[2] < 2 * (hlo + hhi) || g.nb[2] > 2048) return sp;
    // input-driven dimensions: the window (column + 2M) must not wrap onto itself
    if ((hx && 8 + 2 * L > g.Nover[0]) || (hy && 8 + 2 * L > g.Nover[1])) return sp;
    const int xcd_chunk = option_int("NUFFT_XCD_CHUNK", 8);
    const int force_n1 = option_int("NUFFT_SMARCH_N1", 0), force_n2 = option_int("NUFFT_SMARCH_N2", 0), force_nseg = option_int("NUFFT_SMARCH_NSEG", 0);
    const double cz = 0.5 * (hlo + hhi);               // a halo layer's points are visited, but add about half their planes
    const double fixed = 0.15;                         // per task, in layers: ring zero fill, first table, launch
    double best = 1e300;
    std::vector<double> work;
    for (int n2 = n[1]; n2 >= 8; n2 -= 4) {
        if (force_n2 && n2 != force_n2) continue;
        if (n2 + L - 1 > g.Nover[1] || (n2 < n[1] / 2 && n2 + 4 + L - 1 <= g.Nover[1] && !force_n2)) continue;
        for (int n1 = n[0]; n1 >= 8; n1 -= 4) {
            if (force_n1 && n1 != force_n1) continue;
            if (n1 + L - 1 > g.Nover[0] || (n1 < n[0] / 2 && n1 + 4 + L - 1 <= g.Nover[0] && !force_n1)) continue;
            if ((hx && n1 + 2 * L > g.Nover[0]) || (hy && n2 + 2 * L > g.Nover[1])) continue;
            // halo variant: whole columns only, wider than the reach (a cell then receives from one neighbour per side)
            if (halo == 2 && (g.Nover[0] % n1 || g.Nover[1] % n2 || n1 < xreach || n2 < yreach)) continue;
            const int ncx = (g.Nover[0] + n1 - 1) / n1, ncy = (g.Nover[1] + n2 - 1) / n2;
            if ((int64_t)ncx * ncy >= 65536) continue;
            // relative cost of a layer of each column: the points it visits
            std::vector<double> colw((size_t)ncx * ncy);
            double ideal = 0.0;
            for (int ty = 0; ty < ncy; ++ty)
                for (int tx = 0; tx < ncx; ++tx) {
                    const int e1 = std::min(n1, g.Nover[0] - tx * n1), e2 = std::min(n2, g.Nover[1] - ty * n2);
                    // (an input-driven dimension visits only the column's own points)
                    colw[(size_t)ty * ncx + tx] = (double)(hx ? e1 : e1 + L - 1) * (double)(hy ? e2 : e2 + L - 1);
                    ideal += (double)e1 * (double)e2;
                }
            ideal *= (double)g.nb[2] * C / cus;         // every point visited once, the chip evenly busy
            const int max_seg = std::max(1, g.nb[2] / 4);
            for (int nseg = 1; nseg <= std::min(max_seg, 64); ++nseg) {
                if (force_nseg && nseg != force_nseg) continue;
                const int segl = (g.nb[2] + nseg - 1) / nseg, ns = (g.nb[2] + segl - 1) / segl;
                if (ns != nseg && !force_nseg) continue;
                work.clear();
                for (int k = 0; k < ns; ++k) {
                    const int len = std::min(segl, g.nb[2] - k * segl);
                    for (double w : colw) work.push_back(w * (len + cz + fixed));
                }
                double cost = smarch_makespan(work, C, cus, xcd_chunk) / ideal;
                // halo variant: the reach is retired, stored and read again by the FFT pass — cost in proportion to its share of
                // the column (C2, 32 x 32: 0.49 of the grid's bytes, + 0.15 ms of FFT pass beside 1.87 ms of spreading; measured
                // 256^3 / 1e6 points: 16 x 16 x 1 segment 0.48 ms against 0.41 ms for 32 x 32 x 4 without this term)
                if (halo == 2) cost += 0.25 * ((double)(n1 + xreach) * (n2 + yreach) / ((double)n1 * n2) - 1.0);
                if (cost < best) {
                    best = cost;
                    sp.n1 = n1; sp.n2 = n2;
                    sp.ct = ColumnTasks{ncx, ncy, n1 / 4, n2 / 4, ns, segl, ncx * ncy * ns, 1, -hhi, hlo, 0};
                    double vis = 0.0;
                    for (double w : colw) vis += w;
                    sp.visits = vis / ((double)g.Nover[0] * g.Nover[1]);
                    sp.efficiency = sp.visits / cost;
                }
            }
        }
    }
    if (best >= 1e300) return sp;
    sp.hlo = hlo; sp.hhi = hhi;
    if (halo == 2) sp.halo_reals = make_halo_layout(sp.n1, sp.n2, M, is_complex ? 2 : 1, sp.ct.ncolx, sp.ct.ncoly).plane * g.Nover[2];
    sp.lds_bytes = lds;
    sp.threads = n[4];
    sp.eligible = true;
    return sp;
}

// (is_complex = 0 for complex plans whose SMarchPlan::parts is 2)
hipError_t prepare_spread_march(int dtype, int is_complex, int M, int halo) {
    for (int poly = 0; poly < 2; ++poly) {
        int lds = 0, n[5];
        const void* fn = smarch_kernel(dtype, is_complex, M, halo, poly != 0, &lds, n);
        if (!fn) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

const void* spread_kernel_f32r(int D, int M, bool flag, bool other);
const void* spread_kernel_f32c(int D, int M, bool flag, bool other);
const void* spread_kernel_f64r(int D, int M, bool flag, bool other);
const void* spread_kernel_f64c(int D, int M, bool flag, bool other);
const void* interp_kernel_f32r(int D, int M, bool flag, bool other);
const void* interp_kernel_f32c(int D, int M, bool flag, bool other);
const void* interp_kernel_f64r(int D, int M, bool flag, bool other);
const void* interp_kernel_f64c(int D, int M, bool flag, bool other);

const void* spread_fixed_f32r(int D, int M, int* n);
const void* spread_fixed_f32c(int D, int M, int* n);
const void* spread_fixed_f64r(int D, int M, int* n);
const void* spread_fixed_f64c(int D, int M, int* n);
const void* spread_cubes_f32r(int D, int M);
const void* spread_cubes_f32c(int D, int M);
const void* spread_cubes_f64r(int D, int M);
const void* spread_cubes_f64c(int D, int M);
void interp_fixed_dims_f32r(int D, int M, int* n);
void interp_fixed_dims_f32c(int D, int M, int* n);
void interp_fixed_dims_f64r(int D, int M, int* n);
void interp_fixed_dims_f64c(int D, int M, int* n);

void interp_fixed_dims(int dtype, int is_complex, int D, int M, int* n) {
    n[0] = n[1] = n[2] = n[3] = 0;
    if (M < 2 || M > 10 || D < 1 || D > 3) return;
    if (dtype == NUFFT_F32) is_complex ? interp_fixed_dims_f32c(D, M, n) : interp_fixed_dims_f32r(D, M, n);
    else is_complex ? interp_fixed_dims_f64c(D, M, n) : interp_fixed_dims_f64r(D, M, n);
}

// Kernel with the compile-time spreading tile (null: none) and the tile itself (n[0..2], n[3] = row stride).
const void* spread_fixed_kernel(int dtype, int is_complex, int D, int M, int* n) {
    n[0] = n[1] = n[2] = n[3] = n[4] = 0;
    if (M < 2 || M > 10 || D < 1 || D > 3) return nullptr;
    if (dtype == NUFFT_F32) return is_complex ? spread_fixed_f32c(D, M, n) : spread_fixed_f32r(D, M, n);
    return is_complex ? spread_fixed_f64c(D, M, n) : spread_fixed_f64r(D, M, n);
}
void spread_fixed_dims(int dtype, int is_complex, int D, int M, int* n) { (void)spread_fixed_kernel(dtype, is_complex, D, M, n); }
// cube-accumulation variant of that kernel (real data, 3-D, M <= 4), or null
const void* spread_cubes_kernel(int dtype, int is_complex, int D, int M) {
    if (M < 2 || M > 10 || D != 3) return nullptr;
    if (dtype == NUFFT_F32) return is_complex ? spread_cubes_f32c(D, M) : spread_cubes_f32r(D, M);
    return is_complex ? spread_cubes_f64c(D, M) : spread_cubes_f64r(D, M);
}
bool spread_cubes_available(int dtype, int is_complex, int D, int M) { return spread_cubes_kernel(dtype, is_complex, D, M) != nullptr; }

// `flag`: spreading = single-tile axis (wrap variant); interpolation = compile-time tile.
// `other`: window evaluation of the non-default kernels (see needs_other_eval).
static const void* pick(bool interp, int dtype, int is_complex, int D, int M, bool flag, bool other) {
    if (interp) {
        if (dtype == NUFFT_F32) return is_complex ? interp_kernel_f32c(D, M, flag, other) : interp_kernel_f32r(D, M, flag, other);
        return is_complex ? interp_kernel_f64c(D, M, flag, other) : interp_kernel_f64r(D, M, flag, other);
    }
    if (dtype == NUFFT_F32) return is_complex ? spread_kernel_f32c(D, M, flag, other) : spread_kernel_f32r(D, M, flag, other);
    return is_complex ? spread_kernel_f64c(D, M, flag, other) : spread_kernel_f64r(D, M, flag, other);
}

// The polynomial evaluation (FastApproximation of both Kaiser-Bessel kernels) and the sinh form of the
// backwards Kaiser-Bessel kernel live in the default instantiations; everything else in the OTHERK ones.
bool needs_other_eval(int kernel, int evalmode) {
    if (kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) return false;
    if (kernel == NUFFT_KERNEL_KAISER_BESSEL) return evalmode == NUFFT_EVAL_DIRECT;
    return true;
}

static hipError_t prepare(bool interp, int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    // The attribute belongs to the kernel, not to the plan: another plan of the same instantiation with a smaller tile must not lower it
    // under what this plan launches with — always the whole LDS of a gfx950 workgroup (the launch itself asks for the plan's bytes).
    if (lds_bytes > 163840) return hipErrorInvalidValue;
    lds_bytes = 163840;
    for (int wrap = 0; wrap < 2; ++wrap) {
        const void* fn = pick(interp, dtype, is_complex, D, M, wrap != 0, other);
        if (!fn && interp && wrap) continue;       // no compile-time tile for this instantiation
        if (!fn) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (!interp && !other) {
        int n[5];
        const void* fn = spread_fixed_kernel(dtype, is_complex, D, M, n);
        if (fn) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
        }
        fn = spread_cubes_kernel(dtype, is_complex, D, M);
        if (fn) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

hipError_t prepare_spread(int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    return prepare(false, dtype, is_complex, D, M, lds_bytes, other);
}
hipError_t prepare_interp(int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    return prepare(true, dtype, is_complex, D, M, lds_bytes, other);
}

// kernel arguments of components c0 .. c0 + nc - 1
template <typename T>
static TileArgs<T> fill_tile_args(const TileKernelArgs& a, int c0, int nc) {
    const int ncr = a.is_complex ? 2 : 1;
    TileArgs<T> k{};
    k.g = a.g;
    k.sorted = a.sorted;
    k.offsets = a.offsets;
    k.coefs = static_cast<const T*>(a.coefs);
    for (int d = 0; d < 3; ++d) {
        k.beta[d] = (T)a.beta[d];
        k.bop[d] = (T)a.beta_over_pi[d];
    }
    for (int c = 0; c < nc; ++c) {
        k.grid[c] = static_cast<T*>(a.grid) + (int64_t)(c0 + c) * a.grid_stride * ncr;
        k.vin[c] = a.values_in ? static_cast<const T*>(a.values_in[c0 + c]) : nullptr;
        k.vout[c] = a.values_out ? static_cast<T*>(a.values_out[c0 + c]) : nullptr;
    }
    k.prefactor = (T)a.prefactor;
    k.weights = static_cast<const T*>(a.weights);
    k.desc = static_cast<const uint2*>(a.desc);
    k.desc_total = a.desc_total;
    k.xcd_chunk = a.xcd_chunk;
    k.march_flag = nullptr;
    k.evalmode = a.evalmode;
    k.kernel = a.kernel;
    return k;
}

template <typename T>
static hipError_t launch_t(bool interp, const TileKernelArgs& a, hipStream_t stream) {
    bool wrap = false;
    for (int d = 0; d < a.D; ++d) wrap = wrap || a.g.sp.nt[d] == 1;
    const bool other = needs_other_eval(a.kernel, a.evalmode) || a.weights != nullptr;   // general variant
    const void* fn = pick(interp, a.dtype, a.is_complex, a.D, a.M, interp ? (a.fixed_tile != 0 && !other) : wrap, other);
    if (!interp && a.fixed_tile != 0 && !other && !wrap) {
        int n[5];
        const void* ff = spread_fixed_kernel(a.dtype, a.is_complex, a.D, a.M, n);
        if (ff) fn = ff;
        if (ff && a.cubes) {
            const void* fc = spread_cubes_kernel(a.dtype, a.is_complex, a.D, a.M);
            if (fc) fn = fc;
        }
    }
    if (!fn) return hipErrorInvalidValue;
    // z-marching interpolation: launched next to the tile kernel; the flag set_points left on the device (heaviest ring task
    // and total work against the ring's advantage limit, balance.hip) decides which of the two finds work
    // (the ring applies the per-point weights of the callback menu itself; the tile kernel next to it is then the general variant)
    const bool march = interp && a.march != 0 && !needs_other_eval(a.kernel, a.evalmode);
    for (int c0 = 0; c0 < a.C; c0 += kMaxCompPerLaunch) {
        const int nc = (a.C - c0) < kMaxCompPerLaunch ? (a.C - c0) : kMaxCompPerLaunch;
        TileArgs<T> k = fill_tile_args<T>(a, c0, nc);
        k.march_flag = march ? a.march_flag : nullptr;
        void* params[] = {&k};
        hipError_t e = hipLaunchKernel(fn, dim3((unsigned)a.ntiles, (unsigned)nc, 1), dim3((unsigned)a.threads, 1, 1),
                                       params, (size_t)a.lds_bytes, stream);
        if (e != hipSuccess) return e;
        if (march) {
            int lds = 0, n[4];
            const int parts = (a.interp_parts == 2 && a.is_complex) ? 2 : 1;      // complex data part by part through the real kernel
            const int mcplx = parts == 2 ? 0 : a.is_complex;
            const void* mfn = march_kernel(a.dtype, mcplx, a.M, a.evalmode != NUFFT_EVAL_DIRECT, &lds, n);
            MarchGeom mg{};
            mg.parts = parts;
            mg.ntx = a.march_ct.ncolx;
            mg.nty = a.march_ct.ncoly;
            mg.nseg = a.march_ct.nseg;
            mg.segl = a.march_ct.segl;
            mg.ntasks = column_task_table_entries(a.march_ct, a.g.nb[2]);
            mg.flag = a.march_flag;
            mg.tasktab = a.march_tasks;
            mg.coarse_a = a.coarse ? a.coarse_a : nullptr;
            mg.coarse_b = a.coarse ? a.coarse_b : nullptr;
            mg.n1 = 4 * a.march_ct.bxw;      // the column of this plan (= the kernel's own unless the plan shares the spreading window's)
            mg.n2 = 4 * a.march_ct.byw;
            void* mparams[] = {&k, &mg};
            // (parts = 2: both parts of a task side by side on one XCD — march_setup.inc)
            const unsigned gx = parts == 2 ? 2u * (((unsigned)mg.ntasks + 7u) & ~7u) : (unsigned)mg.ntasks;
            e = hipLaunchKernel(mfn, dim3(gx, (unsigned)nc, 1), dim3((unsigned)n[3], 1, 1), mparams, (size_t)lds, stream);
            if (e != hipSuccess) return e;
            if (a.coarse) {
                // plans of the column-layer sort: the staged kernel serves the point sets sorted that way (device flags), the plain one the others
                int slds = 0, sn[4];
                const void* sfn = march_staged_kernel(a.dtype, mcplx, a.M, a.evalmode != NUFFT_EVAL_DIRECT, &slds, sn);
                if (!sfn) return hipErrorInvalidValue;
                e = hipLaunchKernel(sfn, dim3(gx, (unsigned)nc, 1), dim3((unsigned)sn[3], 1, 1), mparams, (size_t)slds, stream);
                if (e != hipSuccess) return e;
            }
        }
    }
    return hipSuccess;
}

// ---- spreading on MFMA patches (patch_kernels.h) ------------------------------------------------------------------
const void* patch_kernel_f32r(int M, bool other, int* lds_bytes, int* pby);
const void* patch_kernel_f32c(int M, bool other, int* lds_bytes, int* pby);
const void* patch_kernel_f64r(int M, bool other, int* lds_bytes, int* pby);
const void* patch_kernel_f64c(int M, bool other, int* lds_bytes, int* pby);

const void* patch32_kernel_f32c(int M, bool other, int* lds_bytes, int* pby);
const void* patch_planar_kernel_f32r(int NP, int M, int* lds_bytes, int* pby);
const void* patch_planar_kernel_f64r(int NP, int M, int* lds_bytes, int* pby);
static const void* patch_planar_kernel(int dtype, int NP, int M, int* lds_bytes, int* pby) {
    return dtype == NUFFT_F32 ? patch_planar_kernel_f32r(NP, M, lds_bytes, pby) : patch_planar_kernel_f64r(NP, M, lds_bytes, pby);
}

static const void* patch_kernel(int dtype, int is_complex, int M, bool other, int* lds_bytes, int* pby) {
    if (dtype == NUFFT_F32) return is_complex ? patch_kernel_f32c(M, other, lds_bytes, pby) : patch_kernel_f32r(M, other, lds_bytes, pby);
    return is_complex ? patch_kernel_f64c(M, other, lds_bytes, pby) : patch_kernel_f64r(M, other, lds_bytes, pby);
}

// Patch decomposition of a plan, or eligible = false: 3-D grids of 4-cell bins whose axes are multiples of the bin
// edge and long enough that the bins a patch visits are distinct and a stencil cannot reach a patch from both
// sides; default window evaluation without per-point weights (those use the LDS-tile kernel).
PatchPlan patch_plan(int dtype, int is_complex, int D, int M, const Geom& g, bool other, bool allow_f32acc, int planar_nc) {
    PatchPlan pp{};
    int lds = 0, pby = 0;
    if (D != 3 || !patch_kernel(dtype, is_complex, M, other, &lds, &pby)) return pp;
    // real plans with ntransforms = 2 / 3: the components together (shared windows and operands), where that kernel exists
    int planar = 0;
    if (!is_complex && !other && (planar_nc == 2 || planar_nc == 3)) {
        int ldsp = 0, pbyp = 0;
        if (patch_planar_kernel(dtype, planar_nc, M, &ldsp, &pbyp)) { planar = planar_nc; lds = ldsp; pby = pbyp; }
    }
    const int clo = floor_div4(1 - M), chi = floor_div4(3 + M), ncb = chi - clo + 1;
    // ComplexF32: the FP32 matrix pipe with Float32 accumulators (octets of 8 planes: dimension 3 a multiple of 8), if its
    // larger patch still fits the grid; the Float64-accumulating kernel otherwise
    bool f32acc = false;
    if (allow_f32acc && dtype == NUFFT_F32 && is_complex && g.Nover[2] % 8 == 0) {
        int lds32 = 0, pby32 = 0;
        if (patch32_kernel_f32c(M, other, &lds32, &pby32) && g.nb[1] >= 2 * (pby32 + ncb)) { f32acc = true; lds = lds32; pby = pby32; }
    }
    const int pb[3] = {4, pby, 1};
    for (int d = 0; d < 3; ++d) {
        if (g.blog[d] != 2 || g.Nover[d] % 4 != 0) return pp;
        if (g.nb[d] < 2 * (pb[d] + ncb)) return pp;
    }
    if (!patch_tasks_supported(g)) return pp;          // (more bin layers than set_points' task splitter holds)
    pp.npx = (g.nb[0] + 3) / 4;
    pp.npy = (g.nb[1] + pby - 1) / pby;
    // segments along dimension 3: enough tasks for ~4 rounds of the 2048 resident waves, at least 8 cube layers each
    // (a segment visits ncb - 1 bin layers beyond its own)
    const int cols = pp.npx * pp.npy;
    const int task_target = option_int("NUFFT_PATCH_TASKS", 8192);
    int nseg = (task_target + cols - 1) / cols;
    const int max_seg = g.nb[2] / 8 > 0 ? g.nb[2] / 8 : 1;
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    pp.segl = (g.nb[2] + nseg - 1) / nseg;
    if (f32acc) pp.segl += pp.segl & 1;               // whole octets per segment
    pp.nseg = (g.nb[2] + pp.segl - 1) / pp.segl;
    pp.ntasks = cols * pp.nseg;
    pp.lds_bytes = lds;
    pp.pby = pby;
    pp.occ = f32acc ? NUFFT_PATCH32_OCC : patch_occupancy(planar ? planar : (is_complex ? 2 : 1), M);
    pp.f32acc = f32acc ? 1 : 0;
    pp.planar = planar;
    pp.eligible = true;
    return pp;
}

hipError_t prepare_spread_patch(int dtype, int is_complex, int M, bool other, int planar_nc) {
    int lds = 0, pby = 0;
    if (planar_nc) {
        const void* pf = patch_planar_kernel(dtype, planar_nc, M, &lds, &pby);
        if (!pf) return hipErrorInvalidValue;
        return hipFuncSetAttribute(pf, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    const void* fn = patch_kernel(dtype, is_complex, M, other, &lds, &pby);
    if (!fn) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (dtype == NUFFT_F32 && is_complex && (fn = patch32_kernel_f32c(M, other, &lds, &pby)))
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}

template <typename T>
static hipError_t launch_patch_t(const TileKernelArgs& a, const PatchPlan& pp, const void* vsorted, int64_t vstride_reals,
                                 const uint32_t* enabled, const uint2* tasktab, hipStream_t stream) {
    int lds = 0, pby = 0;
    const void* fn = pp.planar ? patch_planar_kernel(a.dtype, pp.planar, a.M, &lds, &pby)
                   : (pp.f32acc ? patch32_kernel_f32c(a.M, false, &lds, &pby) : patch_kernel(a.dtype, a.is_complex, a.M, false, &lds, &pby));
    if (!fn) return hipErrorInvalidValue;
    for (int c0 = 0; c0 < a.C; c0 += kMaxCompPerLaunch) {
        int nc = (a.C - c0) < kMaxCompPerLaunch ? (a.C - c0) : kMaxCompPerLaunch;
        PatchArgs<T> k{};
        k.t = fill_tile_args<T>(a, c0, nc);
        k.pg.npx = pp.npx; k.pg.npy = pp.npy; k.pg.nseg = pp.nseg; k.pg.segl = pp.segl;
        k.pg.ntasks = tasktab ? patch_task_table_entries(pp) : pp.ntasks;
        for (int c = 0; c < nc; ++c) k.vsorted[c] = static_cast<const T*>(vsorted) + (int64_t)(c0 + c) * vstride_reals;
        if (pp.planar) {                 // all components in one launch: vsorted is one interleaved buffer, gridDim.y = 1
            k.vsorted[0] = static_cast<const T*>(vsorted);
            nc = 1;
        }
        k.prof = nullptr;
        k.enabled = enabled;
        k.tasktab = tasktab;
#if defined(NUFFT_PATCH_PROFILE)
        static unsigned long long* prof_dev = nullptr;       // development builds: phase cycles of the last launch on stderr
        if (!prof_dev) { (void)hipMalloc(&prof_dev, 8 * sizeof(unsigned long long)); }
        (void)hipMemsetAsync(prof_dev, 0, 8 * sizeof(unsigned long long), stream);
        k.prof = prof_dev;
#endif
        void* params[] = {&k};
        const unsigned nwg = (unsigned)((k.pg.ntasks + kPatchWaves - 1) / kPatchWaves);
        hipError_t e = hipLaunchKernel(fn, dim3(nwg, (unsigned)nc, 1), dim3(kPatchWaves * kWave, 1, 1), params, (size_t)lds, stream);
        if (e != hipSuccess) return e;
#if defined(NUFFT_PATCH_PROFILE)
        unsigned long long h[8];
        (void)hipMemcpyAsync(h, k.prof, sizeof(h), hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        fprintf(stderr, "patch phases (%% of wave cycles): advance+prefetch %.1f  retire %.1f  commit %.1f  eval %.1f  batches %.1f  (total %.3g cycles)\n",
                100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, tot);
        if (h[5]) fprintf(stderr, "patch32 counts: %.4g matrix instructions, %.4g K-batches (%.2f matrix instructions each), %.4g batch-columns\n",
                          (double)h[5], (double)h[6], (double)h[5] / (double)h[6], (double)h[7]);
#endif
    }
    return hipSuccess;
}

hipError_t launch_spread_patch(const TileKernelArgs& a, const PatchPlan& pp, const void* vsorted, int64_t vstride_reals,
                               const uint32_t* enabled, const uint2* tasktab, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_patch_t<float>(a, pp, vsorted, vstride_reals, enabled, tasktab, stream)
                                : launch_patch_t<double>(a, pp, vsorted, vstride_reals, enabled, tasktab, stream);
}

// values of one component in sorted order (times the per-point weights of the callback menu)
template <typename T, int NC>
static hipError_t gather_t(int D, const void* sorted, int64_t np, const void* vin, const void* weights, void* vout,
                           const uint32_t* enabled, hipStream_t stream) {
    if (np <= 0) return hipSuccess;
    const int rec_bytes = (int)sizeof(T) * D + 4 > 16 ? 32 : ((int)sizeof(T) * D + 4 > 8 ? 16 : 8);
    const int idx_off = D * (int)sizeof(T);
    int64_t blocks = (np + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const unsigned char* r = static_cast<const unsigned char*>(sorted);
    const T* vi = static_cast<const T*>(vin);
    const T* w = static_cast<const T*>(weights);
    T* vo = static_cast<T*>(vout);
    switch (rec_bytes) {
        case 8: hipLaunchKernelGGL((gather_values_kernel<T, NC, 8>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, vi, w, vo, enabled); break;
        case 16: hipLaunchKernelGGL((gather_values_kernel<T, NC, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, vi, w, vo, enabled); break;
        default: hipLaunchKernelGGL((gather_values_kernel<T, NC, 32>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, vi, w, vo, enabled); break;
    }
    return hipGetLastError();
}

hipError_t launch_gather_values(int dtype, int is_complex, int D, const void* sorted, int64_t np, const void* vin,
                                const void* weights, void* vout, const uint32_t* enabled, hipStream_t stream) {
    if (dtype == NUFFT_F32) return is_complex ? gather_t<float, 2>(D, sorted, np, vin, weights, vout, enabled, stream)
                                              : gather_t<float, 1>(D, sorted, np, vin, weights, vout, enabled, stream);
    return is_complex ? gather_t<double, 2>(D, sorted, np, vin, weights, vout, enabled, stream)
                      : gather_t<double, 1>(D, sorted, np, vin, weights, vout, enabled, stream);
}

template <typename T, int NC>
static hipError_t gather_planar_t(int D, const void* sorted, int64_t np, const void* const* vin, const void* weights, void* vout,
                                  const uint32_t* enabled, hipStream_t stream) {
    if (np <= 0) return hipSuccess;
    const int rec_bytes = (int)sizeof(T) * D + 4 > 16 ? 32 : ((int)sizeof(T) * D + 4 > 8 ? 16 : 8);
    const int idx_off = D * (int)sizeof(T);
    int64_t blocks = (np + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    PlanarPtrs<T> pv{};
    for (int c = 0; c < NC; ++c) pv.p[c] = static_cast<const T*>(vin[c]);
    const unsigned char* r = static_cast<const unsigned char*>(sorted);
    const T* w = static_cast<const T*>(weights);
    T* vo = static_cast<T*>(vout);
    switch (rec_bytes) {
        case 8: hipLaunchKernelGGL((gather_planar_kernel<T, NC, 8>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, pv, w, vo, enabled); break;
        case 16: hipLaunchKernelGGL((gather_planar_kernel<T, NC, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, pv, w, vo, enabled); break;
        default: hipLaunchKernelGGL((gather_planar_kernel<T, NC, 32>), dim3((unsigned)blocks), dim3(256), 0, stream, r, idx_off, np, pv, w, vo, enabled); break;
    }
    return hipGetLastError();
}

hipError_t launch_gather_planar(int dtype, int D, const void* sorted, int64_t np, const void* const* vin, int C,
                                const void* weights, void* vout, const uint32_t* enabled, hipStream_t stream) {
    if (C == 2) return dtype == NUFFT_F32 ? gather_planar_t<float, 2>(D, sorted, np, vin, weights, vout, enabled, stream)
                                          : gather_planar_t<double, 2>(D, sorted, np, vin, weights, vout, enabled, stream);
    if (C == 3) return dtype == NUFFT_F32 ? gather_planar_t<float, 3>(D, sorted, np, vin, weights, vout, enabled, stream)
                                          : gather_planar_t<double, 3>(D, sorted, np, vin, weights, vout, enabled, stream);
    return hipErrorInvalidValue;
}

// ---- dense-set engine on the same window (dmarch_kernels.h): matrix-pipe register accumulation per bin ----
const void* dmarch_kernel_f32r(int M, bool poly, int* lds_bytes, int* n);
const void* dmarch_kernel_f64r(int M, bool poly, int* lds_bytes, int* n);
static const void* dmarch_kernel(int dtype, int M, bool poly, int* lds_bytes, int* n) {
    return dtype == NUFFT_F32 ? dmarch_kernel_f32r(M, poly, lds_bytes, n) : dmarch_kernel_f64r(M, poly, lds_bytes, n);
}
// plans of the spreading window's halo variant on real data (or complex data part by part), M <= 6, a column the kernel's window holds
bool spread_dense_available(int dtype, int is_complex, int M, bool poly, const SMarchPlan& sp) {
    int lds = 0, n[4];
    if (!sp.eligible || sp.halo != 2 || (is_complex && sp.parts != 2)) return false;
    return dmarch_kernel(dtype, M, poly, &lds, n) != nullptr && sp.n1 <= n[0] && sp.n2 <= n[1];
}
hipError_t prepare_spread_dense(int dtype, int M, bool poly) {
    int lds = 0, n[4];
    const void* fn = dmarch_kernel(dtype, M, poly, &lds, n);
    if (!fn) return hipErrorInvalidValue;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <typename T>
static hipError_t launch_smarch_t(const TileKernelArgs& a, const SMarchPlan& sp, const uint32_t* flag, const uint2* tasktab, uint32_t* halo_state, bool dense, hipStream_t stream) {
    int lds = 0, n[5];
    const int parts = (sp.parts == 2 && a.is_complex) ? 2 : 1;      // complex data part by part through the real kernel
    const void* fn = smarch_kernel(a.dtype, parts == 2 ? 0 : a.is_complex, a.M, sp.halo, a.evalmode != NUFFT_EVAL_DIRECT, &lds, n);
    if (dense) {
        int dn[4];
        fn = dmarch_kernel(a.dtype, a.M, a.evalmode != NUFFT_EVAL_DIRECT, &lds, dn);
        n[4] = dn[2];
    }
    if (!fn) return hipErrorInvalidValue;
    if (sp.halo == 2 && !a.halo) return hipErrorInvalidValue;
    for (int c0 = 0; c0 < a.C; c0 += kMaxCompPerLaunch) {
        const int nc = (a.C - c0) < kMaxCompPerLaunch ? (a.C - c0) : kMaxCompPerLaunch;
        TileArgs<T> k = fill_tile_args<T>(a, c0, nc);
        MarchGeom mg{};
        mg.ntx = sp.ct.ncolx;
        mg.nty = sp.ct.ncoly;
        mg.nseg = sp.ct.nseg;
        mg.segl = sp.ct.segl;
        mg.ntasks = column_task_table_entries(sp.ct, a.g.nb[2]);
        mg.flag = flag;
        mg.tasktab = tasktab;
        mg.n1 = sp.n1;
        mg.n2 = sp.n2;
        mg.halo = sp.halo == 2 ? static_cast<void*>(static_cast<T*>(a.halo) + (int64_t)c0 * parts * sp.halo_reals) : nullptr;
        mg.halo_comp = sp.halo_reals;
        mg.parts = parts;
        mg.halo_state = sp.halo == 2 ? halo_state : nullptr;
        void* params[] = {&k, &mg};
        // (parts = 2: both parts of a task side by side on one XCD — smarch_kernels.h)
        const unsigned gx = parts == 2 ? 2u * (((unsigned)mg.ntasks + 7u) & ~7u) : (unsigned)mg.ntasks;
        hipError_t e = hipLaunchKernel(fn, dim3(gx, (unsigned)nc, 1), dim3((unsigned)n[4], 1, 1), params, (size_t)lds, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_spread_march(const TileKernelArgs& a, const SMarchPlan& sp, const uint32_t* flag, const uint2* tasktab, uint32_t* halo_state, bool dense, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_smarch_t<float>(a, sp, flag, tasktab, halo_state, dense, stream) : launch_smarch_t<double>(a, sp, flag, tasktab, halo_state, dense, stream);
}
hipError_t launch_smarch_halo_add(const TileKernelArgs& a, const SMarchPlan& sp, const uint32_t* flag, hipStream_t stream) {
    if (sp.halo != 2) return hipSuccess;
    if (!a.halo) return hipErrorInvalidValue;
    const int ncr = a.is_complex ? 2 : 1;
    if (sp.parts == 2 && a.is_complex) {
        // planar side buffers of the real and imaginary parts (complex data through the real kernel) onto the interleaved grid
        const HaloLayout h = make_halo_layout(sp.n1, sp.n2, a.M, 1, sp.ct.ncolx, sp.ct.ncoly);
        return launch_halo_add_lines(a.dtype, a.grid, a.halo, a.grid_stride * ncr, sp.halo_reals, a.g.Nover[0], a.g.Nover[1], a.g.Nover[2], a.C, h, flag, stream, true);
    }
    // line by line through LDS (fft_lines.hip: 0.6 ms at C2); the element-wise gather kernel (1.5 ms) where a line does not fit
    const bool gather = option_int("NUFFT_SMARCH_HALO_ADD_GATHER", 0) != 0;
    if (!gather) {
        const HaloLayout h = make_halo_layout(sp.n1, sp.n2, a.M, ncr, sp.ct.ncolx, sp.ct.ncoly);
        hipError_t e = launch_halo_add_lines(a.dtype, a.grid, a.halo, a.grid_stride * ncr, sp.halo_reals, a.g.Nover[0], a.g.Nover[1], a.g.Nover[2], a.C, h, flag, stream);
        if (e != hipErrorInvalidValue) return e;
        (void)hipGetLastError();
    }
    return a.dtype == NUFFT_F32
        ? smarch_halo_add_f32(a.grid, a.halo, a.grid_stride * ncr, sp.halo_reals, a.g, ncr, a.C, sp.n1, sp.n2, a.M, sp.ct.ncolx, sp.ct.ncoly, flag, stream)
        : smarch_halo_add_f64(a.grid, a.halo, a.grid_stride * ncr, sp.halo_reals, a.g, ncr, a.C, sp.n1, sp.n2, a.M, sp.ct.ncolx, sp.ct.ncoly, flag, stream);
}

hipError_t launch_spread(const TileKernelArgs& a, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_t<float>(false, a, stream) : launch_t<double>(false, a, stream);
}
hipError_t launch_interp(const TileKernelArgs& a, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_t<float>(true, a, stream) : launch_t<double>(true, a, stream);
}

}  // namespace nufft
