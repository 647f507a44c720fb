// Pruned 1-D FFT passes over strided lines (lengths 2^a, 1.5 * 2^a, 1.25 * 2^a), fused with deconvolution.
//
// exec_type1! needs only N_d of the Ñ_d modes of the oversampled spectrum (N_d ≈ Ñ_d / σ), exec_type2!
// feeds a spectrum that is zero outside those modes.  The reference runs the dense multi-dimensional FFT
// (rocFFT / FFTW: `_type1_fft!`, `_type2_fft!`, src/NonuniformFFTs.jl:197-211,293-314) and truncates /
// zero-pads in separate passes (copy_deconvolve_to_non_oversampled!, :387-414; fill_with_zeros +
// copy_deconvolve_to_oversampled!, :260-272,453-480).  For real 3-D (and 2-D) plans whose higher dimensions are
// powers of two this file replaces everything after / before the dimension-1 r2c / c2r transform (which
// stays in rocFFT):
//
//   type 1:  rocFFT r2c along dim 1  ->  [pass 2: FFT along dim 2, only for the k1 that are kept, only the
//            kept k2 are stored, times 1/ϕ̂2]  ->  [pass 3: FFT along dim 3, kept k3 only, times
//            normfactor/(ϕ̂1 ϕ̂3), stored straight into the caller's array]
//   type 2:  [pass 3': zero-padded inverse FFT along dim 3 read straight from the caller's array, times
//            1/(ϕ̂1 ϕ̂3)]  ->  [pass 2': zero-padded inverse FFT along dim 2, times 1/ϕ̂2, columns k1 >= N1 of the
//            oversampled spectrum written as zeros]  ->  rocFFT c2r along dim 1
//
// HBM traffic at C2 (256³ -> 512³, Float64): 1.2 GB instead of 4.6 GB (two dense c2c passes + deconvolution).
//
// One workgroup transforms TA consecutive lines (consecutive in the contiguous index `a`), one wave per line:
// global accesses are TA * 16-byte segments, the lines sit in LDS, each wave runs an in-place Stockham FFT
// (radix 8, then 4 or 2, then 3 / 5) on its own line with twiddles from an LDS table.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <vector>

#include "kernels.h"

namespace nufft {

template <typename T> struct Cplx2;
template <> struct Cplx2<float>  { using type = float2; };
template <> struct Cplx2<double> { using type = double2; };

template <typename C> __device__ __forceinline__ C cadd(C a, C b) { C r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { C r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b) { C r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
// multiply by -i (SIGN = -1, forward) or +i (SIGN = +1, backward)
template <int SIGN, typename C> __device__ __forceinline__ C mul_i(C a) {
    C r;
    if (SIGN < 0) { r.x = a.y; r.y = -a.x; } else { r.x = -a.y; r.y = a.x; }
    return r;
}

template <int SIGN, typename C> __device__ __forceinline__ void dft2(C* u) {
    const C a = u[0], b = u[1];
    u[0] = cadd(a, b);
    u[1] = csub(a, b);
}
template <int SIGN, typename C> __device__ __forceinline__ void dft4(C* u) {
    const C e0 = cadd(u[0], u[2]), e1 = csub(u[0], u[2]);
    const C o0 = cadd(u[1], u[3]), o1 = mul_i<SIGN>(csub(u[1], u[3]));
    u[0] = cadd(e0, o0); u[1] = cadd(e1, o1); u[2] = csub(e0, o0); u[3] = csub(e1, o1);
}
template <int SIGN, typename T, typename C> __device__ __forceinline__ void dft8(C* u) {
    const T h = T(0.70710678118654752440);
    C s[4], d[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { s[t] = cadd(u[t], u[t + 4]); d[t] = csub(u[t], u[t + 4]); }
    // d1 *= w8, d2 *= w8^2 = -+i, d3 *= w8^3   (w8 = exp(SIGN * 2πi / 8))
    { C w; w.x = h; w.y = SIGN * h; d[1] = cmul(d[1], w); }
    d[2] = mul_i<SIGN>(d[2]);
    { C w; w.x = -h; w.y = SIGN * h; d[3] = cmul(d[3], w); }
    dft4<SIGN>(s);
    dft4<SIGN>(d);
#pragma unroll
    for (int t = 0; t < 4; ++t) { u[2 * t] = s[t]; u[2 * t + 1] = d[t]; }
}

// radix 3 and 5 (the oversampled sizes are products of 2, 3 and 5: nextprod((2, 3, 5), ...), src/plan.jl:485-498)
template <int SIGN, typename T, typename C> __device__ __forceinline__ void dft3(C* u) {
    const T s3 = T(0.86602540378443864676);            // sin(2 pi / 3)
    const C t1 = cadd(u[1], u[2]);
    C m1; m1.x = u[0].x - T(0.5) * t1.x; m1.y = u[0].y - T(0.5) * t1.y;
    const C d = csub(u[1], u[2]);
    C m2; m2.x = s3 * d.x; m2.y = s3 * d.y;
    const C im2 = mul_i<SIGN>(m2);                     // SIGN * i * sin(2 pi / 3) * (u1 - u2)
    u[0] = cadd(u[0], t1);
    u[1] = cadd(m1, im2);
    u[2] = csub(m1, im2);
}
template <int SIGN, typename T, typename C> __device__ __forceinline__ void dft5(C* u) {
    const T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410);   // cos(2 pi / 5), cos(4 pi / 5)
    const T s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);    // sin(2 pi / 5), sin(4 pi / 5)
    const C a1 = cadd(u[1], u[4]), b1 = csub(u[1], u[4]);
    const C a2 = cadd(u[2], u[3]), b2 = csub(u[2], u[3]);
    C r1, r2, q1, q2;
    r1.x = u[0].x + c1 * a1.x + c2 * a2.x; r1.y = u[0].y + c1 * a1.y + c2 * a2.y;
    r2.x = u[0].x + c2 * a1.x + c1 * a2.x; r2.y = u[0].y + c2 * a1.y + c1 * a2.y;
    q1.x = s1 * b1.x + s2 * b2.x; q1.y = s1 * b1.y + s2 * b2.y;
    q2.x = s2 * b1.x - s1 * b2.x; q2.y = s2 * b1.y - s1 * b2.y;
    const C iq1 = mul_i<SIGN>(q1), iq2 = mul_i<SIGN>(q2);
    C u0; u0.x = u[0].x + a1.x + a2.x; u0.y = u[0].y + a1.y + a2.y;
    u[0] = u0;
    u[1] = cadd(r1, iq1);
    u[4] = csub(r1, iq1);
    u[2] = cadd(r2, iq2);
    u[3] = csub(r2, iq2);
}

__device__ __forceinline__ int lpad(int e) { return e + (e >> 4); }   // one pad element per 16: spreads banks

struct FftLineArgs {
    const void* in;
    void* out;
    int64_t a_total;          // number of valid contiguous indices a
    int64_t a_out;            // backward: output columns (>= a_total; the rest is written as zeros)
    int64_t in_stride_j, in_stride_c;     // elements; j = transform index on the full side, or k' on the pruned side
    int64_t out_stride_j, out_stride_c;
    int nc;                   // number of outer indices c
    int nk;                   // kept modes
    const int32_t* map;       // kept index k' -> FFT index (non_oversampled_indices!)
    const void* fa;           // T[ka]: factor by (a mod ka)
    int ka;
    const void* fk;           // T[nk]: factor by k'
    const void* twiddle;      // complex<T>[N]: exp(SIGN 2πi m / N)
    double scale;             // extra scalar factor (normfactor)
    const void* mult;         // optional T[]: real multiplier indexed like the pruned side (uniform callback), or null
    // rows: a = r * row_a + k1 enumerates rows r of row_a columns of which k1 < row_valid exist; the row strides of the two
    // sides differ (intermediates pad their rows to 128 bytes, the caller's array does not).  row_a = 0: no row structure.
    int row_a, row_valid, row_in, row_out;
};

// One radix-R Stockham stage of a line held in LDS (in place, wave-synchronous).
// TWS: the twiddle table holds the roots of unity of order N * TWS (TWS = 2 for the half-length complex FFT
// inside a real transform, whose table is shared with the real/complex split step).
template <typename T, int N, int R, int P, int SIGN, int TWS = 1>
__device__ __forceinline__ void stage(typename Cplx2<T>::type* line, const typename Cplx2<T>::type* tw, int lane) {
    using C = typename Cplx2<T>::type;
    constexpr int p = P;                              // product of the radices of the earlier stages
    constexpr int NB = N / R;                         // butterflies per line
    constexpr int PER = (NB + kWave - 1) / kWave;     // butterflies per lane
    C u[PER][R];
    int jout[PER];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
        const int i = lane + b * kWave;
        const int k = i % p;
        jout[b] = (i - k) * R + k;
        if (i < NB) {
#pragma unroll
            for (int t = 0; t < R; ++t) u[b][t] = line[lpad(i + t * NB)];
            if (p > 1) {
                const int step = k * (N / (p * R));   // w_{pR}^{k t} = w_N^{k t N / (p R)}; k t N / (p R) < N
#pragma unroll
                for (int t = 1; t < R; ++t) u[b][t] = cmul(u[b][t], tw[(step * t) * TWS]);
            }
            if constexpr (R == 8) dft8<SIGN, T>(u[b]);
            else if constexpr (R == 5) dft5<SIGN, T>(u[b]);
            else if constexpr (R == 4) dft4<SIGN>(u[b]);
            else if constexpr (R == 3) dft3<SIGN, T>(u[b]);
            else dft2<SIGN>(u[b]);
        }
    }
    wave_lds_fence();      // every read of this stage is issued before the first write (same wave, in order)
#pragma unroll
    for (int b = 0; b < PER; ++b) {
        const int i = lane + b * kWave;
        if (i < NB) {
#pragma unroll
            for (int t = 0; t < R; ++t) line[lpad(jout[b] + t * p)] = u[b][t];
        }
    }
    wave_lds_fence();
}

// Stockham stages for N = 2^a 3^b 5^c: radix 8 while possible, then 4 / 2, then 3s and 5s.
template <typename T, int N, int REM, int P, int SIGN, int TWS>
__device__ __forceinline__ void fft_stages(typename Cplx2<T>::type* line, const typename Cplx2<T>::type* tw, int lane) {
    if constexpr (REM > 1) {
        constexpr int R = REM % 8 == 0 ? 8 : (REM % 4 == 0 ? 4 : (REM % 2 == 0 ? 2 : (REM % 3 == 0 ? 3 : 5)));
        static_assert(REM % R == 0, "length must be a product of 2, 3 and 5");
        stage<T, N, R, P, SIGN, TWS>(line, tw, lane);
        fft_stages<T, N, REM / R, P * R, SIGN, TWS>(line, tw, lane);
    }
}

template <typename T, int N, int SIGN, int TWS = 1>
__device__ __forceinline__ void fft_line(typename Cplx2<T>::type* line, const typename Cplx2<T>::type* tw, int lane) {
    fft_stages<T, N, N, 1, SIGN, TWS>(line, tw, lane);
}

#ifndef NUFFT_FFT_PRIO
#define NUFFT_FFT_PRIO 3        // BACKWARD strided passes: the waves that load or store go ahead of those that transform (s_setprio; 0: none).  Round 6, scripts/r6_ah.sh:
                                // deconvolve + pad + dimensions 3, 2 of type 2 at C2 0.38 -> 0.32 ms, ComplexF64 0.71 -> 0.56, C4 1.12 -> 0.98; the forward passes lose 1 % with it
#endif
#if NUFFT_FFT_PRIO
#define NUFFT_FFT_PRIO_MEM() __builtin_amdgcn_s_setprio(NUFFT_FFT_PRIO)
#define NUFFT_FFT_PRIO_ALU() __builtin_amdgcn_s_setprio(0)
#else
#define NUFFT_FFT_PRIO_MEM() do { } while (0)
#define NUFFT_FFT_PRIO_ALU() do { } while (0)
#endif
#ifndef NUFFT_FFT_PRIO_CLINES
#define NUFFT_FFT_PRIO_CLINES 3 // ... and of ComplexF32 plans, backward (cplx_lines_kernel): C3 7.0 -> 6.5 ms for the backward FFT stage (the last pass 3.33 -> 2.90); the forward pass
                                // and ComplexF64 lose 1 % with it and stay without (scripts/r6_ak.sh)
#endif
#if NUFFT_FFT_PRIO_CLINES
#define NUFFT_FFT_PRIOC_MEM() __builtin_amdgcn_s_setprio(NUFFT_FFT_PRIO_CLINES)
#define NUFFT_FFT_PRIOC_ALU() __builtin_amdgcn_s_setprio(0)
#else
#define NUFFT_FFT_PRIOC_MEM() do { } while (0)
#define NUFFT_FFT_PRIOC_ALU() do { } while (0)
#endif
#ifndef NUFFT_FFT_PRIO_LINES
#define NUFFT_FFT_PRIO_LINES 3  // the same in the contiguous-line kernel of dimension 1 of real plans, both directions: C2 r2c pass + halo 0.690 -> 0.678 ms, c2r 0.355 -> 0.342
#endif
#if NUFFT_FFT_PRIO_LINES
#define NUFFT_FFT_PRIOL_MEM() __builtin_amdgcn_s_setprio(NUFFT_FFT_PRIO_LINES)
#define NUFFT_FFT_PRIOL_ALU() __builtin_amdgcn_s_setprio(0)
#else
#define NUFFT_FFT_PRIOL_MEM() do { } while (0)
#define NUFFT_FFT_PRIOL_ALU() do { } while (0)
#endif
// FWD: full input (N along j), pruned output (nk along k').  BWD: pruned input, full output.
// MULT: a real multiplier array (uniform callback menu) is applied on the pruned side.
template <typename T, int N, bool FWD, int TA, bool MULT>
__global__ __launch_bounds__(TA * kWave) void fft_lines_kernel(FftLineArgs a) {
    using C = typename Cplx2<T>::type;
    constexpr int LINE = N + (N >> 4) + 1;            // padded line length (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C* tw = reinterpret_cast<C*>(smem);               // [N]
    C* lines = tw + N;                                // [TA][LINE]

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    constexpr int NT = TA * kWave;
    const int64_t a0 = (int64_t)blockIdx.x * TA;
    const int c = blockIdx.y;
    const C* in = static_cast<const C*>(a.in) + (int64_t)c * a.in_stride_c;
    C* out = static_cast<C*>(a.out) + (int64_t)c * a.out_stride_c;
    const T* fa = static_cast<const T*>(a.fa);
    const T* fk = static_cast<const T*>(a.fk);
    const T scale = (T)a.scale;
    const T* mult = static_cast<const T*>(a.mult);
    // this thread's column (NT is a multiple of TA: e % TA is the same in every loop below)
    const int64_t acol = a0 + tid % TA;
    bool cvalid = acol < a.a_total;
    int64_t in_col = acol, out_col = acol;
    int fidx = 0;
    if (a.row_a > 0) {
        const int64_t r = acol / a.row_a;
        const int k1 = (int)(acol - r * a.row_a);
        cvalid = cvalid && k1 < a.row_valid;
        in_col = r * a.row_in + k1;
        out_col = r * a.row_out + k1;
        fidx = k1;
    } else if (a.ka > 1) {
        fidx = (int)(acol % a.ka);
    }

    if (!FWD && a0 >= a.a_total) {
        // backward: columns beyond the kept ones are zeros of the oversampled spectrum
        C z; z.x = T(0); z.y = T(0);
        for (int e = tid; e < TA * N; e += NT) {
            const int ai = e % TA, j = e / TA;
            if (a0 + ai < a.a_out) out[a0 + ai + (int64_t)j * a.out_stride_j] = z;
        }
        return;
    }

    const C* twg = static_cast<const C*>(a.twiddle);
    for (int i = tid; i < N; i += NT) tw[i] = twg[i];
    if constexpr (!FWD) NUFFT_FFT_PRIO_MEM();
#if defined(NUFFT_FFT_PRIO_FWD_LOAD)
    if constexpr (FWD) __builtin_amdgcn_s_setprio(NUFFT_FFT_PRIO_FWD_LOAD);
#endif

    if (FWD) {
        // ceil(N / 64) loads per thread (rows tid / TA + 64 it of column tid % TA), issued eight at a time before the first is waited for: as a
        // rolled loop — one 8- or 16-byte load in flight per thread, 8-16 KB per CU — the strided passes moved 2.0 TB/s (Float32, one workgroup per
        // CU) to 3.3 TB/s (Float64).  (Also tried: the deconvolution factors and FFT indices of the output loop fetched into registers before the
        // transform — 32 registers across the FFT at N = 1024, spills: C3's last pass 1.45 -> 1.54 ms; not kept.)
        constexpr int kIters = (N + kWave - 1) / kWave, kLoadBatch = kIters >= 8 ? 8 : kIters;
        const int ai = tid % TA, j0 = tid / TA;
        const C* src = in + in_col;
        for (int it0 = 0; it0 < kIters; it0 += kLoadBatch) {
            C v[kLoadBatch];
#pragma unroll
            for (int b = 0; b < kLoadBatch; ++b) {
                const int j = j0 + (it0 + b) * kWave;
                v[b].x = T(0); v[b].y = T(0);
                if (cvalid && j < N) v[b] = src[(int64_t)j * a.in_stride_j];
            }
#pragma unroll
            for (int b = 0; b < kLoadBatch; ++b) {
                const int j = j0 + (it0 + b) * kWave;
                if (j < N) lines[ai * LINE + lpad(j)] = v[b];
            }
        }
    } else {
        C z; z.x = T(0); z.y = T(0);
        for (int e = tid; e < TA * LINE; e += NT) lines[e] = z;
        __syncthreads();
        // (kept modes: a run-time count; four loads in flight per thread)
        if (cvalid) {
            const int ai = tid % TA;
            for (int k0 = tid / TA; k0 < a.nk; k0 += 4 * kWave) {
                C v[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = k0 + b * kWave;
                    v[b].x = T(0); v[b].y = T(0);
                    if (k < a.nk) v[b] = in[in_col + (int64_t)k * a.in_stride_j];
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = k0 + b * kWave;
                    if (k < a.nk) {
                        T f = fa[fidx] * fk[k] * scale;
                        if constexpr (MULT) f *= mult[in_col + (int64_t)k * a.in_stride_j];
                        v[b].x *= f; v[b].y *= f;
                        lines[ai * LINE + lpad(a.map[k])] = v[b];
                    }
                }
            }
        }
    }
    __syncthreads();

#if defined(NUFFT_FFT_PRIO_FWD_LOAD)
    if constexpr (FWD) __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (!FWD) NUFFT_FFT_PRIO_ALU();
    fft_line<T, N, FWD ? -1 : 1>(lines + wave * LINE, tw, lane);
    __syncthreads();
    if constexpr (!FWD) NUFFT_FFT_PRIO_MEM();
#if defined(NUFFT_FFT_PRIO_FWD_STORE)       // (experiment: forward passes, the stores only)
    if constexpr (FWD) __builtin_amdgcn_s_setprio(NUFFT_FFT_PRIO_FWD_STORE);
#endif
#if defined(NUFFT_FFT_PRIO_FWD_LOAD)        // (experiment: forward passes, the loads only)
    if constexpr (FWD) __builtin_amdgcn_s_setprio(0);
#endif

    if (FWD) {
        for (int e = tid; e < TA * a.nk; e += NT) {
            const int ai = e % TA, k = e / TA;
            if (cvalid) {
                C v = lines[ai * LINE + lpad(a.map[k])];
                const int64_t off = out_col + (int64_t)k * a.out_stride_j;
                T f = fa[fidx] * fk[k] * scale;
                if constexpr (MULT) f *= mult[off];
                v.x *= f; v.y *= f;
                out[off] = v;
            }
        }
    } else {
        for (int e = tid; e < TA * N; e += NT) {
            const int ai = e % TA, j = e / TA;
            if (a.row_a > 0) {
                if (cvalid) out[out_col + (int64_t)j * a.out_stride_j] = lines[ai * LINE + lpad(j)];
            } else if (a0 + ai < a.a_out) {
                C v = lines[ai * LINE + lpad(j)];
                if (a0 + ai >= a.a_total) { v.x = T(0); v.y = T(0); }
                out[a0 + ai + (int64_t)j * a.out_stride_j] = v;
            }
        }
    }
}

// Contiguous line <-> LDS with 16-byte accesses per lane: Float32 lines move two complex elements per lane and step (8-byte accesses
// reach about half the rate: C3's dimension-1 pass, 12.9 GB, took 4.9 ms).  n elements, n even, the line 16-byte aligned.
template <typename C>
__device__ __forceinline__ void load_line_wide(C* line, const C* src, int n, int lane) {
    if constexpr (sizeof(C) == 8) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int k = lane; k < n / 2; k += kWave) {
            const float4 v = s4[k];
            C a, b;
            a.x = v.x; a.y = v.y; b.x = v.z; b.y = v.w;
            line[lpad(2 * k)] = a;
            line[lpad(2 * k + 1)] = b;
        }
    } else {
        for (int k = lane; k < n; k += kWave) line[lpad(k)] = src[k];
    }
}
template <typename C>
__device__ __forceinline__ void store_line_wide(C* dst, const C* line, int n, int lane) {
    if constexpr (sizeof(C) == 8) {
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int k = lane; k < n / 2; k += kWave) {
            const C a = line[lpad(2 * k)], b = line[lpad(2 * k + 1)];
            d4[k] = make_float4(a.x, a.y, b.x, b.y);
        }
    } else {
        for (int k = lane; k < n; k += kWave) dst[k] = line[lpad(k)];
    }
}

// ---------------------------------------------------------------------------------------------------
// Dimension 1 of real plans: r2c / c2r of contiguous lines with a *compact* spectrum (only the K1 = N1/2 + 1
// kept modes are stored: row length K1 instead of Ñ1/2 + 1).  A real line of N = 2M samples is transformed as
// one M-point complex FFT of z[n] = x[2n] + i x[2n+1] plus the usual even/odd split.
// ---------------------------------------------------------------------------------------------------
struct RealLineArgs {
    const void* in;
    void* out;
    int64_t nlines;
    int k1;                 // kept modes per line
    int row;                // row stride of the compact spectrum in complex elements (>= k1: rows padded to 128 bytes)
    const void* twiddle;    // complex<T>[N]: exp(SIGN 2πi m / N), N = 2M
    // forward pass behind the halo variant of the spreading ring (smarch_kernels.h): the stencil reach of every column sits in
    // a side buffer and is added to the line while it is loaded (halo != null; the flag says whether the ring served the point set)
    const void* halo;       // T[planes][nty][ntx][record] of this component
    const uint32_t* hflag;
    int ny;                 // lines per plane of the grid (line = z * ny + y)
    HaloLayout hl;
};

// line += the stencil reach that the neighbouring columns left in the side buffer (real data: nc = 1; pairs of reals = the
// complex elements of the line).  Strips of the x reach: one aligned pair per lane and step, distinct cells for distinct columns
// (columns are wider than the reach).  Rows of the y reach (lines within the reach of a column boundary only): the rows of
// neighbouring columns overlap, so even and odd columns take turns (and the last column of an odd count goes alone).
// NC = 1: real lines, a pair = two cells = one complex element of the packed line (n = cells per line); NC = 2: complex lines, a pair = one cell.
// PLANAR (NC = 1 layout, complex lines): complex data spread part by part by the real kernel (smarch_kernels.h, MarchGeom::parts = 2) —
// `halo` holds the real parts' side buffer, `halo2` the imaginary parts'; a pair of each = that part of two neighbouring cells of the
// line, both added in one pass (two whole complex elements per step).
template <typename T, typename C, int NC, bool PLANAR = false>
__device__ __forceinline__ void add_halo_to_line(const void* halo, const HaloLayout& h, int ny, C* line, int64_t line_id, int lane, int n, const void* halo2 = nullptr) {
    typedef T T2 __attribute__((ext_vector_type(2)));
    const int y = (int)(line_id % ny), z = (int)(line_id / ny);
    const int ty = y / h.n2, ly = y - ty * h.n2;
    const T* hz = static_cast<const T*>(halo) + (int64_t)z * h.plane;
    const int64_t d2 = PLANAR ? static_cast<const T*>(halo2) - static_cast<const T*>(halo) : 0;      // from a real-part pair to the imaginary-part pair
    auto add_pair = [&](int x0, const T* src) __attribute__((always_inline)) {       // x0: first cell of the pair
        const T2 v = *reinterpret_cast<const T2*>(src);
        if (x0 < 0) x0 += n;
        if (x0 >= n) x0 -= n;
        if constexpr (PLANAR) {                         // (pairs are aligned and n is even: x0 + 1 < n)
            const T2 w = *reinterpret_cast<const T2*>(src + d2);
            C c0 = line[lpad(x0)], c1 = line[lpad(x0 + 1)];
            c0.x += v.x; c0.y += w.x;
            c1.x += v.y; c1.y += w.y;
            line[lpad(x0)] = c0;
            line[lpad(x0 + 1)] = c1;
            return;
        }
        const int e = NC == 1 ? x0 >> 1 : x0;
        C c = line[lpad(e)];
        c.x += v.x;
        c.y += v.y;
        line[lpad(e)] = c;
    };
    constexpr int CPP = PLANAR ? 2 : 2 / NC;            // cells per pair
    {
        const int sp = h.sw / 2;
        const T* row = hz + (int64_t)ty * h.ntx * h.rec + ly * h.sw;
        for (int idx = lane; idx < h.ntx * sp; idx += kWave) {
            const int tx = idx / sp, pi = idx - tx * sp, i = CPP * pi;       // i: cell index within the strip
            add_pair(i < h.xlo ? tx * h.n1 - h.xlo + i : tx * h.n1 + h.n1 + (i - h.xlo), row + (int64_t)tx * h.rec + 2 * pi);
        }
        wave_lds_fence();
    }
    int sy = -1, rr = 0;
    if (ly < h.yhi) { sy = ty == 0 ? h.nty - 1 : ty - 1; rr = h.ylo + ly; }
    else if (ly >= h.n2 - h.ylo) { sy = ty + 1 == h.nty ? 0 : ty + 1; rr = ly - (h.n2 - h.ylo); }
    if (sy >= 0) {
        const int rp = h.rw / 2;
        const T* row = hz + (int64_t)sy * h.ntx * h.rec + h.n2 * h.sw + rr * h.rw;
        const int neven = h.ntx & ~1;
        for (int par = 0; par < 3; ++par) {
            const int ncol = par == 2 ? (h.ntx & 1) : neven / 2;
            for (int idx = lane; idx < ncol * rp; idx += kWave) {
                const int k = idx / rp, pi = idx - k * rp;
                const int tx = par == 2 ? h.ntx - 1 : 2 * k + par;
                add_pair(tx * h.n1 - h.xlo + CPP * pi, row + (int64_t)tx * h.rec + 2 * pi);
            }
            wave_lds_fence();
        }
    }
}

template <typename T, int M, bool FWD, int TL, bool HALO = false>
__global__ __launch_bounds__(TL * kWave) void real_lines_kernel(RealLineArgs a) {
    using C = typename Cplx2<T>::type;
    constexpr int N = 2 * M;
    constexpr int LINE = M + (M >> 4) + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C* tw = reinterpret_cast<C*>(smem);                       // [N]
    C* lines = tw + N;                                        // [TL][LINE]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const C* twg = static_cast<const C*>(a.twiddle);
    for (int i = tid; i < N; i += TL * kWave) tw[i] = twg[i];
    __syncthreads();
    const int64_t line_id = (int64_t)blockIdx.x * TL + wave;
    if (line_id >= a.nlines) return;
    C* line = lines + wave * LINE;
    if (FWD) {
        const C* zin = reinterpret_cast<const C*>(static_cast<const T*>(a.in) + line_id * N);
        NUFFT_FFT_PRIOL_MEM();
        load_line_wide(line, zin, M, lane);
        wave_lds_fence();
        if constexpr (HALO) {
            if (*a.hflag != 0u) add_halo_to_line<T, C, 1>(a.halo, a.hl, a.ny, line, line_id, lane, N);
        }
        NUFFT_FFT_PRIOL_ALU();
        fft_line<T, M, -1, 2>(line, tw, lane);
        NUFFT_FFT_PRIOL_MEM();
        C* xout = static_cast<C*>(a.out) + line_id * a.row;
        for (int k = lane; k < a.k1; k += kWave) {
            const C zk = line[lpad(k == M ? 0 : k)];
            C zm = line[lpad(k == 0 ? 0 : M - k)];
            zm.y = -zm.y;                                      // conj(Z[M - k])
            const C e = cadd(zk, zm), d = csub(zk, zm);
            // X[k] = (e - i w^k d) / 2,  w = exp(-2πi/N)
            const C wd = cmul(tw[k], d);
            C x;
            x.x = T(0.5) * (e.x + wd.y);
            x.y = T(0.5) * (e.y - wd.x);
            xout[k] = x;
        }
    } else {
        // Z[k] = E'[k] + i O'[k] with E' = X[k] + conj(X[M-k]), O' = w^k (X[k] - conj(X[M-k])), w = exp(+2πi/N).
        // The partner of k is M - k: Z[M-k] = conj(E'[k]) + i conj(O'[k]), so one lane builds both from two
        // global loads and no staging copy of X is needed in LDS (X is zero beyond the kept modes).
        const C* xin = static_cast<const C*>(a.in) + line_id * a.row;
        NUFFT_FFT_PRIOL_MEM();
        for (int k = lane; k <= M / 2; k += kWave) {
            C xk, xm;
            xk.x = xk.y = xm.x = xm.y = T(0);
            if (k < a.k1) xk = xin[k];
            if (M - k < a.k1) xm = xin[M - k];
            // a c2r transform ignores the imaginary parts of the self-conjugate modes k = 0 and k = N/2
            // (FFTW / rocFFT / numpy.irfft semantics: the result is the real part of the Hermitian sum)
            if (k == 0) { xk.y = T(0); xm.y = T(0); }
            xm.y = -xm.y;                                      // conj(X[M - k])
            const C e = cadd(xk, xm);
            const C o = cmul(tw[k], csub(xk, xm));
            C z;
            z.x = e.x - o.y;
            z.y = e.y + o.x;
            line[lpad(k)] = z;
            if (k > 0 && k < M / 2) {
                C zp;                                          // conj(e) + i conj(o)
                zp.x = e.x + o.y;
                zp.y = o.x - e.y;
                line[lpad(M - k)] = zp;
            }
        }
        wave_lds_fence();
        NUFFT_FFT_PRIOL_ALU();
        fft_line<T, M, 1, 2>(line, tw, lane);
        NUFFT_FFT_PRIOL_MEM();
        C* zout = reinterpret_cast<C*>(static_cast<T*>(a.out) + line_id * N);
        store_line_wide(zout, line, M, lane);
    }
}

// Dimension 1 of complex plans: c2c of contiguous lines with a compact spectrum (the K1 = N1 kept modes of each
// line, in the caller's mode order: map[k'] is the FFT index of kept mode k').
struct CplxLineArgs {
    const void* in;
    void* out;
    int64_t nlines;
    int k1;                 // kept modes per line
    const int32_t* map;     // [k1]
    const void* twiddle;    // complex<T>[N]
    // forward pass behind the halo variant of the spreading ring (see RealLineArgs)
    const void* halo;
    const void* halo2;      // complex data spread by the real kernel: `halo` holds the real parts' side buffer, `halo2` the imaginary parts' (hl.nc = 1)
    const uint32_t* hflag;
    int ny;
    HaloLayout hl;
};

template <typename T, int N, bool FWD, int TL, bool HALO = false>
__global__ __launch_bounds__(TL * kWave) void cplx_lines_kernel(CplxLineArgs a) {
    using C = typename Cplx2<T>::type;
    constexpr int LINE = N + (N >> 4) + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    C* tw = reinterpret_cast<C*>(smem);                       // [N]
    C* lines = tw + N;                                        // [TL][LINE]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const C* twg = static_cast<const C*>(a.twiddle);
    for (int i = tid; i < N; i += TL * kWave) tw[i] = twg[i];
    __syncthreads();
    const int64_t line_id = (int64_t)blockIdx.x * TL + wave;
    if (line_id >= a.nlines) return;
    C* line = lines + wave * LINE;
    if (FWD) {
        const C* zin = static_cast<const C*>(a.in) + line_id * N;
        load_line_wide(line, zin, N, lane);
        wave_lds_fence();
        if constexpr (HALO) {
            if (*a.hflag != 0u) {
                if (a.halo2) {
                    add_halo_to_line<T, C, 1, true>(a.halo, a.hl, a.ny, line, line_id, lane, N, a.halo2);
                } else {
                    add_halo_to_line<T, C, 2>(a.halo, a.hl, a.ny, line, line_id, lane, N);
                }
            }
        }
        fft_line<T, N, -1>(line, tw, lane);
        C* xout = static_cast<C*>(a.out) + line_id * a.k1;
        if (sizeof(C) == 8 && (a.k1 & 1) == 0) {        // Float32: two kept modes (16 bytes) per lane and step
            float4* x4 = reinterpret_cast<float4*>(xout);
            for (int k = lane; k < a.k1 / 2; k += kWave) {
                const C u = line[lpad(a.map[2 * k])], v = line[lpad(a.map[2 * k + 1])];
                x4[k] = make_float4((float)u.x, (float)u.y, (float)v.x, (float)v.y);
            }
        } else {
            for (int k = lane; k < a.k1; k += kWave) xout[k] = line[lpad(a.map[k])];
        }
    } else {
        if constexpr (sizeof(T) == 4) NUFFT_FFT_PRIOC_MEM();
        C z; z.x = T(0); z.y = T(0);
        for (int n = lane; n < N; n += kWave) line[lpad(n)] = z;
        wave_lds_fence();
        const C* xin = static_cast<const C*>(a.in) + line_id * a.k1;
        if (sizeof(C) == 8 && (a.k1 & 1) == 0) {
            const float4* x4 = reinterpret_cast<const float4*>(xin);
            for (int k = lane; k < a.k1 / 2; k += kWave) {
                const float4 w = x4[k];
                C u, v;
                u.x = w.x; u.y = w.y; v.x = w.z; v.y = w.w;
                line[lpad(a.map[2 * k])] = u;
                line[lpad(a.map[2 * k + 1])] = v;
            }
        } else {
            for (int k = lane; k < a.k1; k += kWave) line[lpad(a.map[k])] = xin[k];
        }
        wave_lds_fence();
        if constexpr (sizeof(T) == 4) NUFFT_FFT_PRIOC_ALU();
        fft_line<T, N, 1>(line, tw, lane);
        if constexpr (sizeof(T) == 4) NUFFT_FFT_PRIOC_MEM();
        C* zout = static_cast<C*>(a.out) + line_id * N;
        store_line_wide(zout, line, N, lane);
    }
}

// Line lengths instantiated: powers of two and 1.5 x / 1.25 x powers of two (sigma = 2, 1.5, 1.25 on power-of-two
// grids).  Other products of 2, 3 and 5 use the general rocFFT path.
#define NUFFT_FFT_SIZES(X) X(64) X(80) X(96) X(128) X(160) X(192) X(256) X(320) X(384) X(512) X(640) X(768) X(1024)

constexpr size_t kFftLdsLimit = 160 * 1024;      // gfx950: LDS per workgroup

// lines per workgroup of the dimension-1 real passes: 16 below 150 KiB, else as many as fit (Float64 lines of
// 2 x 1024 need 4: 8 would take 172 KB)
template <typename T, int M>
constexpr int real_lines_per_group() {
    constexpr size_t c = 2 * sizeof(T);
    constexpr int LINE = M + (M >> 4) + 1;
    if (c * (16 * LINE + 2 * M) <= 150 * 1024) return 16;
    if (c * (8 * LINE + 2 * M) <= kFftLdsLimit) return 8;
    if (c * (4 * LINE + 2 * M) <= kFftLdsLimit) return 4;
    return 0;
}

template <typename T, int M, bool FWD>
static hipError_t launch_real_m(const RealLineArgs& a, hipStream_t stream) {
    using C = typename Cplx2<T>::type;
    constexpr int LINE = M + (M >> 4) + 1;
    constexpr int TL = real_lines_per_group<T, M>();
    static_assert(TL >= 4 && sizeof(C) * (size_t)(TL * LINE + 2 * M) <= kFftLdsLimit, "line buffers exceed the 160 KiB of LDS");
    const size_t lds = sizeof(C) * (size_t)(TL * LINE + 2 * M);
    // the attribute is per device: remember which devices of this process have it (plans may live on several)
    static std::atomic<unsigned long long> prepared{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    auto fn = real_lines_kernel<T, M, FWD, TL, false>;
    auto fnh = real_lines_kernel<T, M, FWD, TL, FWD>;      // forward: the variant that adds the spreading ring's side buffer
    if (!(prepared.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && FWD) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fnh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        prepared.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((FWD && a.halo) ? fnh : fn, dim3((unsigned)((a.nlines + TL - 1) / TL)), dim3(TL * kWave), lds, stream, a);
    return hipGetLastError();
}

template <typename T, bool FWD>
static hipError_t launch_real_t(int m, const RealLineArgs& a, hipStream_t stream) {
    switch (m) {
#define NUFFT_CASE(NN) case NN: return launch_real_m<T, NN, FWD>(a, stream);
        NUFFT_FFT_SIZES(NUFFT_CASE)
#undef NUFFT_CASE
        default: return hipErrorInvalidValue;
    }
}

static bool size_instantiated(int64_t n) {
    switch (n) {
#define NUFFT_CASE(NN) case NN:
        NUFFT_FFT_SIZES(NUFFT_CASE)
#undef NUFFT_CASE
            return true;
        default: return false;
    }
}

// real line length n = 2 M
bool real_lines_supported(int dtype, int64_t n) {
    (void)dtype;
    return n % 2 == 0 && size_instantiated(n / 2);
}

hipError_t launch_real_lines(int dtype, int64_t n, bool forward, const void* in, void* out, int64_t nlines, int k1, int row,
                             const void* twiddle, hipStream_t stream, const RealLineHalo* halo) {
    RealLineArgs a{};
    a.in = in; a.out = out; a.nlines = nlines; a.k1 = k1; a.row = row; a.twiddle = twiddle;
    if (halo && halo->buffer && forward) {
        if (halo->layout.nc != 1 || halo->ny <= 0) return hipErrorInvalidValue;
        a.halo = halo->buffer; a.hflag = halo->flag; a.ny = halo->ny; a.hl = halo->layout;
    }
    const int m = (int)(n / 2);
    if (dtype == NUFFT_F32) return forward ? launch_real_t<float, true>(m, a, stream) : launch_real_t<float, false>(m, a, stream);
    return forward ? launch_real_t<double, true>(m, a, stream) : launch_real_t<double, false>(m, a, stream);
}

template <typename T, int N, bool FWD>
static hipError_t launch_cplx_n(const CplxLineArgs& a, hipStream_t stream) {
    using C = typename Cplx2<T>::type;
    constexpr int LINE = N + (N >> 4) + 1;
    constexpr int TL = (sizeof(C) * (16 * LINE + N) <= 80 * 1024) ? 16 : 8;
    static_assert(sizeof(C) * (size_t)(TL * LINE + N) <= kFftLdsLimit, "line buffers exceed the 160 KiB of LDS");
    const size_t lds = sizeof(C) * (size_t)(TL * LINE + N);
    auto fn = cplx_lines_kernel<T, N, FWD, TL, false>;
    auto fnh = cplx_lines_kernel<T, N, FWD, TL, FWD>;
    static std::atomic<unsigned long long> prepared{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(prepared.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && FWD) e = hipFuncSetAttribute(reinterpret_cast<const void*>(fnh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        prepared.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((FWD && a.halo) ? fnh : fn, dim3((unsigned)((a.nlines + TL - 1) / TL)), dim3(TL * kWave), lds, stream, a);
    return hipGetLastError();
}

template <typename T, bool FWD>
static hipError_t launch_cplx_t(int n, const CplxLineArgs& a, hipStream_t stream) {
    switch (n) {
#define NUFFT_CASE(NN) case NN: return launch_cplx_n<T, NN, FWD>(a, stream);
        NUFFT_FFT_SIZES(NUFFT_CASE)
#undef NUFFT_CASE
        default: return hipErrorInvalidValue;
    }
}

// grid += side buffer of the spreading window's halo variant, line by line through LDS (the consumer for the stage-level
// nufft_spread and for plans whose first FFT pass is not real_lines_kernel / cplx_lines_kernel): one wave per line of the grid,
// the same add_halo_to_line as the fused passes.  ne: complex-sized elements per line (real data: pairs of cells).
template <typename T, int NC>
__global__ __launch_bounds__(1024) void halo_add_lines_kernel(T* grid, const T* halo, int64_t nlines, int ny, int ne, HaloLayout h, const uint32_t* flag,
                                                              int64_t grid_comp, int64_t halo_comp) {
    using C = typename Cplx2<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (*flag == 0u) return;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
    const int LINE = ne + (ne >> 4) + 1;
    C* line = reinterpret_cast<C*>(smem) + (size_t)wave * LINE;
    const int64_t line_id = (int64_t)blockIdx.x * nw + wave;
    if (line_id >= nlines) return;
    C* g = reinterpret_cast<C*>(grid + (int64_t)blockIdx.y * grid_comp) + line_id * ne;
    load_line_wide(line, g, ne, lane);
    wave_lds_fence();
    if constexpr (NC == 3) {        // complex grid, planar side buffers of its two parts (2 blockIdx.y, 2 blockIdx.y + 1)
        add_halo_to_line<T, C, 1, true>(halo + (int64_t)(2 * blockIdx.y) * halo_comp, h, ny, line, line_id, lane, ne, halo + (int64_t)(2 * blockIdx.y + 1) * halo_comp);
    } else {
        add_halo_to_line<T, C, NC>(halo + (int64_t)blockIdx.y * halo_comp, h, ny, line, line_id, lane, NC == 1 ? 2 * ne : ne);
    }
    store_line_wide(g, line, ne, lane);
}

template <typename T>
static hipError_t launch_halo_add_lines_t(void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, int n1cells, int ny, int nz,
                                          int C, const HaloLayout& h, const uint32_t* flag, hipStream_t stream, bool planar) {
    using Cx = typename Cplx2<T>::type;
    const int ne = (h.nc == 1 && !planar) ? n1cells / 2 : n1cells;
    const int LINE = ne + (ne >> 4) + 1;
    int nw = 16;
    while (nw > 1 && (size_t)nw * LINE * sizeof(Cx) > 72 * 1024) nw >>= 1;
    const size_t lds = (size_t)nw * LINE * sizeof(Cx);
    if (lds > kFftLdsLimit || (h.nc == 1 && (n1cells & 1)) || (planar && h.nc != 1)) return hipErrorInvalidValue;
    const int64_t nlines = (int64_t)ny * nz;
    auto fn = planar ? halo_add_lines_kernel<T, 3> : (h.nc == 1 ? halo_add_lines_kernel<T, 1> : halo_add_lines_kernel<T, 2>);
    static std::atomic<unsigned long long> prepared{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(prepared.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(halo_add_lines_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFftLdsLimit);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(halo_add_lines_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFftLdsLimit);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(halo_add_lines_kernel<T, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFftLdsLimit);
        if (e != hipSuccess) return e;
        prepared.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)((nlines + nw - 1) / nw), (unsigned)C, 1), dim3(nw * kWave), lds, stream, static_cast<T*>(grid),
                       static_cast<const T*>(halo), nlines, ny, ne, h, flag, grid_comp_reals, halo_comp_reals);
    return hipGetLastError();
}

hipError_t launch_halo_add_lines(int dtype, void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, int n1cells, int ny, int nz,
                                 int C, const HaloLayout& h, const uint32_t* flag, hipStream_t stream, bool planar) {
    return dtype == NUFFT_F32 ? launch_halo_add_lines_t<float>(grid, halo, grid_comp_reals, halo_comp_reals, n1cells, ny, nz, C, h, flag, stream, planar)
                              : launch_halo_add_lines_t<double>(grid, halo, grid_comp_reals, halo_comp_reals, n1cells, ny, nz, C, h, flag, stream, planar);
}

hipError_t launch_cplx_lines(int dtype, int64_t n, bool forward, const void* in, void* out, int64_t nlines, int k1,
                             const int32_t* map, const void* twiddle, hipStream_t stream, const RealLineHalo* halo) {
    CplxLineArgs a{};
    a.in = in; a.out = out; a.nlines = nlines; a.k1 = k1; a.map = map; a.twiddle = twiddle;
    if (halo && halo->buffer && forward) {
        if (halo->layout.nc != (halo->buffer2 ? 1 : 2) || halo->ny <= 0 || (halo->buffer2 && (n & 1))) return hipErrorInvalidValue;
        a.halo = halo->buffer; a.halo2 = halo->buffer2; a.hflag = halo->flag; a.ny = halo->ny; a.hl = halo->layout;
    }
    if (dtype == NUFFT_F32) return forward ? launch_cplx_t<float, true>((int)n, a, stream) : launch_cplx_t<float, false>((int)n, a, stream);
    return forward ? launch_cplx_t<double, true>((int)n, a, stream) : launch_cplx_t<double, false>((int)n, a, stream);
}

template <typename T, int N, bool FWD, bool MULT>
static hipError_t launch_n_m(const FftLineArgs& a, hipStream_t stream) {
    using C = typename Cplx2<T>::type;
    constexpr int LINE = N + (N >> 4) + 1;
    // TA lines per workgroup: as many as fit in the LDS limit, at most 16.  Float64: two workgroups per CU (70 KB at N = 512) beat one
    // with 16 lines (measured).  Float32: 16 lines make the strided accesses 128-byte segments instead of 64-byte ones, which is worth
    // one workgroup per CU (C3, N = 1024: deconvolution + padding pass 4.47 -> 3.47 ms, last forward pass 1.66 -> 1.40 ms)
#ifndef NUFFT_FFT_LDS_LIMIT
#define NUFFT_FFT_LDS_LIMIT (sizeof(C) == 8 ? 160 * 1024 - 1024 : 80 * 1024)
#endif
    constexpr int TA = (sizeof(C) * (16 * LINE + N) <= NUFFT_FFT_LDS_LIMIT) ? 16 : ((sizeof(C) * (8 * LINE + N) <= NUFFT_FFT_LDS_LIMIT) ? 8 : 4);
    static_assert(sizeof(C) * (size_t)(TA * LINE + N) <= kFftLdsLimit, "line buffers exceed the 160 KiB of LDS");
    const size_t lds = sizeof(C) * (size_t)(TA * LINE + N);
    auto fn = fft_lines_kernel<T, N, FWD, TA, MULT>;
    // the attribute is per device: remember which devices of this process have it (plans may live on several)
    static std::atomic<unsigned long long> prepared{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(prepared.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        prepared.fetch_or(bit, std::memory_order_relaxed);
    }
    const int64_t acols = FWD ? a.a_total : a.a_out;
    dim3 grid((unsigned)((acols + TA - 1) / TA), (unsigned)a.nc, 1);
    hipLaunchKernelGGL(fn, grid, dim3(TA * kWave), lds, stream, a);
    return hipGetLastError();
}

template <typename T, int N, bool FWD>
static hipError_t launch_n(const FftLineArgs& a, hipStream_t stream) {
    return a.mult ? launch_n_m<T, N, FWD, true>(a, stream) : launch_n_m<T, N, FWD, false>(a, stream);
}

template <typename T, bool FWD>
static hipError_t launch_t(int n, const FftLineArgs& a, hipStream_t stream) {
    switch (n) {
#define NUFFT_CASE(NN) case NN: return launch_n<T, NN, FWD>(a, stream);
        NUFFT_FFT_SIZES(NUFFT_CASE)
#undef NUFFT_CASE
        default: return hipErrorInvalidValue;
    }
}

bool fft_lines_supported(int dtype, int64_t n) {
    (void)dtype;
    return size_instantiated(n);
}

hipError_t launch_fft_lines(int dtype, int64_t n, bool forward, const FftLinePass& p, hipStream_t stream) {
    FftLineArgs a;
    a.in = p.in; a.out = p.out;
    a.a_total = p.a_total; a.a_out = p.a_out;
    a.in_stride_j = p.in_stride_j; a.in_stride_c = p.in_stride_c;
    a.out_stride_j = p.out_stride_j; a.out_stride_c = p.out_stride_c;
    a.nc = p.nc; a.nk = p.nk; a.map = p.map; a.fa = p.fa; a.ka = p.ka; a.fk = p.fk;
    a.twiddle = p.twiddle; a.scale = p.scale; a.mult = p.mult;
    a.row_a = p.row_a; a.row_valid = p.row_valid; a.row_in = p.row_in; a.row_out = p.row_out;
    if (dtype == NUFFT_F32) return forward ? launch_t<float, true>((int)n, a, stream) : launch_t<float, false>((int)n, a, stream);
    return forward ? launch_t<double, true>((int)n, a, stream) : launch_t<double, false>((int)n, a, stream);
}

}  // namespace nufft
