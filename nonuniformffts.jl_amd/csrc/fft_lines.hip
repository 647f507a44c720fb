// Pruned 1-D FFT passes over strided lines (power-of-two lengths), fused with deconvolution.
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
// (radix 8, then 4 or 2) on its own line with twiddles from an LDS table.
#include <hip/hip_runtime.h>

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
};

// One radix-R Stockham stage of a line held in LDS (in place, wave-synchronous).
// TWS: the twiddle table holds the roots of unity of order N * TWS (TWS = 2 for the half-length complex FFT
// inside a real transform, whose table is shared with the real/complex split step).
template <typename T, int N, int R, int SIGN, int TWS = 1>
__device__ __forceinline__ void stage(typename Cplx2<T>::type* line, const typename Cplx2<T>::type* tw, int p, int lane) {
    using C = typename Cplx2<T>::type;
    constexpr int NB = N / R;                         // butterflies per line
    constexpr int PER = (NB + kWave - 1) / kWave;     // butterflies per lane
    C u[PER][R];
    int jout[PER];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
        const int i = lane + b * kWave;
        const int k = i & (p - 1);
        jout[b] = (i - k) * R + k;
        if (i < NB) {
#pragma unroll
            for (int t = 0; t < R; ++t) u[b][t] = line[lpad(i + t * NB)];
            if (p > 1) {
                const int step = k * (N / (p * R));   // w_{pR}^{k t} = w_N^{k t N / (p R)}
#pragma unroll
                for (int t = 1; t < R; ++t) u[b][t] = cmul(u[b][t], tw[((step * t) & (N - 1)) * TWS]);
            }
            if constexpr (R == 8) dft8<SIGN, T>(u[b]);
            else if constexpr (R == 4) dft4<SIGN>(u[b]);
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

template <typename T, int LOGN, int SIGN, int TWS = 1>
__device__ __forceinline__ void fft_line(typename Cplx2<T>::type* line, const typename Cplx2<T>::type* tw, int lane) {
    constexpr int N = 1 << LOGN;
    int p = 1;
    constexpr int N8 = LOGN / 3;
#pragma unroll
    for (int s = 0; s < N8; ++s) { stage<T, N, 8, SIGN, TWS>(line, tw, p, lane); p *= 8; }
    if constexpr (LOGN % 3 == 2) stage<T, N, 4, SIGN, TWS>(line, tw, p, lane);
    if constexpr (LOGN % 3 == 1) stage<T, N, 2, SIGN, TWS>(line, tw, p, lane);
}

// FWD: full input (N along j), pruned output (nk along k').  BWD: pruned input, full output.
// MULT: a real multiplier array (uniform callback menu) is applied on the pruned side.
template <typename T, int LOGN, bool FWD, int TA, bool MULT>
__global__ __launch_bounds__(TA * kWave) void fft_lines_kernel(FftLineArgs a) {
    using C = typename Cplx2<T>::type;
    constexpr int N = 1 << LOGN;
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

    if (FWD) {
        for (int e = tid; e < TA * N; e += NT) {
            const int ai = e % TA, j = e / TA;
            C v; v.x = T(0); v.y = T(0);
            if (a0 + ai < a.a_total) v = in[a0 + ai + (int64_t)j * a.in_stride_j];
            lines[ai * LINE + lpad(j)] = v;
        }
    } else {
        C z; z.x = T(0); z.y = T(0);
        for (int e = tid; e < TA * LINE; e += NT) lines[e] = z;
        __syncthreads();
        for (int e = tid; e < TA * a.nk; e += NT) {
            const int ai = e % TA, k = e / TA;
            if (a0 + ai < a.a_total) {
                const int64_t off = a0 + ai + (int64_t)k * a.in_stride_j;
                C v = in[off];
                T f = fa[(a0 + ai) % a.ka] * fk[k] * scale;
                if constexpr (MULT) f *= mult[off];
                v.x *= f; v.y *= f;
                lines[ai * LINE + lpad(a.map[k])] = v;
            }
        }
    }
    __syncthreads();

    fft_line<T, LOGN, FWD ? -1 : 1>(lines + wave * LINE, tw, lane);
    __syncthreads();

    if (FWD) {
        for (int e = tid; e < TA * a.nk; e += NT) {
            const int ai = e % TA, k = e / TA;
            if (a0 + ai < a.a_total) {
                C v = lines[ai * LINE + lpad(a.map[k])];
                const int64_t off = a0 + ai + (int64_t)k * a.out_stride_j;
                T f = fa[(a0 + ai) % a.ka] * fk[k] * scale;
                if constexpr (MULT) f *= mult[off];
                v.x *= f; v.y *= f;
                out[off] = v;
            }
        }
    } else {
        for (int e = tid; e < TA * N; e += NT) {
            const int ai = e % TA, j = e / TA;
            if (a0 + ai < a.a_out) {
                C v = lines[ai * LINE + lpad(j)];
                if (a0 + ai >= a.a_total) { v.x = T(0); v.y = T(0); }
                out[a0 + ai + (int64_t)j * a.out_stride_j] = v;
            }
        }
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
    const void* twiddle;    // complex<T>[N]: exp(SIGN 2πi m / N), N = 2M
};

template <typename T, int LOGM, bool FWD, int TL>
__global__ __launch_bounds__(TL * kWave) void real_lines_kernel(RealLineArgs a) {
    using C = typename Cplx2<T>::type;
    constexpr int M = 1 << LOGM;
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
        for (int n = lane; n < M; n += kWave) line[lpad(n)] = zin[n];
        wave_lds_fence();
        fft_line<T, LOGM, -1, 2>(line, tw, lane);
        C* xout = static_cast<C*>(a.out) + line_id * a.k1;
        for (int k = lane; k < a.k1; k += kWave) {
            const C zk = line[lpad(k & (M - 1))];
            C zm = line[lpad((M - k) & (M - 1))];
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
        const C* xin = static_cast<const C*>(a.in) + line_id * a.k1;
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
        fft_line<T, LOGM, 1, 2>(line, tw, lane);
        C* zout = reinterpret_cast<C*>(static_cast<T*>(a.out) + line_id * N);
        for (int n = lane; n < M; n += kWave) zout[n] = line[lpad(n)];
    }
}

template <typename T, int LOGM, bool FWD>
static hipError_t launch_real_logm(const RealLineArgs& a, hipStream_t stream) {
    using C = typename Cplx2<T>::type;
    constexpr int M = 1 << LOGM;
    constexpr int LINE = M + (M >> 4) + 1;
    constexpr int TL = (sizeof(C) * (16 * LINE + 2 * M) <= 150 * 1024) ? 16 : 8;
    const size_t lds = sizeof(C) * (size_t)(TL * LINE + 2 * M);
    auto fn = real_lines_kernel<T, LOGM, FWD, TL>;
    static bool prepared = false;
    if (!prepared) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        prepared = true;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)((a.nlines + TL - 1) / TL)), dim3(TL * kWave), lds, stream, a);
    return hipGetLastError();
}

template <typename T, bool FWD>
static hipError_t launch_real_t(int logm, const RealLineArgs& a, hipStream_t stream) {
    switch (logm) {
        case 6: return launch_real_logm<T, 6, FWD>(a, stream);
        case 7: return launch_real_logm<T, 7, FWD>(a, stream);
        case 8: return launch_real_logm<T, 8, FWD>(a, stream);
        case 9: return launch_real_logm<T, 9, FWD>(a, stream);
        case 10: return launch_real_logm<T, 10, FWD>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

bool real_lines_supported(int dtype, int64_t n) {
    (void)dtype;
    if (n < 128 || n > 2048) return false;
    return (n & (n - 1)) == 0;
}

hipError_t launch_real_lines(int dtype, int logn, bool forward, const void* in, void* out, int64_t nlines, int k1,
                             const void* twiddle, hipStream_t stream) {
    RealLineArgs a;
    a.in = in; a.out = out; a.nlines = nlines; a.k1 = k1; a.twiddle = twiddle;
    const int logm = logn - 1;
    if (dtype == NUFFT_F32) return forward ? launch_real_t<float, true>(logm, a, stream) : launch_real_t<float, false>(logm, a, stream);
    return forward ? launch_real_t<double, true>(logm, a, stream) : launch_real_t<double, false>(logm, a, stream);
}

template <typename T, int LOGN, bool FWD, bool MULT>
static hipError_t launch_logn_m(const FftLineArgs& a, hipStream_t stream) {
    using C = typename Cplx2<T>::type;
    constexpr int N = 1 << LOGN;
    constexpr int LINE = N + (N >> 4) + 1;
    // TA lines per workgroup: 16 when they fit in ~150 KB of LDS, else 8 / 4
#ifndef NUFFT_FFT_LDS_LIMIT
#define NUFFT_FFT_LDS_LIMIT (80 * 1024)      // two workgroups per CU (70 KB at N = 512 Float64) beat one with 16 lines: measured
#endif
    constexpr int TA = (sizeof(C) * (16 * LINE + N) <= NUFFT_FFT_LDS_LIMIT) ? 16 : ((sizeof(C) * (8 * LINE + N) <= NUFFT_FFT_LDS_LIMIT) ? 8 : 4);
    const size_t lds = sizeof(C) * (size_t)(TA * LINE + N);
    auto fn = fft_lines_kernel<T, LOGN, FWD, TA, MULT>;
    static bool prepared = false;
    if (!prepared) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        prepared = true;
    }
    const int64_t acols = FWD ? a.a_total : a.a_out;
    dim3 grid((unsigned)((acols + TA - 1) / TA), (unsigned)a.nc, 1);
    hipLaunchKernelGGL(fn, grid, dim3(TA * kWave), lds, stream, a);
    return hipGetLastError();
}

template <typename T, int LOGN, bool FWD>
static hipError_t launch_logn(const FftLineArgs& a, hipStream_t stream) {
    return a.mult ? launch_logn_m<T, LOGN, FWD, true>(a, stream) : launch_logn_m<T, LOGN, FWD, false>(a, stream);
}

template <typename T, bool FWD>
static hipError_t launch_t(int logn, const FftLineArgs& a, hipStream_t stream) {
    switch (logn) {
        case 6: return launch_logn<T, 6, FWD>(a, stream);
        case 7: return launch_logn<T, 7, FWD>(a, stream);
        case 8: return launch_logn<T, 8, FWD>(a, stream);
        case 9: return launch_logn<T, 9, FWD>(a, stream);
        case 10: return launch_logn<T, 10, FWD>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

bool fft_lines_supported(int dtype, int64_t n) {
    if (n < 64 || n > 1024) return false;
    return (n & (n - 1)) == 0;
}

hipError_t launch_fft_lines(int dtype, int logn, bool forward, const FftLinePass& p, hipStream_t stream) {
    FftLineArgs a;
    a.in = p.in; a.out = p.out;
    a.a_total = p.a_total; a.a_out = p.a_out;
    a.in_stride_j = p.in_stride_j; a.in_stride_c = p.in_stride_c;
    a.out_stride_j = p.out_stride_j; a.out_stride_c = p.out_stride_c;
    a.nc = p.nc; a.nk = p.nk; a.map = p.map; a.fa = p.fa; a.ka = p.ka; a.fk = p.fk;
    a.twiddle = p.twiddle; a.scale = p.scale; a.mult = p.mult;
    if (dtype == NUFFT_F32) return forward ? launch_t<float, true>(logn, a, stream) : launch_t<float, false>(logn, a, stream);
    return forward ? launch_t<double, true>(logn, a, stream) : launch_t<double, false>(logn, a, stream);
}

}  // namespace nufft
