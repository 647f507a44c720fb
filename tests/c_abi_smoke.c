/*
 * c_abi_smoke.c — the C ABI of libnufft_mi355x.so driven from plain C through include/nufft_mi355x.h alone (no
 * Python, no ctypes mirror of the structs): create -> set_points -> exec_type1 -> exec_type2 -> destroy on a small 3-D
 * Float64 problem, results checked against the direct sums  û(k) = Σ_j v_j exp(-i k·x_j)  /  v_j = Σ_k û(k) exp(+i k·x_j)
 * computed on the host (the reference's known answers, test/accuracy.jl:119-125,184-200), at the m = 4, σ = 2
 * ceiling 6·10^(-1.9 m) of test/accuracy.jl:33-35.
 *
 * Built by __graft_entry__.build() (gcc + the HIP runtime library for device memory) into tests/c_abi_smoke.
 *   ./c_abi_smoke            full run on device 0
 *   ./c_abi_smoke --host     no GPU: host-only plan (device = -1), struct sizes, error paths
 * Exit code 0 = pass.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nufft_mi355x.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != NUFFT_OK) { \
    fprintf(stderr, "%s:%d: %s -> %d (%s: %s)\n", __FILE__, __LINE__, #call, rc_, nufft_strerror(rc_), nufft_last_error_message()); return 1; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return 1; } } while (0)

static double urand(uint64_t* s) {          /* xorshift64*: deterministic inputs without libc's rand */
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return (double)((*s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

static int host_only(void) {
    if (nufft_sizeof_params() != (int64_t)sizeof(nufft_params) || nufft_sizeof_info() != (int64_t)sizeof(nufft_info)) {
        fprintf(stderr, "struct layout differs between header and library\n");
        return 1;
    }
    nufft_params prm;
    memset(&prm, 0, sizeof prm);
    prm.struct_size = (int32_t)sizeof prm;      /* ABI >= 104: how much of the struct this caller knows */
    prm.dtype = NUFFT_F64; prm.ndim = 3; prm.N[0] = 24; prm.N[1] = 20; prm.N[2] = 16; prm.device = -1;
    nufft_plan* plan = NULL;
    CHECK(nufft_plan_create_ex(&plan, &prm));
    nufft_info info;
    CHECK(nufft_plan_info(plan, &info));
    if (info.N_over[0] != 48 || info.N_over[1] != 40 || info.N_over[2] != 32 || info.N_out[0] != 13 || info.half_support != 4) {
        fprintf(stderr, "unexpected plan geometry\n");
        return 1;
    }
    /* a caller of the ABI <= 102 layout (struct_size = 0, a SHORTER struct in its own memory): the library must not read past it.
       The fields behind `reserved` are poisoned here; a library that read them would refuse N_over = -1 / take the options pointer. */
    {
        nufft_params old = prm;
        old.struct_size = 0;
        old.N_over[0] = -1; old.kernel_param_dim[1] = -5.0; old.options = (const char*)(uintptr_t)0x10;
        nufft_plan* po = NULL;
        CHECK(nufft_plan_create_ex(&po, &old));
        nufft_info io;
        CHECK(nufft_plan_info(po, &io));
        if (io.N_over[0] != 48) { fprintf(stderr, "legacy-size caller: trailing fields were read\n"); return 1; }
        CHECK(nufft_plan_destroy(po));
    }
    /* development switches travel in the struct, not in the environment */
    {
        nufft_params sw = prm;
        sw.options = "NUFFT_BIN_LOG2=3;NUFFT_X=1";
        nufft_plan* ps = NULL;
        CHECK(nufft_plan_create_ex(&ps, &sw));
        nufft_info is;
        CHECK(nufft_plan_info(ps, &is));
        if (is.bin_dims[0] != 8 || strcmp(nufft_plan_options(ps), "NUFFT_BIN_LOG2=3;NUFFT_X=1") != 0) { fprintf(stderr, "options not honoured: %s\n", nufft_plan_options(ps)); return 1; }
        CHECK(nufft_plan_destroy(ps));
        if (info.bin_dims[0] != 4 || strcmp(nufft_plan_options(plan), "") != 0) { fprintf(stderr, "default plan carries options\n"); return 1; }
    }
    double phi[32];
    CHECK(nufft_plan_get_phi_hat(plan, 1, phi, 32));
    if (!(phi[0] > 0.0)) { fprintf(stderr, "phi_hat\n"); return 1; }
    /* errors are return codes, never exceptions: Ñ < 2M (src/plan.jl:545-556), unknown dtype, null plan */
    nufft_params bad = prm; bad.N[0] = 2; bad.N[1] = 2; bad.N[2] = 2; bad.half_support = 8;
    nufft_plan* q = NULL;
    if (nufft_plan_create_ex(&q, &bad) != NUFFT_ERR_SIZE_TOO_SMALL) { fprintf(stderr, "expected SIZE_TOO_SMALL\n"); return 1; }
    if (nufft_set_points(plan, 10, NULL, NULL) == NUFFT_OK) { fprintf(stderr, "host-only plan accepted points\n"); return 1; }
    CHECK(nufft_plan_destroy(plan));
    printf("c_abi_smoke --host: ok (version %d, sizeof params %lld, info %lld)\n", nufft_version(),
           (long long)nufft_sizeof_params(), (long long)nufft_sizeof_info());
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--host") == 0) return host_only();
    if (host_only()) return 1;
    const int64_t N[3] = {24, 20, 16};
    const int64_t Np = 3000;
    const int M = 4;
    nufft_plan* plan = NULL;
    CHECK(nufft_plan_create(&plan, NUFFT_F64, 0, 3, N, M, 2.0, NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL, NUFFT_EVAL_DIRECT, 1, 0,
                            NUFFT_POINT_TRANSFORM_IDENTITY, 0));
    nufft_info info;
    CHECK(nufft_plan_info(plan, &info));
    const int64_t K1 = info.N_out[0], K2 = info.N_out[1], K3 = info.N_out[2], nout = K1 * K2 * K3;

    uint64_t seed = 0x9E3779B97F4A7C15ULL;
    double* x[3];
    double* v = (double*)malloc(sizeof(double) * Np);
    for (int d = 0; d < 3; ++d) {
        x[d] = (double*)malloc(sizeof(double) * Np);
        for (int64_t j = 0; j < Np; ++j) x[d][j] = (urand(&seed) * 3.0 - 1.0) * 2.0 * M_PI;     /* also outside [0, 2π) */
    }
    for (int64_t j = 0; j < Np; ++j) v[j] = urand(&seed) - 0.5;

    void* dx[3]; void* dv; void* du; void* dw;
    for (int d = 0; d < 3; ++d) {
        HIPCHECK(hipMalloc(&dx[d], sizeof(double) * Np));
        HIPCHECK(hipMemcpy(dx[d], x[d], sizeof(double) * Np, hipMemcpyHostToDevice));
    }
    HIPCHECK(hipMalloc(&dv, sizeof(double) * Np));
    HIPCHECK(hipMemcpy(dv, v, sizeof(double) * Np, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&du, 2 * sizeof(double) * nout));
    HIPCHECK(hipMalloc(&dw, sizeof(double) * Np));
    hipStream_t stream;
    HIPCHECK(hipStreamCreate(&stream));

    const void* coords[3] = {dx[0], dx[1], dx[2]};
    CHECK(nufft_set_points(plan, Np, coords, stream));
    const void* vin[1] = {dv};
    void* uout[1] = {du};
    CHECK(nufft_exec_type1(plan, uout, vin, stream));
    const void* uin[1] = {du};
    void* wout[1] = {dw};
    CHECK(nufft_exec_type2(plan, wout, uin, stream));
    HIPCHECK(hipStreamSynchronize(stream));

    double* u = (double*)malloc(2 * sizeof(double) * nout);
    double* w = (double*)malloc(sizeof(double) * Np);
    HIPCHECK(hipMemcpy(u, du, 2 * sizeof(double) * nout, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(w, dw, sizeof(double) * Np, hipMemcpyDeviceToHost));

    /* type 1 against the direct sum on every 7th mode (dimension 1 fastest; k1 = 0..N1/2, k2 / k3 in FFT order) */
    double num = 0, den = 0;
    for (int64_t idx = 0; idx < nout; idx += 7) {
        const int64_t i1 = idx % K1, i2 = (idx / K1) % K2, i3 = idx / (K1 * K2);
        const double k1 = (double)i1, k2 = (double)(i2 < (K2 + 1) / 2 ? i2 : i2 - K2), k3 = (double)(i3 < (K3 + 1) / 2 ? i3 : i3 - K3);
        double re = 0, im = 0;
        for (int64_t j = 0; j < Np; ++j) {
            const double ph = k1 * x[0][j] + k2 * x[1][j] + k3 * x[2][j];
            re += v[j] * cos(ph); im -= v[j] * sin(ph);
        }
        num += (u[2 * idx] - re) * (u[2 * idx] - re) + (u[2 * idx + 1] - im) * (u[2 * idx + 1] - im);
        den += re * re + im * im;
    }
    const double e1 = sqrt(num / den), ceil_m4 = 6.0 * pow(10.0, -1.9 * M);
    /* type 2 of the computed spectrum at every 50th point: v_j = Σ_k h(k1) Re(û(k) exp(i k·x_j)), h = 1 at k1 = 0 and 2
     * otherwise — also at k1 = N1/2, an ordinary mode of the oversampled c2r transform (test/accuracy.jl:184-186:
     * factor = ifelse(iszero(k), 1, 2)) */
    num = den = 0;
    for (int64_t j = 0; j < Np; j += 50) {
        double s = 0;
        for (int64_t idx = 0; idx < nout; ++idx) {
            const int64_t i1 = idx % K1, i2 = (idx / K1) % K2, i3 = idx / (K1 * K2);
            const double k1 = (double)i1, k2 = (double)(i2 < (K2 + 1) / 2 ? i2 : i2 - K2), k3 = (double)(i3 < (K3 + 1) / 2 ? i3 : i3 - K3);
            const double ph = k1 * x[0][j] + k2 * x[1][j] + k3 * x[2][j];
            const double h = i1 == 0 ? 1.0 : 2.0;
            s += h * (u[2 * idx] * cos(ph) - u[2 * idx + 1] * sin(ph));
        }
        num += (w[j] - s) * (w[j] - s);
        den += s * s;
    }
    const double e2 = sqrt(num / den);
    printf("c_abi_smoke: type-1 rel-L2 vs direct sum %.3e, type-2 %.3e (ceiling %.1e)\n", e1, e2, ceil_m4);
    int engine = 0;
    CHECK(nufft_spread_engine_used(plan, &engine, stream));
    CHECK(nufft_plan_destroy(plan));
    for (int d = 0; d < 3; ++d) { (void)hipFree(dx[d]); free(x[d]); }
    (void)hipFree(dv); (void)hipFree(du); (void)hipFree(dw);
    (void)hipStreamDestroy(stream);
    free(v); free(u); free(w);
    if (!(e1 < 2 * ceil_m4) || !(e2 < 2 * ceil_m4) || engine != NUFFT_SPREAD_LDS_TILES) return 1;
    return 0;
}
