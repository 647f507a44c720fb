# NonuniformFFTsMI355XExt.jl — package extension that routes a NonuniformFFTs.jl plan to libnufft_mi355x.so.
#
# Copy to NonuniformFFTs.jl/ext/ and declare it in Project.toml ([weakdeps] AMDGPU, [extensions]
# NonuniformFFTsMI355XExt = "AMDGPU"); INTEGRATION.md walks _PlanNUFFT (src/plan.jl:467-541) line by line and names the method
# every call hits for this backend.  Nothing inside src/ is edited.
# NEVER EXECUTED in the build image (no Julia runtime).  tests/test_julia_shim_static.py checks, against include/nufft_mi355x.h and
# the reference source: every ccall (symbol, return and argument types), the mirrored structs, every NUFFT_* constant, that the
# enum slots of CParams are filled from those constants only, every NonuniformFFTs.* / Kernels.* / AbstractNFFTs.* name the file
# *calls* (exists, with that positional arity), every field it reads from a reference struct, every constructor call of its own
# structs (as many arguments as fields), and that each overload repeats a reference signature in every slot but the backend's
# (so it is strictly more specific: no ambiguity).
module NonuniformFFTsMI355XExt

using NonuniformFFTs
using NonuniformFFTs: PlanNUFFT, NUFFTCallbacks, default_callback, Kernels, AbstractBlockData, AbstractNUFFTData, AbstractNFFTs,
                      HalfSupport, StaticBool, True, False, Direct, AbstractKernel
using NonuniformFFTs.Kernels: AbstractKernelData
using TimerOutputs: @timeit
using Adapt: adapt
using AMDGPU
using KernelAbstractions: KernelAbstractions as KA

const libnufft = "libnufft_mi355x.so"      # nonuniformffts.jl_amd/libnufft_mi355x.so

# ---- constants of include/nufft_mi355x.h (checked one by one against the header) --------------------------------------
const NUFFT_F32 = Int32(0)
const NUFFT_F64 = Int32(1)
const NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL = Int32(0)
const NUFFT_KERNEL_KAISER_BESSEL = Int32(1)
const NUFFT_KERNEL_GAUSSIAN = Int32(2)
const NUFFT_KERNEL_BSPLINE = Int32(3)
const NUFFT_EVAL_DIRECT = Int32(0)
const NUFFT_EVAL_FAST_APPROXIMATION = Int32(1)
const NUFFT_METHOD_SHARED_MEMORY = Int32(0)
const NUFFT_METHOD_GLOBAL_MEMORY = Int32(1)
const NUFFT_POINT_TRANSFORM_IDENTITY = Int32(0)
const NUFFT_POINT_TRANSFORM_NFFT = Int32(1)
const NUFFT_SPREAD_AUTO = Int32(0)
const NUFFT_MI355X_VERSION = Int32(104)             # the ABI this file was written against (nufft_params.struct_size / .options, nufft_set_callbacks)
const NUFFT_ERR_INVALID_ARG = Int32(1)
const NUFFT_ERR_SIZE_TOO_SMALL = Int32(2)
const NUFFT_ERR_DIM_MISMATCH = Int32(3)
const NUFFT_ERR_LDS_TOO_SMALL = Int32(4)
const NUFFT_ERR_UNSUPPORTED = Int32(5)
const NUFFT_ERR_NO_POINTS = Int32(6)
const NUFFT_ERR_NO_DEVICE = Int32(10)

# ---- backend value: a KA.GPU of its own, forwarding allocation to ROCBackend ----------------------------------------
struct MI355XBackend <: KA.GPU
    roc::ROCBackend
end
MI355XBackend() = MI355XBackend(ROCBackend())
# (KA.allocate(backend, T, dims...) and KA.zeros(backend, T, dims...) of KernelAbstractions forward to this tuple form)
KA.allocate(b::MI355XBackend, ::Type{T}, dims::Tuple; kws...) where {T} = KA.allocate(b.roc, T, dims; kws...)
KA.synchronize(b::MI355XBackend) = KA.synchronize(b.roc)

NonuniformFFTs.default_kernel(::MI355XBackend) = BackwardsKaiserBesselKernel()   # ext/NonuniformFFTsAMDGPUExt.jl:54
NonuniformFFTs.default_kernel_evalmode(::MI355XBackend) = Direct()               # ext/NonuniformFFTsAMDGPUExt.jl:56
# default_block_size(Ns, ::GPU), default_gpu_batch_size(::KA.Backend), to_unit_cell(::GPU, x): the reference's generic methods apply
# (src/NonuniformFFTs.jl:58-63, src/gpu_common.jl:7, src/blocking/blocking.jl:7)

lasterr() = unsafe_string(ccall((:nufft_last_error_message, libnufft), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing :
    rc in (NUFFT_ERR_INVALID_ARG, NUFFT_ERR_SIZE_TOO_SMALL, NUFFT_ERR_LDS_TOO_SMALL, NUFFT_ERR_UNSUPPORTED, NUFFT_ERR_NO_POINTS, NUFFT_ERR_NO_DEVICE) ?
        throw(ArgumentError(lasterr())) :
    rc == NUFFT_ERR_DIM_MISMATCH ? throw(DimensionMismatch(lasterr())) : error(lasterr())

# ---- plan-owned state ------------------------------------------------------------------------------------------------
# p.blocks: only what show(::PlanNUFFT) and the keyword checks ask of it (src/plan.jl:375-390) — no tile arrays
struct MI355XBlockData{D, S <: StaticBool, Np} <: AbstractBlockData
    method::Symbol
    block_dims::Dims{D}
    sort_points::S
    batch_size::Val{Np}
end
function NonuniformFFTs.BlockDataGPU(::Type{Z}, backend::MI355XBackend, block_dims::Dims{D}, Ñs::Dims{D}, h::HalfSupport{M},
                                     sort_points::StaticBool; method::Symbol, batch_size::Val) where {Z <: Number, D, M}
    method ∈ (:global_memory, :shared_memory) || throw(ArgumentError("expected gpu_method ∈ (:global_memory, :shared_memory)"))   # src/blocking/gpu.jl:26
    # Plan-time errors at plan time: the reference throws from PlanNUFFT(…) when the LDS cannot hold a tile (src/gpu_common.jl:55-78) and
    # for sizes below 2M (src/plan.jl:545-556).  The device handle is only built at the first set_points! (the kernel data, evaluation
    # mode and point transform are not visible from the functions _PlanNUFFT dispatches on the backend), so what decides those errors
    # here — element type, D, M, Ñs — goes through the library's own parameter checks now, on a HOST-ONLY plan (device = -1: no GPU
    # call, no allocation): D > 3 or M outside 2:10 (NUFFT_ERR_UNSUPPORTED), Ñ < 2M, an LDS budget no tile fits.
    probe_parameters(Z, Ñs, Val(M), method)
    MI355XBlockData(method, block_dims, sort_points, batch_size)
end
NonuniformFFTs.gpu_method(bd::MI355XBlockData) = bd.method                 # show(::PlanNUFFT), src/plan.jl:381
NonuniformFFTs.with_blocking(bd::MI355XBlockData) = true
NonuniformFFTs.get_batch_size(bd::MI355XBlockData) = NonuniformFFTs.get_batch_size(bd.batch_size)      # src/blocking/gpu.jl:38-39
# get_block_dims(bd) = bd.block_dims and get_sort_points(bd) = bd.sort_points: the AbstractBlockData defaults (src/blocking/blocking.jl:3-4)

# What _PlanNUFFT asks of output_field(data) (src/plan.jl:531-535): `first(...)::AbstractArray{<:Complex}` and its `axes`, which
# non_oversampled_indices! requires to be at least as long as ks (src/NonuniformFFTs.jl:322).  A storage-free array of the
# oversampled spectrum's size: (Ñ₁÷2+1, Ñ₂, …) for real data, Ñs for complex data (src/plan.jl:43,55).  Never read.
struct SpectrumShape{T, N} <: AbstractArray{Complex{T}, N}
    dims::Dims{N}
end
Base.size(a::SpectrumShape) = a.dims
Base.getindex(a::SpectrumShape{T, N}, I::Vararg{Int, N}) where {T, N} = zero(Complex{T})

# p.data: the wavenumbers (size(p), check_nufft_uniform_data: src/plan.jl:426, src/NonuniformFFTs.jl:92-103) and the C handle,
# created at the first set_points!.  No KA grids and no rocFFT plans: the handle owns them.
mutable struct MI355XData{Z, N, Nc, W, T} <: AbstractNUFFTData{Z, N, Nc}
    ks::W
    shape::NTuple{Nc, SpectrumShape{T, N}}
    handle::Ptr{Cvoid}
end
NonuniformFFTs.output_field(data::MI355XData) = data.shape                 # a tuple of Nc arrays, as for RealNUFFTData (src/plan.jl:33)
function new_plan_data(::Type{Z}, dims_out::Dims{N}, ks::W, ::Val{Nc}) where {Z, N, W, Nc}
    T = real(Z)
    shape = ntuple(_ -> SpectrumShape{T, N}(dims_out), Val(Nc))
    data = MI355XData{Z, N, Nc, W, T}(ks, shape, C_NULL)
    finalizer(d -> d.handle == C_NULL || ccall((:nufft_plan_destroy, libnufft), Cint, (Ptr{Cvoid},), d.handle), data)
    data
end
# one method per method of the reference (src/plan.jl:37-41, 52-56): the same first-argument types, so that these are strictly
# more specific (a single `::Type{Z}` method would be ambiguous with both)
function NonuniformFFTs.init_plan_data(::Type{T}, backend::MI355XBackend, Ñs::Dims, ks::NTuple, ::Val{Nc};
                                       plan_kwargs) where {T <: AbstractFloat, Nc}
    new_plan_data(T, (Ñs[1] ÷ 2 + 1, Base.tail(Ñs)...), ks, Val(Nc))
end
function NonuniformFFTs.init_plan_data(::Type{Complex{T}}, backend::MI355XBackend, Ñs::Dims, ks::NTuple, ::Val{Nc};
                                       plan_kwargs) where {T <: AbstractFloat, Nc}
    new_plan_data(Complex{T}, Ñs, ks, Val(Nc))
end

# The kernel is the first type parameter of the kernel data (AbstractKernelData{K, M, T}, src/Kernels/Kernels.jl:63); the shape
# parameter is a field of the data — already resolved per dimension by optimal_kernel, explicit or optimal, in the plan's precision:
# β (kaiser_bessel_backwards.jl:84, kaiser_bessel.jl:112), σ = ℓ Δx (gaussian.jl:67,76-78); the B-spline has none.
kernel_id(::AbstractKernelData{BackwardsKaiserBesselKernel}) = NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL
kernel_id(::AbstractKernelData{KaiserBesselKernel}) = NUFFT_KERNEL_KAISER_BESSEL
kernel_id(::AbstractKernelData{GaussianKernel}) = NUFFT_KERNEL_GAUSSIAN
kernel_id(::AbstractKernelData{BSplineKernel}) = NUFFT_KERNEL_BSPLINE
shape_param(g::AbstractKernelData{BackwardsKaiserBesselKernel}) = Float64(g.β)
shape_param(g::AbstractKernelData{KaiserBesselKernel}) = Float64(g.β)
shape_param(g::AbstractKernelData{GaussianKernel}) = Float64(g.σ / Kernels.gridstep(g))
shape_param(g::AbstractKernelData{BSplineKernel}) = 0.0

# nufft_params mirrored field by field (include/nufft_mi355x.h); nufft_sizeof_params() guards the layout
struct CParams
    dtype::Int32; is_complex::Int32; ndim::Int32; N::NTuple{3, Int64}; half_support::Int32; sigma::Float64
    kernel::Int32; evalmode::Int32; ntransforms::Int32; fftshift::Int32; point_transform::Int32; gpu_method::Int32
    device::Int32; tile_dims::NTuple{3, Int32}; lds_budget_bytes::Int32; spread_threads::Int32; interp_threads::Int32
    interp_tile_dims::NTuple{3, Int32}; bin_log2::Int32; spread_method::Int32; kernel_param::Float64; struct_size::Int32; reserved::Int32
    kernel_param_dim::NTuple{3, Float64}; N_over::NTuple{3, Int64}; options::Ptr{UInt8}
end

pad3(f, N, z) = ntuple(d -> d ≤ N ? f(d) : z, Val(3))

function check_library()
    ccall((:nufft_version, libnufft), Cint, ()) ≥ NUFFT_MI355X_VERSION || error("libnufft_mi355x.so is older than ABI $NUFFT_MI355X_VERSION")
    ccall((:nufft_sizeof_params, libnufft), Int64, ()) == sizeof(CParams) || error("nufft_params layout differs from the library's")
    nothing
end

function probe_parameters(::Type{Z}, Ñs::Dims{D}, ::Val{M}, method::Symbol) where {Z <: Number, D, M}
    D ≤ 3 || throw(ArgumentError("MI355XBackend: transforms of more than 3 dimensions are not supported (got $D); use ROCBackend()"))
    check_library()
    T = real(Z)
    zero3 = (Int32(0), Int32(0), Int32(0))
    Ns = pad3(d -> Int64(cld(Ñs[d], 2)), D, Int64(0))          # any N ≤ Ñ: the checks in question depend on Ñ, M, D and the element type only
    probe = CParams(T === Float64 ? NUFFT_F64 : NUFFT_F32, Int32(Z <: Complex), Int32(D), Ns, Int32(M), 0.0,
                  NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL, NUFFT_EVAL_DIRECT, Int32(1), Int32(0), NUFFT_POINT_TRANSFORM_IDENTITY,
                  method === :shared_memory ? NUFFT_METHOD_SHARED_MEMORY : NUFFT_METHOD_GLOBAL_MEMORY,
                  Int32(-1),
                  zero3, Int32(0), Int32(0), Int32(0), zero3, Int32(0), NUFFT_SPREAD_AUTO, 0.0, Int32(sizeof(CParams)), Int32(0),
                  (0.0, 0.0, 0.0), pad3(d -> Int64(Ñs[d]), D, Int64(0)), Ptr{UInt8}(C_NULL))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nufft_plan_create_ex, libnufft), Cint, (Ref{Ptr{Cvoid}}, Ref{CParams}), h, Ref(probe)))
    ccall((:nufft_plan_destroy, libnufft), Cint, (Ptr{Cvoid},), h[])
    nothing
end

# The handle is built from the plan's own fields: the oversampled sizes and shape parameters the reference has already resolved
# go over verbatim (N_over, kernel_param_dim), so the library repeats nothing of src/plan.jl:485-506.  For real data only
# N₁÷2+1 wavenumbers survive in the plan (src/plan.jl:560): N₁ = 2(L - 1) is sent — with Ñ₁ and β given, nothing else depends on
# the parity of N₁ (the retained modes k = 0 … L - 1, their ϕ̂ and the oversampled grid are the same for 2L - 2 and 2L - 1).
function ensure_handle!(p::PlanNUFFT{Z, N, Nc, M, MI355XBackend}) where {Z, N, Nc, M}
    data = p.data
    data.handle == C_NULL || return data.handle
    check_library()
    fold = p.point_transform_fold                     # closure of generate_point_transform_fold_function, src/plan.jl:459-464
    pt = fold.point_transform                         # its captured `point_transform`: identity or _transform_point_convention
    pt === identity || pt === NonuniformFFTs._transform_point_convention ||
        throw(ArgumentError("MI355XBackend: point_transform must be identity or the AbstractNFFTs convention (closures cannot cross the C ABI); use ROCBackend()"))
    T = real(Z)
    Ls = map(length, data.ks)
    Ns = pad3(d -> Int64(Z <: Real && d == 1 ? max(1, 2 * (Ls[1] - 1)) : Ls[d]), N, Int64(0))
    Ñs = pad3(d -> Int64(Kernels.gridsize(p.kernels[d])), N, Int64(0))
    βs = pad3(d -> shape_param(p.kernels[d]), N, 0.0)
    dtype = T === Float64 ? NUFFT_F64 : NUFFT_F32
    evalmode = p.kernel_evalmode isa Direct ? NUFFT_EVAL_DIRECT : NUFFT_EVAL_FAST_APPROXIMATION
    ptrans = pt === identity ? NUFFT_POINT_TRANSFORM_IDENTITY : NUFFT_POINT_TRANSFORM_NFFT
    method = NonuniformFFTs.gpu_method(p.blocks) === :shared_memory ? NUFFT_METHOD_SHARED_MEMORY : NUFFT_METHOD_GLOBAL_MEMORY
    zero3 = (Int32(0), Int32(0), Int32(0))
    prm = CParams(dtype, Int32(Z <: Complex), Int32(N), Ns, Int32(M), Float64(p.σ),
                  kernel_id(first(p.kernels)), evalmode, Int32(Nc), Int32(p.fftshift), ptrans, method,
                  Int32(AMDGPU.device_id(AMDGPU.device()) - 1),
                  zero3, Int32(0), Int32(0), Int32(0), zero3, Int32(0), NUFFT_SPREAD_AUTO, 0.0, Int32(sizeof(CParams)), Int32(0), βs, Ñs,
                  Ptr{UInt8}(C_NULL))                 # (no development switches: the library reads no environment either)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nufft_plan_create_ex, libnufft), Cint, (Ref{Ptr{Cvoid}}, Ref{CParams}), h, Ref(prm)))
    data.handle = h[]
end

# ---- callback menu ---------------------------------------------------------------------------------------------------
struct PointWeights{V <: ROCVector} <: Function; w::V; end           # (v, n) -> v .* w[n]        (src/plan.jl:117-127)
struct ModeFactors{A <: ROCArray} <: Function; f::A; end             # (ŵ, idx) -> ŵ .* f[idx...]  (src/plan.jl:129-143)
(c::PointWeights)(v, n) = oftype(v, v .* c.w[n])
(c::ModeFactors)(w, idx) = oftype(w, w .* c.f[idx...])
struct CCallbacks; point_weights::Ptr{Cvoid}; mode_factors::Ptr{Cvoid}; end
cptr(c::PointWeights) = Ptr{Cvoid}(UInt(pointer(c.w)));  cptr(c::ModeFactors) = Ptr{Cvoid}(UInt(pointer(c.f)))
cptr(::typeof(default_callback)) = C_NULL
function ccallbacks(cb::NUFFTCallbacks)
    (cb.nonuniform isa Union{PointWeights, typeof(default_callback)} && cb.uniform isa Union{ModeFactors, typeof(default_callback)}) ||
        throw(ArgumentError("MI355XBackend: callbacks must be PointWeights / ModeFactors (closures cannot cross the C ABI); use ROCBackend()"))
    Ref(CCallbacks(cptr(cb.nonuniform), cptr(cb.uniform)))
end

stream_ptr() = reinterpret(Ptr{Cvoid}, AMDGPU.stream().stream)
ptrs(xs::NTuple{N, ROCArray}) where {N} = Ptr{Cvoid}[Ptr{Cvoid}(UInt(pointer(x))) for x in xs]
on_device(xs::Tuple) = all(x -> x isa ROCArray, xs) ||
    throw(ArgumentError("MI355XBackend: points, values and uniform arrays must be ROCArrays"))

# ---- set_points!  (src/set_points.jl:33-52 + set_points_impl!, src/blocking/gpu.jl:73-142) ----------------------------
# the reference's signature with the plan narrowed to this backend: strictly more specific; the matrix / vector-of-tuples /
# 1-D forms (src/set_points.jl:55-88) convert their argument and land here
function NonuniformFFTs.set_points!(p::PlanNUFFT{Z, N, Nc, M, MI355XBackend}, xp::NTuple{N, AbstractVector{T}};
                                    kwargs...) where {Z, N, Nc, M, T}
    T === real(Z) || throw(ArgumentError(lazy"input points must have the same accuracy as the created plan (got $T points for a $Z plan)"))
    P_in, P_plan = typeof(xp), eltype(p.points_ref)
    P_in === P_plan || throw(ArgumentError(lazy"""unexpected point container:
        - expected:  points::$P_plan
        - got:       points::$P_in"""))                # src/set_points.jl:36-44
    on_device(xp)
    Np = length(xp[1])
    all(x -> length(x) == Np, xp) || throw(DimensionMismatch("input points must have the same length along all dimensions"))   # src/blocking/gpu.jl:86
    p.points_ref[] = xp                                # the plan keeps the caller's arrays, as the reference does (:45)
    h = ensure_handle!(p)
    @timeit NonuniformFFTs.get_timer_nowarn(p) "Set points" begin
        GC.@preserve xp check(ccall((:nufft_set_points, libnufft), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}),
                                    h, Np, ptrs(xp), stream_ptr()))
        NonuniformFFTs.maybe_synchronise(p)
    end
    p
end

# ---- exec_type1! / exec_type2!  (src/NonuniformFFTs.jl:148-195, 237-291) ---------------------------------------------
# (again the reference's signatures with the plan narrowed; T is the plan's non-uniform element type, Z the uniform one)
# The stages are enqueued one by one through the stage-level entry points, each under the reference's own timer label and followed
# by maybe_synchronise(p) (src/NonuniformFFTs.jl:157-186, 246-283): p.timer shows the same tree as a ROCBackend plan, and with
# synchronise = true the same per-stage times.  nufft_exec_type1 is exactly this sequence (csrc/plan.cpp), so nothing is lost by
# not calling it.  The callback menu is put in force around the stages (nufft_set_callbacks) — the arguments the reference passes
# to spread_from_points! / copy_deconvolve_to_non_oversampled! / copy_deconvolve_to_oversampled! / interpolate!.
# (one literal ccall per stage: the (symbol, library) pair of a ccall must be a constant expression)
spread_deferred(h, a, s) = check(ccall((:nufft_spread_deferred, libnufft), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), h, a, s))
fft_forward(h, s) = check(ccall((:nufft_fft_forward, libnufft), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, s))
deconvolve_truncate(h, a, s) = check(ccall((:nufft_deconvolve_truncate, libnufft), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), h, a, s))
deconvolve_pad(h, a, s) = check(ccall((:nufft_deconvolve_pad, libnufft), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), h, a, s))
fft_backward(h, s) = check(ccall((:nufft_fft_backward, libnufft), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h, s))
interpolate_points(h, a, s) = check(ccall((:nufft_interpolate, libnufft), Cint, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), h, a, s))
set_callbacks(h, cb) = check(ccall((:nufft_set_callbacks, libnufft), Cint, (Ptr{Cvoid}, Ref{CCallbacks}), h, cb))
const no_callbacks = CCallbacks(C_NULL, C_NULL)

function NonuniformFFTs.exec_type1!(ûs_k::NTuple{C, AbstractArray{Z}}, p::PlanNUFFT{T, N, Nc, M, MI355XBackend},
                                    vp::NTuple{C, AbstractVector{T}};
                                    callbacks::NUFFTCallbacks = NUFFTCallbacks()) where {T, Z, C, N, Nc, M}
    Z === complex(T) || throw(ArgumentError(lazy"uniform data must have the same accuracy as the created plan (got $Z values for a $T plan)"))   # :154
    cb = ccallbacks(callbacks)
    timer = NonuniformFFTs.get_timer_nowarn(p)
    @timeit timer "Execute type 1" begin
        NonuniformFFTs.check_nufft_uniform_data(p, ûs_k)          # :92-103
        NonuniformFFTs.check_nufft_nonuniform_data(p, vp)         # :105-114
        on_device(ûs_k); on_device(vp)
        h, s = p.data.handle, stream_ptr()
        GC.@preserve ûs_k vp callbacks begin
            set_callbacks(h, cb)
            try
                @timeit timer "(0) Fill with zeros" begin      # (nothing to enqueue: the spreading kernels store every cell of `us` once)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(1) Spreading" begin
                    spread_deferred(h, ptrs(vp), s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(2) Forward FFT" begin
                    fft_forward(h, s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(3) Deconvolution" begin
                    deconvolve_truncate(h, ptrs(ûs_k), s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
            finally
                set_callbacks(h, Ref(no_callbacks))
            end
        end
    end
    ûs_k
end

function NonuniformFFTs.exec_type2!(vp::NTuple{C, AbstractVector{T}}, p::PlanNUFFT{T, N, Nc, M, MI355XBackend},
                                    ûs_k::NTuple{C, AbstractArray{Z}};
                                    callbacks::NUFFTCallbacks = NUFFTCallbacks()) where {T, Z, C, N, Nc, M}
    Z === complex(T) || throw(ArgumentError(lazy"uniform data must have the same accuracy as the created plan (got $Z values for a $T plan)"))   # :243
    cb = ccallbacks(callbacks)
    timer = NonuniformFFTs.get_timer_nowarn(p)
    @timeit timer "Execute type 2" begin
        NonuniformFFTs.check_nufft_uniform_data(p, ûs_k)
        NonuniformFFTs.check_nufft_nonuniform_data(p, vp)
        on_device(ûs_k); on_device(vp)
        h, s = p.data.handle, stream_ptr()
        GC.@preserve ûs_k vp callbacks begin
            set_callbacks(h, cb)
            try
                @timeit timer "(0) Fill with zeros" begin      # (fused into the next stage: every element of the spectrum is written once)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(1) Deconvolution" begin
                    deconvolve_pad(h, ptrs(ûs_k), s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(2) Backward FFT" begin
                    fft_backward(h, s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
                @timeit timer "(3) Interpolation" begin
                    interpolate_points(h, ptrs(vp), s)
                    NonuniformFFTs.maybe_synchronise(p)
                end
            finally
                set_callbacks(h, Ref(no_callbacks))
            end
        end
    end
    vp
end
# the single-array forms (src/NonuniformFFTs.jl:193-196, 288-291) wrap their arguments in 1-tuples and land here; a plan without
# points has a C_NULL handle: check_nufft_nonuniform_data has then thrown already unless Np = 0, and the library answers
# NUFFT_ERR_INVALID_ARG ("null plan") for that case.

# ---- AbstractNFFTs entry (src/abstractNFFTs.jl:198-247) ---------------------------------------------------------------
# NFFTPlan(xp, Ns; …) takes the backend from the array (KA.get_backend(xp), :214) and refuses any other (:218), so a ROCArray of
# nodes always lands on the stock kernels.  The same constructor with the backend in front:
function NonuniformFFTs.NFFTPlan(backend::MI355XBackend, xp::AbstractMatrix{T}, Ns::Dims;
                                 fftflags = nothing, blocking = true, sortNodes = false,
                                 window = NonuniformFFTs.default_kernel(backend), fftshift = true, precompute = nothing,
                                 kws...) where {T <: AbstractFloat}
    isnothing(precompute) || @warn "Precompute flags are not supported by the NonuniformFFTs backend and will be ignored."
    kws_plan, kws_accuracy = NonuniformFFTs._split_accuracy_params(; kws...)
    m_actual, σ_actual, reltol_actual = AbstractNFFTs.accuracyParams(; kws_accuracy...)
    sort_points = sortNodes ? True() : False()
    block_size = blocking ? NonuniformFFTs.default_block_size(Ns, backend) : nothing
    kernel = window isa AbstractKernel ? window : NonuniformFFTs.convert_window_function(window, backend)
    p = PlanNUFFT(Complex{T}, Ns, HalfSupport(m_actual); backend, σ = T(σ_actual), sort_points, fftshift, block_size, kernel,
                  point_transform = NonuniformFFTs._transform_point_convention, kws_plan...)
    pp = NonuniformFFTs.NFFTPlan(p)
    AbstractNFFTs.nodes!(pp, xp)                       # -> set_points!(p, xp::AbstractMatrix) (:163-165) -> the method above
    pp
end

# plan_nfft through AbstractNFFTs' backend selection: with(nfft_backend => NonuniformFFTsMI355XBackend()) do … end
struct NonuniformFFTsMI355XBackend <: AbstractNFFTs.AbstractNFFTBackend end
function AbstractNFFTs.plan_nfft(::NonuniformFFTsMI355XBackend, ::Type{Q}, xp::AbstractMatrix{T}, Ns::Dims{D};
                                 kwargs...) where {Q, T, D}
    NonuniformFFTs.NFFTPlan(MI355XBackend(), adapt(Q, xp), Ns; kwargs...)      # src/abstractNFFTs.jl:240-247
end

end # module
