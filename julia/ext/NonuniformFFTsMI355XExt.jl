# NonuniformFFTsMI355XExt.jl — package extension that routes a NonuniformFFTs.jl plan to libnufft_mi355x.so.
#
# Copy to NonuniformFFTs.jl/ext/ and declare it in Project.toml ([weakdeps] AMDGPU, [extensions]
# NonuniformFFTsMI355XExt = "AMDGPU"); see INTEGRATION.md for the dispatch points it uses and why nothing inside src/
# is edited.  NEVER EXECUTED in the build image (no Julia runtime): tests/test_julia_shim_static.py checks every ccall
# (symbol, return type, argument types) and the mirrored structs against include/nufft_mi355x.h, and every
# NonuniformFFTs function it overloads against the reference source (name and positional arity).
module NonuniformFFTsMI355XExt

using NonuniformFFTs
using NonuniformFFTs: PlanNUFFT, NUFFTCallbacks, default_callback, Kernels, AbstractBlockData, AbstractNUFFTData,
                      HalfSupport, StaticBool, Direct, get_timer_nowarn, maybe_synchronise
using TimerOutputs: @timeit
using AMDGPU
using KernelAbstractions: KernelAbstractions as KA

const libnufft = "libnufft_mi355x.so"      # nonuniformffts.jl_amd/libnufft_mi355x.so

# ---- backend value: a KA.GPU of its own, forwarding allocation to ROCBackend ----------------------------------------
struct MI355XBackend <: KA.GPU
    roc::ROCBackend
    last_Ns::Base.RefValue{Any}            # written by default_block_size, read by BlockDataGPU (same _PlanNUFFT call)
end
MI355XBackend() = MI355XBackend(ROCBackend(), Ref{Any}(nothing))
KA.allocate(b::MI355XBackend, args...) = KA.allocate(b.roc, args...)
KA.zeros(b::MI355XBackend, args...) = KA.zeros(b.roc, args...)
KA.synchronize(b::MI355XBackend) = KA.synchronize(b.roc)

NonuniformFFTs.default_kernel(::MI355XBackend) = BackwardsKaiserBesselKernel()   # ext/NonuniformFFTsAMDGPUExt.jl:54
NonuniformFFTs.default_kernel_evalmode(::MI355XBackend) = Direct()               # ext/NonuniformFFTsAMDGPUExt.jl:56
function NonuniformFFTs.default_block_size(Ns::Dims, b::MI355XBackend)
    b.last_Ns[] = Ns
    NonuniformFFTs.default_block_size(Ns, b.roc)                                 # src/NonuniformFFTs.jl:59-63
end

lasterr() = unsafe_string(ccall((:nufft_last_error_message, libnufft), Cstring, ()))
check(rc::Cint) = rc == 0 ? nothing :
    rc in (1, 2, 4, 5, 6, 10) ? throw(ArgumentError(lasterr())) :
    rc == 3 ? throw(DimensionMismatch(lasterr())) : error(lasterr())

# ---- plan-owned state ------------------------------------------------------------------------------------------------
mutable struct MI355XBlockData{D} <: AbstractBlockData
    handle::Ptr{Cvoid}                     # C_NULL until the first set_points!
    Ns::Union{Nothing, Dims{D}}            # from default_block_size; nothing if the caller passed block_size itself
    Ñs::Dims{D}
    method::Symbol
    sort_points::StaticBool
end
function NonuniformFFTs.BlockDataGPU(::Type{Z}, b::MI355XBackend, block_dims::Dims{D}, Ñs::Dims{D}, ::HalfSupport,
                                     sort_points::StaticBool; method::Symbol, batch_size::Val) where {Z <: Number, D}
    method ∈ (:global_memory, :shared_memory) || throw(ArgumentError("expected gpu_method ∈ (:global_memory, :shared_memory)"))   # src/blocking/gpu.jl:26
    Ns = b.last_Ns[] isa Dims{D} ? b.last_Ns[] : nothing
    b.last_Ns[] = nothing
    bd = MI355XBlockData{D}(C_NULL, Ns, Ñs, method, sort_points)
    finalizer(o -> o.handle == C_NULL || ccall((:nufft_plan_destroy, libnufft), Cint, (Ptr{Cvoid},), o.handle), bd)
    bd
end
NonuniformFFTs.gpu_method(bd::MI355XBlockData) = bd.method                 # what show(::PlanNUFFT) asks, src/plan.jl:380-388
NonuniformFFTs.with_blocking(::MI355XBlockData) = true
NonuniformFFTs.get_block_dims(::MI355XBlockData) = nothing
NonuniformFFTs.get_sort_points(bd::MI355XBlockData) = bd.sort_points
NonuniformFFTs.get_batch_size(::MI355XBlockData) = 0

struct MI355XData{Z, N, Nc, W, A} <: AbstractNUFFTData{Z, N, Nc}
    ks::W
    stub::A                                 # 1-element ROCArray{complex(real(Z)), N}: output_field for _PlanNUFFT's index_map lines
end
NonuniformFFTs.output_field(d::MI355XData) = d.stub
function NonuniformFFTs.init_plan_data(::Type{Z}, b::MI355XBackend, Ñs::Dims{N}, ks::NTuple, ::Val{Nc}; plan_kwargs) where {Z <: Number, N, Nc}
    stub = KA.zeros(b, complex(real(Z)), ntuple(_ -> 1, Val(N)))
    MI355XData{Z, N, Nc, typeof(ks), typeof(stub)}(ks, (stub,))
end
# (the stock index_map of such a plan is non_oversampled_indices!(…, axes(stub)…): unused — the handle builds its own.)

kernel_id(::BackwardsKaiserBesselKernel) = 0     # NUFFT_KERNEL_* of include/nufft_mi355x.h
kernel_id(::KaiserBesselKernel) = 1
kernel_id(::GaussianKernel) = 2
kernel_id(::BSplineKernel) = 3

# nufft_params mirrored field by field (include/nufft_mi355x.h); nufft_sizeof_params() guards the layout
struct CParams
    dtype::Int32; is_complex::Int32; ndim::Int32; N::NTuple{3, Int64}; half_support::Int32; sigma::Float64
    kernel::Int32; evalmode::Int32; ntransforms::Int32; fftshift::Int32; point_transform::Int32; gpu_method::Int32
    device::Int32; tile_dims::NTuple{3, Int32}; lds_budget_bytes::Int32; spread_threads::Int32; interp_threads::Int32
    interp_tile_dims::NTuple{3, Int32}; bin_log2::Int32; spread_method::Int32; kernel_param::Float64; reserved::NTuple{2, Int32}
end

# Non-oversampled sizes of the plan.  Complex plans: length.(ks).  Real plans: N₁ ∈ {2L - 2, 2L - 1} with L = length(ks[1]);
# recorded by default_block_size, else the candidate whose oversampled size reproduces Ñ₁ (src/plan.jl:491-494).
function plan_Ns(p::PlanNUFFT{Z, N}, σ_wanted) where {Z, N}
    bd = p.blocks
    bd.Ns === nothing || return bd.Ns
    Ls = map(length, p.data.ks)
    Z <: Complex && return Ls
    L = Ls[1]
    cands = filter(n -> 2 * nextprod((2, 3, 5), floor(Int, σ_wanted * ((n + 1) ÷ 2))) == bd.Ñs[1], (2L - 2, 2L - 1))
    isempty(cands) && throw(ArgumentError("cannot recover N₁ of a real plan created with an explicit block_size; omit block_size"))
    (last(cands), Base.tail(Ls)...)
end

function ensure_handle!(p::PlanNUFFT{Z, N, Nc, M}) where {Z, N, Nc, M}
    bd = p.blocks
    bd.handle == C_NULL || return bd.handle
    ccall((:nufft_sizeof_params, libnufft), Int64, ()) == sizeof(CParams) || error("nufft_params layout differs from the library's")
    fold = p.point_transform_fold                     # generate_point_transform_fold_function(point_transform, backend), src/plan.jl:459-464
    pt = fold.point_transform                         # closure field: identity or _transform_point_convention
    pt === identity || pt === NonuniformFFTs._transform_point_convention ||
        throw(ArgumentError("MI355XBackend: point_transform must be identity or the AbstractNFFTs convention (closures cannot cross the C ABI); use ROCBackend()"))
    kern = Kernels.kernel(first(p.kernels))          # the AbstractKernel the data was built from (β / ℓ explicit or nothing)
    σ = Float64(p.σ)                                  # actual σ = max(Ñ ./ N): reproduces Ñ (nextprod is idempotent on 2-3-5 numbers)
    Ns = plan_Ns(p, σ)
    T = real(Z)
    prm = CParams(T === Float64 ? 1 : 0, Z <: Complex, N, ntuple(d -> d ≤ N ? Int64(Ns[d]) : Int64(0), 3), M, σ,
                  kernel_id(kern), p.kernel_evalmode isa Direct ? 0 : 1, Nc, p.fftshift, pt === identity ? 0 : 1,
                  bd.method === :shared_memory ? 1 : 0, AMDGPU.device_id(AMDGPU.device()) - 1,
                  (0, 0, 0), 0, 0, 0, (0, 0, 0), 0, 0, something(Kernels.shape_parameter(kern), 0.0), (0, 0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nufft_plan_create_ex, libnufft), Cint, (Ref{Ptr{Cvoid}}, Ref{CParams}), h, Ref(prm)))
    bd.handle = h[]
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

stream_ptr() = Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream().stream)
ptrs(xs::NTuple{N, ROCArray}) where {N} = Ptr{Cvoid}[Ptr{Cvoid}(UInt(pointer(x))) for x in xs]
const MIPlan{Z, N, Nc, M} = PlanNUFFT{Z, N, Nc, M, MI355XBackend}

# ---- set_points!  (src/set_points.jl:33-52 + set_points_impl!, src/blocking/gpu.jl:73-142) ----------------------------
function NonuniformFFTs.set_points!(p::MIPlan{Z, N}, xp::NTuple{N, ROCVector{T}}; kwargs...) where {Z, N, T}
    T === real(Z) || throw(ArgumentError(lazy"input points must have the same accuracy as the created plan (got $T points for a $Z plan)"))
    Np = length(xp[1])
    all(x -> length(x) == Np, xp) || throw(DimensionMismatch("input points must have the same length along all dimensions"))   # src/blocking/gpu.jl:86
    p.points_ref[] = xp                                # the plan keeps the caller's arrays, as the reference does (:45)
    h = ensure_handle!(p)
    @timeit get_timer_nowarn(p) "Set points" begin
        GC.@preserve xp check(ccall((:nufft_set_points, libnufft), Cint, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}),
                                    h, Np, ptrs(xp), stream_ptr()))
        maybe_synchronise(p)
    end
    p
end

# ---- exec_type1! / exec_type2!  (src/NonuniformFFTs.jl:148-195, 237-291) ---------------------------------------------
function NonuniformFFTs.exec_type1!(ûs_k::NTuple{C, ROCArray{<:Complex}}, p::MIPlan{Z, N, C}, vp::NTuple{C, ROCVector{Z}};
                                    callbacks::NUFFTCallbacks = NUFFTCallbacks()) where {Z, N, C}
    eltype(first(ûs_k)) === complex(Z) || throw(ArgumentError("uniform data must have the same accuracy as the created plan"))   # :154
    NonuniformFFTs.check_nufft_uniform_data(p, ûs_k)          # :92-103
    NonuniformFFTs.check_nufft_nonuniform_data(p, vp)         # :105-114
    cb = ccallbacks(callbacks)
    @timeit get_timer_nowarn(p) "Execute type 1" begin
        GC.@preserve ûs_k vp callbacks check(ccall((:nufft_exec_type1_cb, libnufft), Cint,
            (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ref{CCallbacks}, Ptr{Cvoid}),
            p.blocks.handle, ptrs(ûs_k), ptrs(vp), cb, stream_ptr()))
        maybe_synchronise(p)
    end
    ûs_k
end

function NonuniformFFTs.exec_type2!(vp::NTuple{C, ROCVector{Z}}, p::MIPlan{Z, N, C}, ûs_k::NTuple{C, ROCArray{<:Complex}};
                                    callbacks::NUFFTCallbacks = NUFFTCallbacks()) where {Z, N, C}
    eltype(first(ûs_k)) === complex(Z) || throw(ArgumentError("uniform data must have the same accuracy as the created plan"))   # :243
    NonuniformFFTs.check_nufft_uniform_data(p, ûs_k)
    NonuniformFFTs.check_nufft_nonuniform_data(p, vp)
    cb = ccallbacks(callbacks)
    @timeit get_timer_nowarn(p) "Execute type 2" begin
        GC.@preserve ûs_k vp callbacks check(ccall((:nufft_exec_type2_cb, libnufft), Cint,
            (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ref{CCallbacks}, Ptr{Cvoid}),
            p.blocks.handle, ptrs(vp), ptrs(ûs_k), cb, stream_ptr()))
        maybe_synchronise(p)
    end
    vp
end
# the single-array forms (exec_type1!(ûs::AbstractArray, p, vp::AbstractVector), :125) wrap their arguments in 1-tuples and
# land here unchanged.

end # module
