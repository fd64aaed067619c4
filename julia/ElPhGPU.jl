# ElPhGPU.jl — the Julia side of the drop-in: GPU-backed `<: AbstractModel` wrappers for ElPhDynamics.jl whose operator API
# (`mul!`, `mulM!`, `mulMᵀ!`, `mulMᵀM!`, `mulMMᵀ!`, `ldiv!` in both arities, `update_model!`, `transpose!`), KPM preconditioner
# (`setup!`, `ldiv!(z, P, r)`) and `fourier_accelerate!` are thin `ccall`s into libelphgpu.so (include/elph_gpu.h).
#
# STATUS: NOT EXECUTED.  Julia is installed neither in the build image nor on the GPU box of this project, so this file has never
# been parsed by a Julia compiler.  It is written against the reference's sources as they lie in src/ (file:line cited at every
# method) and against the C ABI that IS tested: tests/abi_c/abi_smoke.c makes exactly these calls, in this order, from plain C
# (gcc, no Python in the process), and tests/test_gpu_parity.py drives the same entry points through ctypes against the oracle.
#
# How a maintainer wires it in (three lines in the reference, nothing else changes):
#   src/ElPhDynamics.jl, after `include("Models.jl")` … `include("FourierAcceleration.jl")`:   include("ElPhGPU.jl")
#   src/ProcessInputFile.jl:216-326 (`initialize_model`), last line:   model = ElPhGPU.gpu(model)      # when the deck asks for it
#   src/ProcessInputFile.jl:473-513 (`initialize_preconditioner`):     P = ElPhGPU.GPUKPMPreconditioner(model, n, buf, c1, c2)
# `HMC.calc_O⁻¹Λϕ!` (HMC.jl:820-915), `GreensFunctions.update!` (:201-234), `LangevinDynamics.calc_dSfdx!` (:350-384) and
# `SpecialUpdates` keep calling `update_model!`, `ldiv!`, `mulMᵀ!`, `setup!` — by dispatch they now land here.
#
# Layouts are the reference's own (Utilities.jl:12-15: `Vector{Float64}` of length Ndim, τ fastest; `Matrix{Int}` 2 × Nbonds
# column-major, 1-based; SSH `cosht/sinht` (Lτ × Nbonds) column-major): Julia arrays are passed as they are.

module ElPhGPU

using LinearAlgebra
using Random
using Logging

import LinearAlgebra: mul!, ldiv!, transpose!

using ..Models: AbstractModel, HolsteinModel, SSHModel
import ..Models: mulM!, mulMᵀ!, mulMᵀM!, mulMMᵀ!, update_model!
using ..KPMPreconditioners: KPMPreconditioner
import ..KPMPreconditioners: setup!
using ..FourierAcceleration: FourierAccelerator
import ..FourierAcceleration: fourier_accelerate!

export GPUHolsteinModel, GPUSSHModel, GPUKPMPreconditioner, gpu, ldiv_batched!, ElphError

"Path of the shared library; `ENV[\"ELPHGPU_LIB\"]` overrides (the in-tree build is elphdynamics_amd/libelphgpu.so)."
const lib = get(ENV, "ELPHGPU_LIB", "libelphgpu.so")

const ELPH_ABI = 1                      # include/elph_gpu.h: elph_abi_version()

struct ElphError <: Exception
    code::Cint                          # ELPH_E_ARG -1, _HIP -2, _STATE -3, _NOGPU -4, _UNSUPPORTED -5
    msg::String
end
Base.showerror(io::IO, e::ElphError) = print(io, "libelphgpu error ", e.code, ": ", e.msg)

"Status check of every call: the library never throws across the boundary, it returns a code and keeps a message."
function chk(rc::Cint)
    rc == 0 && return nothing
    throw(ElphError(rc, unsafe_string(ccall((:elph_last_error, lib), Cstring, ()))))
end

"What the loaded library was built from (source hash, compiler, time): `elph_build_info`."
build_info() = unsafe_string(ccall((:elph_build_info, lib), Cstring, ()))

# ------------------------------------------------------------------------------------------------------------------------------
# Model wrappers.  A wrapper OWNS nothing but the device handle; every host-visible field the callers reach into (SURVEY §8b:
# x, rng, solver.tol/.maxiter, mul_by_M, transposed, Ndof, Ndim, Nph, Nsites, Lτ, Δτ, λ, λ₂, μ, v″, lattice, neighbor_table, …)
# stays in the reference's own struct `host` and is forwarded by getproperty / setproperty!.
# ------------------------------------------------------------------------------------------------------------------------------

abstract type GPUModel{T1,T2,T3,T4} <: AbstractModel{T1,T2,T3,T4} end

"Holstein model whose fermion matrix lives on an MI355X (HolsteinModels.jl:22-314 keeps parameters, x, rng, scratch)."
mutable struct GPUHolsteinModel{T1,T2,T3,T4} <: GPUModel{T1,T2,T3,T4}
    host::HolsteinModel{T1,T2,T3,T4}
    handle::Ptr{Cvoid}
end

"SSH (bond-phonon) model on the device (SSHModels.jl:79-314)."
mutable struct GPUSSHModel{T1,T2,T3,T4} <: GPUModel{T1,T2,T3,T4}
    host::SSHModel{T1,T2,T3,T4}
    handle::Ptr{Cvoid}
    cb_index::Vector{Int64}             # checkerboard_perm[phonon_to_bond[p]]  (1-based position of each phonon's bond)
    t_ph::Vector{Float64}               # t[phonon_to_bond[p]]
    t_bare_cb::Vector{Float64}          # bare hopping of every bond, checkerboard order
end

Base.getproperty(g::GPUModel, s::Symbol) = hasfield(typeof(g), s) ? getfield(g, s) : getproperty(getfield(g, :host), s)
Base.setproperty!(g::GPUModel, s::Symbol, v) = hasfield(typeof(g), s) ? setfield!(g, s, v) : setproperty!(getfield(g, :host), s, v)
Base.propertynames(g::GPUModel) = (fieldnames(typeof(g))..., propertynames(getfield(g, :host))...)

function create_handle(kind::Integer, N::Integer, Lτ::Integer, Nbonds::Integer, neighbor_table, cosht, sinht, device::Integer)
    v = ccall((:elph_abi_version, lib), Cint, ())
    v == ELPH_ABI || error("libelphgpu ABI version $v, this wrapper was written for $ELPH_ABI")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall((:elph_create, lib), Cint,
              (Ref{Ptr{Cvoid}}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Cint),
              h, kind, N, Lτ, Nbonds, neighbor_table, cosht, sinht, device))
    return h[]
end

destroy!(g::GPUModel) = (getfield(g, :handle) == C_NULL || ccall((:elph_destroy, lib), Cint, (Ptr{Cvoid},), getfield(g, :handle)); setfield!(g, :handle, C_NULL); nothing)

"""
    GPUHolsteinModel(m::HolsteinModel; device=0)

`m` must be initialised (`initialize_model!`, HolsteinModels.jl:484-517: `neighbor_table`, `cosht`, `sinht` in checkerboard
order).  Errors with ELPH_E_NOGPU when no gfx950 device is visible: there is no CPU fallback.
"""
function GPUHolsteinModel(m::HolsteinModel{T1,T2,T3,T4}; device::Integer=0) where {T1,T2,T3,T4}
    nt = m.Nbonds > 0 ? m.neighbor_table : zeros(Int64, 2, 0)
    h = create_handle(0, m.Nsites, m.Lτ, m.Nbonds, nt, m.cosht, m.sinht, device)
    g = GPUHolsteinModel{T1,T2,T3,T4}(m, h)
    finalizer(destroy!, g)
    push_solver!(g)
    update_model!(g)
    return g
end

"""
    GPUSSHModel(m::SSHModel; device=0)

`cosht/sinht` are computed on the device from `m.x` at every `update_model!` (no host cosh/sinh, no tables over PCIe).
"""
function GPUSSHModel(m::SSHModel{T1,T2,T3,T4}; device::Integer=0) where {T1,T2,T3,T4}
    h = create_handle(1, m.Nsites, m.Lτ, m.Nbonds, m.neighbor_table, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), device)
    cb_index  = Int64[m.checkerboard_perm[m.phonon_to_bond[p]] for p in 1:m.Nph]
    t_ph      = Float64[m.t[m.phonon_to_bond[p]] for p in 1:m.Nph]
    t_bare_cb = zeros(Float64, m.Nbonds)
    for bond in 1:m.Nbonds
        t_bare_cb[m.checkerboard_perm[bond]] = m.t[bond]
    end
    g = GPUSSHModel{T1,T2,T3,T4}(m, h, cb_index, t_ph, t_bare_cb)
    finalizer(destroy!, g)
    push_solver!(g)
    update_model!(g)
    return g
end

"Wrap whatever `initialize_model` built."
gpu(m::HolsteinModel; device::Integer=0) = GPUHolsteinModel(m, device=device)
gpu(m::SSHModel; device::Integer=0) = GPUSSHModel(m, device=device)

# ---- update_model! ------------------------------------------------------------------------------------------------------------

"update_model!(holstein): expnΔτV = exp(-Δτ(λx + λ₂x² - μ)) on the device — HolsteinModels.jl:526-549"
function update_model!(g::GPUHolsteinModel)
    m = g.host
    chk(ccall((:elph_update_model_holstein, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
              g.handle, m.x, m.λ, m.λ₂, m.μ, m.Δτ))
    return nothing
end

"update_model!(ssh): expΔτμ, t′ = t − (αx + sign(x)α₂x²), cosh/sinh(Δτ t′) per (τ, bond) on the device — SSHModels.jl:510-562.
The equality test of symmetry-equivalent fields (:548-559) stays on the host, as in the reference."
function update_model!(g::GPUSSHModel)
    m = g.host
    for field in 1:m.Ndof
        field′ = m.primary_field[field]
        if field != field′ && !(m.x[field] ≈ m.x[field′])
            error("(x[$field]=$(m.x[field])) != (x[$field′]=$(m.x[field′]))\n")
        end
    end
    chk(ccall((:elph_update_model_ssh_fields, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
              g.handle, m.x, m.Nph, g.cb_index, g.t_ph, m.α, m.α₂, g.t_bare_cb, m.μ, m.Δτ))
    return nothing
end

"Bring `model.cosht` / `model.sinht` of the host struct up to date after a device-side update (readers: KPM diagnostics, dumps)."
function pull_cosh_sinh!(g::GPUModel)
    chk(ccall((:elph_get_cosh_sinh, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), g.handle, g.host.cosht, g.host.sinht))
    return nothing
end

# ---- mul! family — Models.jl:192-248, HolsteinModels.jl:569-684, SSHModels.jl:581-701 ------------------------------------------

for (jl, c) in ((:mulM!, :elph_mulM), (:mulMᵀ!, :elph_mulMT), (:mulMᵀM!, :elph_mulMTM), (:mulMMᵀ!, :elph_mulMMT))
    @eval function $jl(y::AbstractVector{Float64}, g::GPUModel, v::AbstractVector{Float64})
        @assert length(y) == g.Ndim && length(v) == g.Ndim
        chk(ccall(($(QuoteNode(c)), lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), g.handle, y, v))
        return nothing
    end
end

"mul!(y, model, v): honours model.mul_by_M and model.transposed — Models.jl:192-209"
function mul!(y::AbstractVector{Float64}, g::GPUModel, v::AbstractVector{Float64})
    if g.mul_by_M
        g.transposed ? mulMᵀ!(y, g, v) : mulM!(y, g, v)
    else
        g.transposed ? mulMMᵀ!(y, g, v) : mulMᵀM!(y, g, v)
    end
    return nothing
end

"transpose!(model): M ⇆ Mᵀ for mul! — Models.jl:244-248"
transpose!(g::GPUModel) = (g.host.transposed = !g.host.transposed; nothing)

# ---- ldiv! — Models.jl:74-137 (with P), :139-186 (without) ----------------------------------------------------------------------

"callers mutate model.solver.tol between solves (HMC.jl:827-828, restored :912): the current values travel with every solve"
push_solver!(g::GPUModel) = chk(ccall((:elph_solver_set, lib), Cint, (Ptr{Cvoid}, Float64, Int64, Float64),
                                      g.handle, g.solver.tol, g.solver.maxiter, g.solver.κmax))

function log_flag(flag::Integer, err, iters, with_P::Bool)
    # the reference's @info lines — Models.jl:108,117 (with preconditioner), :163,172 (without)
    suffix = with_P ? ", W/ Preconditioner" : ""
    flag == 1 && @info("Hit Max Iters, Residual Error = $err, Iterations = $iters$suffix")
    flag == 2 && @info("Large Residual Error = $err, Iterations = $iters$suffix")
    if flag > 0
        logger = global_logger()
        hasproperty(logger, :stream) && flush(logger.stream)
    end
    return nothing
end

"""
    ldiv!(x, model, b; maxiter=0) -> (iters, residual_error, flag)                       Models.jl:139-186
    ldiv!(x, model, b, P; maxiter=0) -> (iters, residual_error, flag)                    Models.jl:74-137

Solve MᵀM⋅x = b from the caller's `x` (callers pass zeros: HMC.jl:854) with the reference's stop rule, true-residual check, flags
0 / 1 (hit maxiter) / 2 (false convergence), zero-fill of `x` when flag > 0 and — with a preconditioner — the un-preconditioned
retry with 10·maxiter.  `P == I` takes the first form, as Models.jl:83-86.  `model.mul_by_M`/`transposed` select what is solved
in the reference (M, Mᵀ, MᵀM, MMᵀ: `mul!`); every caller on the path solves MᵀM (HMC.jl:851-886, GreensFunctions.jl:225,
LangevinDynamics.jl:374) and that is what the library solves — anything else is refused here rather than answered wrongly.
"""
function ldiv!(x::AbstractVector{Float64}, g::GPUModel, b::AbstractVector{Float64}; maxiter::Int=0)::Tuple{Int,Float64,Int}
    return _ldiv!(x, g, b, false, maxiter)
end

function ldiv!(x::AbstractVector{Float64}, g::GPUModel, b::AbstractVector{Float64}, P; maxiter::Int=0)::Tuple{Int,Float64,Int}
    P == I && return _ldiv!(x, g, b, false, maxiter)
    P isa GPUKPMPreconditioner || error("ldiv!(x, ::GPUModel, b, P): P must be I or a GPUKPMPreconditioner of this model")
    P.model === g || error("the preconditioner belongs to another model")
    return _ldiv!(x, g, b, true, maxiter)
end

function _ldiv!(x, g::GPUModel, b, use_P::Bool, maxiter::Int)
    (g.mul_by_M || g.transposed) && error("libelphgpu solves MᵀM⋅x = b (mul_by_M = false, transposed = false)")
    push_solver!(g)
    iters = Ref{Int64}(0); err = Ref{Float64}(0.0); flag = Ref{Cint}(0)
    chk(ccall((:elph_ldiv, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Int64, Ref{Int64}, Ref{Float64}, Ref{Cint}),
              g.handle, x, b, use_P ? 1 : 0, maxiter, iters, err, flag))
    log_flag(flag[], err[], iters[], use_P)
    return Int(iters[]), err[], Int(flag[])
end

"""
    ldiv_batched!(X, model, B[, P]; maxiter=0) -> (iters[], residual_error[], flag[])

`size(B, 2)` right-hand sides in one call (columns): the two pseudofermion solves of `calc_O⁻¹Λϕ!` (HMC.jl:851-886), the nᵥ solves
of `GreensFunctions.update!` (:201-234).  Every column follows exactly the single-solve recurrences and stop rule.
"""
function ldiv_batched!(X::AbstractMatrix{Float64}, g::GPUModel, B::AbstractMatrix{Float64}, P=I; maxiter::Int=0)
    push_solver!(g)
    n = size(B, 2)
    iters = zeros(Int64, n); err = zeros(Float64, n); flag = zeros(Cint, n)
    chk(ccall((:elph_ldiv_batched, lib), Cint,
              (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cint, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Cint}),
              g.handle, n, X, B, P == I ? 0 : 1, maxiter, iters, err, flag))
    for k in 1:n
        log_flag(flag[k], err[k], iters[k], !(P == I))
    end
    return iters, err, Int.(flag)
end

"solve!(x, A, b, cg[, P]) without ldiv!'s wrapper — IterativeSolvers.jl:153-234, 239-314 — returns the iteration count"
function solve!(x::AbstractVector{Float64}, g::GPUModel, b::AbstractVector{Float64}, solver, P=I;
                maxiter::Int=0, tol::Float64=0.0, κmax::Float64=0.0)::Int
    iters = Ref{Int64}(0)
    chk(ccall((:elph_cg_solve, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Int64, Float64, Cint, Ref{Int64}, Ptr{Float64}),
              g.handle, x, b, iszero(tol) ? solver.tol : tol, iszero(maxiter) ? solver.maxiter : maxiter,
              iszero(κmax) ? solver.κmax : κmax, P == I ? 0 : 1, iters, C_NULL))
    return Int(iters[])
end

# ---- KPM preconditioner — KPMPreconditioners.jl:219-235 (type), :259-321 (setup!), :426-481 (apply) ------------------------------

"Tag type: the expansion (Ē, Arnoldi bounds, orders, Chebyshev coefficients) lives inside the model's device handle."
mutable struct GPUKPMPreconditioner{T<:GPUModel}
    model::T
    active::Bool
    λ_lo::Float64
    λ_hi::Float64
    transposed::Bool
end

"GPUKPMPreconditioner(model, n, buf, c1, c2): arguments of SymmetricKPMPreconditioner (KPMPreconditioners.jl:224)"
function GPUKPMPreconditioner(g::GPUModel, n::Int, buf::Float64, c1::Float64, c2::Float64)
    chk(ccall((:elph_kpm_create, lib), Cint, (Ptr{Cvoid}, Cint, Float64, Float64, Float64), g.handle, n, buf, c1, c2))
    return GPUKPMPreconditioner(g, false, 0.0, 2.0, false)
end

"setup!(P): τ-average Ē, Arnoldi eigenvalue bounds from two start vectors drawn from model.rng (KPMPreconditioners.jl:859-861,
902-904), acceptance test, orders and coefficients — :259-321.  Call after update_model!, as HMC.jl:834 does."
function setup!(P::GPUKPMPreconditioner)
    g = P.model
    N = g.Nsites
    b_max = randn(g.rng, N)
    b_min = randn(g.rng, N)
    act = Ref{Cint}(0); lo = Ref{Float64}(0.0); hi = Ref{Float64}(0.0)
    chk(ccall((:elph_kpm_setup, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ref{Cint}, Ref{Float64}, Ref{Float64}),
              g.handle, b_max, b_min, NaN, NaN, act, lo, hi))
    P.active = act[] != 0
    P.λ_lo = lo[]; P.λ_hi = hi[]
    return nothing
end

"ldiv!(z, P, r): z = P⁻¹ r (twisted τ-FFT, per-ω Chebyshev series, inverse) — KPMPreconditioners.jl:426-481; a copy when inactive"
function ldiv!(z::AbstractVector{Float64}, P::GPUKPMPreconditioner, r::AbstractVector{Float64})
    chk(ccall((:elph_kpm_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), P.model.handle, z, r))
    return nothing
end

Base.:(==)(P::GPUKPMPreconditioner, ::UniformScaling) = false
Base.:(==)(::UniformScaling, P::GPUKPMPreconditioner) = false

# ---- Fourier acceleration — FourierAcceleration.jl:91-143 ------------------------------------------------------------------------

"""
    fourier_accelerate!(v′, fa, model, v, power; use_mass=false)

v′ = iFFT_τ( D^power ∘ FFT_τ(v) ), D = fa.M (use_mass) or fa.Q, real in, real out (FourierAcceleration.jl:128-135) on the device
of `model`.  The reference's method has no model argument (`fourier_accelerate!(v′, fa, v, power)`); HMC.jl:386,656,715 and
LangevinDynamics.jl pass `fa` built from the same model, so the call sites gain the one argument — or keep FFTW: both give the
same numbers to 1e-13.
"""
function fourier_accelerate!(v′::AbstractVector{Float64}, fa::FourierAccelerator{Float64}, g::GPUModel, v::AbstractVector{Float64},
                             power::Float64; use_mass::Bool=false)
    chk(ccall((:elph_fourier_accelerate, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Int64),
              g.handle, v′, v, use_mass ? fa.M : fa.Q, power, fa.N))
    return nothing
end

"in-place form — FourierAcceleration.jl:137-141"
fourier_accelerate!(v::AbstractVector{Float64}, fa::FourierAccelerator{Float64}, g::GPUModel, power::Float64; use_mass::Bool=false) =
    fourier_accelerate!(v, fa, g, v, power, use_mass=use_mass)

# ---- twisted transforms — TimeFreqFFTs.jl:55-73, 112-130 -------------------------------------------------------------------------

"τ_to_ω!(ν, model, v): ν = FFT_τ(Θ ∘ v), Θ_τ = exp(-iπ(τ-1)/Lτ); ν complex, length Ndim"
function τ_to_ω!(ν::AbstractVector{ComplexF64}, g::GPUModel, v::AbstractVector{Float64})
    chk(ccall((:elph_tau_to_omega, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{Float64}), g.handle, ν, v))
    return nothing
end

"ω_to_τ!(v, model, ν): v = real(conj(Θ) ∘ iFFT_τ(ν))"
function ω_to_τ!(v::AbstractVector{Float64}, g::GPUModel, ν::AbstractVector{ComplexF64})
    chk(ccall((:elph_omega_to_tau, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{ComplexF64}), g.handle, v, ν))
    return nothing
end

# ---- sizes, as Models.jl:254-284 ---------------------------------------------------------------------------------------------------

Base.eltype(g::GPUModel) = eltype(g.host)
Base.size(g::GPUModel) = size(g.host)
Base.size(g::GPUModel, d::Int) = size(g.host, d)
Base.length(g::GPUModel) = length(g.host)

end # module
