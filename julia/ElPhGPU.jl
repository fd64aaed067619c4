# ElPhGPU.jl — the Julia side of the drop-in: the reference's OWN model values (`HolsteinModel`, `SSHModel`) keep every method they
# have, and the fermion-matrix path of a model that has been `attach!`ed to an MI355X runs in libelphgpu.so (include/elph_gpu.h).
#
# Design (round 5; the round-4 wrapper type `GPUHolsteinModel <: AbstractModel` is gone — it lost every caller that dispatches on the
# concrete model type: update_Λ!, mulΛ!, calc_Sb, special_update!, … 33 signatures, tests/test_julia_dispatch.py lists them):
#
#   * No new model type.  `ElPhGPU.attach!(model)` creates the device handle and records it in a registry keyed by the model's
#     identity (`IdDict`).  The model stays a `HolsteinModel{Float64,Float64,…}` / `SSHModel{Float64,Float64,…}`: HMC.jl, PhononAction.jl,
#     SpecialUpdates.jl, Measurements.jl, SimulationSummary.jl, InitializePhonons.jl, read_/write_phonons!, Serialization of checkpoints —
#     all untouched and all still applicable.
#   * No method is overwritten.  This file ADDS methods whose signatures are strictly more specific than the reference's
#     (`Vector{Float64}` where the reference says `AbstractVector{T}`, `HolsteinModel{Float64,Float64}` where it says
#     `HolsteinModel{T1,T2}`), so Julia's dispatch prefers them for the real-valued production case, nothing is redefined (no
#     "method overwritten" warning, legal during precompilation on Julia ≥ 1.10), and complex or `SubArray` arguments still reach the
#     reference's methods.  Every added method starts with a registry lookup and, when the model is not attached (or the call is one
#     the library does not serve: M⋅x = b through GMRES/BiCGStab, a Left/Right preconditioner), hands over to the reference's method
#     with `invoke` — the CPU path is always one `detach!` away and is what an unattached model runs.
#   * The host fields stay current.  `update_model!` first runs the reference's method (host `expnΔτV` / `cosht`, `sinht`, `t′`,
#     `expΔτμ` and the equal-fields check) and then uploads exactly those numbers (`elph_set_expV` / `elph_update_model_ssh`), so the
#     device matrix is bit-identical to the host one and every reader of the host fields (KPM diagnostics, measurements, a fallback
#     through `invoke`) sees the current configuration.  `attach!(model, host_sync = false)` computes exp / cosh / sinh on the device
#     instead and leaves the host copies stale until `pull!(model)` — for drivers that never read them.
#
# Served on the device (file:line = the reference method the added method stands in front of):
#   update_model!(model)                      HolsteinModels.jl:526-549, SSHModels.jl:510-562
#   mulM!, mulMᵀ!                             HolsteinModels.jl:569-626,631-684, SSHModels.jl:581-640,646-701
#   mulMᵀM!, mulMMᵀ!  (hence mul!)            Models.jl:215-238 (mul! :192-209 dispatches onto these four)
#   muldMdx!(dMdx, u, model, v)               HolsteinModels.jl:691-755, SSHModels.jl:707-829
#   ldiv!(x, model, b[, P]; maxiter)          Models.jl:74-137,139-186   -> (iters, residual_error, flag), @info lines, zero-fill, retry
#   solve!(x, model, b, cg[, P]; …)           IterativeSolvers.jl:153-234,239-314
#   setup!(P), ldiv!(z, P, r)                 KPMPreconditioners.jl:259-321,426-481  for P::SymmetricKPMPreconditioner of an attached model
#   calc_O⁻¹Λϕ!(hmc, model, P, power)         HMC.jl:820-915: the two pseudofermion solves as ONE batch of two right-hand sides
#   fourier_accelerate!(v′, fa, v, power)     FourierAcceleration.jl:131-137 for an accelerator registered with `attach!(fa, model)`
#   update!(model, hmc, fa, P)                HMC.jl:307-337 — opt-in (`attach!(model, resident_hmc = true)`): the whole trajectory in one call
#
# Wiring in the reference (two lines; nothing else changes):
#   src/ElPhDynamics.jl, after include("ProcessInputFile.jl"):                                    include("ElPhGPU.jl")
#   src/ElPhDynamics.jl `simulate`, after process_input_file / process_checkpoint returned (:103-118):   ElPhGPU.attach!(model)
#
# STATUS: NOT EXECUTED.  Julia is installed neither in the build image nor on the GPU box of this project: this file has never been
# parsed by a Julia compiler.  What IS checked, on every test run: tests/test_julia_dispatch.py (every `ccall` below against the
# prototypes of include/elph_gpu.h — symbol, argument count, argument types; block structure of this file; every reference function
# whose signature names a concrete model type is either left alone or specialised here; every added signature is more specific than
# the reference method it names; EVERY `invoke(f, Tuple{…}, …)` tuple is a subtype of the reference method it must reach — same arity,
# each element equal to or narrower than the reference's, every type variable the reference shares between arguments bound to one
# value — and of no method of this file (tests/julia_types.py restates the needed part of Julia's subtype relation; round 5's eight
# bare `HolsteinModel` / `SSHModel` / `AbstractModel` tuples are kept there as cases the check must reject)), and the calls themselves,
# in this order, from plain C (tests/abi_c/abi_smoke.c) and through ctypes (tests/test_gpu_parity.py, tests/test_gpu_muldmdx.py)
# against the oracle on the GPU.  FIRST RUN with Julia and a GPU: `ElPhGPU.selftest(model)` on a 4 × 4 model (end of this file) — the
# attached and the detached pass of the whole operator API, compared.
#
# The fall-back idiom: `invoke(f, Tuple{AbstractVector{Float64}, typeof(m), AbstractVector{Float64}}, y, m, v)`.  The model's CONCRETE
# type is a subtype of whatever the reference's signature says about the model (`HolsteinModel{T1,T3}`, `SSHModel{T1,T2}`,
# `AbstractModel{T1,T2}` with T2 tied to the vectors), and `AbstractVector` keeps the tuple out of the methods added here (which
# take `Vector{Float64}`), so the reference's method is the most specific one that covers it.  Where the model is the ONLY argument
# (`update_model!`) `typeof(m)` would select the method added here; those fall-backs name the reference's own signature type instead
# (`Tuple{HolsteinModel}`, `Tuple{SSHModel}` == `Tuple{SSHModel{T1,T2}} where {T1,T2}`: each variable occurs once).

module ElPhGPU

using LinearAlgebra
using Random
using Logging

import LinearAlgebra: ldiv!

using ..Utilities: get_index
using ..IterativeSolvers: ConjugateGradient
import ..IterativeSolvers: solve!
using ..Models: AbstractModel, HolsteinModel, SSHModel
import ..Models: mulM!, mulMᵀ!, mulMᵀM!, mulMMᵀ!, muldMdx!, update_model!
using ..KPMPreconditioners: KPMPreconditioner, SymmetricKPMPreconditioner, KPMExpansion
import ..KPMPreconditioners: setup!
using ..FourierAcceleration: FourierAccelerator
import ..FourierAcceleration: fourier_accelerate!
using ..HMC: HybridMonteCarlo, update_Λ!, mulΛ!
import ..HMC: calc_O⁻¹Λϕ!, update!

export attach!, detach!, attached, pull!, ldiv_batched!, selftest, ElphError

"Path of the shared library; `ENV[\"ELPHGPU_LIB\"]` overrides (the in-tree build is elphdynamics_amd/libelphgpu.so)."
const lib = get(ENV, "ELPHGPU_LIB", "libelphgpu.so")

"include/elph_gpu.h: ELPH_ABI_VERSION this file was written against; checked for equality when the first model is attached."
const ELPH_ABI = 2

"The models whose fermion matrix the library holds: real parameters, real matrix elements (`is_complex = false`)."
const GPUModel = Union{HolsteinModel{Float64,Float64},SSHModel{Float64,Float64}}

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

"What the loaded library was built from (source hash, compiler, time)."
build_info() = unsafe_string(ccall((:elph_build_info, lib), Cstring, ()))

# ------------------------------------------------------------------------------------------------------------------------------
# The registry: model identity -> device state.  The model itself is not changed, wrapped or subtyped.
# ------------------------------------------------------------------------------------------------------------------------------

mutable struct Entry
    handle::Ptr{Cvoid}
    host_sync::Bool                     # update_model! keeps the host fields current (default) or computes on the device only
    batch_pseudofermions::Bool          # calc_O⁻¹Λϕ!: ϕ₊ and ϕ₋ as one batch of two right-hand sides
    kpm::Any                            # the KPMExpansion whose parameters elph_kpm_create received (nothing: none yet)
    kpm_ready::Bool                     # setup!(P) has run on the device (the expansion of e.kpm exists there)
    X2::Matrix{Float64}                 # Ndim × 2 staging for batched solves (no allocation in the hot loop)
    B2::Matrix{Float64}
    q::Matrix{Float64}                  # Lτ × Nbonds bond brackets of muldMdx! (SSH)
    bmax::Vector{Float64}               # Arnoldi start vectors of setup!(P)
    bmin::Vector{Float64}
    cb_index::Vector{Int64}             # SSH, host_sync = false: checkerboard position of every phonon's bond, 1-based
    t_ph::Vector{Float64}               #   bare hopping of every phonon's bond
    t_bare_cb::Vector{Float64}          #   bare hopping of every bond, checkerboard order
    gpu_calls::Int                      # served on the device / handed to the reference's method — `status(model)`
    fallbacks::Int
    resident_hmc::Bool                  # update!(model, hmc, fa, P): the whole trajectory on the device (elph_hmc_update)
    hmc_fa::Any                         # the FourierAccelerator the device-side HMC state was created with (nothing: none yet)
    hmc_Msum::Float64                   # checksums of fa.M and model.μ at that time: a change re-creates the state / pushes μ
    hmc_μsum::Float64
    kpm_randn::Vector{Float64}          # (Nt + 2) pairs of Arnoldi start vectors of one update
    energies::Vector{Float64}           # H₀, H₁, S, K, P_accept of the last resident update
    device::Int                         # the GPU the handle lives on (selftest re-attaches with the caller's options)
end

const REGISTRY = IdDict{Any,Entry}()
const FA_REGISTRY = IdDict{Any,Any}()   # FourierAccelerator -> its model

"The device state of `model`, or `nothing` when it is not attached (then every method below defers to the reference's)."
entry(model) = get(REGISTRY, model, nothing)::Union{Nothing,Entry}

"`true` when the fermion matrix of `model` lives on a GPU."
attached(model) = haskey(REGISTRY, model)

"(calls served on the device, calls handed to the reference's CPU method) since `attach!`."
function status(model)
    e = entry(model)
    return e === nothing ? (0, 0) : (e.gpu_calls, e.fallbacks)
end

function create_handle(kind::Integer, N::Integer, Lτ::Integer, Nbonds::Integer, neighbor_table, cosht, sinht, device::Integer)
    v = ccall((:elph_abi_version, lib), Cint, ())
    v == ELPH_ABI || error("libelphgpu speaks ABI $v, ElPhGPU.jl was written for $ELPH_ABI ($(build_info()))")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall((:elph_create, lib), Cint,
              (Ref{Ptr{Cvoid}}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Cint),
              h, kind, N, Lτ, Nbonds, neighbor_table, cosht, sinht, device))
    return h[]
end

"""
    attach!(model; device = 0, host_sync = true, batch_pseudofermions = true, resident_hmc = false) -> model

Put the fermion matrix of an initialised model (`initialize_model!`, HolsteinModels.jl:484-517 / SSHModels.jl:348-505:
`neighbor_table`, `cosht`, `sinht` in checkerboard order) on GPU `device` and route the operator API of this model there.  Throws
`ElphError(ELPH_E_NOGPU)` when no gfx950 device is visible — the library has no CPU path; the reference's own methods are the CPU
path and stay in force for every model that is not attached.  Attaching twice is a no-op.  `resident_hmc = true`: `update!(model, hmc, fa, P)`
runs the whole trajectory on the device (see `update!` below for what that changes).
"""
function attach!(model::GPUModel; device::Integer=0, host_sync::Bool=true, batch_pseudofermions::Bool=true, resident_hmc::Bool=false)
    attached(model) && return model
    N, Lτ, Nb = model.Nsites, model.Lτ, model.Nbonds
    if model isa HolsteinModel
        nt = Nb > 0 ? model.neighbor_table : zeros(Int64, 2, 0)
        h = create_handle(0, N, Lτ, Nb, nt, model.cosht, model.sinht, device)
        cb_index = Int64[]; t_ph = Float64[]; t_bare_cb = Float64[]
    else
        h = create_handle(1, N, Lτ, Nb, model.neighbor_table, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), device)
        cb_index = Int64[model.checkerboard_perm[model.phonon_to_bond[p]] for p in 1:model.Nph]
        t_ph = Float64[model.t[model.phonon_to_bond[p]] for p in 1:model.Nph]
        t_bare_cb = zeros(Float64, Nb)
        for bond in 1:Nb
            t_bare_cb[model.checkerboard_perm[bond]] = model.t[bond]
        end
    end
    e = Entry(h, host_sync, batch_pseudofermions, nothing, false, zeros(model.Ndim, 2), zeros(model.Ndim, 2),
              zeros(Lτ, model isa SSHModel ? Nb : 0), zeros(N), zeros(N), cb_index, t_ph, t_bare_cb, 0, 0,
              resident_hmc, nothing, 0.0, 0.0, Float64[], zeros(5), Int(device))
    REGISTRY[model] = e
    update_model!(model)
    return model
end

"Register a FourierAccelerator built from `model` (FourierAcceleration.jl:47-82) so that `fourier_accelerate!` runs on its device."
function attach!(fa::FourierAccelerator{Float64}, model::GPUModel)
    attached(model) || error("attach!(fa, model): attach the model first")
    FA_REGISTRY[fa] = model
    return fa
end

"""
    detach!(model)

Free the device handle; from here on every call runs the reference's own methods again.  A preconditioner that was set up on the
device forgets its bounds (λ_lo, λ_hi back to the constructor's 0 and 2, KPMPreconditioners.jl:68-69), so that the next `setup!` on
the CPU recomputes its coefficients instead of trusting numbers it never computed.
"""
function detach!(model)
    e = entry(model)
    e === nothing && return nothing
    if e.kpm !== nothing
        op = e.kpm
        op.λ_lo = 0.0; op.λ_hi = 2.0; op.λ_avg = 1.0; op.λ_mag = 1.0
    end
    for (fa, m) in collect(FA_REGISTRY)
        m === model && delete!(FA_REGISTRY, fa)
    end
    e.handle == C_NULL || ccall((:elph_destroy, lib), Cint, (Ptr{Cvoid},), e.handle)
    e.handle = C_NULL
    delete!(REGISTRY, model)
    return nothing
end

function detach_all!()
    for m in collect(keys(REGISTRY))
        detach!(m)
    end
    return nothing
end

function __init__()
    atexit(detach_all!)
    return nothing
end

served!(e::Entry) = (e.gpu_calls += 1; nothing)
deferred!(e::Union{Nothing,Entry}) = (e === nothing || (e.fallbacks += 1); nothing)

# ------------------------------------------------------------------------------------------------------------------------------
# update_model!
# ------------------------------------------------------------------------------------------------------------------------------

"update_model!(holstein): expnΔτV = exp(-Δτ(λx + λ₂x² - μ)) — HolsteinModels.jl:526-549 — on the host AND on the device (same bits)"
function update_model!(m::HolsteinModel{Float64,Float64})
    e = entry(m)
    if e === nothing || e.host_sync
        invoke(update_model!, Tuple{HolsteinModel}, m)
        e === nothing && return nothing
        chk(ccall((:elph_set_expV, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), e.handle, m.expnΔτV))
    else
        chk(ccall((:elph_update_model_holstein, lib), Cint,
                  (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
                  e.handle, m.x, m.λ, m.λ₂, m.μ, m.Δτ))
    end
    served!(e)                          # (an expansion set up for an earlier configuration stays usable until the next setup!, as in
                                        #  the reference, where ldiv!(x, model, b, P) applies whatever setup!(P) last left in P)
    return nothing
end

"update_model!(ssh): expΔτμ, t′ = t − (αx + sign(x)α₂x²), cosh/sinh(Δτ t′) per (τ, bond), equal-fields check — SSHModels.jl:510-562"
function update_model!(m::SSHModel{Float64,Float64})
    e = entry(m)
    if e === nothing || e.host_sync
        invoke(update_model!, Tuple{SSHModel}, m)
        e === nothing && return nothing
        # model.cosht / .sinht are (Lτ × Nbonds) column-major = [bond][τ], the layout elph_update_model_ssh takes
        chk(ccall((:elph_update_model_ssh, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                  e.handle, m.cosht, m.sinht, m.expΔτμ))
    else
        for field in 1:m.Ndof           # the equal-fields check stays on the host, as in the reference (:548-559)
            field′ = m.primary_field[field]
            if field != field′ && !(m.x[field] ≈ m.x[field′])
                error("(x[$field]=$(m.x[field])) != (x[$field′]=$(m.x[field′]))\n")
            end
        end
        chk(ccall((:elph_update_model_ssh_fields, lib), Cint,
                  (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
                  e.handle, m.x, m.Nph, e.cb_index, e.t_ph, m.α, m.α₂, e.t_bare_cb, m.μ, m.Δτ))
    end
    served!(e)
    return nothing
end

"""
    pull!(model)

`host_sync = false` only: bring the host copies of the device-computed matrix elements up to date (SSH: `model.cosht`, `model.sinht`
through `elph_get_cosh_sinh`; Holstein: `model.expnΔτV` by the reference's own update_model!, the host exp being what it would have
computed anyway).
"""
function pull!(m::GPUModel)
    e = entry(m)
    e === nothing && return nothing
    if m isa HolsteinModel
        invoke(update_model!, Tuple{HolsteinModel}, m)
    else
        chk(ccall((:elph_get_cosh_sinh, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, m.cosht, m.sinht))
        @. m.expΔτμ = exp(m.Δτ * m.μ)
    end
    return nothing
end

# ------------------------------------------------------------------------------------------------------------------------------
# mul! family — HolsteinModels.jl:569-684, SSHModels.jl:581-701, Models.jl:215-238.  `mul!` (Models.jl:192-209) needs no method of
# its own: it dispatches on model.mul_by_M / model.transposed onto these four.
# ------------------------------------------------------------------------------------------------------------------------------

"y = M⋅v"
function mulM!(y::Vector{Float64}, m::HolsteinModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulM!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulM, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

function mulM!(y::Vector{Float64}, m::SSHModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulM!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulM, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

"y = Mᵀ⋅v"
function mulMᵀ!(y::Vector{Float64}, m::HolsteinModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulMᵀ!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulMT, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

function mulMᵀ!(y::Vector{Float64}, m::SSHModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulMᵀ!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulMT, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

"y = MᵀM⋅v in one fused pass (the reference goes through model.v′, Models.jl:215-224)"
function mulMᵀM!(y::Vector{Float64}, m::GPUModel, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulMᵀM!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulMTM, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

"y = MMᵀ⋅v (Models.jl:229-238)"
function mulMMᵀ!(y::Vector{Float64}, m::GPUModel, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(mulMMᵀ!, Tuple{AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, y, m, v)
    @assert length(y) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_mulMMT, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, y, v))
    served!(e)
    return nothing
end

# ------------------------------------------------------------------------------------------------------------------------------
# muldMdx! — dMdx[field] = uᵀ⋅(∂M/∂x_field)⋅v; calc_dSfdx! calls it with (u, v) = (M⋅O⁻¹Λϕ±, O⁻¹Λϕ±) (HMC.jl:799,804),
# LangevinDynamics.calc_dSfdx! with (g, M⁻¹g) (:378)
# ------------------------------------------------------------------------------------------------------------------------------

"muldMdx!(dMdx, u, holstein, v) — HolsteinModels.jl:691-755"
function muldMdx!(dMdx::Vector{Float64}, u::Vector{Float64}, m::HolsteinModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(muldMdx!, Tuple{AbstractVector{Float64},AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, dMdx, u, m, v)
    @assert length(dMdx) == m.Ndim && length(u) == m.Ndim && length(v) == m.Ndim
    chk(ccall((:elph_muldMdx_holstein, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
              e.handle, dMdx, u, v, m.x, m.λ, m.λ₂, m.Δτ))
    served!(e)
    return nothing
end

"muldMdx!(dMdx, u, ssh, v) — SSHModels.jl:707-829: the bond brackets ⟨c_n|…|b_n⟩ come from the device (the two checkerboard
recursions per time slice), ∂K/∂x, the τ = 1 sign and the sum over equivalent fields are applied here exactly as at :797-826"
function muldMdx!(dMdx::Vector{Float64}, u::Vector{Float64}, m::SSHModel{Float64,Float64}, v::Vector{Float64})
    e = entry(m)
    e === nothing && return invoke(muldMdx!, Tuple{AbstractVector{Float64},AbstractVector{Float64},typeof(m),AbstractVector{Float64}}, dMdx, u, m, v)
    @assert length(dMdx) == m.Ndof && length(u) == m.Ndim && length(v) == m.Ndim
    q = e.q                             # q[τ, n] = c_j b_i + c_i b_j for checkerboard bond n: (Lτ × Nbonds) column-major = elph_muldMdx_ssh's q_out
    chk(ccall((:elph_muldMdx_ssh, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), e.handle, q, u, v))
    Lτ, Δτ = m.Lτ, m.Δτ
    fill!(dMdx, 0.0)
    @inbounds for n in 1:m.Nbonds
        bond = m.inv_checkerboard_perm[n]
        phonon = m.bond_to_phonon[bond]
        phonon == 0 && continue
        for τ in 1:Lτ
            field = get_index(τ, phonon, Lτ)
            dKdx = m.α[phonon] + 2 * m.α₂[phonon] * m.x[field]
            dmdx = Δτ * dKdx * q[τ, n]
            if τ == 1
                dmdx = -dmdx
            end
            dMdx[m.primary_field[field]] += dmdx
        end
    end
    @views @. dMdx = dMdx[m.primary_field]
    served!(e)
    return nothing
end

# ------------------------------------------------------------------------------------------------------------------------------
# ldiv! — Models.jl:74-137 (with P), :139-186 (without); solve! — IterativeSolvers.jl:153-234, 239-314
# ------------------------------------------------------------------------------------------------------------------------------

"callers mutate model.solver.tol between solves (HMC.jl:827-828, restored :912): the current values travel with every solve"
function push_solver!(e::Entry, m::GPUModel)
    chk(ccall((:elph_solver_set, lib), Cint, (Ptr{Cvoid}, Float64, Int64, Float64),
              e.handle, m.solver.tol, m.solver.maxiter, m.solver.κmax))
    return nothing
end

function log_flag(flag::Integer, err, iters, with_P::Bool)
    # the reference's @info lines — Models.jl:108,117 (with preconditioner), :163,172 (without)
    suffix = with_P ? "W/ Preconditioner" : "W/O Preconditioner"
    flag == 1 && @info("Hit Max Iters, Residual Error = $err, Iterations = $iters, $suffix")
    flag == 2 && @info("Large Residual Error = $err, Iterations = $iters, $suffix")
    if flag > 0
        logger = global_logger()
        hasproperty(logger, :stream) && flush(logger.stream)
    end
    return nothing
end

"Is this a solve the library serves?  MᵀM⋅x = b by conjugate gradients (every caller on the path: HMC.jl:851-886, GreensFunctions.jl:225,
LangevinDynamics.jl:374); M⋅x = b / Mᵀ⋅x = b through GMRES / BiCGStab stay with the reference's solvers (which then call the mat-vecs above)."
servable(m::GPUModel) = !m.mul_by_M && !m.transposed && m.solver isa ConjugateGradient

"0: P is the identity; 1: P is the symmetric KPM preconditioner of this model, set up on the device; -1: anything else"
function precond_mode(e::Entry, m::GPUModel, P)
    P === I && return 0
    P isa UniformScaling && return (P == I ? 0 : -1)
    if P isa SymmetricKPMPreconditioner && P.expansion.model === m && e.kpm === P.expansion && e.kpm_ready && !P.transposed
        return 1
    end
    return -1
end

function gpu_ldiv!(x::Vector{Float64}, e::Entry, m::GPUModel, b::Vector{Float64}, use_P::Bool, maxiter::Int)
    @assert length(x) == m.Ndim && length(b) == m.Ndim
    push_solver!(e, m)
    iters = Ref{Int64}(0); err = Ref{Float64}(0.0); flag = Ref{Cint}(0)
    chk(ccall((:elph_ldiv, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint, Int64, Ref{Int64}, Ref{Float64}, Ref{Cint}),
              e.handle, x, b, use_P ? 1 : 0, maxiter, iters, err, flag))
    # elph_ldiv has already done what Models.jl:96-134 does (true residual, flag, zero-fill, the un-preconditioned retry with
    # 10·maxiter): only the log lines are left
    log_flag(flag[], err[], iters[], use_P)
    served!(e)
    return Int(iters[]), err[], Int(flag[])
end

"""
    ldiv!(x, model, b; maxiter = 0)    -> (iters, residual_error, flag)                    Models.jl:139-186
    ldiv!(x, model, b, P; maxiter = 0) -> (iters, residual_error, flag)                    Models.jl:74-137

MᵀM⋅x = b from the caller's `x` (callers pass zeros: HMC.jl:854) with the reference's stop rule, true-residual check, flags 0 / 1
(hit maxiter) / 2 (false convergence), zero-fill of `x` when flag > 0 and — with a preconditioner — the un-preconditioned retry with
10⋅maxiter.
"""
function ldiv!(x::Vector{Float64}, m::GPUModel, b::Vector{Float64}; maxiter::Int=0)::Tuple{Int,Float64,Int}
    e = entry(m)
    if e === nothing || !servable(m)
        deferred!(e)
        return invoke(ldiv!, Tuple{AbstractVector,typeof(m),AbstractVector}, x, m, b; maxiter=maxiter)
    end
    return gpu_ldiv!(x, e, m, b, false, maxiter)
end

function ldiv!(x::Vector{Float64}, m::GPUModel, b::Vector{Float64}, P; maxiter::Int=0)::Tuple{Int,Float64,Int}
    e = entry(m)
    mode = (e === nothing || !servable(m)) ? -1 : precond_mode(e, m, P)
    if mode < 0
        deferred!(e)
        return invoke(ldiv!, Tuple{AbstractVector,typeof(m),AbstractVector,Any}, x, m, b, P; maxiter=maxiter)
    end
    return gpu_ldiv!(x, e, m, b, mode == 1, maxiter)
end

"""
    ldiv_batched!(X, model, B[, P]; maxiter = 0) -> (iters[], residual_error[], flag[])

`size(B, 2)` right-hand sides in one call (columns): the two pseudofermion solves of `calc_O⁻¹Λϕ!` (HMC.jl:851-886), the nᵥ solves of
`GreensFunctions.update!` (:201-234).  Every column follows exactly the single-solve recurrences, stop rule and flag logic.  No
reference method exists for this (the reference solves one at a time); an unattached model loops over `ldiv!`.
"""
function ldiv_batched!(X::Matrix{Float64}, m::GPUModel, B::Matrix{Float64}, P=I; maxiter::Int=0)
    n = size(B, 2)
    @assert size(X) == size(B) && size(B, 1) == m.Ndim
    iters = zeros(Int64, n); err = zeros(Float64, n); flag = zeros(Cint, n)
    e = entry(m)
    mode = (e === nothing || !servable(m)) ? -1 : precond_mode(e, m, P)
    if mode < 0
        deferred!(e)
        for k in 1:n
            xk = X[:, k]
            it, er, fl = ldiv!(xk, m, B[:, k], P, maxiter=maxiter)
            X[:, k] = xk
            iters[k] = it; err[k] = er; flag[k] = fl
        end
        return iters, err, Int.(flag)
    end
    push_solver!(e, m)
    chk(ccall((:elph_ldiv_batched, lib), Cint,
              (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cint, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Cint}),
              e.handle, n, X, B, mode, maxiter, iters, err, flag))
    for k in 1:n
        log_flag(flag[k], err[k], iters[k], mode == 1)
    end
    served!(e)
    return iters, err, Int.(flag)
end

function gpu_solve!(x::Vector{Float64}, e::Entry, m::GPUModel, b::Vector{Float64}, cg::ConjugateGradient{Float64,Float64}, use_P::Bool,
                    maxiter::Int, tol::Float64, κmax::Float64)
    @assert length(x) == m.Ndim && length(b) == m.Ndim
    iters = Ref{Int64}(0)
    chk(ccall((:elph_cg_solve, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Int64, Float64, Cint, Ref{Int64}, Ptr{Float64}),
              e.handle, x, b, iszero(tol) ? cg.tol : tol, iszero(maxiter) ? cg.maxiter : maxiter,
              iszero(κmax) ? cg.κmax : κmax, use_P ? 1 : 0, iters, C_NULL))
    served!(e)
    return Int(iters[])
end

"solve!(x, A, b, cg; maxiter, tol, κmax) -> iters — IterativeSolvers.jl:239-314 (no true-residual check, no flags: that is ldiv!'s)"
function solve!(x::Vector{Float64}, m::GPUModel, b::Vector{Float64}, cg::ConjugateGradient{Float64,Float64};
                maxiter::Int=0, tol::Float64=0.0, κmax::Float64=0.0)::Int
    e = entry(m)
    if e === nothing || !servable(m)
        deferred!(e)
        return invoke(solve!, Tuple{AbstractVector{Float64},Any,AbstractVector{Float64},ConjugateGradient{Float64,Float64}}, x, m, b, cg;
                      maxiter=maxiter, tol=tol, κmax=κmax)
    end
    return gpu_solve!(x, e, m, b, cg, false, maxiter, tol, κmax)
end

"solve!(x, A, b, cg, P; maxiter, tol, κmax) -> iters — IterativeSolvers.jl:153-234"
function solve!(x::Vector{Float64}, m::GPUModel, b::Vector{Float64}, cg::ConjugateGradient{Float64,Float64}, P;
                maxiter::Int=0, tol::Float64=0.0, κmax::Float64=0.0)::Int
    e = entry(m)
    mode = (e === nothing || !servable(m)) ? -1 : precond_mode(e, m, P)
    if mode < 0
        deferred!(e)
        return invoke(solve!, Tuple{AbstractVector{Float64},Any,AbstractVector{Float64},ConjugateGradient{Float64,Float64},Any}, x, m, b, cg, P;
                      maxiter=maxiter, tol=tol, κmax=κmax)
    end
    return gpu_solve!(x, e, m, b, cg, mode == 1, maxiter, tol, κmax)
end

# ------------------------------------------------------------------------------------------------------------------------------
# KPM preconditioner — the reference's own SymmetricKPMPreconditioner (KPMPreconditioners.jl:219-235) of an attached model
# ------------------------------------------------------------------------------------------------------------------------------

const GPUSymmetricKPM = SymmetricKPMPreconditioner{Float64,Float64,<:GPUModel}

"""
    setup!(P)                                                                             KPMPreconditioners.jl:259-321

τ-average Ē (update_A!, :332-381), the two Arnoldi runs for e_max and 1/e_min (:845-942) from start vectors drawn from `model.rng` — the
same 2·Nsites `randn(rng, Float64)` calls in the same order as :859-861 and :902-904, so the stream of `model.rng` advances exactly as
in a CPU run — the acceptance test, λ_lo/λ_hi with the `buf` hysteresis, orders and Chebyshev coefficients (:259-321), all on the
device.  `P.expansion.active`, `.λ_lo`, `.λ_hi`, `.λ_avg`, `.λ_mag` and `.order` mirror the device's values for the readers of those
fields; the coefficients themselves stay on the device.
"""
function setup!(P::GPUSymmetricKPM)
    op = P.expansion
    m = op.model
    e = entry(m)
    if e === nothing
        return invoke(setup!, Tuple{KPMPreconditioner}, P)
    end
    if e.kpm !== op                     # first use of this expansion on this handle: its n, buf, c1, c2 (KPMPreconditioners.jl:63-93)
        chk(ccall((:elph_kpm_create, lib), Cint, (Ptr{Cvoid}, Cint, Float64, Float64, Float64), e.handle, op.n, op.buf, op.c1, op.c2))
        e.kpm = op
    end
    N = m.Nsites
    for i in 1:N
        e.bmax[i] = randn(m.rng, Float64)
    end
    for i in 1:N
        e.bmin[i] = randn(m.rng, Float64)
    end
    act = Ref{Cint}(0); lo = Ref{Float64}(0.0); hi = Ref{Float64}(0.0)
    chk(ccall((:elph_kpm_setup, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ref{Cint}, Ref{Float64}, Ref{Float64}),
              e.handle, e.bmax, e.bmin, NaN, NaN, act, lo, hi))
    op.active = act[] != 0
    if op.active
        op.λ_lo = lo[]; op.λ_hi = hi[]
        op.λ_avg = (op.λ_hi + op.λ_lo) / 2
        op.λ_mag = (op.λ_hi - op.λ_lo) / 2
        total = Ref{Int64}(0)
        chk(ccall((:elph_kpm_orders, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), e.handle, op.order, total))
    end
    e.kpm_ready = true
    served!(e)
    return nothing
end

"ldiv!(z, P, r): z = P⁻¹⋅r (twisted τ-FFT, per-ω Chebyshev series, inverse) — KPMPreconditioners.jl:426-481; a copy when inactive"
function ldiv!(z::Vector{Float64}, P::GPUSymmetricKPM, r::Vector{Float64})
    op = P.expansion
    e = entry(op.model)
    if e === nothing || e.kpm !== op || !e.kpm_ready || P.transposed
        deferred!(e)
        return invoke(ldiv!, Tuple{AbstractVector{Float64},KPMPreconditioner,AbstractVector{Float64}}, z, P, r)
    end
    @assert length(z) == op.model.Ndim && length(r) == op.model.Ndim
    chk(ccall((:elph_kpm_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, z, r))
    served!(e)
    return nothing
end

# ------------------------------------------------------------------------------------------------------------------------------
# calc_O⁻¹Λϕ! — HMC.jl:820-915 with the two pseudofermion systems as one batch of two right-hand sides (same matrix, same
# tolerance): every other line is the reference's, in the reference's order
# ------------------------------------------------------------------------------------------------------------------------------

function calc_O⁻¹Λϕ!(hmc::HybridMonteCarlo{Float64}, model::GPUModel, preconditioner, power::Float64)::Tuple{Int,Int}
    e = entry(model)
    mode = (e === nothing || !e.batch_pseudofermions || !servable(model)) ? -1 : 0
    if mode < 0
        return invoke(calc_O⁻¹Λϕ!, Tuple{HybridMonteCarlo{Float64},AbstractModel{Float64,Float64},Any,Float64}, hmc, model, preconditioner, power)
    end
    Λϕ₊, Λϕ₋, O⁻¹Λϕ₊, O⁻¹Λϕ₋ = hmc.Λϕ₊, hmc.Λϕ₋, hmc.O⁻¹Λϕ₊, hmc.O⁻¹Λϕ₋
    tol = model.solver.tol
    model.solver.tol = tol^power                                          # :827-828
    hmc.iters = 0
    setup!(preconditioner)                                                # :834 (a no-op method exists for I, KPMPreconditioners.jl:323)
    mode = precond_mode(e, model, preconditioner)
    if mode < 0                                                           # a preconditioner the library does not hold: one at a time
        model.solver.tol = tol
        return invoke(calc_O⁻¹Λϕ!, Tuple{HybridMonteCarlo{Float64},AbstractModel{Float64,Float64},Any,Float64}, hmc, model, preconditioner, power)
    end
    update_Λ!(hmc, model)                                                 # :840-842
    mulΛ!(Λϕ₊, hmc.ϕ₊, hmc, model)
    mulΛ!(Λϕ₋, hmc.ϕ₋, hmc, model)
    X, B = e.X2, e.B2
    fill!(X, 0.0)                                                         # fill!(O⁻¹Λϕ±, 0.0), :854,883
    copyto!(view(B, :, 1), Λϕ₊)
    copyto!(view(B, :, 2), Λϕ₋)
    model.transposed = false
    iters, err, flags = ldiv_batched!(X, model, B, preconditioner)
    copyto!(O⁻¹Λϕ₊, view(X, :, 1))
    hmc.iters += iters[1]
    flag = flags[1]
    if iszero(flag)                                                       # the second system counts only if the first converged, :880
        copyto!(O⁻¹Λϕ₋, view(X, :, 2))
        hmc.iters += iters[2]
        flag = flags[2]
    end
    if iszero(flag)
        hmc.iters = cld(hmc.iters, 2)                                     # :907-909
    end
    model.solver.tol = tol                                                # :912
    return hmc.iters, flag
end

# ------------------------------------------------------------------------------------------------------------------------------
# update!(model, hmc, fa, P) — HMC.jl:307-337 (standard_update! :343-463, multitimestep_update! :469-638) with the WHOLE trajectory in one
# library call: x, v, ϕ±, O⁻¹Λϕ±, dS/dx stay on the device for its length.  Opt-in (`attach!(model, resident_hmc = true)`), because it is
# the one method here that replaces caller code rather than an operator:
#   * the random numbers are drawn from model.rng HERE, in the reference's order — R (refresh_v!, :655), R₊, R₋ (refresh_ϕ!, :675-676), one
#     pair of Arnoldi start vectors per setup!(P) (KPMPreconditioners.jl:859-861, 902-904; Nt + 2 of them), the uniform of the Metropolis test
#     (:441) — so a trajectory that is not killed consumes the stream exactly as a CPU run; a killed one (flag > 0) has drawn all Nt + 2 pairs
#     where the reference stops early;
#   * hmc.log / hmc.verbose (update_log) are not served: such an `hmc` takes the reference's method;
#   * after the call model.x, hmc.v, hmc.accepted, hmc.H / .S / .K, hmc.iters, hmc.updates and the host copy of exp(−ΔτV) are current; the
#     work vectors hmc.ϕ±, hmc.O⁻¹Λϕ±, hmc.dSdx are NOT (they live on the device).
# ------------------------------------------------------------------------------------------------------------------------------

"(re)create the device-side HMC state for this accelerator (elph_hmc_create / elph_hmc_create_ssh: ω, ω₄, couplings, μ, Δτ, fa.M)"
function hmc_state!(e::Entry, m::GPUModel, fa::FourierAccelerator{Float64})
    Msum = sum(fa.M)
    if e.hmc_fa !== fa || e.hmc_Msum != Msum
        if m isa HolsteinModel
            chk(ccall((:elph_hmc_create, lib), Cint,
                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                      e.handle, m.ω, m.ω₄, m.λ, m.λ₂, m.μ, m.Δτ, fa.M))
        else
            chk(ccall((:elph_hmc_create_ssh, lib), Cint,
                      (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
                      e.handle, m.Nph, m.ω, m.ω₄, e.cb_index, e.t_ph, m.α, m.α₂, e.t_bare_cb, m.μ, m.Δτ, fa.M))
            if any(f -> m.primary_field[f] != f, 1:m.Ndof)      # phonon types of one name share their fields (SSHModels.jl:480-502)
                prim = Int64[div(m.primary_field[(p - 1) * m.Lτ + 1] - 1, m.Lτ) for p in 1:m.Nph]      # 0-based primary column of every phonon
                chk(ccall((:elph_hmc_set_shared_fields, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), e.handle, prim))
            end
        end
        e.hmc_fa = fa; e.hmc_Msum = Msum; e.hmc_μsum = sum(m.μ)
    elseif e.hmc_μsum != sum(m.μ)      # the chemical-potential tuner moved μ (MuFinder.jl:68-107)
        chk(ccall((:elph_hmc_set_mu, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), e.handle, m.μ))
        e.hmc_μsum = sum(m.μ)
    end
    return nothing
end

function update!(model::GPUModel, hmc::HybridMonteCarlo{Float64}, fa::FourierAccelerator{Float64}, preconditioner)::Tuple{Bool,Float64}
    e = entry(model)
    use_P = -1
    if e !== nothing && e.resident_hmc && servable(model) && !hmc.log && hmc.Ndof > 0 &&
       !(model isa HolsteinModel && !isempty(model.ωᵢⱼ))      # (dispersive modes: not on the device)
        if preconditioner === I
            use_P = 0
        elseif preconditioner isa GPUSymmetricKPM && preconditioner.expansion.model === model && !preconditioner.transposed
            op = preconditioner.expansion
            if e.kpm !== op
                chk(ccall((:elph_kpm_create, lib), Cint, (Ptr{Cvoid}, Cint, Float64, Float64, Float64), e.handle, op.n, op.buf, op.c1, op.c2))
                e.kpm = op
            end
            use_P = 1
        end
    end
    if use_P < 0
        deferred!(e)
        return invoke(update!, Tuple{AbstractModel{Float64,Float64},HybridMonteCarlo{Float64},FourierAccelerator{Float64},Any}, model, hmc, fa, preconditioner)
    end
    hmc_state!(e, model, fa)
    push_solver!(e, model)
    chk(ccall((:elph_hmc_set_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, model.x, hmc.v))
    # the random numbers of one update, from model.rng in the reference's order
    R = hmc.y
    randn!(R, model)                                                      # refresh_v!, HMC.jl:655 (bond phonons: v[primary_field])
    randn!(model.rng, hmc.R₊)                                             # refresh_ϕ!, :675-676
    randn!(model.rng, hmc.R₋)
    N, Nt = model.Nsites, hmc.Nt
    if use_P == 1
        length(e.kpm_randn) == 2 * N * (Nt + 2) || resize!(e.kpm_randn, 2 * N * (Nt + 2))
        for k in 1:(2 * N * (Nt + 2))                                     # per setup!(P): N scalars for e_max, N for e_min
            e.kpm_randn[k] = randn(model.rng, Float64)
        end
    end
    u = rand(model.rng)                                                   # :441
    acc = Ref{Cint}(0); its = Ref{Float64}(0.0); flag = Ref{Cint}(0)
    chk(ccall((:elph_hmc_update, lib), Cint,
              (Ptr{Cvoid}, Float64, Int64, Cint, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64,
               Ref{Cint}, Ref{Float64}, Ptr{Float64}, Ref{Cint}),
              e.handle, hmc.Δt, Nt, hmc.Nb, hmc.α, use_P, R, hmc.R₊, hmc.R₋, use_P == 1 ? pointer(e.kpm_randn) : Ptr{Float64}(C_NULL), u,
              acc, its, e.energies, flag))
    chk(ccall((:elph_hmc_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), e.handle, model.x, hmc.v))
    # the host copy of the matrix elements follows the field (the device rebuilt its own inside the update)
    if model isa HolsteinModel
        invoke(update_model!, Tuple{HolsteinModel}, model)
    else
        invoke(update_model!, Tuple{SSHModel}, model)
    end
    hmc.accepted = acc[] != 0
    hmc.H = e.energies[2]; hmc.S = e.energies[3]; hmc.K = e.energies[4]
    hmc.iters = round(Int, its[])
    hmc.t = Nt                                                            # where `for hmc.t in 1:Nt` (:394, :535) leaves it
    hmc.updates += 1                                                      # :329
    served!(e)
    return hmc.accepted, its[]
end

# ------------------------------------------------------------------------------------------------------------------------------
# Fourier acceleration — FourierAcceleration.jl:131-137 (real in, real out: the form HMC.jl:386,656,715 and LangevinDynamics.jl use)
# ------------------------------------------------------------------------------------------------------------------------------

"v′ = iFFT_τ( D^power ∘ FFT_τ(v) ), D = fa.M (use_mass) or fa.Q, on the device of the model `fa` was registered with"
function fourier_accelerate!(v′::Vector{Float64}, fa::FourierAccelerator{Float64}, v::Vector{Float64}, power::Float64; use_mass::Bool=false)
    m = get(FA_REGISTRY, fa, nothing)
    e = m === nothing ? nothing : entry(m)
    if e === nothing
        return invoke(fourier_accelerate!, Tuple{AbstractVector{Float64},FourierAccelerator{Float64},AbstractVector{Float64},Float64}, v′, fa, v, power;
                      use_mass=use_mass)
    end
    @assert length(v′) == fa.N * fa.L && length(v) == fa.N * fa.L
    chk(ccall((:elph_fourier_accelerate, lib), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Int64),
              e.handle, v′, v, use_mass ? fa.M : fa.Q, power, fa.N))
    served!(e)
    return nothing
end

# ------------------------------------------------------------------------------------------------------------------------------
# selftest — the first thing to run where Julia and a GPU meet (this file was written where neither existed)
# ------------------------------------------------------------------------------------------------------------------------------

relerr(a, b) = norm(a .- b) / max(norm(b), floatmin(Float64))

"""
    selftest(model; P = I, rtol_mul = 1e-12, rtol_solve = 1e-10, solver_tol = 1e-13, verbose = true) -> NamedTuple

Run the operator API of `model` twice on the same seeded vectors — attached (the library) and detached (the reference's own methods,
reached through the `invoke` fall-backs of this file) — and compare: `update_model!`, `mulM!`, `mulMᵀ!`, `mulMᵀM!`, `mulMMᵀ!`,
`muldMdx!` to `rtol_mul`; `ldiv!(x, model, b)` (and `ldiv!(x, model, b, P)` when a `SymmetricKPMPreconditioner` of this model is
given) to `rtol_solve`, the solver's tolerance tightened to `solver_tol` for the duration so that both solutions sit at the fixed point.
Use a small model (4 × 4 sites, Lτ ≈ 20): the detached pass is the CPU code.  The model's phonon field, solver settings and attachment
(same device, same options) are as before on return; `model.rng` advances only if `P` is given (two `setup!`s).  Throws an
`ErrorException` naming the first quantity out of tolerance; returns the measured errors and iteration counts otherwise.

What a failure means: a MethodError in the detached pass — an `invoke` tuple of this file does not select the reference's method (the
static check tests/test_julia_dispatch.py restates Julia's rule and may be wrong where this run is right); a numerical mismatch in
the attached pass — the layout handed to `elph_create` (checkerboard order of `neighbor_table`, `cosht`/`sinht`) is not what this
version of the reference's `initialize_model!` produces.
"""
function selftest(model::GPUModel; P=I, rtol_mul::Float64=1e-12, rtol_solve::Float64=1e-10, solver_tol::Float64=1e-13, verbose::Bool=true)
    n = model.Ndim
    rng = MersenneTwister(20260131)
    v = randn(rng, n); u = randn(rng, n); b = randn(rng, n)
    e0 = entry(model)
    opts = e0 === nothing ? (device=0, host_sync=true, batch_pseudofermions=true, resident_hmc=false) :
           (device=e0.device, host_sync=e0.host_sync, batch_pseudofermions=e0.batch_pseudofermions, resident_hmc=e0.resident_hmc)
    saved = (model.mul_by_M, model.transposed, model.solver.tol)
    model.mul_by_M = false; model.transposed = false; model.solver.tol = solver_tol
    with_P = P isa SymmetricKPMPreconditioner
    function pass()
        update_model!(model)
        yM = zeros(n); mulM!(yM, model, v)
        yMᵀ = zeros(n); mulMᵀ!(yMᵀ, model, v)
        yMᵀM = zeros(n); mulMᵀM!(yMᵀM, model, v)
        yMMᵀ = zeros(n); mulMMᵀ!(yMMᵀ, model, v)
        d = zeros(model isa HolsteinModel ? n : model.Ndof); muldMdx!(d, u, model, v)
        x = zeros(n); it, _, fl = ldiv!(x, model, b)
        xP = zeros(n); itP = 0; flP = 0
        if with_P
            setup!(P)
            itP, _, flP = ldiv!(xP, model, b, P)
        end
        return (yM=yM, yMᵀ=yMᵀ, yMᵀM=yMᵀM, yMMᵀ=yMMᵀ, dMdx=d, x=x, iters=it, flag=fl, xP=xP, itersP=itP, flagP=flP)
    end
    local g, c, served
    try
        e0 === nothing && attach!(model)
        g = pass()
        served = status(model)
        detach!(model)
        c = pass()                          # every call: added method -> registry miss -> invoke -> the reference's method
    finally
        model.mul_by_M, model.transposed, model.solver.tol = saved
        attached(model) && e0 === nothing && detach!(model)
        if e0 !== nothing && !attached(model)
            attach!(model; opts...)
        end
    end
    errs = (mulM=relerr(g.yM, c.yM), mulMᵀ=relerr(g.yMᵀ, c.yMᵀ), mulMᵀM=relerr(g.yMᵀM, c.yMᵀM), mulMMᵀ=relerr(g.yMMᵀ, c.yMMᵀ),
            muldMdx=relerr(g.dMdx, c.dMdx), ldiv=relerr(g.x, c.x), ldiv_P=with_P ? relerr(g.xP, c.xP) : 0.0,
            iters=(g.iters, c.iters), iters_P=(g.itersP, c.itersP), flags=(g.flag, c.flag, g.flagP, c.flagP), device_calls=served[1])
    verbose && @info "ElPhGPU.selftest" errs build=build_info()
    served[1] >= 7 || error("selftest: the attached pass was not served by the library (device calls: $(served[1]))")
    for k in (:mulM, :mulMᵀ, :mulMᵀM, :mulMMᵀ, :muldMdx)
        errs[k] <= rtol_mul || error("selftest: $k differs between the library and the reference's method: $(errs[k]) > $rtol_mul")
    end
    all(iszero, errs.flags) || error("selftest: a solve did not converge (flags $(errs.flags)); use a smaller model or a looser solver_tol")
    errs.ldiv <= rtol_solve || error("selftest: ldiv! solutions differ: $(errs.ldiv) > $rtol_solve (iterations $(errs.iters))")
    errs.ldiv_P <= rtol_solve || error("selftest: preconditioned ldiv! solutions differ: $(errs.ldiv_P) > $rtol_solve (iterations $(errs.iters_P))")
    return errs
end

end # module
