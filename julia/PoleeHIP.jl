# PoleeHIP.jl -- thin ccall layer over libpolee_hip.so (include/polee_hip.h).
#
# This is the reference-side binding a Polee maintainer would add: it keeps the Julia
# function API of the hot path (names and argument order of src/ptt.jl, src/likelihood.jl,
# src/likelihood-approximation.jl, src/approx-sampler.jl) and routes it to the MI355X
# kernels, replacing PyCall + TensorFlow + hsb_ops.so.  NOT TESTED in the build container
# (no Julia there); the Python mirror polee_amd/core.py exercises the same C entry points.
module PoleeHIP

const LIB = get(ENV, "POLEE_HIP_LIB", joinpath(@__DIR__, "..", "polee_amd", "csrc", "libpolee_hip.so"))

struct PoleeHIPError <: Exception
    status::Cint
    msg::String
end

function check(status::Cint, ctx::Ptr{Cvoid}=C_NULL)
    status == 0 && return
    msg = unsafe_string(ccall((:polee_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    # POLEE_ERR_NONFINITE (4) mirrors `@assert isfinite(...)` (likelihood-approximation.jl:559)
    status == 4 ? throw(AssertionError(msg)) : throw(PoleeHIPError(status, msg))
end

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:polee_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[])
        finalizer(c -> ccall((:polee_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.h), c)
        return c
    end
end

# ---- PolyaTreeTransform (src/ptt.jl:6-27, 89-116) ------------------------------------
mutable struct PolyaTreeTransform
    h::Ptr{Cvoid}
    ctx::Context
    n::Int
    function PolyaTreeTransform(ctx::Context, parent_idxs::Vector{Int32}, output_idxs::Vector{Int32})
        @assert length(parent_idxs) == length(output_idxs)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve parent_idxs output_idxs check(
            ccall((:polee_ptt_create, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                  ctx.h, parent_idxs, output_idxs, length(parent_idxs), r), ctx.h)
        t = new(r[], ctx, div(length(parent_idxs) + 1, 2))
        finalizer(t -> ccall((:polee_ptt_destroy, LIB), Cvoid, (Ptr{Cvoid},), t.h), t)
        return t
    end
end

"transform!(t, ys, xs, Val(compute_ladj)) -- src/ptt.jl:125-160"
function transform!(t::PolyaTreeTransform, ys::Vector{Float64}, xs::Vector{Float32},
                    ::Val{compute_ladj}=Val(false)) where {compute_ladj}
    ladj = Ref{Float64}(0.0)
    GC.@preserve ys xs check(
        ccall((:polee_ptt_transform, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float32}, Ptr{Float64}),
              t.h, ys, 1, xs, compute_ladj ? ladj : C_NULL), t.ctx.h)
    return ladj[]
end

"transform_gradients!(t, ys, y_grad, x_grad) -- src/ptt.jl:167-209"
function transform_gradients!(t::PolyaTreeTransform, ys::Vector{Float64}, y_grad::AbstractVector,
                              x_grad::Vector{Float64}; with_ladj::Bool=true)
    tmp = Vector{Float64}(undef, t.n - 1)
    GC.@preserve ys x_grad tmp check(
        ccall((:polee_ptt_transform_gradients, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32, Cint, Ptr{Float64}),
              t.h, ys, x_grad, 1, with_ladj, tmp), t.ctx.h)
    y_grad .= tmp   # the reference's y_grad is Float32 (likelihood-approximation.jl:466)
    return nothing
end
transform_gradients_no_ladj!(t, ys, y_grad, x_grad) = transform_gradients!(t, ys, y_grad, x_grad, with_ladj=false)

"inverse_transform!(t, xs, ys) -- src/ptt.jl:257-285"
function inverse_transform!(t::PolyaTreeTransform, xs::Vector{Float32}, ys::Vector{Float64})
    ladj = Ref{Float64}(0.0)
    GC.@preserve xs ys check(
        ccall((:polee_ptt_inverse_transform, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float64}, Ptr{Float64}),
              t.h, xs, 1, ys, ladj), t.ctx.h)
    return ladj[]
end

# ---- X + log_likelihood (src/likelihood.jl:2-56, src/sparse.jl) -----------------------
mutable struct DeviceSample
    h::Ptr{Cvoid}
    ctx::Context
    m::Int
    n::Int
    "X::SparseMatrixCSC{Float32,UInt32} exactly as RNASeqSample holds it (src/rnaseq_sample.jl:11)"
    function DeviceSample(ctx::Context, m, n, colptr::Vector{UInt32}, rowval::Vector{UInt32},
                          nzval::Vector{Float32}; ks::Union{Nothing,Vector{Int64}}=nothing)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve colptr rowval nzval ks check(
            ccall((:polee_loglik_create, LIB), Cint,
                  (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{Float32}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                  ctx.h, m, n, colptr, 4, rowval, nzval, ks === nothing ? C_NULL : ks, r), ctx.h)
        s = new(r[], ctx, m, n)
        finalizer(s -> ccall((:polee_loglik_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.h), s)
        return s
    end
end

"log_likelihood(..., xs, x_grad, Val(gradonly)) -- src/likelihood.jl:36-56 (frag_probs scratch lives on the GPU)"
function log_likelihood(s::DeviceSample, xs::Vector{Float32}, x_grad::Vector{Float64},
                        ::Val{gradonly}) where {gradonly}
    lp = Ref{Float64}(0.0)
    GC.@preserve xs x_grad check(
        ccall((:polee_loglik_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float64}, Ptr{Float64}),
              s.h, xs, 1, x_grad, gradonly ? C_NULL : lp), s.ctx.h)
    return lp[]
end

# ---- approximate_likelihood (src/likelihood-approximation.jl:395-624) -----------------
# Mirrors `struct polee_vi_opts`; obtain defaults with polee_vi_default_opts.
mutable struct ViOpts
    num_steps::Int32; num_mc_samples::Int32; use_efflen_jacobian::Int32; gradonly::Int32
    seed::UInt64; z0::Ptr{Float32}; y_eps::Float64
    adam_initial_learning_rate::Float64; adam_learning_rate_decay::Float64; adam_min_learning_rate::Float64
    adam_eps::Float64; adam_rv::Float64; adam_rm::Float64
    max_mu_step::Float64; max_omega_step::Float64; max_alpha_step::Float64
    profile::Int32; deterministic::Int32
    gene_of::Ptr{Int32}   # optional: gene index of every transcript (0-based, -1 = none known) = gene_noninformative
    ViOpts() = (o = new(); ccall((:polee_vi_default_opts, LIB), Cvoid, (Ref{ViOpts},), o); o)
end

"""
approximate_likelihood(::LogitSkewNormalPTTApprox, sample) replacement: returns the params Dict
("mu", "omega", "alpha") exactly as likelihood-approximation.jl:615-623 does.
"""
function approximate_likelihood(s::DeviceSample, t::PolyaTreeTransform, efflens::Vector{Float32};
                                use_efflen_jacobian::Bool=true, seed::Integer=123456789,
                                gene_transcripts::Union{Nothing,Dict{String,Vector{Int}}}=nothing,  # gene_noninformative
                                deterministic::Bool=false)
    o = ViOpts(); o.use_efflen_jacobian = use_efflen_jacobian; o.seed = seed; o.deterministic = deterministic
    # the reference's Dict{gene id -> transcript indexes} (likelihood-approximation.jl:475-487) as gene_of[n]
    gene_of = Int32[]
    if gene_transcripts !== nothing && !isempty(gene_transcripts)
        gene_of = fill(Int32(-1), s.n)
        for (gi, idxs) in enumerate(values(gene_transcripts)), i in idxs
            gene_of[i] = gi - 1
        end
        o.gene_of = pointer(gene_of)
    end
    mu = Vector{Float32}(undef, s.n - 1); omega = similar(mu); alpha = similar(mu)
    GC.@preserve efflens mu omega alpha gene_of check(
        ccall((:polee_vi_fit, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ref{ViOpts}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
              s.h, t.h, efflens, o, mu, omega, alpha, C_NULL), s.ctx.h)
    return Dict{String,Vector}("mu" => mu, "omega" => omega, "alpha" => alpha)
end

"rand!(als, xs) -- src/approx-sampler.jl:37-44"
function rand_draws!(t::PolyaTreeTransform, mu::Vector{Float32}, sigma::Vector{Float32}, alpha::Vector{Float32},
                     xs::Matrix{Float32}; seed::Integer=rand(UInt64))   # xs is n x ndraws (column = one draw)
    GC.@preserve mu sigma alpha xs check(
        ccall((:polee_sampler_draw, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, UInt64, Ptr{Float32}),
              t.h, mu, sigma, alpha, C_NULL, size(xs, 2), seed, xs), t.ctx.h)
    return xs
end

"x0 of load_samples_hdf5 -- src/estimate.jl:436-455: mean of N draws (y clamped, / efflens, renormalised)"
function initial_values(t::PolyaTreeTransform, mu::Vector{Float32}, sigma::Vector{Float32}, alpha::Vector{Float32},
                        efflens::Vector{Float32}, N::Integer=30; seed::Integer=rand(UInt64))
    x0 = Vector{Float32}(undef, length(mu) + 1)
    GC.@preserve mu sigma alpha efflens x0 check(
        ccall((:polee_sampler_initial_values, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, UInt64, Ptr{Float32}),
              t.h, mu, sigma, alpha, efflens, C_NULL, N, seed, x0), t.ctx.h)
    return x0
end

"one sample of posterior_mean(loaded_samples, N) -- src/approx-sampler.jl:86-117"
function posterior_mean(t::PolyaTreeTransform, mu::Vector{Float32}, sigma::Vector{Float32}, alpha::Vector{Float32},
                        N::Integer=100; seed::Integer=rand(UInt64))
    pm = Vector{Float32}(undef, length(mu) + 1)
    GC.@preserve mu sigma alpha pm check(
        ccall((:polee_sampler_posterior_mean, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, UInt64, Ptr{Float32}),
              t.h, mu, sigma, alpha, C_NULL, N, seed, pm), t.ctx.h)
    return pm
end

"one sample of Statistics.quantile(loaded_samples, transforms, qs, N) -- src/approx-sampler.jl:50-83; n x length(qs)"
function quantiles(t::PolyaTreeTransform, mu::Vector{Float32}, sigma::Vector{Float32}, alpha::Vector{Float32},
                   qs::Vector{Float64}=[0.01, 0.99], N::Integer=100; seed::Integer=rand(UInt64))
    out = Matrix{Float32}(undef, length(mu) + 1, length(qs))
    GC.@preserve mu sigma alpha qs out check(
        ccall((:polee_sampler_quantiles, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, UInt64, Ptr{Float64}, Int32,
               Ptr{Float32}),
              t.h, mu, sigma, alpha, C_NULL, N, seed, qs, length(qs), out), t.ctx.h)
    return out
end

"effective_length_jacobian_adjustment!(efflens, xs, xls, x_grad) -- src/likelihood.jl:93-110"
function effective_length_jacobian_adjustment!(ctx::Context, efflens::Vector{Float32}, xs::Vector{Float32},
                                               xls::Vector{Float32}, x_grad::Vector{Float64})
    GC.@preserve efflens xs xls x_grad check(
        ccall((:polee_efflen_jacobian_adjustment, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int32, Int64, Ptr{Float64}, Ptr{Float32}),
              ctx.h, efflens, xs, 1, length(xs), x_grad, xls), ctx.h)
    return 0.0
end

"gene_noninformative_prior!(efflens, xls, xl_grad, xs, x_grad, gene_transcripts) -- src/likelihood.jl:114-159"
function gene_noninformative_prior!(ctx::Context, efflens::Vector{Float32}, xls::Vector{Float32}, xs::Vector{Float32},
                                    x_grad::Vector{Float64}, gene_transcripts::Dict{String, Vector{Int}})
    gene_of = fill(Int32(-1), length(xs))
    for (gi, idxs) in enumerate(values(gene_transcripts)), i in idxs
        gene_of[i] = Int32(gi - 1)
    end
    GC.@preserve efflens xls xs x_grad gene_of check(
        ccall((:polee_gene_noninformative_prior, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int32, Int64, Ptr{Int32}, Ptr{Float64}),
              ctx.h, efflens, xls, xs, 1, length(xs), gene_of, x_grad), ctx.h)
    return 0.0
end

# ---- one sample over several GPUs (one Julia process per GPU) -------------------------
"128-byte id for polee_comm_create; rank 0 creates it and the caller broadcasts it (MPI.jl, a file, ...)"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(ccall((:polee_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end

mutable struct Comm
    h::Ptr{Cvoid}
    ctx::Context
    function Comm(ctx::Context, nranks::Integer, rank::Integer, id::Vector{UInt8})
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve id check(ccall((:polee_comm_create, LIB), Cint,
                                    (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
                                    ctx.h, nranks, rank, id, out), ctx.h)
        c = new(out[], ctx)
        finalizer(x -> ccall((:polee_comm_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), c)
        return c
    end
end

end # module
