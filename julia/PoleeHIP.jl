# PoleeHIP.jl -- thin ccall layer over libpolee_hip.so (include/polee_hip.h).
#
# This is the reference-side binding a Polee maintainer would add: it keeps the Julia
# function API of the hot path (names and argument order of src/ptt.jl, src/likelihood.jl,
# src/likelihood-approximation.jl, src/approx-sampler.jl) and routes it to the MI355X
# kernels, replacing PyCall + TensorFlow + hsb_ops.so.  NOT EXECUTED in the build container
# (no Julia there); the Python mirror polee_amd/core.py exercises the same C entry points, and
# julia/runtests.jl is the test-suite a box with Julia + an MI355X runs (it reads the reference's own
# HDF5 fixtures, test/dataset/mBr_M_6w_1.*.h5).
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
                                deterministic::Union{Nothing,Bool}=nothing)
    # (polee_vi_opts.deterministic: 0 = the library's rule, on exactly when the sample is shared by more than one rank; 1 on; -1 off)
    o = ViOpts(); o.use_efflen_jacobian = use_efflen_jacobian; o.seed = seed
    o.deterministic = deterministic === nothing ? 0 : (deterministic ? 1 : -1)
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

# ---- the VI loop as a handle (polee_vi_create / run / sync): step the fit, inspect it, shard it ----------------
mutable struct ViStats
    steps_done::Int32
    nonfinite_step::Int32
    loglik_kernel_ms_avg::Float64
    loglik_kernel_launches::Int64
    last_elbo::Float64
    last_lp_mean::Float64
    loglik_pass_ms_avg::Float64
    ViStats() = new(0, 0, 0.0, 0, 0.0, 0.0, 0.0)
end

"State of one fit of approximate_likelihood(::LogitSkewNormalPTTApprox, ...) (likelihood-approximation.jl:395-624)"
mutable struct LikelihoodApproximationFit
    h::Ptr{Cvoid}
    sample::DeviceSample
    t::PolyaTreeTransform
    function LikelihoodApproximationFit(s::DeviceSample, t::PolyaTreeTransform, efflens::Vector{Float32}, opts::ViOpts=ViOpts())
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve efflens check(ccall((:polee_vi_create, LIB), Cint,
                                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ref{ViOpts}, Ref{Ptr{Cvoid}}),
                                         s.h, t.h, efflens, opts, out), t.ctx.h)
        f = new(out[], s, t)
        finalizer(x -> ccall((:polee_vi_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), f)
        return f
    end
end

"enqueue nsteps iterations (no host synchronisation)"
run!(f::LikelihoodApproximationFit, nsteps::Integer) =
    check(ccall((:polee_vi_run, LIB), Cint, (Ptr{Cvoid}, Int32), f.h, nsteps), f.t.ctx.h)
"wait for the stream; throws AssertionError on a non-finite gradient (likelihood-approximation.jl:559)"
sync!(f::LikelihoodApproximationFit) = check(ccall((:polee_vi_sync, LIB), Cint, (Ptr{Cvoid},), f.h), f.t.ctx.h)

function params(f::LikelihoodApproximationFit)
    n = f.t.n
    mu, omega, alpha = (Vector{Float32}(undef, n - 1) for _ in 1:3)
    GC.@preserve mu omega alpha check(ccall((:polee_vi_get_params, LIB), Cint,
                                            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), f.h, mu, omega, alpha), f.t.ctx.h)
    return mu, omega, alpha
end

function set_params!(f::LikelihoodApproximationFit, mu::Vector{Float32}, omega::Vector{Float32}, alpha::Vector{Float32})
    GC.@preserve mu omega alpha check(ccall((:polee_vi_set_params, LIB), Cint,
                                            (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), f.h, mu, omega, alpha), f.t.ctx.h)
end

function stats(f::LikelihoodApproximationFit)
    st = ViStats()
    check(ccall((:polee_vi_get_stats, LIB), Cint, (Ptr{Cvoid}, Ref{ViStats}), f.h, st), f.t.ctx.h)
    return st
end

"per-step ELBO and mean log-likelihood (gradonly == 0 only)"
function trace(f::LikelihoodApproximationFit)
    k = Int(stats(f).steps_done)
    elbo, lp = Vector{Float64}(undef, k), Vector{Float64}(undef, k)
    GC.@preserve elbo lp check(ccall((:polee_vi_get_trace, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), f.h, elbo, lp), f.t.ctx.h)
    return elbo, lp
end

"row-sharded fit: this handle's sample holds one rank's block of fragments (SURVEY 8(e)(1))"
set_comm!(f::LikelihoodApproximationFit, c) =
    check(ccall((:polee_vi_set_comm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), f.h, c === nothing ? C_NULL : c.h), f.t.ctx.h)

# ---- tree construction: hclust + order_nodes (src/hclust.jl:193-319, 361-389) -----------------------------------
"X in CSC, 1-based (colptr UInt32 or UInt64), as in the likelihood-matrix HDF5 -> (node_parent_idxs, node_js)"
function hclust(m::Integer, n::Integer, colptr::Union{Vector{UInt32},Vector{UInt64}}, rowval::Vector{UInt32};
                parallel::Bool=false, device::Union{Nothing,Context}=nothing)
    parents, js = Vector{Int32}(undef, 2n - 1), Vector{Int32}(undef, 2n - 1)
    if device !== nothing   # the rounds variant built on the GPU: the same arrays as parallel = true
        GC.@preserve colptr rowval parents js check(
            ccall((:polee_hclust_parallel_device, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
                  device.h, m, n, colptr, sizeof(eltype(colptr)), rowval, parents, js), device.h)
    elseif parallel   # rounds of mutually-best merges on all host threads: a variant, not hclust.jl's tree node for node
        GC.@preserve colptr rowval parents js check(
            ccall((:polee_hclust_parallel, LIB), Cint, (Int64, Int64, Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
                  m, n, colptr, sizeof(eltype(colptr)), rowval, parents, js))
    else
        GC.@preserve colptr rowval parents js check(
            ccall((:polee_hclust, LIB), Cint, (Int64, Int64, Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
                  m, n, colptr, sizeof(eltype(colptr)), rowval, parents, js))
    end
    return parents, js
end

"release the scratch blocks the host-side builders keep between samples"
host_cache_trim() = ccall((:polee_host_cache_trim, LIB), Cvoid, ())

# ---- density of the fitted approximations: replaces create_tensorflow_variables! (src/estimate.jl:502-556) +
#      RNASeqApproxLikelihoodDist (src/polee_approx_likelihood.py:367-450) -------------------------------------------
"""
    ApproxLikelihood(ctx, vars)

`vars` is `LoadedSamples.variables` (estimate.jl:502-556): "efflen" [S,n], "la_mu" / "la_sigma" / "la_alpha" [S,n-1],
"left_index" / "right_index" / "leaf_index" [S,N] (or [1,N] for one shared tree), as ROW-MAJOR C arrays -- i.e. pass
`permutedims` of Julia's column-major matrices, or vectors already laid out sample by sample.
"""
mutable struct ApproxLikelihood
    h::Ptr{Cvoid}
    ctx::Context
    S::Int
    n::Int
    function ApproxLikelihood(ctx::Context, S::Integer, n::Integer, efflen::Vector{Float32}, la_mu::Vector{Float32},
                              la_sigma::Vector{Float32}, la_alpha::Vector{Float32}, left_index::Vector{Int32},
                              right_index::Vector{Int32}, leaf_index::Vector{Int32}; shared_tree::Bool=false)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve efflen la_mu la_sigma la_alpha left_index right_index leaf_index check(
            ccall((:polee_approx_create, LIB), Cint,
                  (Ptr{Cvoid}, Int32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Ptr{Int32},
                   Ptr{Int32}, Cint, Ref{Ptr{Cvoid}}),
                  ctx.h, S, n, efflen, la_mu, la_sigma, la_alpha, left_index, right_index, leaf_index, shared_tree, out), ctx.h)
        a = new(out[], ctx, S, n)
        finalizer(x -> ccall((:polee_approx_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), a)
        return a
    end
end

"log_prob(ap, x; grad) -- x Float32 [S*n] (sample by sample) unnormalised log expression -> lp [S] (, d lp / d x)"
function log_prob(ap::ApproxLikelihood, x::Vector{Float32}; grad::Bool=false)
    lp = Vector{Float32}(undef, ap.S)
    g = grad ? Vector{Float32}(undef, ap.S * ap.n) : Float32[]
    GC.@preserve x lp g check(ccall((:polee_approx_logprob, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                                    ap.h, x, lp, grad ? pointer(g) : C_NULL), ap.ctx.h)
    return grad ? (lp, g) : lp
end

"rnaseq_approx_likelihood_sampler (polee_approx_likelihood.py:35-59): one draw per sample -> x Float32 [S*n]"
function sample(ap::ApproxLikelihood; seed::Integer=0, z0::Union{Nothing,Vector{Float32}}=nothing)
    x = Vector{Float32}(undef, ap.S * ap.n)
    GC.@preserve x z0 check(ccall((:polee_approx_sample, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, UInt64, Ptr{Float32}),
                                  ap.h, z0 === nothing ? C_NULL : pointer(z0), seed, x), ap.ctx.h)
    return x
end

"RNASeqGeneApproxLikelihoodDist (polee_gene_expression.py:14-90); gene_of: 0-based gene of every transcript"
function gene_log_prob(ap::ApproxLikelihood, x_gene::Vector{Float32}, x_isoform::Vector{Float32}, gene_of::Vector{Int32},
                       num_genes::Integer; grad::Bool=false)
    lp = Vector{Float32}(undef, ap.S)
    gg = grad ? Vector{Float32}(undef, ap.S * num_genes) : Float32[]
    gi = grad ? Vector{Float32}(undef, ap.S * ap.n) : Float32[]
    GC.@preserve x_gene x_isoform gene_of lp gg gi check(
        ccall((:polee_approx_gene_logprob, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Int32}, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
              ap.h, x_gene, x_isoform, gene_of, num_genes, lp, grad ? pointer(gg) : C_NULL, grad ? pointer(gi) : C_NULL), ap.ctx.h)
    return grad ? (lp, gg, gi) : lp
end

"approximate_feature_likelihood (polee_gene_expression.py:191-222): (loc, scale) Float32 [S*F]; pairs are 1-based"
function feature_moments(ap::ApproxLikelihood, feature_idxs::Vector{Int32}, transcript_idxs::Vector{Int32}, F::Integer;
                         num_mean_draws::Integer=1000, num_var_draws::Integer=1000, seed::Integer=0)
    loc, scale = Vector{Float32}(undef, ap.S * F), Vector{Float32}(undef, ap.S * F)
    GC.@preserve feature_idxs transcript_idxs loc scale check(
        ccall((:polee_approx_feature_moments, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Int64, Int32, Int32, Int32, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
              ap.h, feature_idxs, transcript_idxs, length(feature_idxs), F, num_mean_draws, num_var_draws, seed, C_NULL, loc, scale),
        ap.ctx.h)
    return loc, scale
end

# ---- regression model: replaces the python-object protocol Regression(...).fit(niter)
#      (src/regression.jl:319-327, models/polee_regression.py:18-340) ------------------------------------------------
mutable struct Regression
    h::Ptr{Cvoid}
    ctx::Context
    S::Int
    F::Int
    n::Int
    """design Float32 [S*F], x_init Float32 [S*n] (row-major); hinges = nothing: choose_knots as
    RNASeqTranscriptLinearRegression does (models/polee_regression.py:436-440)"""
    function Regression(ctx::Context, ap::Union{Nothing,ApproxLikelihood}, S::Integer, F::Integer, n::Integer,
                        design::Vector{Float32}, x_init::Vector{Float32}, sample_scales::Vector{Float32};
                        x_init_mean::Union{Nothing,Vector{Float32}}=nothing, hinges::Union{Nothing,Vector{Float32}}=nothing,
                        degree::Integer=15, bandwidth::Real=1.0, x_bias_loc0::Real=log(1 / n), x_bias_scale0::Real=12.0,
                        use_distortion::Bool=true, scale_penalty::Real=1.0, use_point_estimates::Bool=false)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve design x_init sample_scales x_init_mean hinges check(
            ccall((:polee_regression_create, LIB), Cint,
                  (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32},
                   Int32, Cfloat, Cfloat, Cfloat, Cint, Cfloat, Cint, Ref{Ptr{Cvoid}}),
                  ctx.h, ap === nothing ? C_NULL : ap.h, S, F, n, design, x_init,
                  x_init_mean === nothing ? C_NULL : pointer(x_init_mean), sample_scales,
                  hinges === nothing ? C_NULL : pointer(hinges), degree, bandwidth, x_bias_loc0, x_bias_scale0, use_distortion,
                  scale_penalty, use_point_estimates, out), ctx.h)
        r = new(out[], ctx, S, F, n)
        finalizer(x -> ccall((:polee_regression_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), r)
        return r
    end
end

num_params(r::Regression) = Int(ccall((:polee_regression_num_params, LIB), Int64, (Ptr{Cvoid},), r.h))

"fit(niter) (models/polee_regression.py:303-340); returns the loss trace"
function fit!(r::Regression, niter::Integer; seed::Integer=0)
    losses = Vector{Float32}(undef, niter)
    GC.@preserve losses check(ccall((:polee_regression_fit, LIB), Cint, (Ptr{Cvoid}, Int32, UInt64, Ptr{Float32}, Ptr{Float32}),
                                    r.h, niter, seed, C_NULL, losses), r.ctx.h)
    return losses
end

"the flat parameter vector (order documented in include/polee_hip.h)"
function flat_params(r::Regression)
    p = Vector{Float32}(undef, num_params(r))
    GC.@preserve p check(ccall((:polee_regression_get_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, p), r.ctx.h)
    return p
end

"""
(qx_loc[S,n], qw_loc[F,n], qw_scale[F,n], qx_bias[n], qx_scale[n]) as `Regression(...).fit(niter)` returns them
(src/regression.jl:319-327): slices of the flat vector, softplus applied where the reference applies it.
"""
function fit_results(r::Regression; degree::Integer=15)
    p = flat_params(r)
    S, F, n = r.S, r.F, r.n
    softplus(x) = log1p(exp(x))
    o = 4 + F * degree + 2 * degree                     # scalars, distortion and mean-variance coefficients
    blk(k) = reshape(p[o + k * F * n + 1:o + (k + 1) * F * n], n, F)'   # k-th [F][n] array (row-major) as an F x n matrix
    qw_loc, qw_scale = blk(8), softplus.(blk(9))
    o += 10 * F * n
    qx_bias = p[o + 1:o + n]
    qx_scale = softplus.(p[o + 2n + 1:o + 3n])
    o += 4n
    qx_loc = reshape(p[o + 1:o + S * n], n, S)'
    return qx_loc, qw_loc, qw_scale, qx_bias, qx_scale
end

"samples sharded over ranks: one all-reduce of (F+2) n statistics per step (SURVEY 8(e)(2))"
set_comm!(r::Regression, c) =
    check(ccall((:polee_regression_set_comm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), r.h, c === nothing ? C_NULL : c.h), r.ctx.h)

"""
gene-level likelihood on a model created over the genes (ap = nothing): RNASeqGeneLinearRegression
(models/polee_regression.py:533-600), or with `F_isoform` (S x Fi) RNASeqGeneIsoformLinearRegression (:656-877).
`gene_of`: 1-based gene of every transcript; `x_isoform_init`: S x nt.
"""
function set_gene_likelihood!(r::Regression, ap::ApproxLikelihood, gene_of::AbstractVector{<:Integer},
                              x_isoform_init::AbstractMatrix; F_isoform::Union{Nothing,AbstractMatrix}=nothing)
    g = Int32.(gene_of .- 1)
    xi = Matrix{Float32}(x_isoform_init')            # row-major [S][nt]
    if F_isoform === nothing
        GC.@preserve g xi check(ccall((:polee_regression_set_gene_likelihood, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}), r.h, ap.h, g, xi), r.ctx.h)
    else
        Fi = Matrix{Float32}(F_isoform')             # row-major [S][Fi]
        GC.@preserve g xi Fi check(ccall((:polee_regression_set_gene_isoform_likelihood, LIB), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Int32),
            r.h, ap.h, g, xi, Fi, Int32(size(F_isoform, 2))), r.ctx.h)
    end
    return r
end

"""
RNASeqJointLinearRegression (models/polee_regression.py:879-1283) on a model created over the genes (ap = nothing, no
distortion, scale penalty 5e-4): gene block with a one-level horseshoe prior and kernel-regression weights of the sampled
bias, plus a regression over `num_splice_features` splice features that reaches the transcripts through the 0/1 matrix given
by the 1-based pairs (`feature_is`, `feature_js`) -- the arguments of models/joint-regression.jl.
"""
function set_joint_likelihood!(r::Regression, ap::ApproxLikelihood, gene_of::AbstractVector{<:Integer},
                               x_isoform_init::AbstractMatrix, num_splice_features::Integer,
                               feature_is::AbstractVector{<:Integer}, feature_js::AbstractVector{<:Integer})
    length(feature_is) == length(feature_js) || throw(ArgumentError("feature_is and feature_js must pair up"))
    g = Int32.(gene_of .- 1)
    xi = Matrix{Float32}(x_isoform_init')            # row-major [S][nt]
    pt = Int32.(feature_is .- 1)
    pf = Int32.(feature_js .- 1)
    GC.@preserve g xi pt pf check(ccall((:polee_regression_set_joint_likelihood, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Int32, Ptr{Int32}, Ptr{Int32}, Int64),
        r.h, ap.h, g, xi, Int32(num_splice_features), pt, pf, Int64(length(pt))), r.ctx.h)
    return r
end

"Adam step size of fit (2e-3 as created; the joint model sets 1e-3, models/polee_regression.py:1222)"
set_learning_rate!(r::Regression, lr::Real) =
    check(ccall((:polee_regression_set_learning_rate, LIB), Cint, (Ptr{Cvoid}, Cfloat), r.h, Float32(lr)), r.ctx.h)

"the isoform block of a gene-level / gene-isoform model as a flat vector (order: include/polee_hip.h)"
function isoform_params(r::Regression)
    n = Int(ccall((:polee_regression_num_isoform_params, LIB), Int64, (Ptr{Cvoid},), r.h))
    p = Vector{Float32}(undef, n)
    GC.@preserve p check(ccall((:polee_regression_get_isoform_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, p), r.ctx.h)
    return p
end

# ---- host-staged communicator: the all-reduce is the caller's (e.g. MPI.Allreduce!) ----------------------------------
mutable struct HostComm
    h::Ptr{Cvoid}
    ctx::Context
    allreduce::Function     # (buf::Vector{Float32} or Vector{Float64}) -> sums it over the ranks in place
    cfun::Base.CFunction
    function HostComm(ctx::Context, nranks::Integer, rank::Integer, allreduce::Function)
        cb = function (user::Ptr{Cvoid}, buf::Ptr{Cvoid}, count::Int64, is_f64::Cint)::Cint
            try
                a = is_f64 != 0 ? unsafe_wrap(Array, Ptr{Float64}(buf), count) : unsafe_wrap(Array, Ptr{Float32}(buf), count)
                allreduce(a)
                return Cint(0)
            catch
                return Cint(1)
            end
        end
        cfun = @cfunction($cb, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint))
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:polee_comm_create_host, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                    ctx.h, nranks, rank, cfun, C_NULL, out), ctx.h)
        c = new(out[], ctx, allreduce, cfun)
        finalizer(x -> ccall((:polee_comm_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), c)
        return c
    end
end


# =====================================================================================================================
# Round 4: the rest of the C ABI (every export of libpolee_hip.so now has a wrapper; julia/runtests.jl exercises them
# against the golden fixtures).  Reference call sites in the docstrings.
# =====================================================================================================================

version() = unsafe_string(ccall((:polee_version, LIB), Cstring, ()))
synchronize(ctx::Context) = check(ccall((:polee_ctx_synchronize, LIB), Cint, (Ptr{Cvoid},), ctx.h), ctx.h)
"the context's hipStream_t (for callers that enqueue their own HIP work behind the library's)"
stream(ctx::Context) = ccall((:polee_ctx_stream, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), ctx.h)
"(free, total) bytes of the context's GPU"
function mem_info(ctx::Context)
    f, t = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:polee_ctx_mem_info, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), ctx.h, f, t), ctx.h)
    return f[], t[]
end
"HIP-event stopwatch on the context's stream"
timer_start(ctx::Context) = check(ccall((:polee_ctx_timer_start, LIB), Cint, (Ptr{Cvoid},), ctx.h), ctx.h)
function timer_stop(ctx::Context)
    ms = Ref{Float64}(0.0)
    check(ccall((:polee_ctx_timer_stop, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), ctx.h, ms), ctx.h)
    return ms[]
end
"cap (MB) of the host builders' scratch-block cache; cap_mb < 0 only queries.  Returns the cap in force."
host_cache_configure(cap_mb::Integer=-1) = ccall((:polee_host_cache_configure, LIB), Int64, (Int64,), cap_mb)
host_cache_bytes() = ccall((:polee_host_cache_bytes, LIB), Int64, ())
"device bytes the library keeps for its next allocations (host_cache_trim() frees them too)"
device_cache_bytes() = ccall((:polee_device_cache_bytes, LIB), Int64, ())

# ---- trees from the TF-side index arrays; the three TF custom ops (src/tensorflow_ext/hsb_ops.cpp) -------------------
"make_inverse_ptt_params(t) -- src/estimate.jl:461-500: (left_index, right_index, leaf_index), 0-based, -1 = none"
function make_inverse_ptt_params(parent_idxs::Vector{Int32}, output_idxs::Vector{Int32})
    N = length(parent_idxs)
    l, r, f = (Vector{Int32}(undef, N) for _ in 1:3)
    GC.@preserve parent_idxs output_idxs l r f check(
        ccall((:polee_make_inverse_ptt_params, LIB), Cint, (Ptr{Int32}, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
              parent_idxs, output_idxs, N, l, r, f))
    return l, r, f
end

"a tree handle from the index arrays the TF ops take (hsb_ops.cpp:17-60)"
function tree_from_index(ctx::Context, left_index::Vector{Int32}, right_index::Vector{Int32}, leaf_index::Vector{Int32})
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve left_index right_index leaf_index check(
        ccall((:polee_ptt_create_from_index, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
              ctx.h, left_index, right_index, leaf_index, length(left_index), out), ctx.h)
    t = PolyaTreeTransformHandle(out[], ctx)
    return t
end
"a bare tree handle (trees created from index arrays have no parent / js arrays on the Julia side)"
mutable struct PolyaTreeTransformHandle
    h::Ptr{Cvoid}
    ctx::Context
    function PolyaTreeTransformHandle(h::Ptr{Cvoid}, ctx::Context)
        t = new(h, ctx)
        finalizer(x -> ccall((:polee_ptt_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), t)
        return t
    end
end
const AnyTree = Union{PolyaTreeTransform,PolyaTreeTransformHandle}
num_leaves(t::AnyTree) = Int(ccall((:polee_ptt_n, LIB), Int32, (Ptr{Cvoid},), t.h))

"TF op HSB (hsb_ops.cpp:62-121): y_logit Float32 [B*(n-1)] row-major -> x Float32 [B*n]"
function hsb(t::AnyTree, y_logit::Vector{Float32}, B::Integer=1)
    x = Vector{Float32}(undef, B * num_leaves(t))
    GC.@preserve y_logit x check(ccall((:polee_hsb, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float32}), t.h, y_logit, B, x), t.ctx.h)
    return x
end
"TF op InvHSB (hsb_ops.cpp:123-250): x [B*n] -> (y Float64 [B*(n-1)], ladj Float32 [B])"
function inv_hsb(t::AnyTree, x::Vector{Float32}, B::Integer=1)
    y, ladj = Vector{Float64}(undef, B * (num_leaves(t) - 1)), Vector{Float32}(undef, B)
    GC.@preserve x y ladj check(ccall((:polee_inv_hsb, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float64}, Ptr{Float32}),
                                      t.h, x, B, y, ladj), t.ctx.h)
    return y, ladj
end
"TF op InvHSBGrad (hsb_ops.cpp:252-402): -> backprops Float32 [B*n]"
function inv_hsb_grad(t::AnyTree, y_grad::Vector{Float64}, ladj_grad::Vector{Float32}, y::Vector{Float64}, B::Integer=1)
    bp = Vector{Float32}(undef, B * num_leaves(t))
    GC.@preserve y_grad ladj_grad y bp check(
        ccall((:polee_inv_hsb_grad, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float32}, Ptr{Float64}, Int32, Ptr{Float32}),
              t.h, y_grad, ladj_grad, y, B, bp), t.ctx.h)
    return bp
end

# ---- the reparameterisations as standalone calls (src/logitnormal.jl, src/sinh_arcsinh.jl, src/kumaraswamy.jl) -------
"logit_normal_transform!(mu, sigma, zs, ys, Val(compute_ladj)) -- src/logitnormal.jl:8-20"
function logit_normal_transform!(ctx::Context, mu::Vector{Float32}, sigma::Vector{Float32}, zs::Vector{Float32},
                                 ys::Vector{Float64}, ::Val{compute_ladj}=Val(false)) where {compute_ladj}
    ladj = Ref{Float64}(0.0)
    GC.@preserve mu sigma zs ys check(
        ccall((:polee_logit_normal_transform, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float64}, Ptr{Float64}),
              ctx.h, mu, sigma, zs, length(zs), ys, compute_ladj ? ladj : C_NULL), ctx.h)
    return ladj[]
end
"logit_normal_transform_gradients!(zs, ys, mu, sigma, y_grad, z_grad, mu_grad, sigma_grad) -- src/logitnormal.jl:38-55 (accumulates)"
function logit_normal_transform_gradients!(ctx::Context, zs::Vector{Float32}, ys::Vector{Float64}, sigma::Vector{Float32},
                                           y_grad::Vector{Float32}, z_grad::Union{Nothing,Vector{Float32}},
                                           mu_grad::Vector{Float32}, sigma_grad::Vector{Float32})
    GC.@preserve zs ys sigma y_grad z_grad mu_grad sigma_grad check(
        ccall((:polee_logit_normal_transform_gradients, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float64}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
              ctx.h, zs, ys, sigma, y_grad, length(zs), z_grad === nothing ? C_NULL : pointer(z_grad), mu_grad, sigma_grad), ctx.h)
    return nothing
end
"sinh_asinh_transform!(alpha, zs0, zs, Val(compute_ladj)) -- src/sinh_arcsinh.jl:10-23"
function sinh_asinh_transform!(ctx::Context, alpha::Vector{Float32}, zs0::Vector{Float32}, zs::Vector{Float32},
                               ::Val{compute_ladj}=Val(false)) where {compute_ladj}
    ladj = Ref{Float64}(0.0)
    GC.@preserve alpha zs0 zs check(
        ccall((:polee_sinh_asinh_transform, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float64}),
              ctx.h, alpha, zs0, length(zs0), zs, compute_ladj ? ladj : C_NULL), ctx.h)
    return ladj[]
end
"sinh_asinh_transform_gradients!(zs0, alpha, z_grad, alpha_grad) -- src/sinh_arcsinh.jl:29-38 (accumulates)"
function sinh_asinh_transform_gradients!(ctx::Context, zs0::Vector{Float32}, alpha::Vector{Float32}, z_grad::Vector{Float32},
                                         alpha_grad::Vector{Float32})
    GC.@preserve zs0 alpha z_grad alpha_grad check(
        ccall((:polee_sinh_asinh_transform_gradients, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}),
              ctx.h, zs0, alpha, z_grad, length(zs0), alpha_grad), ctx.h)
    return nothing
end
"kumaraswamy_transform!(as, bs, zs, ys, Val(compute_ladj)) -- src/kumaraswamy.jl:9-36"
function kumaraswamy_transform!(ctx::Context, as::Vector{Float32}, bs::Vector{Float32}, zs::Vector{Float32},
                                ys::Vector{Float64}, ::Val{compute_ladj}=Val(false)) where {compute_ladj}
    ladj = Ref{Float64}(0.0)
    GC.@preserve as bs zs ys check(
        ccall((:polee_kumaraswamy_transform, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float64}, Ptr{Float64}),
              ctx.h, as, bs, zs, length(zs), ys, compute_ladj ? ladj : C_NULL), ctx.h)
    return ladj[]
end
"kumaraswamy_transform_gradients!(zs, as, bs, y_grad, a_grad, b_grad) -- src/kumaraswamy.jl:39-71 (accumulates)"
function kumaraswamy_transform_gradients!(ctx::Context, zs::Vector{Float32}, as::Vector{Float32}, bs::Vector{Float32},
                                          y_grad::Vector{Float32}, a_grad::Vector{Float32}, b_grad::Vector{Float32})
    GC.@preserve zs as bs y_grad a_grad b_grad check(
        ccall((:polee_kumaraswamy_transform_gradients, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float32}),
              ctx.h, zs, as, bs, y_grad, length(zs), a_grad, b_grad), ctx.h)
    return nothing
end

# ---- the sample from the rows of X; layout information; deterministic mode ------------------------------------------
"X given by rows (Xt = SparseMatrixCSC(transpose(X)), likelihood-approximation.jl:407): tcolptr UInt64 [m+1] 1-based"
function DeviceSampleFromXt(ctx::Context, m, n, tcolptr::Vector{UInt64}, trowval::Vector{UInt32}, tnzval::Vector{Float32};
                            ks::Union{Nothing,Vector{Int64}}=nothing)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve tcolptr trowval tnzval ks check(
        ccall((:polee_loglik_create_from_xt, LIB), Cint,
              (Ptr{Cvoid}, Int64, Int64, Ptr{UInt64}, Ptr{UInt32}, Ptr{Float32}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
              ctx.h, m, n, tcolptr, trowval, tnzval, ks === nothing ? C_NULL : pointer(ks), r), ctx.h)
    return wrap_sample(r[], ctx, m, n)
end
"true: the device layout of this sample was built by the device builder (csrc/psell_device.hip); false: by the host builder"
built_on_device(s) = ccall((:polee_loglik_built_on_device, LIB), Cint, (Ptr{Cvoid},), s.h) != 0
"adopts a polee_loglik handle (DeviceSample's inner constructor always builds from CSC)"
function wrap_sample(h::Ptr{Cvoid}, ctx::Context, m, n)
    s = DeviceSampleHandle(h, ctx, Int(m), Int(n))
    finalizer(x -> ccall((:polee_loglik_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), s)
    return s
end
mutable struct DeviceSampleHandle
    h::Ptr{Cvoid}
    ctx::Context
    m::Int
    n::Int
end
const AnySample = Union{DeviceSample,DeviceSampleHandle}

"X by columns on the device, uploaded once for the tree and the layout (polee_devx_upload): `sample_from_devx`, `hclust_from_devx`"
mutable struct DeviceX
    h::Ptr{Cvoid}
    ctx::Context
    m::Int
    n::Int
    function DeviceX(ctx::Context, m, n, colptr::Union{Vector{UInt32},Vector{UInt64}}, rowval::Vector{UInt32}, nzval::Vector{Float32})
        r = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve colptr rowval nzval check(
            ccall((:polee_devx_upload, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{UInt32}, Ptr{Float32}, Ref{Ptr{Cvoid}}),
                  ctx.h, m, n, colptr, sizeof(eltype(colptr)), rowval, isempty(nzval) ? C_NULL : pointer(nzval), r), ctx.h)
        x = new(r[], ctx, m, n)
        finalizer(x -> ccall((:polee_devx_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h), x)
        return x
    end
end
"the values after the fact (DeviceX(..., nzval) with an empty nzval uploads colptr + rowval only: enough for the tree)"
upload_values!(x::DeviceX, nzval::Vector{Float32}) =
    GC.@preserve nzval check(ccall((:polee_devx_upload_values, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), x.h, nzval), x.ctx.h)
function sample_from_devx(ctx::Context, x::DeviceX; ks::Union{Nothing,Vector{Int64}}=nothing, nzval::Union{Nothing,Vector{Float32}}=nothing)
    r = Ref{Ptr{Cvoid}}(C_NULL)   # nzval: the values, if the handle has none yet (uploaded beside the layout's first kernels)
    GC.@preserve ks nzval check(ccall((:polee_loglik_create_from_devx, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                                      ctx.h, x.h, nzval === nothing ? C_NULL : nzval, ks === nothing ? C_NULL : ks, r), ctx.h)
    s = DeviceSampleHandle(r[], ctx, x.m, x.n)
    finalizer(s -> ccall((:polee_loglik_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.h), s)
    return s
end
function hclust_from_devx(ctx::Context, x::DeviceX)
    parents, js = Vector{Int32}(undef, 2 * x.n - 1), Vector{Int32}(undef, 2 * x.n - 1)
    GC.@preserve parents js check(ccall((:polee_hclust_parallel_device_from_devx, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}),
                                        ctx.h, x.h, parents, js), ctx.h)
    return parents, js
end

"mirror of `polee_loglik_info` (include/polee_hip.h): field order and types must stay in step with the header"
mutable struct LoglikInfo
    m::Int64; n::Int64; nnz::Int64; num_slices::Int64; num_tiles::Int64; padded_nnz::Int64
    device_bytes::Int64; stream_bytes::Int64; num_empty_rows::Int64; max_row_nnz::Int32; max_tile_cols::Int32
    stream_rows::NTuple{8,Int64}; stream_nnz::NTuple{8,Int64}; stream_tiles::NTuple{8,Int64}; stream_bytes_hbm::NTuple{8,Int64}
    dict_entries::Int64
    LoglikInfo() = new(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, ntuple(_ -> 0, 8), ntuple(_ -> 0, 8), ntuple(_ -> 0, 8), ntuple(_ -> 0, 8), 0)
end
function info(s::AnySample)
    i = LoglikInfo()
    check(ccall((:polee_loglik_get_info, LIB), Cint, (Ptr{Cvoid}, Ref{LoglikInfo}), s.h, i), s.ctx.h)
    return i
end
"bitwise reproducible gradient sums (fixed reduction order instead of float atomics)"
set_deterministic!(s::AnySample, on::Bool=true) =
    check(ccall((:polee_loglik_set_deterministic, LIB), Cint, (Ptr{Cvoid}, Cint), s.h, on), s.ctx.h)
"debug: every slice through the per-tile kernel (the second algorithm of the cross-check tests)"
debug_force_mixed!(s::AnySample, on::Bool=true) =
    check(ccall((:polee_debug_loglik_force_mixed, LIB), Cint, (Ptr{Cvoid}, Cint), s.h, on), s.ctx.h)

"K expression vectors at once: xs Float32 [n, K] (a column per draw) -> (lp [K], x_grad Float64 [n, K])"
function log_likelihood_batch(s::AnySample, xs::Matrix{Float32}; gradonly::Bool=false)
    K = size(xs, 2)
    g, lp = Matrix{Float64}(undef, size(xs, 1), K), Vector{Float64}(undef, K)
    GC.@preserve xs g lp check(
        ccall((:polee_loglik_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float64}, Ptr{Float64}),
              s.h, xs, K, g, gradonly ? C_NULL : pointer(lp)), s.ctx.h)
    return lp, g
end

"OptimizePTTApprox (likelihood-approximation.jl:149-242): maximum-likelihood point estimate -> (xs, zs)"
function optimize_ptt(s::AnySample, t::AnyTree, efflens::Vector{Float32}, num_steps::Integer=500)
    xs, zs = Vector{Float32}(undef, s.n), Vector{Float32}(undef, s.n - 1)
    GC.@preserve efflens xs zs check(
        ccall((:polee_optimize_ptt, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Int32, Ptr{Float32}, Ptr{Float32}),
              s.h, t.h, efflens, num_steps, xs, zs), s.ctx.h)
    return xs, zs
end

# ---- the VI handle's test hooks ---------------------------------------------------------------------------------------
"""
One iteration's quantities WITHOUT applying the update (the parity tests' hook): K draws of the current parameters ->
(xs [n,K], x_grad [n,K], y_grad [n-1,K], mu_grad, omega_grad, alpha_grad [n-1], lp [K], ladj [K]).
"""
function eval_gradients(f::LikelihoodApproximationFit, K::Integer)
    n = f.t.n
    xs, xg, yg = Matrix{Float32}(undef, n, K), Matrix{Float64}(undef, n, K), Matrix{Float64}(undef, n - 1, K)
    mg, og, ag = (Vector{Float32}(undef, n - 1) for _ in 1:3)
    lp, ladj = Vector{Float64}(undef, K), Vector{Float64}(undef, K)
    GC.@preserve xs xg yg mg og ag lp ladj check(
        ccall((:polee_vi_eval_gradients, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}, Ptr{Float64}),
              f.h, xs, xg, yg, mg, og, ag, lp, ladj), f.t.ctx.h)
    return xs, xg, yg, mg, og, ag, lp, ladj
end
"the N(0,1) draws iteration `step` uses (device RNG or the caller's table): Float32 [n-1, K]"
function export_noise(f::LikelihoodApproximationFit, step::Integer, K::Integer)
    z = Matrix{Float32}(undef, f.t.n - 1, K)
    GC.@preserve z check(ccall((:polee_vi_export_noise, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float32}), f.h, step, z), f.t.ctx.h)
    return z
end

# ---- communicators: what the transport reports; a host-side all-reduce through it ------------------------------------
const AnyComm = Union{Comm,HostComm}
comm_rank(c::AnyComm) = Int(ccall((:polee_comm_rank, LIB), Int32, (Ptr{Cvoid},), c.h))
comm_size(c::AnyComm) = Int(ccall((:polee_comm_size, LIB), Int32, (Ptr{Cvoid},), c.h))
"(transport, count, rank): transport 1 = RCCL (count / rank = ncclCommCount / ncclCommUserRank), 2 = host-staged"
function comm_info(c::AnyComm)
    t, n, r = Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:polee_comm_info, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), c.h, t, n, r), c.ctx.h)
    return Int(t[]), Int(n[]), Int(r[])
end
"sum of a Float32 vector over the ranks, in place"
function allreduce_sum!(c::AnyComm, buf::Vector{Float32})
    GC.@preserve buf check(ccall((:polee_allreduce_sum_f32, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), c.h, buf, length(buf)), c.ctx.h)
    return buf
end

# ---- density of the approximations on device buffers; splicing moments ------------------------------------------------
"as log_prob with DEVICE pointers (Ptr{Float32} into memory of the context's GPU): for callers with their own HIP code"
log_prob_device(ap::ApproxLikelihood, d_x::Ptr{Float32}, d_lp::Ptr{Float32}, d_x_grad::Ptr{Float32}=Ptr{Float32}(C_NULL)) =
    check(ccall((:polee_approx_logprob_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}), ap.h, d_x, d_lp, d_x_grad), ap.ctx.h)

"approximate_splicing_likelihood (polee_splicing.py:14-113): (loc, scale) Float32 [S*F]; index pairs (feature, transcript), 1-based, flattened"
function splicing_moments(ap::ApproxLikelihood, feature_indices::Vector{Int32}, antifeature_indices::Vector{Int32}, F::Integer;
                          num_mean_draws::Integer=1000, num_var_draws::Integer=1000, seed::Integer=0)
    loc, scale = Vector{Float32}(undef, ap.S * F), Vector{Float32}(undef, ap.S * F)
    GC.@preserve feature_indices antifeature_indices loc scale check(
        ccall((:polee_approx_splicing_moments, LIB), Cint,
              (Ptr{Cvoid}, Ptr{Int32}, Int64, Ptr{Int32}, Int64, Int32, Int32, Int32, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
              ap.h, feature_indices, div(length(feature_indices), 2), antifeature_indices, div(length(antifeature_indices), 2), F,
              num_mean_draws, num_var_draws, seed, C_NULL, loc, scale), ap.ctx.h)
    return loc, scale
end

# ---- regression model: evaluation hook, parameter access, the Normal-likelihood variant --------------------------------
num_noise(r::Regression) = Int(ccall((:polee_regression_num_noise, LIB), Int64, (Ptr{Cvoid},), r.h))
"loss (and gradient with respect to the flat parameter vector) at one draw of the latents: the tests' hook"
function eval_loss(r::Regression; noise::Union{Nothing,Vector{Float32}}=nothing, seed::Integer=0, grad::Bool=false)
    loss = Ref{Float32}(0f0)
    g = grad ? Vector{Float32}(undef, num_params(r)) : Float32[]
    GC.@preserve noise g check(
        ccall((:polee_regression_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, UInt64, Ref{Float32}, Ptr{Float32}),
              r.h, noise === nothing ? C_NULL : pointer(noise), seed, loss, grad ? pointer(g) : C_NULL), r.ctx.h)
    return grad ? (loss[], g) : loss[]
end
function set_flat_params!(r::Regression, p::Vector{Float32})
    @assert length(p) == num_params(r)
    GC.@preserve p check(ccall((:polee_regression_set_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, p), r.ctx.h)
end
function set_isoform_params!(r::Regression, p::Vector{Float32})
    GC.@preserve p check(ccall((:polee_regression_set_isoform_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, p), r.ctx.h)
end
function isoform_grad(r::Regression)
    n = Int(ccall((:polee_regression_num_isoform_params, LIB), Int64, (Ptr{Cvoid},), r.h))
    g = Vector{Float32}(undef, n)
    GC.@preserve g check(ccall((:polee_regression_get_isoform_grad, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, g), r.ctx.h)
    return g
end
"RNASeqNormalTranscriptLinearRegression (models/polee_regression.py:490-531): x ~ Normal(loc, scale) per sample, [S*n] row-major"
function set_normal_likelihood!(r::Regression, loc::Vector{Float32}, scale::Vector{Float32})
    GC.@preserve loc scale check(ccall((:polee_regression_set_normal_likelihood, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}), r.h, loc, scale), r.ctx.h)
end
# classify (models/polee_regression.py:342-413), the device part: the relaxed design rows of one step go in with set_design!, fit
# (one step) moves only the flat parameters [first, last] (1-based, inclusive; an empty range freezes them all), and
# design_gradient returns d loss / d design of the last evaluation.  The relaxed one-hot rows, their density and the logits' Adam are
# host arithmetic on S x F numbers (polee_amd/regression.py classify is the worked example).
function set_design!(r::Regression, design::Matrix{Float32})
    d = permutedims(design)  # C order [S][F]
    GC.@preserve d check(ccall((:polee_regression_set_design, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, d), r.ctx.h)
end
function set_trainable!(r::Regression, first::Integer, last::Integer)
    check(ccall((:polee_regression_set_trainable, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), r.h, Int64(first - 1), Int64(last)), r.ctx.h)
end
function design_gradient(r::Regression, S::Integer, F::Integer)
    g = Matrix{Float32}(undef, F, S)
    GC.@preserve g check(ccall((:polee_regression_design_grad, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, g), r.ctx.h)
    return permutedims(g)
end

"the kernel-regression weights of the mean-variance prior, Float32 [degree*n] (models/polee_regression.py:436-460)"
function regression_weights(r::Regression, degree::Integer=15)
    w = Vector{Float32}(undef, degree * r.n)
    GC.@preserve w check(ccall((:polee_regression_weights, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), r.h, w), r.ctx.h)
    return w
end

# ---- X construction (src/rnaseq_sample.jl:390-524 with the SimplisticFragModel, src/fragmodel.jl:23-169) --------------
struct XbTranscripts
    n::Int32; seq::Ptr{Int32}; strand::Ptr{Int8}; exon_ptr::Ptr{Int64}; exon_first::Ptr{Int64}; exon_last::Ptr{Int64}
end
struct XbFragments
    m::Int64; seq::Ptr{Int32}; strand::Ptr{Int8}
    m1_left::Ptr{Int64}; m1_right::Ptr{Int64}; m2_left::Ptr{Int64}; m2_right::Ptr{Int64}
    m1_is_flag16::Ptr{UInt8}; cig1_ptr::Ptr{Int64}; cig2_ptr::Ptr{Int64}; cig_op::Ptr{UInt8}; cig_len::Ptr{Int32}
end
struct XbFragModel
    fraglen_pmf::Ptr{Float32}; fraglen_cdf::Ptr{Float32}; fraglen_median::Int32; strand_specificity::Float32; alt_frag_model::Int32
end
"""
    build_likelihood_matrix(ctx, T, F, pmf, cdf, median; strand_specificity, alt_frag_model)

T: NamedTuple (seq, strand, exon_ptr, exon_first, exon_last), F: NamedTuple (seq, strand, m1_left, m1_right, m2_left,
m2_right, m1_is_flag16, cig1_ptr, cig2_ptr, cig_op, cig_len) in the layout of include/polee_hip.h (m1 = the LEFTMOST mate,
as transcripts.jl:288-297 orders them).  Returns (tcolptr, trowval, tnzval, effective_lengths, row_fragment): the rows of X.
"""
function build_likelihood_matrix(ctx::Context, T, F, pmf::Vector{Float32}, cdf::Vector{Float32}, median::Integer;
                                 strand_specificity::Real=0.9, alt_frag_model::Bool=false, return_sample::Bool=false, return_tree::Bool=false)
    n = length(T.seq); m = length(F.seq)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve T F pmf cdf begin
        ts = XbTranscripts(n, pointer(T.seq), pointer(T.strand), pointer(T.exon_ptr), pointer(T.exon_first), pointer(T.exon_last))
        fs = XbFragments(m, pointer(F.seq), pointer(F.strand), pointer(F.m1_left), pointer(F.m1_right), pointer(F.m2_left),
                         pointer(F.m2_right), pointer(F.m1_is_flag16), pointer(F.cig1_ptr), pointer(F.cig2_ptr), pointer(F.cig_op),
                         pointer(F.cig_len))
        ms = XbFragModel(pointer(pmf), pointer(cdf), median, strand_specificity, alt_frag_model)
        check(ccall((:polee_xbuild_run, LIB), Cint, (Ptr{Cvoid}, Ref{XbTranscripts}, Ref{XbFragments}, Ref{XbFragModel}, Ref{Ptr{Cvoid}}),
                    ctx.h, ts, fs, ms, out), ctx.h)
    end
    h = out[]
    try
        rows, nnz = Ref{Int64}(0), Ref{Int64}(0)
        check(ccall((:polee_xbuild_sizes, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    h, rows, nnz, C_NULL, C_NULL, C_NULL), ctx.h)
        tcolptr, trowval, tnzval = Vector{UInt64}(undef, rows[] + 1), Vector{UInt32}(undef, nnz[]), Vector{Float32}(undef, nnz[])
        efflens, rowfrag = Vector{Float32}(undef, n), Vector{Int64}(undef, rows[])
        GC.@preserve tcolptr trowval tnzval efflens rowfrag check(
            ccall((:polee_xbuild_get, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int64}),
                  h, tcolptr, trowval, tnzval, efflens, rowfrag), ctx.h)
        if return_tree   # the clustering tree (rounds variant) from the result on the device
            parents, js = Vector{Int32}(undef, 2n - 1), Vector{Int32}(undef, 2n - 1)
            GC.@preserve parents js check(ccall((:polee_hclust_parallel_device_from_xbuild, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}),
                                                ctx.h, h, parents, js), ctx.h)
            return tcolptr, trowval, tnzval, efflens, rowfrag, parents, js
        end
        if return_sample  # the likelihood handle straight from the result on the device (X never visits the host on its way in)
            r = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:polee_loglik_create_from_xbuild, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ref{Ptr{Cvoid}}), ctx.h, h, C_NULL, r), ctx.h)
            return tcolptr, trowval, tnzval, efflens, rowfrag, wrap_sample(r[], ctx, rows[], n)
        end
        return tcolptr, trowval, tnzval, efflens, rowfrag
    finally
        ccall((:polee_xbuild_destroy, LIB), Cvoid, (Ptr{Cvoid},), h)
    end
end

# ---- the same under the BiasedFragModel (src/fragmodel.jl:174-445 with a TRAINED bias model; training stays upstream) ----------
struct XbBiasModel
    tseq_ptr::Ptr{Int64}; tseq::Ptr{UInt8}; seqbias_len::Int32; ps_ctx::Int32
    orders_left::Ptr{Int32}; orders_right::Ptr{Int32}; ps_left::Ptr{Float32}; ps_right::Ptr{Float32}
    gc_nbins::Int32; gc_bins::Ptr{Float32}; pos_p::Float64; pos_terms::Ptr{Float64}; pos_maxtlen::Int32
    num_fraglens::Int32; high_prob_fraglens::Ptr{Int32}; m1_reverse::Ptr{UInt8}
end
"""
    build_likelihood_matrix_biased(ctx, T, F, pmf, cdf, median, B; strand_specificity, alt_frag_model, return_bias)

As build_likelihood_matrix with the reference's default BiasedFragModel (polee_xbuild_run_biased): B is a NamedTuple in the layout of
polee_xb_biasmodel (include/polee_hip.h): tseq_ptr, tseq (0 A, 1 C, 2 G, 3 T, 4 other), orders_left / orders_right (Int32 [20], -1 =
position not in the model), ps_left / ps_right (Float32 [20][4][ps_ctx] row-major), gc_bins, high_prob_fraglens, and optionally
pos_p / pos_terms (PositionalBiasModel; `nothing` = use_pos_bias false) and m1_reverse (`nothing` when every fragment is paired).
Returns (tcolptr, trowval, tnzval, effective_lengths, row_fragment) and, with return_bias, the transcripts' left / right bias
vectors (polee_xbuild_get_bias, laid out like tseq).
"""
function build_likelihood_matrix_biased(ctx::Context, T, F, pmf::Vector{Float32}, cdf::Vector{Float32}, median::Integer, B;
                                        strand_specificity::Real=0.9, alt_frag_model::Bool=false, return_bias::Bool=false)
    n = length(T.seq); m = length(F.seq)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    pos_terms = get(B, :pos_terms, nothing); m1_reverse = get(B, :m1_reverse, nothing)
    ps_ctx = div(length(B.ps_left), 20 * 4)
    GC.@preserve T F pmf cdf B pos_terms m1_reverse begin
        ts = XbTranscripts(n, pointer(T.seq), pointer(T.strand), pointer(T.exon_ptr), pointer(T.exon_first), pointer(T.exon_last))
        fs = XbFragments(m, pointer(F.seq), pointer(F.strand), pointer(F.m1_left), pointer(F.m1_right), pointer(F.m2_left),
                         pointer(F.m2_right), pointer(F.m1_is_flag16), pointer(F.cig1_ptr), pointer(F.cig2_ptr), pointer(F.cig_op),
                         pointer(F.cig_len))
        ms = XbFragModel(pointer(pmf), pointer(cdf), median, strand_specificity, alt_frag_model)
        bm = XbBiasModel(pointer(B.tseq_ptr), pointer(B.tseq), 20, ps_ctx, pointer(B.orders_left), pointer(B.orders_right),
                         pointer(B.ps_left), pointer(B.ps_right), length(B.gc_bins), pointer(B.gc_bins),
                         pos_terms === nothing ? 0.0 : Float64(B.pos_p), pos_terms === nothing ? Ptr{Float64}(C_NULL) : pointer(pos_terms),
                         pos_terms === nothing ? 0 : length(pos_terms), length(B.high_prob_fraglens), pointer(B.high_prob_fraglens),
                         m1_reverse === nothing ? Ptr{UInt8}(C_NULL) : pointer(m1_reverse))
        check(ccall((:polee_xbuild_run_biased, LIB), Cint,
                    (Ptr{Cvoid}, Ref{XbTranscripts}, Ref{XbFragments}, Ref{XbFragModel}, Ref{XbBiasModel}, Ref{Ptr{Cvoid}}),
                    ctx.h, ts, fs, ms, bm, out), ctx.h)
    end
    h = out[]
    try
        rows, nnz = Ref{Int64}(0), Ref{Int64}(0)
        check(ccall((:polee_xbuild_sizes, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    h, rows, nnz, C_NULL, C_NULL, C_NULL), ctx.h)
        tcolptr, trowval, tnzval = Vector{UInt64}(undef, rows[] + 1), Vector{UInt32}(undef, nnz[]), Vector{Float32}(undef, nnz[])
        efflens, rowfrag = Vector{Float32}(undef, n), Vector{Int64}(undef, rows[])
        GC.@preserve tcolptr trowval tnzval efflens rowfrag check(
            ccall((:polee_xbuild_get, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt32}, Ptr{Float32}, Ptr{Float32}, Ptr{Int64}),
                  h, tcolptr, trowval, tnzval, efflens, rowfrag), ctx.h)
        if return_bias
            len = Int(B.tseq_ptr[end])
            left, right = Vector{Float32}(undef, len), Vector{Float32}(undef, len)
            GC.@preserve left right check(ccall((:polee_xbuild_get_bias, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float64}),
                                                h, left, right, C_NULL), ctx.h)
            return tcolptr, trowval, tnzval, efflens, rowfrag, left, right
        end
        return tcolptr, trowval, tnzval, efflens, rowfrag
    finally
        ccall((:polee_xbuild_destroy, LIB), Cvoid, (Ptr{Cvoid},), h)
    end
end

"fast_log of the tree kernels (csrc/scan.hpp), element-wise on the device: the tests compare it with Base.log"
function debug_fast_log(ctx::Context, x::Vector{Float64})
    out = similar(x)
    GC.@preserve x out check(ccall((:polee_debug_fast_log, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}), ctx.h, x, length(x), out), ctx.h)
    return out
end

end # module
