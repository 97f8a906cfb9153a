# runtests.jl -- what a box with Julia + an MI355X runs against libpolee_hip.so through julia/PoleeHIP.jl.
#
#   POLEE_REFERENCE_DIR=/path/to/polee julia julia/runtests.jl
#
# Data: the reference's OWN fixtures, test/dataset/mBr_M_6w_1.likelihood-matrix.h5 (src/rnaseq_sample.jl:505-519) and
# mBr_M_6w_1.prep.h5 (src/likelihood-approximation.jl:61-87), read with HDF5.jl (a dependency of the reference).  The
# checks are the Julia twins of tests/test_gpu_parity.py / test_gpu_pin.py: the reference formulas evaluated in plain Julia
# (the loops of src/ptt.jl, src/sparse.jl, src/likelihood.jl restated inline below, a few lines each) against the
# device results.  NOT EXECUTED in the build container (no Julia there): written against include/polee_hip.h.
using Test, HDF5, Random
include(joinpath(@__DIR__, "PoleeHIP.jl"))
using .PoleeHIP

const REF = get(ENV, "POLEE_REFERENCE_DIR", joinpath(@__DIR__, "..", "..", "reference"))
lm = h5open(joinpath(REF, "test", "dataset", "mBr_M_6w_1.likelihood-matrix.h5"), "r") do f
    (m = Int(read(f["m"])[1]), n = Int(read(f["n"])[1]), colptr = UInt32.(read(f["colptr"])), rowval = UInt32.(read(f["rowval"])),
     nzval = Float32.(read(f["nzval"])), efflens = Float32.(read(f["effective_lengths"])))
end
pr = h5open(joinpath(REF, "test", "dataset", "mBr_M_6w_1.prep.h5"), "r") do f
    (mu = Float32.(read(f["mu"])), omega = Float32.(read(f["omega"])), alpha = Float32.(read(f["alpha"])),
     parents = Int32.(read(f["node_parent_idxs"])), js = Int32.(read(f["node_js"])))
end
m, n = lm.m, lm.n

# --- the reference's loops, inline (host side, Float64 accumulation as sparse.jl:13-17,32-36 with Vector{Float64}) -----
function ref_log_likelihood(xs::Vector{Float32})
    fp = zeros(Float64, m)
    for j in 1:n, k in lm.colptr[j]:lm.colptr[j+1]-1
        fp[lm.rowval[k]] += xs[j] * lm.nzval[k]          # pAt_mul_B! (Float32 product, Float64 sum)
    end
    g = zeros(Float64, n)
    for j in 1:n, k in lm.colptr[j]:lm.colptr[j+1]-1
        g[j] += lm.nzval[k] / fp[lm.rowval[k]]           # pAt_mulinv_B!
    end
    return sum(log, fp), g
end
function ref_tree(parents, js)                            # src/ptt.jl:89-116: children from the parent array
    N = length(parents); left = zeros(Int, N); right = zeros(Int, N)
    for i in 2:N
        p = parents[i]
        right[p] == 0 ? (right[p] = i) : (left[p] = i)    # nodes arrive right child first (hclust.jl:361-389)
    end
    return left, right
end
function ref_transform(ys::Vector{Float64})               # src/ptt.jl:125-160
    left, right = ref_tree(pr.parents, pr.js)
    N = length(pr.parents); us = zeros(Float64, N); us[1] = 1.0; xs = zeros(Float32, n); k = 1; ladj = 0.0
    for i in 1:N
        if pr.js[i] != 0
            xs[pr.js[i]] = max(Float32(us[i]), 1f-16)
        else
            us[left[i]] = ys[k] * us[i]; us[right[i]] = (1 - ys[k]) * us[i]; ladj += log(us[i]); k += 1
        end
    end
    return xs, ladj
end

ctx = PoleeHIP.Context(0)
@info PoleeHIP.version()
t = PoleeHIP.PolyaTreeTransform(ctx, pr.parents, pr.js)
s = PoleeHIP.DeviceSample(ctx, m, n, lm.colptr, lm.rowval, lm.nzval)

@testset "layout" begin
    i = PoleeHIP.info(s)
    @test i.m == m && i.n == n && i.nnz == length(lm.nzval)
    @test sum(i.stream_nnz) == i.nnz && sum(i.stream_rows) == m
    @test i.stream_rows[8] == 10740                       # fragments compatible with one transcript: collapsed (stream S)
end

@testset "transform! / inverse_transform! (ptt.jl:125-160, 257-285)" begin
    rng = MersenneTwister(1)
    ys = clamp.(rand(rng, n - 1), 1e-6, 1 - 1e-6)
    xs = Vector{Float32}(undef, n)
    ladj = PoleeHIP.transform!(t, ys, xs, Val(true))
    xr, lr = ref_transform(ys)
    @test xs ≈ xr rtol = 1e-6
    @test ladj ≈ lr rtol = 1e-10
    y2 = similar(ys)
    PoleeHIP.inverse_transform!(t, xs, y2)
    @test y2 ≈ ys rtol = 1e-4                             # (xs went through Float32)
    # the TF-side ops on the same tree: HSB(logit y) = transform(y); InvHSB undoes it
    l, r, f = PoleeHIP.make_inverse_ptt_params(pr.parents, pr.js)
    ti = PoleeHIP.tree_from_index(ctx, l, r, f)
    x_hsb = PoleeHIP.hsb(ti, Float32.(log.(ys ./ (1 .- ys))))
    @test x_hsb ≈ xr rtol = 1e-4
    y3, _ = PoleeHIP.inv_hsb(ti, xs)
    @test y3 ≈ ys rtol = 1e-4
end

@testset "log_likelihood (likelihood.jl:36-56)" begin
    rng = MersenneTwister(2)
    x = Float32.(rand(rng, n)); x ./= sum(x)
    g = Vector{Float64}(undef, n)
    lp = PoleeHIP.log_likelihood(s, x, g, Val(false))
    lpr, gr = ref_log_likelihood(x)
    @test lp ≈ lpr rtol = 1e-6
    @test g ≈ gr rtol = 1e-4
    lps, gs = PoleeHIP.log_likelihood_batch(s, hcat(x, x))
    @test lps[1] ≈ lpr rtol = 1e-6 && gs[:, 2] ≈ gr rtol = 1e-4
    PoleeHIP.set_deterministic!(s, true)
    g1 = similar(g); g2 = similar(g)
    PoleeHIP.log_likelihood(s, x, g1, Val(true)); PoleeHIP.log_likelihood(s, x, g2, Val(true))
    @test g1 == g2                                         # bitwise reproducible
    PoleeHIP.set_deterministic!(s, false)
end

@testset "reparameterisations (logitnormal.jl, sinh_arcsinh.jl)" begin
    rng = MersenneTwister(3)
    z0 = randn(rng, Float32, n - 1); sigma = exp.(pr.omega)
    zs = similar(z0); ys = Vector{Float64}(undef, n - 1)
    l1 = PoleeHIP.sinh_asinh_transform!(ctx, pr.alpha, z0, zs, Val(true))
    @test zs ≈ sinh.(asinh.(z0) .+ pr.alpha) rtol = 1e-5
    l2 = PoleeHIP.logit_normal_transform!(ctx, pr.mu, sigma, zs, ys, Val(true))
    @test ys ≈ 1 ./ (1 .+ exp.(-(Float64.(pr.mu) .+ Float64.(zs) .* sigma))) rtol = 1e-5
    @test isfinite(l1) && isfinite(l2)
end

@testset "approximate_likelihood (likelihood-approximation.jl:395-624)" begin
    fit = PoleeHIP.approximate_likelihood(s, t, lm.efflens)
    @test all(isfinite, fit["mu"]) && all(isfinite, fit["omega"]) && all(isfinite, fit["alpha"])
    # the fit explains the data as well as the reference's own: posterior means correlate
    pm_ours = PoleeHIP.posterior_mean(t, fit["mu"], exp.(fit["omega"]), fit["alpha"], 200; seed = 1)
    pm_ref = PoleeHIP.posterior_mean(t, pr.mu, exp.(pr.omega), pr.alpha, 200; seed = 1)
    c = sum((log.(pm_ours) .- sum(log.(pm_ours)) / n) .* (log.(pm_ref) .- sum(log.(pm_ref)) / n)) /
        sqrt(sum(abs2, log.(pm_ours) .- sum(log.(pm_ours)) / n) * sum(abs2, log.(pm_ref) .- sum(log.(pm_ref)) / n))
    @test c > 0.99
    # stationarity of the reference's parameters under the device gradient (tests/test_gpu_pin.py, coarse form)
    o = PoleeHIP.ViOpts(); o.num_mc_samples = 8; o.max_mu_step = 0; o.max_omega_step = 0; o.max_alpha_step = 0; o.num_steps = 2001
    f = PoleeHIP.LikelihoodApproximationFit(s, t, lm.efflens, o)
    PoleeHIP.set_params!(f, pr.mu, pr.omega, pr.alpha)
    acc = zeros(n - 1)
    for _ in 1:2000
        acc .+= PoleeHIP.eval_gradients(f, 8)[4]
        PoleeHIP.run!(f, 1)
    end
    PoleeHIP.sync!(f)
    @test sum(abs, acc ./ 2000) / (n - 1) < 0.1             # |mean mu gradient| per node; a single draw's is ~1.5
end

@testset "tree construction, point estimate, host cache" begin
    parents, js = PoleeHIP.hclust(m, n, lm.colptr, lm.rowval)
    @test length(parents) == 2n - 1 && sort(js[js .> 0]) == collect(1:n)
    t2 = PoleeHIP.PolyaTreeTransform(ctx, parents, js)
    xs, zs = PoleeHIP.optimize_ptt(s, t2, lm.efflens, 100)
    @test sum(xs) ≈ 1 rtol = 1e-3
    @test PoleeHIP.host_cache_configure(-1) >= 0
    PoleeHIP.host_cache_trim()
    @test PoleeHIP.host_cache_bytes() == 0
    x = rand(10) .+ 0.1
    @test PoleeHIP.debug_fast_log(ctx, x) ≈ log.(x) rtol = 1e-14
end

@testset "device builders (csrc/psell_device.hip, csrc/hclust_device.hip)" begin
    # the layout of the fixture is built on the device by default; the tree of the rounds variant is the same from the host and the GPU
    @test PoleeHIP.built_on_device(s)
    ph, jh = PoleeHIP.hclust(m, n, lm.colptr, lm.rowval; parallel=true)
    pd, jd = PoleeHIP.hclust(m, n, lm.colptr, lm.rowval; device=ctx)
    @test ph == pd && jh == jd
    @test PoleeHIP.device_cache_bytes() >= 0
    PoleeHIP.host_cache_trim()
    @test PoleeHIP.device_cache_bytes() == 0
end
