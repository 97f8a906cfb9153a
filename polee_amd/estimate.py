"""Model-entry API: experiment specification -> LoadedSamples (src/estimate.jl:9-59, 224-252, 338-556;
struct src/rnaseq_sample.jl:531-560).  Downstream models reach the approximate likelihoods only through
`LoadedSamples.variables` -- the seven arrays of create_tensorflow_variables! -- so this is the boundary that
lets them drop in unchanged.  Here `variables` holds NumPy arrays plus, under "approx", the device handle
(RNASeqApproxLikelihood) that evaluates the likelihood term on the GPU in place of TensorFlow."""
import base64

import numpy as np

from . import h5io
from .core import RNASeqApproxLikelihood, make_inverse_ptt_params


def read_specification(spec, point_estimates_key=None, max_num_samples=None, rng=None):
    """read_specification (estimate.jl:9-59) -> (filenames, sample_names, sample_factors)."""
    prep_file_suffix = spec.get("prep_file_suffix", ".likelihood.h5")
    sample_names, filenames, sample_factors = [], [], []
    for sample in spec["samples"]:
        name = sample["name"]
        sample_names.append(name)
        if point_estimates_key is None:
            filenames.append(sample.get("file", str(name) + prep_file_suffix))
        else:
            if "point-estimates" not in sample:
                raise ValueError("Sample %s has no point estimate files psecified." % name)
            if point_estimates_key not in sample["point-estimates"]:
                raise ValueError("Sample %s has no point estimates specified with key %s" % (name, point_estimates_key))
            filenames.append(sample["point-estimates"][point_estimates_key])
        sample_factors.append({str(k): str(v) for k, v in (sample.get("factors") or {}).items()})
    num_samples = len(filenames)
    if max_num_samples is not None and max_num_samples < num_samples:
        rng = rng or np.random.default_rng()
        p = rng.permutation(num_samples)[:max_num_samples]
        filenames = [filenames[i] for i in p]
        sample_names = [sample_names[i] for i in p]
        sample_factors = [sample_factors[i] for i in p]
    return filenames, sample_names, sample_factors


class LoadedSamples:
    """src/rnaseq_sample.jl:531-560."""

    def __init__(self, efflen_values, x0_values, la_mu_values, la_sigma_values, la_alpha_values, left_index,
                 right_index, leaf_index, sample_filenames):
        self.efflen_values, self.x0_values, self.log_x0_std = efflen_values, x0_values, None
        self.la_mu_values, self.la_sigma_values, self.la_alpha_values = la_mu_values, la_sigma_values, la_alpha_values
        self.left_index, self.right_index, self.leaf_index = left_index, right_index, leaf_index
        self.variables, self.init_feed_dict = {}, {}
        self.sample_factors, self.sample_names, self.sample_filenames = [], [], list(sample_filenames)


def create_variables(ls, ctx=None):
    """create_tensorflow_variables! (estimate.jl:502-556) without TensorFlow: the same seven keys, plus the GPU
    handle that consumes them."""
    ls.variables.clear()
    ls.variables.update(efflen=ls.efflen_values, la_mu=ls.la_mu_values, la_sigma=ls.la_sigma_values,
                        la_alpha=ls.la_alpha_values, left_index=ls.left_index, right_index=ls.right_index,
                        leaf_index=ls.leaf_index)
    ls.variables["approx"] = RNASeqApproxLikelihood(ls.variables, ctx=ctx)
    return ls


def load_samples_hdf5(filenames, n, gffhash=None, ptt_filename=None, check_gff_hash=True, using_device=True,
                      num_init_draws=30, seed=123456789, ctx=None, init_noise=None):
    """load_samples_hdf5 (estimate.jl:338-499).  `n` = number of transcripts (length(ts) in the reference),
    `gffhash` = raw hash bytes of the annotation (ts_metadata.gffhash) for the consistency check."""
    S, N = len(filenames), 2 * n - 1
    efflen = np.empty((S, n), np.float32)
    mu, sigma, alpha = (np.empty((S, n - 1), np.float32) for _ in range(3))
    shared = ptt_filename is not None
    T = 1 if shared else S
    left, right, leaf = (np.empty((T, N), np.int32) for _ in range(3))
    if shared:
        left[0], right[0], leaf[0] = make_inverse_ptt_params(*h5io.read_transformation(ptt_filename))
    trees = []  # (node_parent_idxs, node_js) of every sample (for the initial-value draws)
    for i, filename in enumerate(filenames):
        s = h5io.read_prepared_sample(filename)  # raises on a version mismatch (estimate.jl:388)
        if s["n"] != n:
            raise ValueError("Prepared sample %s has a different number of transcripts than provided GFF3 file." % filename)
        if check_gff_hash and gffhash is not None and base64.b64decode(s["metadata"].get("gffhash", "")) != bytes(gffhash):
            raise ValueError("%s:\nGFF3 file is not the same as the one used for sample preparation.\n"
                             "Filename of original GFF3 file: %s" % (filename, s["metadata"].get("gfffilename", "")))
        mu[i], sigma[i], alpha[i] = s["mu"], np.exp(s["omega"]), s["alpha"]
        efflen[i] = s["effective_lengths"]
        if not shared:
            if s["node_parent_idxs"] is None:
                raise ValueError("%s holds no tree and no --ptt-tree file was given" % filename)
            left[i], right[i], leaf[i] = make_inverse_ptt_params(s["node_parent_idxs"], s["node_js"])
            trees.append((s["node_parent_idxs"], s["node_js"]))
    ls = LoadedSamples(efflen, np.zeros((S, n), np.float32), mu, sigma, alpha, left, right, leaf, filenames)
    if using_device:
        create_variables(ls, ctx)
        # reasonable initial values: the mean of 30 draws from each approximation, drawn as the reference draws them
        # (estimate.jl:436-455: y clamped to LIKAP_Y_EPS, transform!, / effective lengths, renormalised), on the GPU;
        # `init_noise` (optional, [S][num_init_draws][n-1]) replaces the device RNG (parity tests)
        from .core import ApproxLikelihoodSampler, PolyaTreeTransform
        als = ApproxLikelihoodSampler()
        als.seed(seed)
        shared_tree = None
        for i in range(S):
            if shared:
                if shared_tree is None:
                    shared_tree = PolyaTreeTransform(*h5io.read_transformation(ptt_filename), ctx=ctx)
                t = shared_tree
            else:
                t = PolyaTreeTransform(trees[i][0], trees[i][1], ctx=ctx)
            als.set_transform(t, mu[i], sigma[i], alpha[i])
            z0 = None if init_noise is None else np.asarray(init_noise, np.float32)[i]
            ls.x0_values[i] = als.initial_values(efflen[i], num_init_draws, z0=z0)
    return ls


def load_samples_from_specification(spec, n, gffhash=None, ptt_filename=None, max_num_samples=None, batch_size=None,
                                    check_gff_hash=True, using_device=True, ctx=None):
    """load_samples_from_specification (estimate.jl:224-252); `spec` is the parsed experiment YAML."""
    filenames, sample_names, sample_factors = read_specification(spec, max_num_samples=max_num_samples)
    if "transformation" in spec and ptt_filename is None:
        ptt_filename = spec["transformation"]
    ls = load_samples_hdf5(filenames, n, gffhash, ptt_filename, check_gff_hash=check_gff_hash,
                           using_device=using_device, ctx=ctx)
    ls.sample_factors, ls.sample_names = sample_factors, sample_names
    return ls


def quantile(loaded_samples, transforms, qs=(0.01, 0.99), N=100, seed=123456789):
    """Statistics.quantile(loaded_samples, transforms, qs, N) (src/approx-sampler.jl:50-83): element-wise quantiles
    of the approximated likelihood, [len(qs), num_samples, n]."""
    from .core import ApproxLikelihoodSampler
    ls = loaded_samples
    S, n = ls.x0_values.shape
    out = np.empty((len(qs), S, n), np.float32)
    als = ApproxLikelihoodSampler()
    als.seed(seed)
    for i in range(S):
        als.set_transform(transforms[i], ls.la_mu_values[i], ls.la_sigma_values[i], ls.la_alpha_values[i])
        out[:, i, :] = als.quantile(qs, N)
    return out


def posterior_mean(loaded_samples, N=100, seed=123456789, ctx=None):
    """posterior_mean(loaded_samples, N) (src/approx-sampler.jl:86-117): [num_samples, n]; the trees are read from
    the samples' prepared files."""
    from .core import ApproxLikelihoodSampler, PolyaTreeTransform
    from . import h5io
    ls = loaded_samples
    pm = np.empty_like(np.asarray(ls.x0_values, np.float32))
    als = ApproxLikelihoodSampler()
    als.seed(seed)
    for i, fn in enumerate(ls.sample_filenames):
        d = h5io.read_prepared_sample(fn)
        t = PolyaTreeTransform(d["node_parent_idxs"], d["node_js"], ctx=ctx)
        als.set_transform(t, ls.la_mu_values[i], ls.la_sigma_values[i], ls.la_alpha_values[i])
        pm[i] = als.posterior_mean(N)
    return pm
