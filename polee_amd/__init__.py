"""polee_amd -- host side of the MI355X approximate-likelihood engine for Polee.

A Python mirror of the reference's Julia interface for the hot path (Julia is not
available in the build image; julia/PoleeHIP.jl holds the ccall wrappers a maintainer
would use).  Names, argument meaning and error behaviour follow the reference:

  PolyaTreeTransform, transform!, transform_gradients!, inverse_transform!   src/ptt.jl
  make_inverse_ptt_params                                                    src/ptt.jl:293-309
  hsb / inv_hsb / inv_hsb_grad  (TF custom ops)                              src/tensorflow_ext/hsb_ops.cpp
  RNASeqSample.X + log_likelihood / factored_log_likelihood                  src/likelihood.jl
  approximate_likelihood(LogitSkewNormalPTTApprox(), sample)                 src/likelihood-approximation.jl
  ApproxLikelihoodSampler / rand!                                            src/approx-sampler.jl
  RNASeqApproxLikelihood(...).log_prob, rnaseq_approx_likelihood_sampler     src/polee_approx_likelihood.py
  RNASeqLinearRegression / RNASeqTranscriptLinearRegression(...).fit         models/polee_regression.py
  build_likelihood_matrix (X from alignments, SimplisticFragModel)           src/rnaseq_sample.jl:58-121, src/fragmodel.jl

All numerics run in libpolee_hip.so on the GPU; nothing here computes on the CPU.
"""
from ._lib import PoleeError, NonFiniteError, lib, LIB_PATH  # noqa: F401
from .core import (Context, Comm, HostComm, version, hclust, host_cache_trim, sample_and_tree, PolyaTreeTransform, make_inverse_ptt_params, hsb, inv_hsb, inv_hsb_grad,  # noqa: F401
                   RNASeqSample, DeviceX, log_likelihood, factored_log_likelihood,
                   effective_length_jacobian_adjustment, gene_noninformative_prior, LogitSkewNormalPTTApprox, approximate_likelihood,
                   LikelihoodApproximationFit, ApproxLikelihoodSampler, RNASeqApproxLikelihood,
                   rnaseq_approx_likelihood_sampler, LIKAP_NUM_STEPS, LIKAP_NUM_MC_SAMPLES,
                   logit_normal_transform, logit_normal_transform_gradients, sinh_asinh_transform,
                   sinh_asinh_transform_gradients, kumaraswamy_transform, kumaraswamy_transform_gradients,
                   OptimizePTTApprox, optimize_likelihood, list_nodes)

from . import h5io, estimate  # noqa: F401,E402
from .estimate import LoadedSamples, load_samples_from_specification, load_samples_hdf5, read_specification  # noqa: F401,E402
from .regression import (RNASeqLinearRegression, RNASeqTranscriptLinearRegression, RNASeqNormalTranscriptLinearRegression, RNASeqGeneLinearRegression, RNASeqGeneIsoformLinearRegression, RNASeqJointLinearRegression, estimate_sample_scales,  # noqa: F401,E402
                         find_minimum_effect_size, write_regression_effects)
from .salmon import load_salmon_likelihood, SalmonLikelihood  # noqa: F401,E402
from .cohort import approximate_likelihood_cohort, approximate_likelihood_cohort_processes  # noqa: F401,E402
from .xbuild import build_likelihood_matrix  # noqa: F401,E402
