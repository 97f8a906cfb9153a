"""The second half of `polee prep-sample` on the GPU: likelihood matrix in, prepared sample out.

Mirrors approximate_likelihood(approximation, sample, output_filename; use_efflen_jacobian,
tree_topology_input_filename) (src/likelihood-approximation.jl:32-60) for a sample whose likelihood matrix was
written with `--likelihood-matrix` (src/rnaseq_sample.jl:505-519): read X and the effective lengths, build the
Polya tree (hclust, or read it from a `--ptt-tree` file, :428-436), fit the approximation on the GPU
(polee_vi_*), write the prepared-sample HDF5 (write_approximation, :61-87) that `polee model ...` /
load_samples_from_specification read.  The ingest that produces X (BAM -> fragments -> X, bias models) stays with
the reference.

    python -m polee_amd.prep likelihood-matrix.h5 -o prepared-sample.h5 [--tree-method auto|cluster|cluster_parallel|cluster_device|cluster_auto|sequential]
        [--ptt-tree tree.h5] [--no-efflen-jacobian] [--seed N] [--device D]

--tree-method (default auto): "cluster" is the reference's heuristic merge for merge (src/hclust.jl:193-319) -- a serial
priority-queue loop, 3.7 s on one host thread at 200 000 transcripts.  "cluster_auto" / "cluster_parallel" / "cluster_device" build
the ROUNDS variant (mutually-best pairs merged per round; 82-86 % of its clades are the reference tree's, the fit on it is as good:
tests/test_gpu_hclust.py), the same tree from the host threads or the GPU, 0.09-0.45 s.  "auto" takes the reference's order up to
AUTO_EXACT_MAX_N transcripts (where it costs well under a second) and the rounds variant above, and says which tree it built.

`polee prep-salmon` (src/main.jl:723-750): the factored likelihood of `salmon quant -d` output on a given tree:

    python -m polee_amd.prep --salmon salmon_quant_dir --ptt-tree tree.h5 --transcript-ids ids.txt -o prepared-sample.h5

(`ids.txt`: one transcript id per line in the tree's order, i.e. the `transcript_ids` dataset of the PTT file).
"""
import argparse
import sys
import time

from . import h5io
from .core import (Context, LogitSkewNormalPTTApprox, PolyaTreeTransform, RNASeqSample, approximate_likelihood,
                   sample_and_tree)


# --tree-method auto: the reference-order tree up to this many transcripts, the rounds variant above (the
# out-of-the-box path at GENCODE size was ten times slower than the documented one)
AUTO_EXACT_MAX_N = 20000


def resolve_tree_method(method, n):
    """(treemethod, note) for --tree-method `method` on n transcripts."""
    if method != "auto":
        return method, None
    if n <= AUTO_EXACT_MAX_N:
        return "cluster", "tree: the reference's merge order (hclust.jl), n = %d <= %d" % (n, AUTO_EXACT_MAX_N)
    return "cluster_auto", ("tree: the ROUNDS variant of the heuristic (cluster_auto: host threads or GPU, the same tree), n = %d > %d; "
                            "--tree-method cluster builds the reference's merge order instead (serial: seconds at this size)"
                            % (n, AUTO_EXACT_MAX_N))


def approximate_likelihood_to_file(approx, likelihood_matrix_filename, output_filename, use_efflen_jacobian=True,
                                   tree_topology_input_filename=None, seed=123456789, ctx=None, args=""):
    """Returns the params dict that was written (mu, omega, alpha, node_parent_idxs, node_js) plus timings."""
    ctx = ctx or Context(0)
    t0 = time.time()
    lm = h5io.read_likelihood_matrix(likelihood_matrix_filename)
    t_read = time.time() - t0
    note = None
    if approx.treemethod == "auto":
        tm, note = resolve_tree_method("auto", int(lm["n"]))
        approx = LogitSkewNormalPTTApprox(tm)
    t0 = time.time()
    if tree_topology_input_filename is not None:  # likelihood-approximation.jl:428-433
        sample = RNASeqSample(lm["m"], lm["n"], lm["colptr"], lm["rowval"], lm["nzval"], lm["effective_lengths"], ctx=ctx)
        parents, js = h5io.read_transformation(tree_topology_input_filename)
        tree = PolyaTreeTransform(parents, js, ctx=ctx)
    else:  # the tree heuristic runs beside the device layout build (two independent host stages)
        sample, tree = sample_and_tree(approx, lm["m"], lm["n"], lm["colptr"], lm["rowval"], lm["nzval"],
                                       lm["effective_lengths"], ctx=ctx)
    t_layout = time.time() - t0
    t0 = time.time()
    params = approximate_likelihood(approx, sample, tree, use_efflen_jacobian=use_efflen_jacobian, seed=seed)
    t_fit = time.time() - t0
    h5io.write_approximation(output_filename, lm["m"], lm["n"], lm["effective_lengths"], params, args=args)
    params["timings"] = {"read_s": t_read, "device_layout_and_tree_s": t_layout, "fit_s": t_fit}
    params["tree_method"] = "file" if tree_topology_input_filename is not None else approx.treemethod
    params["tree_note"] = note if tree_topology_input_filename is None else None
    return params


def approximate_salmon_likelihood_to_file(approx, salmon_dir, transcript_ids, tree_filename, output_filename,
                                          use_efflen_jacobian=True, seed=123456789, ctx=None, args=""):
    """polee_prep_salmon (src/main.jl:723-750): load_salmon_likelihood -> factored fit on the given tree -> file."""
    from .salmon import load_salmon_likelihood
    ctx = ctx or Context(0)
    t0 = time.time()
    s = load_salmon_likelihood(salmon_dir, transcript_ids)
    t_read = time.time() - t0
    t0 = time.time()
    sample = s.to_sample(ctx=ctx)
    t_layout = time.time() - t0
    parents, js = h5io.read_transformation(tree_filename)
    if len(js) != 2 * s.n - 1:
        raise ValueError("the tree has %d nodes, salmon reports %d transcripts" % (len(js), s.n))
    tree = PolyaTreeTransform(parents, js, ctx=ctx)
    t0 = time.time()
    params = approximate_likelihood(approx, sample, tree, use_efflen_jacobian=use_efflen_jacobian, seed=seed)
    t_fit = time.time() - t0
    h5io.write_approximation(output_filename, s.m, s.n, s.efflens, params, args=args)
    params["timings"] = {"read_s": t_read, "device_layout_s": t_layout, "tree_and_fit_s": t_fit}
    return params


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m polee_amd.prep", description=__doc__.split("\n\n")[0])
    ap.add_argument("likelihood_matrix", metavar="likelihood-matrix.h5", nargs="?")
    ap.add_argument("--salmon", default=None, metavar="salmon_quant_dir", help="prep-salmon: salmon quant -d output")
    ap.add_argument("--transcript-ids", default=None, metavar="ids.txt", help="with --salmon: transcript ids, tree order")
    ap.add_argument("-o", "--output", default="prepared-sample.h5", metavar="prepared-sample.h5")
    ap.add_argument("--tree-method", default="auto", choices=["auto", "cluster", "cluster_parallel", "cluster_device", "cluster_auto", "sequential"],
                    help="auto (default): the reference's merge order up to %d transcripts, the rounds variant (cluster_auto) above" % AUTO_EXACT_MAX_N)
    ap.add_argument("--ptt-tree", default=None, metavar="tree.h5", help="use this tree topology (polee fit-tree output)")
    ap.add_argument("--no-efflen-jacobian", action="store_true")
    ap.add_argument("--seed", type=int, default=123456789)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.salmon is not None:
        if a.ptt_tree is None or a.transcript_ids is None:
            ap.error("--salmon needs --ptt-tree and --transcript-ids")
        with open(a.transcript_ids) as f:
            ids = [line.rstrip("\n") for line in f if line.strip()]
        params = approximate_salmon_likelihood_to_file(LogitSkewNormalPTTApprox("static"), a.salmon, ids, a.ptt_tree,
                                                       a.output, use_efflen_jacobian=not a.no_efflen_jacobian,
                                                       seed=a.seed, ctx=Context(a.device),
                                                       args=" ".join(argv or sys.argv[1:]))
        t = params["timings"]
        print("wrote %s (n=%d): salmon read %.2f s, device layout %.2f s, fit %.2f s"
              % (a.output, len(params["mu"]) + 1, t["read_s"], t["device_layout_s"], t["tree_and_fit_s"]))
        return 0
    if a.likelihood_matrix is None:
        ap.error("give a likelihood matrix or --salmon")
    params = approximate_likelihood_to_file(LogitSkewNormalPTTApprox(a.tree_method), a.likelihood_matrix, a.output,
                                            use_efflen_jacobian=not a.no_efflen_jacobian,
                                            tree_topology_input_filename=a.ptt_tree, seed=a.seed,
                                            ctx=Context(a.device), args=" ".join(argv or sys.argv[1:]))
    t = params["timings"]
    if params.get("tree_note"):
        print(params["tree_note"])
    print("wrote %s (n=%d): read %.2f s, device layout and tree (%s) %.2f s, fit %.2f s"
          % (a.output, len(params["mu"]) + 1, t["read_s"], params["tree_method"], t["device_layout_and_tree_s"], t["fit_s"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
