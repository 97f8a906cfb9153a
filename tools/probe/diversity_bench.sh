#!/bin/bash
# Set-diversity sweep of the sparse pass on the GPU box: bench.py at C2 size with the generator as built, per-entry dropout
# 0.1 / 0.3, literal per-fragment subsets, and the tiled real fixture.  One JSON line per case under gpurun_out/diversity_<tag>/.
# usage: tools/probe/diversity_bench.sh <tag> [workload (c2|small)] [extra bench args]
TAG=${1:-run}; WL=${2:-c2}; shift; shift
OUT=gpurun_out/diversity_$TAG
mkdir -p $OUT
COMMON="--steps 20 --warmup 5 --cpu-steps 0 $*"
python3 bench.py --workload $WL --generator patterns --no-by-input $COMMON > $OUT/p0.json 2> $OUT/p0.err
python3 bench.py --workload $WL --generator patterns --set-diversity 0.1 $COMMON > $OUT/p0.1.json 2> $OUT/p0.1.err
python3 bench.py --workload $WL --generator patterns --set-diversity 0.3 $COMMON > $OUT/p0.3.json 2> $OUT/p0.3.err
python3 bench.py --workload $WL --generator literal --no-by-input $COMMON > $OUT/literal.json 2> $OUT/literal.err
python3 bench.py --workload fixture $COMMON > $OUT/fixture.json 2> $OUT/fixture.err
python3 - $OUT <<'PY'
import json, sys, os
d = sys.argv[1]
print("| input | nnz/row | layout bytes/nnz | CSR bytes/nnz | share of nnz: dense<=16 / masked<=16 / dense 17..32 / masked 17..32 / mixed<=15 / mixed (2nd launch) | kernel ms | pass ms | physical GB/s (frac) | CSR-equivalent GB/s | VI it/s |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name in ("p0", "p0.1", "p0.3", "literal", "fixture"):
    try:
        j = json.loads([l for l in open(os.path.join(d, name + ".json")) if l.startswith("{")][-1])
    except Exception as e:
        print("|", name, "| failed:", e, "|"); continue
    r = j["roofline"]
    print("| %s | %.2f | %.2f | %.2f | %s | %.4f | %.4f | %.0f (%.2f) | %.0f | %.0f |" % (
        name, j["config"]["nnz"] / float(j["config"]["workload"].split("m=")[1].split(" ")[0]), r["layout_bytes_per_nnz"], r["csr_bytes_per_nnz"],
        " / ".join("%.3f" % x for x in r["stream_share_of_nnz"][:6]), r["kernel_ms_avg"], r["pass_ms_avg"], r["pass_physical_GBs"],
        r["pass_physical_GBs"] / 8000.0, r["pass_effective_GBs"], j["value"]))
PY
