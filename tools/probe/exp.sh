for v in 2 3 4; do echo "S=$v"; timeout 900 python bench.py --steps 50 --warmup 5 --cpu-steps 0 --samples-per-gpu $v 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; done
python bench.py --workload c3 --steps 300 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/bench_c3.json; cat gpurun_out/final/bench_c3.json | cut -c1-200
