for v in "" "POLEE_STREAM_DECOUPLED=1"; do echo "MODE=$v"; env $v timeout 300 python bench.py --steps 50 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; done
echo DET; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-steps 0 --deterministic 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"
echo C5; timeout 600 python bench.py --workload c5 --steps 20 --warmup 3 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
