for v in 0 128; do echo "ABLATE=$v"; POLEE_DBG_ABLATE=$v timeout 300 python bench.py --steps 50 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; done
