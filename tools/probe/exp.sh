for v in 0; do timeout 300 python bench.py --steps 100 --warmup 10 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'], d['detail']['num_tiles'])"; done
