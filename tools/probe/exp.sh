for v in 0 300; do echo "prewarm=$v"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --prewarm $v 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"; done
