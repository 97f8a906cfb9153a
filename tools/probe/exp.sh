for v in "POLEE_TILE_A1=64" "POLEE_TILE_A1=96" "POLEE_TILE_A1=128" "POLEE_TILE_A1=192" "POLEE_TILE_A1=252"; do echo "MODE=$v"; env $v timeout 300 python bench.py --steps 50 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])"; done
