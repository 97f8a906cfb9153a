for v in 0 4096; do POLEE_DBG_ABLATE=$v timeout 600 python tools/probe/time_fit.py 2>&1 | tail -1; done
