#!/bin/bash
# Round 4: biased X construction on the GPU; A/B of the builtin LDS-DMA build; workgroups per CU on small inputs.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_xbuild.py -m gpu -x -q -s > $OUT/xbuild_tests.log 2>&1; tail -6 $OUT/xbuild_tests.log
# builtin DMA: correctness on the sparse-pass tests, then time
POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/libpolee_hip_dmabuiltin.so timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_diversity.py -m gpu -x -q > $OUT/dmabuiltin_tests.log 2>&1; tail -3 $OUT/dmabuiltin_tests.log
bash tools/probe/r04_sweep.sh dma "|POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/libpolee_hip_dmabuiltin.so||POLEE_HIP_LIB=$GRAFT_REPO_ROOT/polee_amd/csrc/libpolee_hip_dmabuiltin.so" literal p0
bash tools/probe/r04_sweep.sh wgs "|POLEE_STREAM_WGS_PER_CU=3|POLEE_STREAM_WGS_PER_CU=2||POLEE_STREAM_WGS_PER_CU=3" fixture literal
