LOGLIK_TUNE_FLAGS = -mllvm -greedy-regclass-priority-trumps-globalness -mllvm -join-splitedges -mllvm -structurizecfg-skip-uniform-regions=true
HIPCC_VERSION = HIP version: 7.2.26015-fc0010cf6a
