#!/bin/bash
# the alternate kernel paths behind the environment switches still pass the trainer-level suites: env_matrix.sh [tag rNN]
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_env_matrix.txt
echo "# tests/test_gpu_step.py + test_gpu_ae.py + test_gpu_lpips.py + test_gpu_inference.py under each alternate-path switch (scripts/env_matrix.sh), kernel sources $(cd $GRAFT_REPO_ROOT && python3 -c 'import bench; print(bench.csrc_sha())' 2>/dev/null)" > $OUT
for E in "AESR_BN_FUSED=0" "AESR_BN_FUSED_MAX_IMAGES=1000" "AESR_WINO_RING=0" "AESR_WINO_RING=1" "AESR_FUSE_EVAL_BN=0" "AESR_RING_KSPLIT=1" "AESR_LPIPS_FOLD=0" "AESR_WINO_RES=0" "AESR_FUSE_STEM=0" "AESR_DEFER_REDUCE=0" "AESR_LAZY_ZERO=0" "AESR_FOLD_UPSAMPLE=0"; do
  R=$(cd $GRAFT_REPO_ROOT && env $E timeout -k 10 500 python3 -m pytest tests/test_gpu_step.py tests/test_gpu_ae.py tests/test_gpu_lpips.py tests/test_gpu_inference.py -q -p no:cacheprovider 2>&1 | tail -1)
  echo "$E: $R" | tee -a $OUT
done
