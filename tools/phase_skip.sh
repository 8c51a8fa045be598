#!/bin/bash
# Diagnostic build only (make -C social_stgcnn_amd/csrc DIAG=1): kernel times with phases switched off (WRONG results,
# timing only) -- the latency of a lone scene-wave (128 scenes, wave path) and the paired time (2048 scenes), per phase.
#   F: 16 = st_gcn block only (no convs)        B: 4 = no block tail, 1024 = no input-gradient convs, 512 = no dz passes
export STG_USE_DIAG_LIB=1
for b in "--batch 128 --wave-path" ""; do
  for skip in 0 16 4 1024 512; do
    echo "== $b skip=$skip"
    STG_DEBUG_SKIP=$skip TAG=ps tools/gpu.sh prof $b --no-extras --steps 10 --repeats 3 2>&1 | grep -E "txp_(fwd|bwd)_x6"
  done
done
