#!/bin/bash
# Class bounds of the team launch against the batch size (run through gpurun from the repo root; DIAGNOSTIC build:
# make -C social_stgcnn_amd/csrc DIAG=1): a scene of up to v1 pedestrians belongs to one wave, up to v2 to two, beyond to
# four.  Synthetic V = 32 and the eth/train crowd histogram, per-kernel device times -> profiles/r03_team_bounds.log.
# The bounds the library picks by itself (team_geom, txp_x6.hip) come from this table.
export STG_USE_DIAG_LIB=1
for b in 128 256 512 768 1024 1536 2048; do
  for vv in "32 64" "16 32" "8 16"; do
    set -- $vv
    STG_TEAM=1 STG_TEAM_V1=$1 STG_TEAM_V2=$2 TAG=tb tools/gpu.sh ksweep "--batch $b" "--batch $b --ragged shuffled" | sed "s/^/v1=$1 v2=$2 /"
  done
done
