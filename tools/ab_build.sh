#!/bin/bash
# Same-box A/B of a kernel change (boxes differ by +-3 %, most single changes by less): builds the WORKING TREE's sources into
# csrc/libstgcnn_hip_diag.so WITHOUT -DSTG_DIAG (the name the binding loads under STG_USE_DIAG_LIB=1) while
# csrc/libstgcnn_hip.so stays what `make` last built from the committed sources:
#   git stash; make -C social_stgcnn_amd/csrc; git stash pop; tools/ab_build.sh
#   gpurun -- 'for i in 1 2; do STG_USE_DIAG_LIB=1 TAG=abA tools/gpu.sh ksweep "--peds 32"; TAG=abB tools/gpu.sh ksweep "--peds 32"; done'
# (afterwards: make -C social_stgcnn_amd/csrc DIAG=1 restores the diagnostic library)
set -e
cd "$(dirname "$0")/../social_stgcnn_amd/csrc"
mkdir -p ab
make OBJDIR=ab LIB=libstgcnn_hip_diag.so -j6 | grep -E "error|warning: v" || true
rm -rf ab
md5sum libstgcnn_hip.so libstgcnn_hip_diag.so
