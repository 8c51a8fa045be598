#!/bin/bash
# Static check of the bf16-pipe kernels' ISA (no GPU needed): s_waitcnt vmcnt(...) instructions located between the
# first and the last MFMA of each kernel -- i.e. inside the tile loops, where a vmcnt wait also waits for the
# acknowledgement of the previous tile's stores.  Expected: only the explicit drains in front of the loops.
set -e
cd "$(dirname "$0")/../social_stgcnn_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -S --cuda-device-only txp_wave.hip -o /tmp/txp_wave.s
for k in 17txp_fwd_x6_kernelILi4ELb0E 17txp_bwd_x6_kernelILi4ELb0E 17txp_fwd_x6_kernelILi4ELb1E 17txp_bwd_x6_kernelILi4ELb1E; do
  awk -v k="$k" 'index($0,k) && /^_ZN/ && /:/ {f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/txp_wave.s > /tmp/k.s
  first=$(grep -n v_mfma /tmp/k.s | head -1 | cut -d: -f1); last=$(grep -n v_mfma /tmp/k.s | tail -1 | cut -d: -f1)
  n=$(awk -v a="$first" -v b="$last" 'NR>a && NR<b && /s_waitcnt vmcnt/' /tmp/k.s | wc -l)
  n0=$(awk -v a="$first" -v b="$last" 'NR>a && NR<b && /s_waitcnt vmcnt\(0\)/' /tmp/k.s | wc -l)
  echo "$k: $(grep -c v_mfma /tmp/k.s) MFMAs, $(grep -c 'global_store' /tmp/k.s) global stores; between the first and the last MFMA: $n vmcnt waits, $n0 of them vmcnt(0)"
done
