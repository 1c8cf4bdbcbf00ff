#!/bin/bash
# compile the device code only and report registers / scratch of the lean kernels (no GPU needed)
cd /tmp && mkdir -p asm && cd asm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=on -std=c++17 -I /root/repo/include -x hip /root/repo/loik_amd/csrc/loik_host.hip -S --cuda-device-only -o cur.s "$@" 2>/dev/null
for k in k_leanIdLb1ELb0 k_leanIdLb1ELb1 k_leanIfLb1ELb0 k_tailIdLb1; do
  awk "/^_ZN5loikb[0-9]*${k}EE.*:/,/; Occupancy/" cur.s > k_$k.s
  echo "$k: scratch instrs $(grep -c 'scratch_' k_$k.s), $(grep -E '; ScratchSize|; NumVgprs|; Occupancy|; NumAgprs' k_$k.s | tr '\n' ' ')"
done
