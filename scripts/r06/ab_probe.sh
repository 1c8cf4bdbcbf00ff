#!/bin/bash
# A/B inside ONE gpurun call: the single time-sliced launch (LOIKB_FLAT_PROBE=0) against probe + finish, several probe lengths and batches
cd ${GRAFT_REPO_ROOT:-.}
for B in 65536 32768 131072; do
  for P in 0 192 256 288 320 384 448; do
    TAG="[probe $P]" LOIKB_FLAT_PROBE=$P timeout 300 python scripts/r06/quick_probe.py $B 7
  done
done
for M in 16 64 128; do TAG="[probe 320 mark $M]" LOIKB_FLAT_PROBE=320 LOIKB_FLAT_PROBE_MARK=$M timeout 300 python scripts/r06/quick_probe.py 65536 7; done
TAG="[probe 320, full table]" LOIKB_FLAT_BUILD=0 LOIKB_FLAT_PROBE=320 timeout 300 python scripts/r06/quick_probe.py 65536 7
TAG="[probe 0, full table]" LOIKB_FLAT_BUILD=0 LOIKB_FLAT_PROBE=0 timeout 300 python scripts/r06/quick_probe.py 65536 7
TAG="[ordered repeat]" LOIKB_FLAT_ORDER=1 timeout 300 python scripts/r06/quick_probe.py 65536 7
for B in 16384 262144; do for P in 0 320; do TAG="[probe $P, slices forced]" LOIKB_FLAT_SLICE=288 LOIKB_FLAT_SLICE2=96 LOIKB_FLAT_PROBE=$P timeout 300 python scripts/r06/quick_probe.py $B 6; done; done
