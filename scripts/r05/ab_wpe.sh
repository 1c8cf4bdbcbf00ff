#!/bin/bash
# A/B of k_flat2 built for two / three wavefronts per SIMD (LOIKB_FLAT_WPE) inside ONE gpurun call
cd ${GRAFT_REPO_ROOT:-.}
for w in 2 3; do
  export LOIKB_FLAT_WPE=$w
  TAG="[wpe $w]" python scripts/r04/lone.py 2>&1 | head -1
  TAG="[wpe $w ordered]" python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[wpe $w arrival]" LOIKB_FLAT_ORDER=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[wpe $w arrival unsliced]" LOIKB_FLAT_ORDER=0 LOIKB_FLAT_SLICE=0 python scripts/r03/quick_headline.py 65536 6 | tail -1
  TAG="[wpe $w]" python scripts/r03/quick_headline.py 262144 4 | tail -1
done
