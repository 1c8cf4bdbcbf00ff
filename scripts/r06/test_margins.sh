#!/bin/bash
# how much of its tolerances each end-to-end comparison of the GPU suite uses (tests/helpers.py::assert_end_to_end with LOIKB_TEST_MARGINS)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_q}; mkdir -p $O; cd $GRAFT_REPO_ROOT; rm -f $O/test_margins.txt
LOIKB_TEST_MARGINS=$O/test_margins.txt timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
sort $O/test_margins.txt | uniq > $O/test_margins_sorted.txt; wc -l $O/test_margins_sorted.txt
