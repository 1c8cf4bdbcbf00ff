import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loik_amd
from loik_amd import workloads
for B in (96,):
    wl = workloads.talos_c3(B)
    m, prm = wl["model"], wl["params"]
    try:
        print("B", B, "create"); s = loik_amd.BatchedLoik(m, B, **prm)
    except Exception as e:
        print("FAILED", e)
