"""Time the stage kernels of the bench workload alone (env knobs are read by the library at first use, so one
process per setting):  python tools/stage_bench.py [stage ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

w = dict(N=2, C=512, Co=512, H=97, W=97, k=3, pad=4, dil=4, dg=1)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    k, v = kv.split("=")
    w[k] = int(v)
only = [a for a in sys.argv[1:] if "=" not in a]
t = bench.make_inputs(w, torch.device("cuda:0"), 0)
for r in bench.kernel_rooflines(w, t, 20)[0]:
    if not only or r["kernel"] in only:
        print("%-14s %.4f ms  %8.1f %s  frac %.3f" % (r["kernel"], r["ms"], r["achieved"], r["unit"], r["frac"]))
