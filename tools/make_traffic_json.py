"""profiles/rNN_hbm_traffic.json from two rocprofv3 --pmc passes over tools/stage_bench.py (default bench workload):
    python tools/make_traffic_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <commit> <out.json>
Per-launch averages.  traffic_bytes = 2 * FETCH_SIZE + WRITE_SIZE in bytes: rocprofv3 prints both counters in units of
1000 B ("KB"); on gfx950 wide coalesced reads are tallied at half (MI355X_MICROARCH.md, HBM section), the factor is
calibrated for the 16-B-per-lane loads of the GEMMs and is an upper bound for the 4-B gathers of im2col / scatter.
The digest of csrc/ ties the figures to the kernels they were measured on (bench.py drops them when it differs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

KERNELS = {   # bench.py stage name -> substring of the kernel's name
    "im2col": "im2col_tile_kernel<true",
    "gemm_out": "gemm_f32_mfma<true, false, false>",
    "gemm_gcol": "gemm_f32_mfma<false, false, false>",
    "col2im_coord": "col2im_coord_tile9_adaptive_kernel<true",
    "gemm_gweight": "gemm_f32_mfma<true, true, true>",
}


def averages(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = averages(sys.argv[1], "FETCH_SIZE"), averages(sys.argv[2], "WRITE_SIZE")
    out = {"_note": __doc__.split("\n", 3)[3].strip(), "commit": sys.argv[3], "csrc_sha256": bench.csrc_digest(),
           "workload": "N=2, C=512->512, 97x97, pad=dil=4 (bench.py default)", "kernels": {}}
    for stage, sub in KERNELS.items():
        fk = [v for k, v in fetch.items() if sub in k]
        wk = [v for k, v in write.items() if sub in k]
        if len(fk) != 1 or len(wk) != 1:
            raise SystemExit("kernel %r: %d fetch / %d write matches" % (sub, len(fk), len(wk)))
        out["kernels"][stage] = {"fetch_size_kb": round(fk[0], 1), "write_size_kb": round(wk[0], 1),
                                 "traffic_bytes": int(round((2 * fk[0] + wk[0]) * 1000))}
    with open(sys.argv[4], "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["kernels"]))


if __name__ == "__main__":
    main()
