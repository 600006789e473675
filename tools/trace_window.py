"""Steady-state breakdown of a rocprofv3 kernel trace: kernel time in the last `window_ms` of the run (the MIOpen
find phase and the warm-up are at the front).  usage: python tools/trace_window.py trace.csv window_ms steps_in_window"""
import collections
import csv
import sys


def main():
    path, win, steps = sys.argv[1], float(sys.argv[2]) * 1e6, int(sys.argv[3])
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
                for r in csv.DictReader(open(path)))
    # the run's epilogue (result checks, prints) may trail the last step: anchor on the last ddcn kernel if there is one
    ddcn_ends = [e for _, e, n in ev if "ddcn::" in n]
    end = max(ddcn_ends) if ddcn_ends else ev[-1][1]
    sel = [e for e in ev if end - win <= e[0] <= end]
    tot = sum(e[1] - e[0] for e in sel)
    d = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        key = ("ddcn:" + n.replace("(anonymous namespace)::", "").split("ddcn::")[1].split("(")[0][:44]
               if "ddcn::" in n else n[:64])
        d[key][0] += e - s
        d[key][1] += 1
    print("window %.1f ms = %d steps, GPU busy %.1f ms" % (win / 1e6, steps, tot / 1e6))
    print("per step:   ms  calls  kernel")
    for k, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:24]:
        print("       %7.2f %6.1f  %s" % (t / 1e6 / steps, c / steps, k))
    dd = sum(t for k, (t, c) in d.items() if k.startswith("ddcn:"))
    print("ddcn kernels: %.2f ms per step = %.1f %% of GPU busy time" % (dd / 1e6 / steps, 100 * dd / tot))


if __name__ == "__main__":
    main()
