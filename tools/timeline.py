"""One mini-batch of the PPO update as a timeline, from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):
which kernels overlap (queues = HIP-graph branches), where the device idles, what the critical path is.

    python tools/timeline.py <..._kernel_trace.csv> [--anchor loss_kernel] [--which -2]

The window runs from the start of the `--which`-th launch of the anchor kernel to the start of the next one."""
import argparse
import csv


def short(name):
    for cut in ("(", "<"):
        if cut in name and not name.startswith("void"):
            name = name.split(cut)[0]
    name = name.replace("void ", "")
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        mt = name.split("_MT")[1].split("_")[0] if "_MT" in name else "?"
        return f"hipBLASLt {name.split('_')[1 if name.startswith('Cijk') else 2]} MT{mt}"
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default="loss_kernel")
    ap.add_argument("--which", type=int, default=-3)
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if r[2].startswith(args.anchor)]
    a, b = anchors[args.which], anchors[args.which + 1]
    t0 = rows[a][0]
    win = rows[a:b]
    period = rows[b][0] - t0
    print(f"# window: launch {args.which} of {args.anchor} to the next one: {period / 1e3:.1f} us, {len(win)} kernels")
    print(f"# {'start':>8s} {'dur':>7s} {'gap':>6s}  q  kernel          (gap = device idle before this kernel: start - latest end of anything earlier)")
    busy_end, idle, ksum = t0, 0, 0
    for s, e, n, q in win:
        gap = max(0, s - busy_end)
        idle += gap
        ksum += e - s
        print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap / 1e3:6.1f}  {q}  {short(n)}")
        busy_end = max(busy_end, e)
    tail = max(0, rows[b][0] - busy_end)
    print(f"# sum of kernel durations {ksum / 1e3:.1f} us; device idle inside the window {(idle + tail) / 1e3:.1f} us "
          f"({100.0 * (idle + tail) / period:.1f} %); overlap {(ksum - (period - idle - tail)) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
