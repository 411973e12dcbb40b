"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small CSV + markdown
table: per-kernel calls, total and average duration, share of GPU kernel time.

    python tools/rocpd_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_kernel_stats
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return f"rocBLAS/Tensile sgemm MT{mt.group(1) if mt else ''}"
    m = re.match(r"(void )?([\w:<>, ]+?)\(", name)
    base = (m.group(2) if m else name).replace("at::native::", "")
    return base[:90]


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, tot, avg, pct in rows:
        a = agg.setdefault(short(name), [0, 0.0, 0.0])
        a[0] += calls
        a[1] += tot
        a[2] += pct
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(out + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for k, (calls, tot, pct) in items:
            w.writerow([k, calls, f"{tot/1e3:.3f}", f"{tot/calls:.2f}", f"{pct:.3f}"])  # rocpd durations are in us
    with open(out + ".md", "w") as f:
        f.write("| kernel | calls | total ms | avg us | % of kernel time |\n|---|---|---|---|---|\n")
        for k, (calls, tot, pct) in items[:25]:
            f.write(f"| `{k}` | {calls} | {tot/1e3:.1f} | {tot/calls:.1f} | {pct:.2f} |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
