"""Launch sequence of ONE network evaluation from a rocprofv3 kernel trace (rocpd sqlite): the dispatches between the last two
se3_step_kernel launches, in order, with grid and duration -- which layer costs what, and how many launches an evaluation takes.

    python tools/rocpd_sequence.py gpurun_out/prof_x/.../bench_results.db out.md
"""
import re
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    pick = lambda *names: next(n for n in names if n in cols)  # noqa: E731
    name, start, end = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
    gx = next((n for n in ("grid_x", "grid_size_x", "grid_size") if n in cols), None)
    wx = next((n for n in ("workgroup_x", "workgroup_size_x", "workgroup_size") if n in cols), None)
    sel = f"select {name}, {start}, {end}, {gx or 0}, {wx or 1} from kernels order by {start}"
    rows = list(c.execute(sel))
    marks = [i for i, r in enumerate(rows) if "se3_step_kernel" in r[0]]
    lo, hi = marks[-2] + 1, marks[-1] + 1
    seq = rows[lo:hi]
    t0 = seq[0][1]
    lines = ["| # | kernel | workgroups | start us | dur us | gap before us |", "|---|---|---|---|---|---|"]
    prev_end = None
    tot = 0.0
    for i, (n, s, e, g, w) in enumerate(seq):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n).replace("at::native::", "")
        n = n.split("(")[0][:70]
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        lines.append(f"| {i} | `{n}` | {g // max(w, 1)} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} |")
        prev_end = e
        tot += (e - s) / 1e3
    span = (seq[-1][2] - t0) / 1e3
    lines.append("")
    lines.append(f"{len(seq)} launches, {tot:.0f} us of kernel time in a {span:.0f} us span")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-3:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
