"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py: the window between the starts of two launches
of a once-per-step kernel (default k_hg_acc = hash-grid backward) isolates whole optimisation steps from set-up work.
usage: step_window.py kernel_trace.csv [first_step last_step] -> CSV on stdout (name, calls/step, ms/step, avg_us)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "kernel_name"
s_k = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "start_timestamp"
e_k = "End_Timestamp" if "End_Timestamp" in rows[0] else "end_timestamp"
rows.sort(key=lambda r: int(r[s_k]))
marks = [int(r[s_k]) for r in rows if "k_hg_acc" in r[name_k]]
a = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 4
b = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 1
t0, t1, n = marks[a], marks[b], b - a
acc = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in rows:
    s = int(r[s_k])
    if t0 <= s < t1:
        d = int(r[e_k]) - s
        acc[r[name_k]][0] += 1
        acc[r[name_k]][1] += d
        busy += d
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls_per_step", "ms_per_step", "avg_us", f"# window = {n} steps, {(t1 - t0) / n / 1e6:.2f} ms/step wall, {busy / n / 1e6:.2f} ms/step kernel time"])
for k, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k[:160], f"{c / n:.1f}", f"{d / n / 1e6:.3f}", f"{d / c / 1e3:.1f}"])
