"""Timeline of one steady-state LM iteration from a rocprofv3 --kernel-trace CSV:
per kernel start offset, duration and the gap since the previous kernel ended
(kernels on the side stream overlap: their gap is negative).

  python tools/timeline.py gpurun_out/prof_x [iteration_index_from_end]
"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        n = n[n.find("k_"):].split("(")[0].split("<")[0] if "k_" in n else n[:30]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# iterations are delimited by k_scalars (last kernel of an iteration)
ends = [i for i, r in enumerate(rows) if r[2] == "k_scalars"]
if len(ends) < back + 2:
    sys.exit("not enough iterations in the trace")
a, b = ends[-back - 2] + 1, ends[-back - 1] + 1
it = rows[a:b]
t0 = rows[a - 1][1]          # end of the previous iteration's last kernel
prev_end = t0
print("%-20s %10s %10s %10s" % ("kernel", "start_us", "dur_us", "gap_us"))
tot_gap = 0.0
busy = 0.0
for s, e, n in it:
    gap = (s - prev_end) / 1e3
    print("%-20s %10.2f %10.2f %10.2f" % (n, (s - t0) / 1e3, (e - s) / 1e3, gap))
    if gap > 0:
        tot_gap += gap
    busy += (e - max(s, prev_end)) / 1e3 if e > prev_end else 0.0
    prev_end = max(prev_end, e)
print("iteration: %.2f us wall, %.2f us idle gaps, %.2f us covered by kernels, %d launches"
      % ((prev_end - t0) / 1e3, tot_gap, busy, len(it)))
# mean iteration time over the steady-state window
span = [(rows[ends[k + 1]][1] - rows[ends[k]][1]) / 1e3 for k in range(len(ends) - 12, len(ends) - 2)]
print("mean of 10 steady iterations: %.2f us" % (sum(span) / len(span)))
