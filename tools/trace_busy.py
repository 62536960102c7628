"""GPU busy time from a rocprofv3 --kernel-trace CSV: the union of the kernel intervals (what the device was occupied for), the
sum of the kernel durations (> union when kernels overlap on several streams) and the span from the first start to the last end.

    python tools/trace_busy.py <dir with *kernel_trace.csv> [skip_first_fraction]
"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5          # drop the warm-up / verification half of the run
iv = iv[int(len(iv) * skip):]
union = 0
cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
total = sum(e - s for s, e in iv)
span = max(e for _, e in iv) - iv[0][0]
print("kernels %d  span %.1f ms  busy (union) %.1f ms = %.0f %% of the span  sum of durations %.1f ms  overlap factor %.2f"
      % (len(iv), span / 1e6, union / 1e6, 100.0 * union / span, total / 1e6, total / max(union, 1)))
