"""launches per train step by kernel = (calls in a K2-step run - calls in a K1-step run) / (K2 - K1), from two
rocprofv3 --stats kernel_stats.csv files of the same bench command (tools/collect_profiles.sh 1 / 1b); torch's own
kernels (at::native / rocclr) are listed one by one, the library's kernels as a total.
    python tools/launches_per_step.py A.csv K1 B.csv K2"""
import csv, sys
a, k1, b, k2 = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
rd = lambda f: {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
A, B = rd(a), rd(b)
rows, lib_calls, lib_ns, aten_calls, aten_ns = [], 0.0, 0.0, 0.0, 0.0
for n, (cb, tb) in B.items():
    ca, ta = A.get(n, (0, 0.0))
    per, ns = (cb - ca) / (k2 - k1), (tb - ta) / (k2 - k1)
    if per <= 0:
        continue
    if "at::" in n or "rocclr" in n or "elementwise_kernel" in n:
        rows.append((per, ns, n))
        aten_calls += per
        aten_ns += ns
    else:
        lib_calls += per
        lib_ns += ns
print(f"per train step: {lib_calls:.1f} launches of the library's kernels ({lib_ns / 1e6:.2f} ms), "
      f"{aten_calls:.1f} launches of torch's own kernels ({aten_ns / 1e6:.3f} ms):")
for per, ns, n in sorted(rows, reverse=True):
    print(f"  {per:6.1f} x {ns / per / 1e3:8.1f} us  {n[:140]}")
