"""Per-kernel time per step from a rocprofv3 --kernel-trace --stats csv (kernel_stats.csv).
usage: python tools/kstats.py <dir> <steps+warmup> [top]"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = 0.0
out = []
for r in rows:
    ns = float(r["TotalDurationNs"]); n = int(r["Calls"])
    tot += ns
    out.append((ns / steps / 1e6, n / steps, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
out.sort(reverse=True)
print(f"kernel sum {tot / steps / 1e6:.3f} ms/step over {steps:.0f} steps")
for ms, n, avg, name in out[:top]:
    print(f"{ms:7.3f} ms  {n:7.1f} calls  {avg:8.1f} us  {name}")
