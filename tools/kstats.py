"""Pretty-print a rocprofv3 *_kernel_stats.csv: python tools/kstats.py <csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total {tot / 1e3 / steps:.1f} us/step, {calls / steps:.1f} launches/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].replace("impnn::(anonymous namespace)::", "").replace("void at::native::", "at::")[:64]
    print(f"{n:64s} {int(r['Calls']) / steps:6.1f}/step {int(r['TotalDurationNs']) / 1e3 / steps:8.1f} us/step "
          f"{float(r['AverageNs']) / 1e3:7.2f} us {float(r['Percentage']):5.1f}%")
