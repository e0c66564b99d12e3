"""Print a rocprofv3 kernel_stats.csv as time per frame (tuning aid): tools/kstats.py <csv> <frames>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    n = r["Name"].replace("ffp::", "").replace("(anonymous namespace)::", "")[:64]
    print(f"{n:64s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  per frame {float(r['TotalDurationNs']) / frames / 1e3:8.1f} us {r['Percentage']}%")
