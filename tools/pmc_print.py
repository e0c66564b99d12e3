"""Per-kernel means of every counter in rocprofv3 counter_collection CSVs: python tools/pmc_print.py <csv>... [--min-us N]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
    seen = set()
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("ffp::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            a = acc[k][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d = dur[k]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, cs in acc.items():
    us = dur[k][1] / max(dur[k][0], 1)
    if us < 20:
        continue
    print(f"== {k}  avg {us:.1f} us over {dur[k][0]} dispatch-passes")
    m = {c: v[1] / v[0] for c, v in cs.items()}
    for c in sorted(m):
        print(f"   {c:34s} {m[c]:16.0f}")
    if "SQ_BUSY_CYCLES" in m:
        cyc = m["SQ_BUSY_CYCLES"] / 32.0
        print(f"   -> dispatch cycles {cyc:.0f}, clock {cyc / (us * 1e3):.2f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print(f"   -> mfma pipe busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}")
        if "SQ_LDS_IDX_ACTIVE" in m: print(f"   -> lds array busy {m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.3f}")
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        w = m["SQ_WAVE_CYCLES"]
        print("   -> wave state: " + " ".join(f"{n} {m.get(c, 0) / w:.3f}" for n, c in [("issuing", "SQ_ACTIVE_INST_ANY"), ("parked", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("lds_stall", "SQ_WAIT_INST_LDS")]))
