"""Per-kernel MFMA / VALU / LDS utilisation and wave-state split from rocprofv3 SQ counter passes (north_star: "rocprof-reported
... MFMA utilisation against gfx950 peak accompany each kernel") -> profiles/<round>_pmc_util.json.

Collection (each pass its own run, counters + --kernel-trace only):
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_sq1 -o p --output-format csv -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU \
              SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d gpurun_out/pmc_sq2 ...
    python tools/pmc_util.py r02 gpurun_out/pmc_sq1/p_counter_collection.csv gpurun_out/pmc_sq2/p_counter_collection.csv

Definitions (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip's 1024 SIMDs;
SQ_BUSY_CYCLES counts per shader engine (32) -> cycles of the dispatch = SQ_BUSY_CYCLES / 32; wave-state counters are in
quad-cycles. mfma_util = MFMA busy / (1024 x dispatch cycles) is the fraction of the dense-MFMA peak AT THE CLOCK THE CHIP HELD;
clock_ghz = dispatch cycles / wall time shows how far below the 2.4 GHz of the quoted peak that was.
"""
import csv, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import variant  # noqa: E402


def short(sym):
    v = variant(sym)
    if v:
        return v
    s = sym.replace("ffp::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return s.split("(")[0][:48]


def main():
    tag = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for path in sys.argv[2:]:
        seen = set()
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                a = acc[k][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
                key = (path, r["Dispatch_Id"])
                if key not in seen and r["Counter_Name"] in ("SQ_BUSY_CYCLES", "SQ_INSTS_MFMA"):
                    seen.add(key)
                    d = dur[(k, path)]
                    d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    out = {"_source": "rocprofv3 --kernel-trace --pmc <SQ counters> (separate passes) over bench.py; per-dispatch means; tools/pmc_util.py", "kernels": {}}
    for k, cs in acc.items():
        m = {c: v[1] / v[0] for c, v in cs.items() if v[0]}
        us = [d[1] / d[0] for (kk, _), d in dur.items() if kk == k and d[0]]
        e = {"launches_profiled": int(max(v[0] for v in cs.values())), "avg_us": round(sum(us) / len(us), 2) if us else None}
        if "SQ_BUSY_CYCLES" in m and us:
            cyc = m["SQ_BUSY_CYCLES"] / 32.0
            e["dispatch_cycles"] = round(cyc)
            e["clock_ghz"] = round(cyc / (us[0] * 1e3), 2)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_util"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
            if "SQ_LDS_IDX_ACTIVE" in m:
                e["lds_array_busy"] = round(m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 4)
            if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
            w = m["SQ_WAVE_CYCLES"]
            e["wave_state"] = {"issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / w, 3), "parked_waitcnt_barrier": round(m.get("SQ_WAIT_ANY", 0) / w, 3),
                               "issue_stalled": round(m.get("SQ_WAIT_INST_ANY", 0) / w, 3)}
        if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"]:
            e["valu_per_mfma"] = round(m.get("SQ_INSTS_VALU", 0) / m["SQ_INSTS_MFMA"], 2)
            e["lds_per_mfma"] = round(m.get("SQ_INSTS_LDS", 0) / m["SQ_INSTS_MFMA"], 2)
        out["kernels"][k] = e
    path = os.path.join(ROOT, "profiles", tag + "_pmc_util.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    top = sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("avg_us") or 0) * kv[1]["launches_profiled"])[:12]
    for k, e in top:
        print(f"{k:40s} n={e['launches_profiled']:5d} avg {e.get('avg_us')} us  mfma_util {e.get('mfma_util')}  clock {e.get('clock_ghz')}  lds {e.get('lds_array_busy')}  {e.get('wave_state')}")
    print("wrote", path)


if __name__ == "__main__":
    main()
