"""Aggregate rocprofv3 PMC passes into HBM bytes per launch for each conv kernel variant -> profiles/<round>_pmc_traffic.json (argv[3], default r02).

Collection (two separate passes, counters only, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv
FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled.
"""
import csv, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def variant(sym: str):
    if "conv_trunk_kernel" in sym:
        return "f16_k3s1_trunk"
    if "conv_rows16pc_kernel" in sym:
        return "f16_k3s1_rows16pc"
    if "conv_rows16_kernel" in sym:
        return "f16_k3s1_rows16"
    m = re.search(r"conv_pw_kernel<(\d+), *(\d+), *(\d+), *(\d+), *(true|false)(?:, *(true|false))?>", sym) or \
        re.search(r"conv_pw_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E(?:Lb([01])E)?", sym)
    if m:                                                   # waves, pixel fragments, channel tiles (ring depth and the x2-source flag do not name a variant)
        nw, mi, nt = int(m.group(1)), int(m.group(2)), int(m.group(3))
        stream = m.group(6) in ("true", "1")
        return f"f32x3_k1s1_pw{mi}x{nt}" + ("s" if stream else "w" if nw == 8 else "")
    m = re.search(r"conv_k3d_kernel<(\d+), *(\d+), *(\d+), *(\d+), *(\d+)>", sym) or re.search(r"conv_k3d_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", sym)
    if m:                                                   # stride, wave grid, pixel fragments and channel tiles per wave (conv_k3d.hip)
        s_, wm, wn, mi, niw = (int(v) for v in m.groups())
        name = {(2, 2, 2, 2): "d128", (2, 2, 2, 1): "d64", (4, 1, 1, 1): "d32", (2, 2, 4, 2): "d128x256", (2, 2, 4, 1): "d64x256"}.get((wm, wn, mi, niw), f"d{wm}{wn}{mi}{niw}")
        return f"f32x3_k3s{s_}_{name}"
    m = re.search(r"conv_rows_kernel<(\d+), *(\d+)>", sym) or re.search(r"conv_rows_kernelILi(\d+)ELi(\d+)E", sym)
    if m:
        return "f16_k3s1_rows"
    if re.search(r"conv_mfma_kernel<[^>]*, *true>", sym) or re.search(r"conv_mfma_kernelI.*Lb1EEE", sym):
        return "f32x3_k3s2_stem_fused"                     # model.0 computed in model.1's loader
    m = re.search(r"conv_mfma_kernel<([^,]+), *(\d+), *(\d+), *(\d+), *(\d+), *(\d+), *(\d+), *(\d+)(?:, *false)?>", sym)
    if m:
        t, ks, s, wm, wn, mi, niw, kc = m.group(1), *map(int, m.groups()[1:])
    else:
        m = re.search(r"conv_mfma_kernelI(f|DF16_|NS_2X3E)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", sym)
        if not m:
            return None
        t, ks, s, wm, wn, mi, niw, kc = m.group(1), *map(int, m.groups()[1:])
    dt = "f32x3" if "X3" in t else ("f16" if ("16" in t or "half" in t.lower()) else "f32")
    full = 2 if s == 2 else 4            # MI of the full-size wide shape; narrow full = full / 2 (>= 1)
    if wm == 2:
        shape = "wide" if mi == full else "wideH"
    else:
        nfull = max(full // 2, 1)
        shape = ("narrow2" if niw == 2 else "narrow1") + ("" if mi == nfull else "H")
    return f"{dt}_k{ks}s{s}_{shape}"


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0, 0.0, ""])      # launches, counter sum, time sum, symbol
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            v = variant(r["Kernel_Name"])
            if v is None:
                continue
            a = acc[v]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            a[3] = r["Kernel_Name"]
    return acc


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over `python bench.py --steps 4 --warmup 2 "
                      "--no-cpu-baseline`; KiB units x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced "
                      "reads); per-launch means; aggregated by tools/pmc_traffic.py", "kernels": {}}
    for v in sorted(f):
        n, s, t, sym = f[v]
        fetch = s / n * 1024 * 2
        wr = (w[v][1] / w[v][0] * 1024) if v in w and w[v][0] else 0.0
        out["kernels"][v] = {"launches": n, "avg_us": t / n, "hbm_fetch_bytes_per_launch_x2corrected": fetch, "hbm_write_bytes_per_launch": wr,
                             "symbol": sym, "hbm_bytes_per_launch": fetch + wr}
    # argv[4]: stdout of the SAME command under the FETCH pass, run with --conv-totals: the library's own count of every conv launch of that process and
    # its algorithmic FLOPs / bytes -> per-launch means over the same launches the counters summed (one population for traffic / algorithmic bytes)
    if len(sys.argv) > 4:
        tot = {}
        try:
            with open(sys.argv[4]) as fh:
                for line in fh:
                    line = line.strip()
                    if line.startswith("{") and "conv_totals" in line:
                        tot = json.loads(line).get("conv_totals", {})
        except Exception as e:      # noqa: BLE001
            print("no conv_totals:", e)
        out["_population"] = ("algorithmic_bytes_per_launch / flops_per_launch: ffp_conv_totals_* printed by the profiled command itself (bench.py --conv-totals) — every "
                              "launch of the process, hipGraph replays included; launches_counted_by_the_library should equal launches (the counters' own launch count)")
        for v, k in out["kernels"].items():
            t = tot.get(v)
            if t and t["launches"]:
                k["launches_counted_by_the_library"] = t["launches"]
                k["algorithmic_bytes_per_launch"] = t["bytes"] / t["launches"]
                k["flops_per_launch"] = t["flops"] / t["launches"]
                k["traffic_over_algorithmic"] = round(k["hbm_bytes_per_launch"] / max(k["algorithmic_bytes_per_launch"], 1.0), 3)
    path = os.path.join(ROOT, "profiles", (sys.argv[3] if len(sys.argv) > 3 else "r02") + "_pmc_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, "with", len(out["kernels"]), "kernel variants")


if __name__ == "__main__":
    main()
