"""Per-kernel sums of the SQ / GRBM counters of scripts/pmc_sq.sh passes -> one JSON.

usage: pmc_sq.py <dir with p1/ p2/ p3/ rocprofv3 csv output> <out.json> "<command string>"
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles summed over the SQs, GRBM_GUI_ACTIVE is summed over
the 8 XCDs (effective clock = GRBM_GUI_ACTIVE / 8 / kernel time).
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    n = name.split("(")[0]
    n = n.replace("nnmpc::", "").replace("(anonymous namespace)::", "")
    return n.split("<")[0] if "asm_lambda" not in n and "gemm" not in n else n


def main():
    root, out, cmd = sys.argv[1:4]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    times = collections.defaultdict(float)
    for p in sorted(glob.glob(root + "/p[0-9]")):
        cc = glob.glob(p + "/**/*counter_collection.csv", recursive=True)
        if not cc:
            continue
        seen = set()
        for r in csv.DictReader(open(cc[0])):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if p.endswith("p1") and key not in seen:
                seen.add(key)
                launches[k] += 1
        kt = glob.glob(p + "/**/*kernel_trace.csv", recursive=True)
        if kt and p.endswith("p1"):
            for r in csv.DictReader(open(kt[0])):
                times[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = {"command": cmd, "units": __doc__.split("Units")[1].strip(), "kernels": {}}
    for k in sorted(acc, key=lambda k: -times.get(k, 0.0)):
        c = acc[k]
        d = {"launches": launches.get(k, 0), "time_us_under_pmc": times.get(k, 0.0), "counters": dict(c)}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["share_of_wave_cycles"] = {n: c[n] / wc for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                 "SQ_ACTIVE_INST_VALU") if n in c}
        if times.get(k) and "GRBM_GUI_ACTIVE" in c:
            npass = len(glob.glob(root + "/p[0-9]"))
            d["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / npass / 8 / (times[k] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                d["mfma_busy_over_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
        if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            d["per_wave"] = {n: c[n] / c["SQ_WAVES"] for n in c if n.startswith("SQ_INSTS") or n == "SQ_WAVE_CYCLES"}
        res["kernels"][k] = d
    json.dump(res, open(out, "w"), indent=1)
    for k, d in list(res["kernels"].items())[:14]:
        print(k, d["launches"], f"{d['time_us_under_pmc']:.0f} us", d.get("share_of_wave_cycles"), d.get("per_wave"))


if __name__ == "__main__":
    main()
