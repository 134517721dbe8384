"""Dev tool: per-kernel MFMA utilisation and wave-state split from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
 SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE) over `tools/stage_bench.py`.

  python tools/pmc_mfma.py counter_collection.csv OUT.json

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE is the
sum over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 / wall time is also the effective clock (MI355X_MICROARCH.md)."""
import csv, json, sys, collections


def main():
    path, out = sys.argv[1:3]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    wall = collections.defaultdict(dict)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(k in name for k in ("mfma", "fewch", "conv3x3s1", "scatter", "rows_kernel", "first_", "hwc_pad", "rans_")):
                continue
            key = (name, int(r["Grid_Size"]))
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            wall[key][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = []
    for key, c in sorted(per.items(), key=lambda kv: -sum(wall[kv[0]].values())):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        ns = sum(wall[key].values()) / len(wall[key])
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        row = {"kernel": key[0], "grid_threads": key[1], "dispatches": len(wall[key]), "avg_ns_under_pmc": ns,
               "counters_mean": m}
        if gui:
            row["effective_clock_GHz"] = gui / 8 / ns
            row["MfmaUtil"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4)
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            row["wave_state_frac"] = {k: m[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in m}
        res.append(row)
    json.dump(res, open(out, "w"), indent=1)
    for r in res:
        print(r["kernel"], r["grid_threads"], "util=%.3f" % r.get("MfmaUtil", -1), "clk=%.2f" % r.get("effective_clock_GHz", -1),
              {k: round(v, 3) for k, v in r.get("wave_state_frac", {}).items()})


if __name__ == "__main__":
    main()
