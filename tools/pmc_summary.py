#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs into the two artefacts kept under profiles/:

  python tools/pmc_summary.py --sq DIR --fetch DIR --write DIR --tcc DIR --tag round1_d

  profiles/<tag>_pmc_summary.csv   pass,kernel,counter,mean_per_launch,launches  (+ derived MFMA utilisation rows)
  profiles/<round>_traffic.json    HBM bytes per launch per kernel (round = the tag up to its first "_"), read by bench.py for
                                   roofline.traffic (labelled there as scaled from this committed pass)

Each DIR is the -d directory of ONE rocprofv3 pass (the counters do not fit one pass and gpurun refuses
--pmc together with trace domains), collected with `bench.py --steps 1 --warmup 1 --batch 256 --cpu-clouds 0`:
  --sq     SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  --fetch  FETCH_SIZE          --write  WRITE_SIZE          --tcc  TCC_HIT_sum TCC_MISS_sum
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(it tallies 128-B requests at 64 B); Infinity-Cache hits are counted, not excluded.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """kernel name WITH its template arguments (dec_main_kernel<false, 2> is the f32 decoder, <true, 2> the bf16x3 one: a pass that runs
    every arithmetic mode must not average them), without the parameter list"""
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])          # rows may be split per XCD / dimension
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
            per_dispatch[(r["Dispatch_Id"], "_duration_ns")] = float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (disp, ctr), v in per_dispatch.items():
            acc[names[disp]][ctr].append(v)
    return acc


def main():
    ap = argparse.ArgumentParser()
    for k in ("sq", "fetch", "write", "tcc"):
        ap.add_argument("--" + k, required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--batch", type=int, default=256, help="clouds per launch the passes ran at (bench.py --batch)")
    args = ap.parse_args()
    passes = {k: load(getattr(args, k)) for k in ("sq", "fetch", "write", "tcc")}
    mine = [k for k in passes["sq"] if not k.startswith("at::") and "rocclr" not in k]

    rows = []
    for pname, acc in passes.items():
        for kern in sorted(acc):
            if kern not in mine:
                continue
            for ctr, vals in sorted(acc[kern].items()):
                if not ctr.startswith("_"):
                    rows.append((pname, kern, ctr, sum(vals) / len(vals), len(vals)))
    # derived: MFMA-pipe utilisation.  SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over the 1024 SIMDs, GRBM_GUI_ACTIVE
    # sums active cycles over the 8 XCDs, so utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024); the clock under load is
    # GUI_ACTIVE / 8 / kernel duration.
    for kern in mine:
        c = passes["sq"][kern]
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            if mf > 0:
                rows.append(("derived", kern, "MFMA_pipe_utilisation", mf / (gui * 128.0), len(c["GRBM_GUI_ACTIVE"])))
        if c.get("GRBM_GUI_ACTIVE") and c.get("_duration_ns"):
            # clock the chip held during the dispatch (it lowers its clock under load: MI355X_MICROARCH.md, DVFS): cycles / ns
            clk = [a / 8.0 / d for a, d in zip(c["GRBM_GUI_ACTIVE"], c["_duration_ns"]) if d >= 1e6]     # reads high below ~0.3 ms
            if clk:
                rows.append(("derived", kern, "clock_GHz_under_load", sum(clk) / len(clk), len(clk)))
    out = os.path.join(ROOT, "profiles", args.tag + "_pmc_summary.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "counter", "mean_per_launch", "launches"])
        w.writerows(rows)

    derived = {(k, c): v for (pn, k, c, v, _) in rows if pn == "derived"}
    kernels = {}
    for kern in mine:
        fs = passes["fetch"].get(kern, {}).get("FETCH_SIZE")
        ws = passes["write"].get(kern, {}).get("WRITE_SIZE")
        if not fs or not ws:
            continue
        fk, wk = sum(fs) / len(fs), sum(ws) / len(ws)
        ent = {"FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        t = passes["tcc"].get(kern, {})
        if t.get("TCC_HIT_sum") and t.get("TCC_MISS_sum"):
            h, m = sum(t["TCC_HIT_sum"]), sum(t["TCC_MISS_sum"])
            ent["l2_hit_rate"] = h / (h + m) if h + m else None
        if (kern, "MFMA_pipe_utilisation") in derived:
            ent["mfma_pipe_busy"] = derived[(kern, "MFMA_pipe_utilisation")]
        if (kern, "clock_GHz_under_load") in derived:
            ent["clock_ghz_under_load"] = derived[(kern, "clock_GHz_under_load")]
        kernels[kern] = ent
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate passes, bench.py --steps 1 --warmup 1 "
            "--batch 256 --cpu-clouds 0 (all arithmetic modes and both octree modes in one pass; tools/pmc_summary.py, tag %s). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE "
            "doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); Infinity-Cache hits are counted." % args.tag)
    # ONE pass tag per file (round-3 review: a file that merged kernels of two passes could not be read as one measurement): the passes
    # are collected WITHOUT --one-mode, so a single tag covers the kernels of all three arithmetic modes
    tpath = os.path.join(ROOT, "profiles", args.tag.split("_")[0] + "_traffic.json")
    for v in kernels.values():
        v["pass_tag"] = args.tag
    with open(tpath, "w") as f:
        json.dump({"note": note, "batch": args.batch, "pass_tag": args.tag, "kernels": kernels}, f, indent=1)
    print(out)
    for k, v in kernels.items():
        print(f"{k:34s} hbm {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB  l2 hit {v.get('l2_hit_rate')}")


if __name__ == "__main__":
    main()
