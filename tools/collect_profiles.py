#!/usr/bin/env python3
"""Copy the outputs of tools/profile_round.sh (gpurun_out/<tag>/) into profiles/ under the round's names and reduce the PMC passes.

    python tools/collect_profiles.py --src gpurun_out/r4p --round round4 [--pmc-tag round4_a]

  bench.json / bench_under_rocprof.json                 -> profiles/<round>_bench.json, <round>_bench_under_rocprof.json
  trace/**/kernel_stats.csv, kernel_trace.csv           -> profiles/<round>_kernel_stats.csv, <round>_kernel_trace.csv
  sq/ fetch/ write/ tcc/ (rocprofv3 --pmc passes)       -> tools/pmc_summary.py --tag <pmc-tag>: profiles/<pmc-tag>_pmc_summary.csv, <round>_traffic.json
  <workload>.json, trace_<workload>/**/kernel_stats.csv -> profiles/<round>_<workload>_bench.json, <round>_<workload>_kernel_stats.csv"""
import argparse
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        raise SystemExit(f"{path}: no JSON line")
    return lines[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True)
    ap.add_argument("--round", required=True)
    ap.add_argument("--pmc-tag", default=None)
    a = ap.parse_args()
    src, P = os.path.join(ROOT, a.src) if not os.path.isabs(a.src) else a.src, os.path.join(ROOT, "profiles")
    done = []

    def put_json(name, dst):
        f = os.path.join(src, name)
        if os.path.exists(f):
            open(os.path.join(P, dst), "w").write(last_json_line(f))
            done.append(dst)

    def put_csv(pattern, dst):
        hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
        if hits:
            shutil.copyfile(hits[-1], os.path.join(P, dst))
            done.append(dst)

    put_json("bench.json", f"{a.round}_bench.json")
    put_json("bench_under_rocprof.json", f"{a.round}_bench_under_rocprof.json")
    put_csv("trace/**/*kernel_stats.csv", f"{a.round}_kernel_stats.csv")
    put_csv("trace/**/*kernel_trace.csv", f"{a.round}_kernel_trace.csv")
    put_csv("trace_headline/**/*kernel_stats.csv", f"{a.round}_headline_kernel_stats.csv")
    for wl in ("s3dis", "pppfbatch256", "pppetraingraph", "ipdaetrain"):
        put_json(f"{wl}.json", f"{a.round}_{wl}_bench.json")
        put_csv(f"trace_{wl}/**/*kernel_stats.csv", f"{a.round}_{wl}_kernel_stats.csv")
    if all(os.path.isdir(os.path.join(src, d)) for d in ("sq", "fetch", "write", "tcc")):
        tag = a.pmc_tag or a.round + "_a"
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] +
                              sum([["--" + d, os.path.join(src, d)] for d in ("sq", "fetch", "write", "tcc")], []) + ["--tag", tag])
        done += [f"{tag}_pmc_summary.csv", f"{a.round}_traffic.json"]
    print("profiles/: " + ", ".join(done))


if __name__ == "__main__":
    main()
