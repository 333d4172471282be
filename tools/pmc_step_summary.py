"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py` into HBM bytes per step and per kernel
family.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced reads, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.

    python tools/pmc_step_summary.py <fetch_counter_csv> <write_counter_csv> <steps> > profiles/rNN/pmc_step.json
"""
import csv, json, sys


def family(name: str) -> str:
    for key, fam in (("gemm_bf16", "gemm"), ("gemm_grouped", "gemm"), ("splitk_reduce", "gemm"), ("attn_", "attention"), ("layernorm", "layernorm"),
                     ("colsum", "colsum"), ("colreduce", "colsum"), ("adam", "adam")):
        if key in name:
            return fam
    return "other"


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:110]


by_kernel: dict = {}  # (kernel, grid) -> {counter: KiB, "n": dispatches}


def load(path: str, counter: str):
    out = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        fam = family(r.get("Kernel_Name", ""))
        out[fam] = out.get(fam, 0.0) + float(r["Counter_Value"])
        if fam == "gemm":
            ent = by_kernel.setdefault((short(r["Kernel_Name"]), int(r.get("Grid_Size", 0) or 0)), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
            ent[counter] += float(r["Counter_Value"])
            if counter == "FETCH_SIZE":
                ent["n"] += 1
    return out


fetch, write, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), float(sys.argv[3])
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res = {"steps_profiled": steps, "unit": "bytes per step", "fetch_correction": "x2 (gfx950 wide-read calibration)",
       # bench.py reports these bytes only while the GEMM sources (bench.GEMM_SOURCES) are what they were measured on
       "gemm_source_sha256_16": bench._gemm_source_hash(),
       "families": {}}
tot_r = tot_w = 0.0
for fam in sorted(set(fetch) | set(write)):
    r = fetch.get(fam, 0.0) * 1024.0 * 2.0 / steps
    w = write.get(fam, 0.0) * 1024.0 / steps
    res["families"][fam] = {"read": round(r), "written": round(w), "total": round(r + w)}
    tot_r += r
    tot_w += w
# the GEMM family by kernel instantiation and grid (one row per launch shape class), bytes per step
rows = []
for (name, grid), ent in by_kernel.items():
    rows.append({"kernel": name, "grid_threads": grid, "launches_per_step": round(ent["n"] / steps, 2),
                 "read": round(ent["FETCH_SIZE"] * 2048.0 / steps), "written": round(ent["WRITE_SIZE"] * 1024.0 / steps)})
res["gemm_by_launch_shape"] = sorted(rows, key=lambda r: -(r["read"] + r["written"]))[:24]
res["all_kernels"] = {"read": round(tot_r), "written": round(tot_w), "total": round(tot_r + tot_w)}
print(json.dumps(res, indent=1))
