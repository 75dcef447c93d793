"""Turns two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE, collected separately as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<round>_pmc_summary.csv and the per-launch HBM
traffic records bench.py quotes (profiles/pmc_pipeline.json: every kernel of one step, and their sum).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc/fetch -o r -- python bench.py --steps 1 --warmup 0 --cpu-sample 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc/write -o r -- python bench.py --steps 1 --warmup 0 --cpu-sample 0
    python scripts/pmc_summary.py gpurun_out/pmc/fetch gpurun_out/pmc/write profiles r01b 100000000

Corrections (gfx950, calibrated in THIS repo on kernels with exactly known traffic): FETCH_SIZE is in KB and
under-reports coalesced 8 B/lane loads by 2x (rs_hist_kernel reads exactly 8 B x N); WRITE_SIZE (KB) is exact
(synth_kernel writes exactly 24 B x N).  The summary keeps the raw and corrected columns side by side.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("dropest::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)\(", name)
    name = m.group(1) if m else name.split("(")[0]
    return name.replace("(anonymous namespace)::", "")


def load(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + directory)
    # a kernel launched with several grid sizes (the radix passes also sort the few million barcode records) is split
    # by grid size: averaging a 1e8-record launch with a 4e6-record one says nothing about either
    by_grid = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = by_grid[short(row["Kernel_Name"])][int(row["Grid_Size"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    acc = {}
    for k, grids in by_grid.items():
        if len(grids) == 1:
            acc[k] = list(grids.values())[0]
        else:
            for g, a in grids.items():
                acc["%s [grid %d]" % (k, g)] = a
    return acc


def main():
    fetch_dir, write_dir, out_dir, tag, n_records = sys.argv[1:6]
    n_records = int(float(n_records))
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw)
        fr = f / nf if nf else 0.0
        wr = w / nw if nw else 0.0
        rows.append((k, n, fr, 2.0 * fr, wr, (2.0 * fr + wr) * 1024.0))
    path = os.path.join(out_dir, tag + "_pmc_summary.csv")
    with open(path, "w") as out:
        out.write("# rocprofv3 PMC summary (%s): FETCH_SIZE and WRITE_SIZE in separate passes over one bench.py step of %d reads.\n" % (tag, n_records))
        out.write("# Units: KB (1024 B) per launch, averaged over the launches of the kernel.  FETCH_KB_corrected = 2 x raw (gfx950;\n")
        out.write("# calibrated on rs_hist_kernel = 8 B x N reads), WRITE_KB is exact (calibrated on synth_kernel = 24 B x N writes).\n")
        out.write("kernel,launches,FETCH_KB_raw,FETCH_KB_corrected,WRITE_KB,hbm_bytes_per_launch\n")
        for r in rows:
            out.write('"%s",%d,%.1f,%.1f,%.1f,%.4g\n' % r)
    print("wrote", path)
    # calibration check (kernels whose traffic is known exactly) + the records bench.py quotes
    by = {r[0]: r for r in rows}
    for probe, exact in (("ss_hist_l1_kernel", 8.0 * n_records), ("rs_hist_kernel", 8.0 * n_records), ("cb_insert_hot_kernel", None)):
        if exact is None:
            continue
        hit = max((r for k, r in by.items() if k.startswith(probe)), key=lambda r: r[2], default=None)
        if hit:
            print("calibration: %s FETCH raw %.1f KB vs exact %.1f KB -> x%.3f" % (probe, hit[2], exact / 1024, exact / 1024 / hit[2]))
    # everything one step launched, generator kernels aside: the pipeline's measured HBM traffic
    skip = ("synth_kernel", "__amd_rocclr")
    per_kernel = {k: {"launches": r[1], "hbm_bytes_per_launch": r[5]} for k, r in by.items() if not k.startswith(skip)}
    copies = sum(r[1] * r[5] for k, r in by.items() if k.startswith("__amd_rocclr"))
    # bench.py --steps 1 --warmup 0 runs the pass twice (the timed step + the kernel-table step): kernels launched once per
    # pass tell how many passes the trace holds
    passes = max([v["launches"] for k, v in per_kernel.items() if k.startswith(("cb_insert_kernel", "cb_insert_hot_kernel", "build_keys_kernel", "build_keys_scatter_kernel"))] + [1])
    total = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in per_kernel.values()) / passes
    copies /= passes
    for v in per_kernel.values():
        v["launches_per_step"] = v.pop("launches") / passes
    sort = "splitter" if any(k.startswith("ss_local_kernel") for k in by) else "lsd"
    rec = {"workload": os.environ.get("DROPEST_PMC_WORKLOAD", "c2"), "reads_per_gpu": n_records, "sort": sort, "tag": tag,
           "hbm_bytes_per_step": total, "runtime_copies_and_fills_bytes": copies, "per_kernel": per_kernel,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over ONE bench.py step (--steps 1 --warmup 0); "
                     "bytes = 1024 x (2 x FETCH_SIZE + WRITE_SIZE): FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section; checked on "
                     "ss_hist_l1 = 8 B x N reads), WRITE_SIZE exact (checked on synth_kernel = 24 B x N writes); "
                     "see profiles/%s_pmc_summary.csv" % tag}
    rec["passes_in_trace"] = passes
    with open(os.path.join(out_dir, os.environ.get("DROPEST_PMC_FILE", "pmc_pipeline.json")), "w") as out:
        json.dump(rec, out, indent=1)
    print("pipeline: %.3f GB per step over %d kernels (%s sort, %d passes in the trace); runtime copies / fills %.3f GB per step" % (total / 1e9, len(per_kernel), sort, passes, copies / 1e9))


if __name__ == "__main__":
    main()
