#!/usr/bin/env python3
"""bench.py -- Mreads/s of the dropEst Estimation hot path (packed reads resident in HBM -> final count
matrix on the host) on N x MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): synthetic 10x v2 stream, 1e8 reads per GPU, 5 000 real cells per
GPU-share drawn from the 10x 737K whitelist, 16 bp CB + 10 bp UMI, 30 000 genes, no CB merge
(DummyMergeStrategy), default N-UMI merge, -L eEBA.  A "step" is one full pass: barcode table, first-seen
cell ids, UMI de-duplication (radix sort + segmented reduces), per-cell sizes, real/filtered cells, and
both count matrices copied to the host.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# (HSA_ENABLE_INTERRUPT is left at the ROCm default: the library's host waits poll the stream's completion signal themselves,
# csrc/util.h stream_wait -- a pass has about a dozen of them and a thread blocked on an interrupt takes 50-100 us, on the
# loaded hosts of the GPU boxes sometimes milliseconds, to run again.)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# Candidates for the dominant kernel, timed with HIP events inside the timed region: the partition / finishing-sort
# kernels of the splitter sort (dropest_amd/csrc/k_ssort.h), the scatter pass of the LSD sort (k_radix.h) and the barcode
# table build.  Algorithmic bytes per launch are the ones DESIGN.md §2 states per kernel (e.g. a scatter pass =
# 2 x (8 B key + value bytes) per record).
DOMINANT = "ss_scatter|ss_local|ss_hist|ss_compact|rs_scatter|cb_insert|build_keys"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=float, default=float(os.environ.get("DROPEST_BENCH_READS", 1e8)),
                    help="reads per GPU (C2: 1e8)")
    ap.add_argument("--cells", type=int, default=0, help="real cells per GPU-share (default: 5000 for c2, 50000 for c3)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"],
                    help="c2 (default, the metric's configuration): 10x v2, UMI 10, no CB merge; "
                         "c3: 10x v3, UMI 12, -m + whitelist merge (use --reads 1e9 for BASELINE's size); "
                         "c4: inDrop v3, split 8+8 barcode, UMI 8, -m + whitelist merge (BASELINE: 4 GPUs x 1.25e8 reads)")
    ap.add_argument("--poisson", action="store_true", help="c3 / c4: -M (PoissonRealBarcodesMergeStrategy) instead of -m, single GPU")
    ap.add_argument("--no-whitelist", action="store_true", help="c3 / c4: -m WITHOUT the barcode whitelist (SimpleMergeStrategy, single GPU)")
    ap.add_argument("--merge-umi", action="store_true", help="-u: directional UMI correction (single-GPU configs only)")
    ap.add_argument("--sharded", action="store_true",
                    help="N = 1 only: run the pass through the sharded runner (partition by owner + RCCL all-to-all to itself + "
                         "shared result buffer) instead of the plain context: what a second GPU would add, measured on one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the launch (barrier, broadcast of the RCCL id, max over ranks).  gloo + --one-device + "
                         "DROPEST_SHARD_DATAPLANE=shm: N processes on ONE GPU, the device data of the exchange through shared memory -- a "
                         "first-contact check of the multi-GPU command on a one-GPU box, not a measurement")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (with --backend gloo and DROPEST_SHARD_DATAPLANE=shm)")
    ap.add_argument("--dump-matrices", default="", help="rank 0: write cm / cm_raw of the last timed step (colptr, rows, values, column barcodes) to this .npz")
    ap.add_argument("--no-secondary", action="store_true", help="default run only: skip the C3 line at 1e9 reads (secondary.c3_1e9)")
    ap.add_argument("--push-sample", type=float, default=float(os.environ.get("DROPEST_BENCH_PUSH_SAMPLE", 5e7)),
                    help="reads pushed from pinned host memory for the PCIe-inclusive rates (c2, N=1; 0 disables)")
    ap.add_argument("--cpu-sample", type=float, default=float(os.environ.get("DROPEST_BENCH_CPU_SAMPLE", 6e6)),
                    help="reads of the same stream timed on the CPU oracle (rank 0, N=1 only; 0 disables)")
    return ap.parse_args()


def matrix_form():
    """What the step hands to the host (DROPEST_BENCH_MATRIX_FORM): "u32" (default: dropest_count_matrix_csc, the dgCMatrix slots i / x as
    32-bit arrays -- what ResultsPrinter::create_matrix fills, ResultsPrinter.cpp:433-442), "u32_direct" (the same arrays copied over PCIe as
    they are instead of as bytes widened on host threads), "u16", "bytes" (the wire forms alone: their consumer still has to decode)."""
    return os.environ.get("DROPEST_BENCH_MATRIX_FORM", "u32")


def one_step(ctx, form=None):
    """One pass of the hot path over the resident stream; returns what lands on the host: both count matrices in CSC form and the
    filtered cells.  The matrices arrive in the byte form (include/dropest_amd.h: dropest_matrix_bytes -- one byte of row delta and one
    byte of count per entry plus two exact lists; lossless, decoded by dropest_matrix_bytes_widen or by the reader itself: the facade's
    ResultsPrinter turns entries into doubles either way); form = "u16" / "u32" selects the 16-bit / 32-bit forms."""
    form = form or matrix_form()
    ctx.set_matrix_wire(form != "u32_direct")
    ctx.set_raw_matrix_prefetch({"u32": 0, "u32_direct": 0, "u16": 1, "bytes": 2}[form] if not os.environ.get("DROPEST_BENCH_NO_PREFETCH") else -1)
    ctx.reset_results()
    ctx.set_initialized()
    ctx.merge_and_filter()
    if form == "u16" and not ctx.narrow_matrix_possible():
        form = "u32"
    code = {"u32": 0, "u32_direct": 0, "u16": 1, "bytes": 2}[form]
    if not os.environ.get("DROPEST_BENCH_NO_PREFETCH"):
        ctx.prefetch_raw_matrix(form=code)    # cm_raw's copy to the host runs under the preparation of cm
    if form == "bytes":
        cm = ctx.count_matrix_csc_bytes(filtered=True)
        cm_raw = ctx.count_matrix_csc_bytes(filtered=False)
    elif form == "u16":
        cm = ctx.count_matrix_csc_narrow(filtered=True)
        cm_raw = ctx.count_matrix_csc_narrow(filtered=False)
    else:
        cm = ctx.count_matrix_csc(filtered=True)
        cm_raw = ctx.count_matrix_csc(filtered=False)
    # the metric names the merge targets beside the matrix: the (source, target) pairs of the CB merge are fetched inside the step
    # (host-resident already after merge_and_filter: two arrays of 8 bytes per merged cell, none at C2, ~2.4e6 pairs at C3)
    return cm, cm_raw, ctx.filtered_cells(), ctx.merge_target_pairs()


def dump_matrices(path, out, sharded):
    """cm / cm_raw of a step as plain arrays (the first-contact test compares an N-process run with the N = 1 run of the same stream)."""
    from dropest_amd.multi import widen_shard_matrix
    arrays = {}
    for name, m in (("cm", out[0]), ("cm_raw", out[1])):
        if sharded:
            colptr, rows, vals, bc = widen_shard_matrix(m)
            arrays.update({name + "_colptr": np.asarray(colptr, np.uint64), name + "_rows": np.array(rows, np.uint32), name + "_vals": np.array(vals, np.uint32),
                           name + "_barcodes": np.array(bc, np.uint64)})
        else:
            colptr, rows, vals = m[0], m[1], m[2]
            arrays.update({name + "_colptr": np.asarray(colptr, np.uint64), name + "_rows": np.array(rows, np.uint32), name + "_vals": np.array(vals, np.uint32)})
    np.savez(path, **arrays)


def nnz_of(m):
    return int(m.nnz) if hasattr(m, "nnz") else int(len(m[1]))


def form_text(cm, cm_raw):
    if hasattr(cm, "nnz"):
        return ("CSC in pinned host memory, byte form (dropest_matrix_bytes): u32 colptr + u8 row delta + u8 value + exact lists "
                "(rows listed: %d + %d, values listed: %d + %d of %d + %d entries)"
                % (cm.n_row_listed, cm_raw.n_row_listed, cm.n_value_listed, cm_raw.n_value_listed, cm.nnz, cm_raw.nnz))
    if len(cm) in (5, 6):    # (context: 5 arrays, sharded runner: 6 with the column barcodes)
        return ("CSC in pinned host memory, u32 colptr + u16 row index + u16 value + exact overflow list (%d + %d entries beyond 65534)"
                % (len(cm[-2]), len(cm_raw[-2])))
    return ("CSC in pinned host memory, u32 colptr + u32 row index + u32 value (the dgCMatrix slots i / x; sent over PCIe as bytes and widened "
            "by the library's host threads inside the step)")


def cpu_baseline(stream, n_sample, cfg, name="C2"):
    """The CPU oracle (oracle/dropest_oracle.cpp, a single-threaded restatement of the reference) timed on the
    first n_sample reads of the same stream: ingest + set_initialized + merge_and_filter + both matrices."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from oracle import Oracle
    cb, umi, gene, aux = parity.canonical_stream(*stream.generate_host(0, n_sample))
    m = cfg.get("merge")
    mkw = dict(merge_kind=1, barcodes_kind=m["barcodes_kind"], barcodes_file=m["barcodes_file"],
               min_merge_fraction=m.get("min_merge_fraction", 0.2)) if m else dict(merge_kind=0)
    o = Oracle(min_genes_before=cfg["min_before"], min_genes_after=cfg["min_after"], **mkw)
    t0 = time.perf_counter()
    o.add_packed(cb, umi, gene, aux)
    t1 = time.perf_counter()
    o.set_initialized(); o.merge_and_filter()
    o.count_matrix(filtered=True); o.count_matrix(filtered=False)
    t2 = time.perf_counter()
    return {"value": round(n_sample / (t2 - t0) / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the same synthetic %s stream%s; oracle/dropest_oracle.cpp single thread; "
                      "ingest %.2f s + finalize/matrices %.2f s; host has %d cores"
                      % (n_sample, name, " (with the whitelist CB merge)" if m else "", t1 - t0, t2 - t1, os.cpu_count() or 0)}


# kernel-stat name (dropest_kernel_stats) -> start of the kernel's name in a rocprofv3 trace
ROCPROF_NAME = {"cb_insert": "cb_insert_", "build_keys": "build_keys_kernel", "build_keys+L1": "build_keys_scatter_kernel", "ss_compact:cell_gene": "ss_compact_cg_kernel", "ss_local:keys": "ss_local_kernel<0",
                "ss_local:key+1B": "ss_local_kernel<1", "ss_scatter:L1:keys": ("ss_scatter_res_l1_kernel<0", "ss_scatter_l1_kernel<0"),
                "ss_scatter:L2:keys": ("ss_scatter_res_l2_kernel<0", "ss_scatter_l2_kernel<0"),
                "ss_scatter:L1:key+1B": ("ss_scatter_res_l1_kernel<1", "ss_scatter_l1_kernel<1"), "ss_scatter:L2:key+1B": ("ss_scatter_res_l2_kernel<1", "ss_scatter_l2_kernel<1"),
                "ss_hist:L1": "ss_hist_l1_kernel",
                "ss_hist:L2": "ss_hist_l2_kernel", "rs_scatter:keys": "rs_scatter_kernel_t<512, 8, false, 0, 8>",
                "rs_scatter:key+1B": "rs_scatter_kernel_t<512, 8, false, 1, 8>", "rs_scatter": "rs_scatter_kernel_t<512, 16, true, 4, 8>"}


def pmc_record(config, reads_per_gpu, sort):
    """The counter record (scripts/pmc_summary.py: profiles/pmc_pipeline*.json) taken on this workload, size and sort path, if there is one."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_pipeline*.json"))):
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        if rec.get("workload") == config and rec.get("reads_per_gpu") == reads_per_gpu and rec.get("sort") == sort:
            rec["file"] = "profiles/" + os.path.basename(path)
            return rec
    return None


def pmc_kernel_bytes(stat_name, config, reads_per_gpu, sort):
    rec, prefix = pmc_record(config, reads_per_gpu, sort), ROCPROF_NAME.get(stat_name)
    if not rec or not prefix:
        return None
    hits = [v["hbm_bytes_per_launch"] for k, v in rec["per_kernel"].items() if k.startswith(prefix)]   # (prefix: a string or a tuple of them)
    return max(hits) if hits else None          # several grid sizes: the main launch


def push_rates(stream, local_rank, n_push, cfg_kw):
    """The boundary's host path: n_push reads of the same stream in PINNED host arrays pushed through
    dropest_push_reads (8 Mi-read batches, copied from in place on the copy stream) into a fresh context, then one pass.
    Returns the push rate alone and the rate of push + pass -- the PCIe-inclusive figures; never `value`."""
    import ctypes as C
    from dropest_amd import capi
    L = capi.lib()
    dev = stream.generate_device(local_rank, first=0, n=n_push)
    host = dev.to_host()
    dev.free()
    for a in host:
        d = C.c_void_p()
        if L.dropest_host_register(local_rank, a.ctypes.data, a.nbytes, C.byref(d)) != 0:
            return None
    try:
        best_push, best_all = None, None
        for _ in range(3):
            ctx = capi.Context(device=local_rank, **cfg_kw)
            L.dropest_reserve_reads(ctx.h, n_push)
            t0 = time.perf_counter()
            B = 8 << 20
            for at in range(0, n_push, B):
                ctx.push_reads(*[a[at:at + B] for a in host])
            t1 = time.perf_counter()
            one_step(ctx)
            t2 = time.perf_counter()
            best_push = min(best_push or 1e9, t1 - t0); best_all = min(best_all or 1e9, t2 - t0)
            ctx.close()
        out = {"reads": n_push, "push_pinned_Mreads_per_s": round(n_push / best_push / 1e6, 1), "push_pinned_GB_per_s": round(n_push * 24 / best_push / 1e9, 1),
               "push_plus_pass_Mreads_per_s": round(n_push / best_all / 1e6, 1),
               "note": "pinned host arrays -> dropest_push_reads (8 Mi-read batches) -> set_initialized -> merge_and_filter -> both matrices; best of 3"}
        out.update(facade_add_record_rate())
        out.update(bam_ingest_rate())
        return out
    finally:
        for a in host:
            L.dropest_host_unregister(local_rank, a.ctypes.data)


def bam_ingest_rate(n=250_000, copies=48, threads=16):
    """BAM file -> container through the native reader (scripts/bench_bam_ingest.py: BGZF inflate, record boundaries, tag parsing, 2-bit
    packing, push): n synthetic 10x-style records written `copies` times into one file.  {} when the tool is not built or
    DROPEST_BENCH_NO_BAM is set."""
    import subprocess
    if os.environ.get("DROPEST_BENCH_NO_BAM") or not os.path.exists(os.path.join(ROOT, "tests", "cpp", "bam_to_counts")):
        return {}
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_bam_ingest.py"), str(n)], capture_output=True, text=True, timeout=240,
                             env=dict(os.environ, COPIES=str(copies), THREADS=str(threads)))
        best = None
        for line in res.stdout.splitlines():
            d = json.loads(line)
            if d.get("path") == "bulk":
                best = d
        if not best:
            return {}
        return {"bam_ingest_Mreads_per_s": best["ingest_mreads_per_s"], "bam_ingest_reads": best["reads"], "bam_ingest_threads": threads,
                "bam_ingest_file_MB": best["bam_mb"], "bam_ingest_note": "BAM -> container, native reader, %d host threads; the record-by-record path and other thread counts: scripts/bench_bam_ingest.py" % threads}
    except Exception:   # noqa: BLE001 (a missing tool or a timeout must not cost the bench line)
        return {}


def facade_add_record_rate(n=4_000_000):
    """The C++ facade's add_record(ReadInfo) -- strings in, what BamProcessor::save_read calls per read -- on one host thread
    (tests/cpp/add_record_rate.cpp, built next to the facade); {} when the tool is not built."""
    import subprocess
    tool = os.path.join(ROOT, "tests", "cpp", "add_record_rate")
    if not os.path.exists(tool):
        return {}
    try:
        res = subprocess.run([tool, str(n)], capture_output=True, text=True, timeout=120)
        d = json.loads(res.stdout.strip().splitlines()[-1])
        return {"facade_add_record_Mreads_per_s": d["add_record_Mreads_per_s"], "facade_add_record_reads": d["reads"]}
    except Exception:
        return {}


def measure(args, config, reads_per_gpu, cells, steps, warmup, world, rank, local_rank, dist, force_sharded, with_ingest):
    """Times `steps` passes of one workload; returns the bench line (rank 0) or None."""
    import torch
    from dropest_amd import capi
    from dropest_amd.synth import SynthStream

    total_reads = reads_per_gpu * world
    cfg = {"min_before": 20, "min_after": 100}    # configs/10x.xml:26-27
    c3, c4 = config == "c3", config == "c4"
    merge = c3 or c4
    if not cells:
        cells = 50000 if c3 else 5000
    wl_name = "indrop_v3" if c4 else "10x_aug_2016_split"
    wl = os.path.join(ROOT, "dropest_amd", "data", "barcodes", wl_name)
    if merge:
        cfg["merge"] = {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": wl, "min_merge_fraction": 0.2}
    stream = SynthStream(n_reads=total_reads, n_cells=cells * world, n_genes=30000, cb_len=16, whitelist=wl_name,
                         umi_len=12 if c3 else (8 if c4 else 10), stream_id={"c2": 2, "c3": 3, "c4": 4}[config])

    dev = run = None
    if world == 1 and not force_sharded:
        from dropest_amd.capi import Context
        dev = stream.generate_device(local_rank, first=0, n=reads_per_gpu)
        ukw = dict(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL) if args.merge_umi else {}
        if merge and args.no_whitelist:
            ctx = Context(device=local_rank, merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=2, min_merge_fraction=0.2,
                          min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"], **ukw)
        elif merge:
            ctx = Context(device=local_rank, merge_kind=capi.MERGE_POISSON_REAL if args.poisson else capi.MERGE_REAL_BARCODES,
                          barcodes_kind=capi.BARCODES_CONST,
                          barcodes_file=wl, min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"],
                          min_merge_fraction=0.2, **ukw)
        else:
            ctx = Context(device=local_rank, merge_kind=capi.MERGE_NONE, min_genes_before_merge=cfg["min_before"],
                          min_genes_after_merge=cfg["min_after"], **ukw)
        ctx.push_reads_device(*dev.ptrs, dev.n, adopt=True)
        step = lambda: one_step(ctx)   # noqa: E731
        get_stats = ctx.kernel_stats
        set_prof = ctx.set_profiling
        get_layout = ctx.sort_layout
    else:
        from dropest_amd.multi import ShardedRun
        ukw = dict(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL) if args.merge_umi else {}
        run = ShardedRun(stream, rank, world, local_rank, reads_per_gpu, cfg, dist, **ukw)
        if force_sharded:
            run.shard.set_option("force_exchange", 1)
        step = run.step
        get_stats = run.kernel_stats
        get_layout = run.ctx.sort_layout
        ctx = run.ctx
        def set_prof(on, only=None):
            run.set_profiling(on, only=only)
            # phase times: in the untimed table pass the device is synchronised at every phase boundary (a diagnostic);
            # inside the timed region the phases are host wall times only
            run.shard.set_option("trace", 1 if (on and only is None) else 0)
            run.shard.set_option("reset_phase_stats", 1)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    # inside the timed region only the dominant kernel carries HIP events (two events around every launch of a pass cost
    # ~0.5 ms per step); the table of all kernels and host stages comes from a separate pass after the clock has stopped
    set_prof(True, only=DOMINANT)
    fence()
    t0 = time.perf_counter()
    step_ms = []
    out = None
    for _ in range(steps):
        ts = time.perf_counter()
        out = step()
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 3))     # host clock; a step ends with its results on the host
    fence()
    elapsed = time.perf_counter() - t0
    stats = get_stats()
    table_steps = max(1, min(3, steps))
    set_prof(True)
    for _ in range(table_steps):
        step()
    fence()
    table = get_stats()
    shard_phases = run.shard.phase_stats() if (world > 1 or force_sharded) else {}
    sizes = ctx.table_sizes()
    set_prof(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0 and args.dump_matrices:
        dump_matrices(args.dump_matrices, out, world > 1 or force_sharded)

    line = None
    if rank == 0:
        ms_per_step = elapsed / max(1, steps) * 1e3
        value = total_reads / (elapsed / max(1, steps)) / 1e6
        # dominant kernel = the candidate with the largest share of the timed region
        # (only launches whose stat name STARTS with a candidate prefix carry events: the small sorts of the cell ids and of
        # the splitter sample, "cell_ids:..." / "ss_sample:...", do not)
        cands = {k: v for k, v in stats.items() if not k.startswith("host:") and v["launches"]}
        frac_of = lambda v: v["bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS    # noqa: E731  (algorithmic bytes over event time over the peak)
        dom_name, tied = "rs_scatter", []
        if cands:
            # the largest by time -- and when several sit within 3 % of it (C3: four kernels of ~11.6 ms), the one FURTHEST below its
            # roofline among them: the line must not pick the flattering one of a tie
            top_ms = max(v["ms"] for v in cands.values())
            tied = sorted(k for k, v in cands.items() if v["ms"] >= 0.97 * top_ms)
            dom_name = min(tied, key=lambda k: frac_of(cands[k]))
        dom = stats.get(dom_name, {"launches": 0, "ms": 0.0, "bytes": 0.0})
        roof = None
        if dom["launches"]:
            avg_ms = dom["ms"] / dom["launches"]
            achieved = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
            traffic = pmc_kernel_bytes(dom_name, config, reads_per_gpu, get_layout()["sort"])
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": None if traffic is None else "%s: rocprofv3 --pmc passes of this workload taken by "
                                      "scripts/refresh_profiles.sh (builder-run, not counters of this run)" % (pmc_record(config, reads_per_gpu, get_layout()["sort"]) or {}).get("file"),
                    "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                    "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"]}
            # the candidates next to it (same events, same region): at C2 three kernels of about 1 ms each take turns at the top from box to box
            roof["tied_within_3_percent"] = tied
            roof["candidates"] = {k: {"ms_per_step": round(v["ms"] / max(1, steps), 4),
                                      "frac": round(v["bytes"] / v["launches"] / (v["ms"] / v["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                  for k, v in sorted(cands.items(), key=lambda kv: -kv[1]["ms"])[:6] if v["ms"] > 0}
        # whole-pipeline roofline: compulsory traffic = every packed record read once, every output item
        # written once: 24 B/read + 24 B/molecule + 12 B/matrix entry + 40 B/cell, over the sum of the kernel times
        cm_nnz, raw_nnz = nnz_of(out[0]), nnz_of(out[1])
        compulsory = 24.0 * sizes["reads"] + 24.0 * sizes["molecules"] + 12.0 * (cm_nnz + raw_nnz) / max(1, world) + 40.0 * sizes["cells"]
        t_kernels_ms = sum(v["ms"] for k, v in table.items() if not k.startswith("host:")) / table_steps
        rec = pmc_record(config, reads_per_gpu, get_layout()["sort"])
        measured = rec["hbm_bytes_per_step"] if rec else None
        if roof is not None and t_kernels_ms > 0:
            # every kernel of the pass weighted by its time: sum of the algorithmic bytes DESIGN.md section 2 states per kernel over the sum of
            # the kernel times of the table pass (events on every launch), against the peak
            alg_all = sum(v["bytes"] for k, v in table.items() if not k.startswith("host:")) / table_steps
            roof["time_weighted_frac"] = round(alg_all / (t_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["time_weighted"] = {"algorithmic_bytes_per_step": alg_all, "kernel_ms_per_step": round(t_kernels_ms, 3),
                                     "what": "sum over all kernels of the pass of algorithmic bytes / sum of their event times / %g GB/s" % HBM_PEAK_GBS}
            roof["pipeline"] = {"compulsory_bytes": compulsory, "kernel_ms_per_step": round(t_kernels_ms, 3),
                                "achieved_GBps": round(compulsory / (t_kernels_ms * 1e-3) / 1e9, 1),
                                "achieved_frac": round(compulsory / (t_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "measured_bytes": measured,
                                "measured_hbm_frac": None if measured is None else round(measured / (t_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "amplification": None if measured is None else round(measured / compulsory, 2),
                                "bytes_per_read_compulsory": round(compulsory / max(1, sizes["reads"]), 1),
                                "bytes_per_read_measured": None if measured is None else round(measured / max(1, sizes["reads"]), 1),
                                "molecules": sizes["molecules"], "cells": sizes["cells"]}
        kernels = {k: {"ms_per_step": round(v["ms"] / table_steps, 4), "launches_per_step": v["launches"] / table_steps}
                   for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]) if not k.startswith("host:")}
        host_stages = {k[5:]: round(v["ms"] / table_steps, 3) for k, v in table.items() if k.startswith("host:")}
        exchange = None
        if world > 1 or force_sharded:
            host_stages.update({"shard:" + k: round(v["ms"] / max(1, v["steps"]), 3) for k, v in shard_phases.items()})
            a2a = shard_phases.get("all_to_all")
            if a2a and a2a["ms"] > 0:
                links = max(1, world - 1)      # xGMI is point-to-point: one link per peer
                rec = shard_phases.get("exchange_record_bytes", {}).get("bytes", 28.0)
                exchange = {"record_bytes": rec, "bytes_out_per_gpu_per_step": a2a["bytes"] / max(1, a2a["steps"]), "ms_per_step": round(a2a["ms"] / max(1, a2a["steps"]), 3),
                            "bytes_a_gpu_would_put_on_its_links_at_world_N": {str(N): reads_per_gpu * rec * (N - 1) / N for N in (2, 4, 8)},
                            "GB_per_s_per_gpu": round(a2a["bytes"] / a2a["ms"] / 1e6, 1), "GB_per_s_per_link": round(a2a["bytes"] / a2a["ms"] / 1e6 / links, 1),
                            "links": links, "transport": "RCCL grouped ncclSend/ncclRecv over xGMI" if world > 1 else "one GPU: no peer -- the block a shard keeps is unpacked where the partition left it (no copy), so nothing moves here; the links' share is the table above"}
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            cpu = cpu_baseline(stream, int(min(args.cpu_sample, total_reads)), cfg, config.upper())
        ingest = None
        if with_ingest and world == 1 and not force_sharded and args.push_sample > 0 and not merge:
            ingest = push_rates(stream, local_rank, int(min(args.push_sample, total_reads)),
                                dict(merge_kind=capi.MERGE_NONE, min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"]))
        cm = out[0]
        forms = None
        if world == 1 and not force_sharded and matrix_form() == "u32" and not os.environ.get("DROPEST_BENCH_NO_FORMS"):
            # the same step with the matrices leaving in the other forms (3 steps each, after the timed region): the wire forms alone (their
            # consumer still has to decode them) and the 32-bit arrays copied over PCIe as they are
            forms = {}
            for f in ("bytes", "u16", "u32_direct"):
                one_step(ctx, f)
                fence(); t0 = time.perf_counter()
                for _ in range(3):
                    one_step(ctx, f)
                fence(); forms[f + "_ms_per_step"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            forms["note"] = ("value / ms_per_step are measured with the 32-bit slots on the host (bytes on the wire, widened under the copy); "
                             "bytes / u16: the step ends with the wire form on the host, undecoded; u32_direct: 8 bytes per entry over PCIe")
            out = one_step(ctx, "u32")
            cm = out[0]
        line = {
            "metric": "Mreads/s processed to final count matrix", "value": round(value, 2), "unit": "Mreads/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("C3: synthetic 10x v3, %d reads/GPU, %d cells/GPU, 16bp CB + 12bp UMI, 30000 genes, "
                                    "-m + 10x whitelist (RealBarcodes merge), -L eEBA" if c3 else
                                    "C4: synthetic inDrop v3, %d reads/GPU, %d cells/GPU, 8+8bp split CB + 8bp UMI, 30000 genes, "
                                    "-m + inDrop v3 whitelist (RealBarcodes merge), -L eEBA" if c4 else
                                    "C2: synthetic 10x v2, %d reads/GPU, %d cells/GPU, 16bp CB + 10bp UMI, 30000 genes, "
                                    "no CB merge, -L eEBA") % (reads_per_gpu, cells) + (", -u" if args.merge_umi else "")
                                   + (", no whitelist (SimpleMergeStrategy)" if args.no_whitelist else "")
                                   + (", -M (Poisson decisions)" if args.poisson else ""),
                       "reads_total": total_reads, "parallelism": "cb-hash-shard x%d%s" % (world, " (sharded runner, forced exchange)" if force_sharded else ""),
                       "cm_nnz": nnz_of(cm), "cm_raw_nnz": raw_nnz, "filtered_cells": int(len(out[2])),
                       "merge_targets_fetched_per_step": int(len(out[3][0])) if len(out) > 3 and out[3] is not None else None, "sort_layout": get_layout(),
                       "matrix_form": form_text(cm, out[1]), "matrix_forms": forms},
            "roofline": roof, "cpu_baseline": cpu, "exchange": exchange, "host_ingest": ingest, "step_ms": step_ms, "kernels_ms_per_step": kernels,
            "kernel_table": "separate pass of %d steps after the timed region, events on every launch" % table_steps,
            "host_stage_wall_ms_per_step": host_stages,
        }
    # give the device memory back before another workload is measured
    out = None
    if run is not None:
        run.close()
    else:
        ctx.close()
    if dev is not None:
        dev.free()
    return line


def measure_bam_ingest(n_reads=250_000, copies=64, threads=16):
    """BAM file -> CellsDataContainer (tests/cpp/bam_to_counts: BamController + the facade), host reader against the device path, on one
    synthetic 10x-style BAM (CB / UB / GX tags, 98 bases; the records written `copies` times into one file); and the inflate kernel alone."""
    import ctypes as C
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bam_writer as bw
    from dropest_amd import capi
    from dropest_amd.build import build_facade
    from dropest_amd.synth import SynthStream
    build_facade()
    tool = os.path.join(ROOT, "tests", "cpp", "bam_to_counts")
    s = SynthStream(n_reads=n_reads, n_cells=500, n_genes=5000, umi_len=10)
    cb, umi, gene, aux = s.generate_host()
    cbs = {int(c): capi.unpack_code(c) for c in np.unique(cb)}
    recs, recs_uq = [], []
    for i in range(n_reads):
        tags = [("CB", "Z", cbs[int(cb[i])]), ("UB", "Z", capi.unpack_code(umi[i]))]
        if gene[i] != capi.NO_GENE:
            tags.append(("GX", "Z", "ENSG%011d" % gene[i]))
        recs.append(bw.record(int(aux[i]) & 0xFFFF, i, "A00000:1:HXXXX:1:1101:%d:%d" % (i, i), seq="ACGT" * 24 + "AC", tags=tags))
        recs_uq.append(bw.record(int(aux[i]) & 0xFFFF, i, "A00000:1:HXXXX:1:1101:%d:%d" % (i, i), seq="ACGT" * 24 + "AC", tags=tags + [("UQ", "Z", "FFFFFFFFFF")]))
    tmp = tempfile.mkdtemp()
    try:
        bam = os.path.join(tmp, "synth.bam")
        bw.write_bam(bam, [("chr%d" % i, 10_000_000) for i in range(25)], recs, repeat=copies)
        out = {"reads": n_reads * copies, "bam_MB": round(os.path.getsize(bam) / 1e6, 1), "host_threads": threads,
               "what": "ingest_ms of tests/cpp/bam_to_counts: BAM file (page cache) -> every accepted read in the container's device arrays"}
        for label, env in (("host_reader", {}), ("device", {"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TRACE": "1"})):
            best = None
            for _ in range(2):
                res = subprocess.run([tool, os.path.join(tmp, "out"), "filled", "20", "100", "-", str(threads), bam], capture_output=True, text=True,
                                     env=dict(os.environ, **env), timeout=600)
                if res.returncode:
                    raise RuntimeError(res.stderr[-300:])
                st = json.loads(res.stdout.strip().splitlines()[-1])
                if best is None or st["ingest_ms"] < best["ingest_ms"]:
                    best = st
                    trace = [ln for ln in res.stderr.splitlines() if "device path" in ln]
            assert best["saved"] == n_reads * copies
            out[label + "_ingest_ms"] = best["ingest_ms"]
            out[label + "_Mreads_per_s"] = round(n_reads * copies / best["ingest_ms"] / 1e3, 1)
            if label == "device" and trace:
                out["device_trace"] = trace[0][6:]
        L = capi.lib()
        L.dropest_bgzf_inflate_buffer.restype = C.c_int
        L.dropest_bgzf_inflate_buffer.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64,
                                                  C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int]
        blob = np.fromfile(bam, np.uint8)
        n_out, n_blocks, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        status = np.zeros(len(blob) // 26 + 1, np.uint32)
        if L.dropest_bgzf_inflate_buffer(0, blob.ctypes.data, len(blob), None, 1 << 62, C.byref(n_out), status.ctypes.data, len(status), C.byref(n_blocks), C.byref(ms), 3) == 0:
            out.update(inflate_kernel_ms=round(ms.value, 3), inflated_GB=round(n_out.value / 1e9, 3), inflate_GB_per_s=round(n_out.value / 1e6 / ms.value, 1),
                       blocks=int(n_blocks.value), blocks_refused=int((status[:n_blocks.value] != 0).sum()), crc32_checked_on_device=True)
        out["x_host_reader"] = round(out["host_reader_ingest_ms"] / out["device_ingest_ms"], 2)
        # the same records with a UMI quality tag (UQ: what a 10x BAM carries): both readers keep one quality row per read
        bam_uq = os.path.join(tmp, "synth_uq.bam")
        bw.write_bam(bam_uq, [("chr%d" % i, 10_000_000) for i in range(25)], recs_uq, repeat=copies)
        uq = {}
        for label, env in (("host_reader", {}), ("device", {"DROPEST_BAM_DEVICE": "1"})):
            res = subprocess.run([tool, os.path.join(tmp, "out"), "filled", "20", "100", "-", str(threads), bam_uq], capture_output=True, text=True,
                                 env=dict(os.environ, **env), timeout=600)
            if res.returncode:
                raise RuntimeError(res.stderr[-300:])
            st = json.loads(res.stdout.strip().splitlines()[-1])
            assert st["saved"] == n_reads * copies
            uq[label + "_ingest_ms"] = st["ingest_ms"]; uq[label + "_Mreads_per_s"] = round(n_reads * copies / st["ingest_ms"] / 1e3, 1)
        out["with_uq_tags"] = uq
        # ... and with bases drawn at random and binned qualities (75 % 'F', the rest ':' ',' '#'): such a file deflates ~3.2 x, as real 10x BAMs do,
        # not 10.8 x like the one above (one sequence, no qualities) -- most DEFLATE symbols are literals then, and the inflate kernel is bound by its
        # instructions per symbol (profiles/NOTES_r05.md, section 13)
        rng = np.random.default_rng(5)
        nib = rng.choice(np.array([1, 2, 4, 8], np.uint8), (n_reads, 98))
        packed_seq = ((nib[:, 0::2] << 4) | nib[:, 1::2]).astype(np.uint8)
        quals = rng.choice(np.array([37, 25, 11, 2], np.uint8), (n_reads, 98), p=[0.75, 0.12, 0.08, 0.05])
        recs_real = []
        for i, r in enumerate(recs):
            r = bytearray(r)
            o = 36 + r[12] + 4              # block_size + the fixed fields + the name (l_read_name, with its NUL) + one CIGAR operation
            r[o:o + 49] = packed_seq[i].tobytes(); r[o + 49:o + 147] = quals[i].tobytes()
            recs_real.append(bytes(r))
        bam_real = os.path.join(tmp, "synth_real.bam")
        real_copies = max(1, copies // 2)
        bw.write_bam(bam_real, [("chr%d" % i, 10_000_000) for i in range(25)], recs_real, repeat=real_copies)
        real = {"reads": n_reads * real_copies, "bam_MB": round(os.path.getsize(bam_real) / 1e6, 1)}
        for label, env in (("host_reader", {}), ("device", {"DROPEST_BAM_DEVICE": "1"})):
            res = subprocess.run([tool, os.path.join(tmp, "out"), "filled", "20", "100", "-", str(threads), bam_real], capture_output=True, text=True,
                                 env=dict(os.environ, **env), timeout=600)
            if res.returncode:
                raise RuntimeError(res.stderr[-300:])
            st = json.loads(res.stdout.strip().splitlines()[-1])
            assert st["saved"] == n_reads * real_copies
            real[label + "_ingest_ms"] = st["ingest_ms"]; real[label + "_Mreads_per_s"] = round(n_reads * real_copies / st["ingest_ms"] / 1e3, 1)
        blob = np.fromfile(bam_real, np.uint8)
        status = np.zeros(len(blob) // 26 + 1, np.uint32)
        if L.dropest_bgzf_inflate_buffer(0, blob.ctypes.data, len(blob), None, 1 << 62, C.byref(n_out), status.ctypes.data, len(status), C.byref(n_blocks), C.byref(ms), 3) == 0:
            real.update(inflate_kernel_ms=round(ms.value, 3), inflate_GB_per_s=round(n_out.value / 1e6 / ms.value, 1), deflate_ratio=round(n_out.value / len(blob), 2),
                        blocks_refused=int((status[:n_blocks.value] != 0).sum()))
        out["file_that_deflates_3x"] = real
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def finish_line(line):
    """The numbers that matter stand twice in the line: as scalars inside `config` (what a reader of the parsed record keeps) and as the
    LAST key, `summary` (what a reader of the line's tail keeps); the kernel / stage tables stand in between."""
    if line is None:
        return line
    sec = line.get("secondary") or {}
    c3, sh = sec.get("c3_1e9") or {}, sec.get("c2_sharded_runner") or {}
    summary = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "n_gpus": line["n_gpus"],
               "roofline_kernel": (line.get("roofline") or {}).get("kernel"), "roofline_frac": (line.get("roofline") or {}).get("frac"),
               "roofline_time_weighted_frac": (line.get("roofline") or {}).get("time_weighted_frac"),
               "kernel_ms_per_step": ((line.get("roofline") or {}).get("pipeline") or {}).get("kernel_ms_per_step"),
               "bytes_per_read_measured": ((line.get("roofline") or {}).get("pipeline") or {}).get("bytes_per_read_measured"),
               "merge_targets_fetched_per_step": line["config"].get("merge_targets_fetched_per_step")}
    if "value" in c3:
        summary.update(c3_1e9_value=c3["value"], c3_1e9_ms_per_step=c3["ms_per_step"],
                       c3_1e9_roofline_kernel=(c3.get("roofline") or {}).get("kernel"), c3_1e9_roofline_frac=(c3.get("roofline") or {}).get("frac"),
                       c3_1e9_roofline_time_weighted_frac=(c3.get("roofline") or {}).get("time_weighted_frac"),
                       c3_1e9_kernel_ms_per_step=((c3.get("roofline") or {}).get("pipeline") or {}).get("kernel_ms_per_step"),
                       c3_1e9_merge_targets=c3.get("config", {}).get("merge_targets_fetched_per_step"))
    elif "error" in c3:
        summary["c3_1e9_error"] = c3["error"][:200]
    if "value" in sh:
        summary.update(c2_sharded_value=sh["value"], c2_sharded_ms_per_step=sh["ms_per_step"], c2_sharded_x_plain=sh.get("x_plain"),
                       c2_sharded_x_plain_same_end_point=sh.get("x_plain_same_end_point"), c2_sharded_end_point=sh.get("end_point_short"))
    elif "error" in sh:
        summary["c2_sharded_error"] = sh["error"][:200]
    bi = sec.get("bam_ingest") or {}
    if "device_Mreads_per_s" in bi:
        summary.update(bam_device_Mreads_per_s=bi["device_Mreads_per_s"], bam_host_reader_Mreads_per_s=bi["host_reader_Mreads_per_s"], bam_inflate_GB_per_s=bi.get("inflate_GB_per_s"))
        if "with_uq_tags" in bi:
            summary.update(bam_uq_device_Mreads_per_s=bi["with_uq_tags"].get("device_Mreads_per_s"), bam_uq_host_reader_Mreads_per_s=bi["with_uq_tags"].get("host_reader_Mreads_per_s"))
        if "file_that_deflates_3x" in bi:
            r3 = bi["file_that_deflates_3x"]
            summary.update(bam_3x_device_Mreads_per_s=r3.get("device_Mreads_per_s"), bam_3x_host_reader_Mreads_per_s=r3.get("host_reader_Mreads_per_s"), bam_3x_inflate_GB_per_s=r3.get("inflate_GB_per_s"))
    elif "error" in bi:
        summary["bam_ingest_error"] = bi["error"][:200]
    for k, v in summary.items():           # scalars only: they survive in the parsed record's `config`
        if k not in ("value", "unit", "ms_per_step", "n_gpus") and v is not None:
            line["config"][k] = v
    line["summary"] = summary              # last key of the line
    return line


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    from dropest_amd import capi

    if not torch.cuda.is_available() or capi.lib().dropest_dev_count() < 1:
        raise SystemExit("bench.py needs a GPU: the dropEst hot path has no CPU implementation")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    force_sharded = world == 1 and (args.sharded or os.environ.get("DROPEST_BENCH_FORCE_SHARDED") == "1")
    saved_stdout = None
    if world > 1 or force_sharded:
        # RCCL prints a version banner on stdout when the first communicator comes up; stdout carries exactly one
        # JSON line, so everything before it goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    if world > 1:
        # torch.distributed carries the launch only: the barrier around the timed region and the broadcast of the RCCL
        # unique id; every collective of the pass itself is issued by the library (csrc/shard_run.h)
        import torch.distributed as dist
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rccl_came_up = False
    line = measure(args, args.config, int(args.reads), args.cells, args.steps, args.warmup, world, rank, local_rank, dist, force_sharded, True)
    # The largest single-GPU configuration of BASELINE.json (configs[2], "C3": 1e9 reads, 50 000 cells, whitelist CB merge) rides
    # along with the default run as `secondary.c3_1e9` -- same clock, same fences, 5 timed steps after 1 warm-up (24 GB of reads).
    if (line is not None and world == 1 and not force_sharded and args.config == "c2" and int(args.reads) == 100_000_000 and not args.no_secondary
            and not args.merge_umi):
        try:
            full_sample, args.cpu_sample = args.cpu_sample, min(args.cpu_sample, 1.5e6)   # (the oracle with the whitelist merge runs at ~0.12 Mreads/s: 12 s for 1.5e6 reads)
            try:
                sec = measure(args, "c3", 1_000_000_000, 50000, 5, 1, world, rank, local_rank, dist, False, False)
            finally:
                args.cpu_sample = full_sample
            line["secondary"] = {"c3_1e9": {k: sec[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "step_ms", "config", "roofline", "cpu_baseline",
                                                                "kernels_ms_per_step", "host_stage_wall_ms_per_step")}}
        except Exception as e:   # the primary line must not be lost to the secondary workload
            line["secondary"] = {"c3_1e9": {"error": "%s: %s" % (type(e).__name__, e)}}
        # ... and the primary workload once more through the sharded runner with the exchange forced (what `--sharded` runs: partition by
        # owner, RCCL all-to-all to itself, unpack, device-planned cm_raw, byte-form matrices into the shared buffer): what a second
        # GPU would add to every pass, under the same clock.  `x_plain` = its ms_per_step / the primary line's.
        try:
            sys.stdout.flush()
            keep = os.dup(1)
            os.dup2(2, 1)          # (RCCL's banner)
            cpu_sample, args.cpu_sample = args.cpu_sample, 0
            try:
                sh = measure(args, "c2", 100_000_000, args.cells, args.steps, args.warmup, world, rank, local_rank, dist, True, False)
            finally:
                args.cpu_sample = cpu_sample
                sys.stdout.flush()
                import ctypes
                ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in the C library's buffer: out with it while fd 1 is stderr
                os.dup2(keep, 1)
                os.close(keep)
                rccl_came_up = True
            plain_bytes = (line["config"].get("matrix_forms") or {}).get("bytes_ms_per_step")
            slots = "dgCMatrix slots" in sh["config"]["matrix_form"]      # (the default since round 5: the sharded step ends where the plain one does)
            same = line["ms_per_step"] if slots else plain_bytes
            line["secondary"]["c2_sharded_runner"] = dict({k: sh[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "step_ms", "exchange")},
                                                          x_plain=round(sh["ms_per_step"] / line["ms_per_step"], 3),
                                                          x_plain_same_end_point=None if not same else round(sh["ms_per_step"] / same, 3),
                                                          end_point_short="u32 slots" if slots else "bytes",
                                                          end_point=("the sharded step ends with the 32-bit dgCMatrix slots in the node-shared host buffer, like the plain step of the headline: every "
                                                                     "shard's columns cross its own PCIe link as bytes and are widened into the shared slots by its host threads under the copy -- "
                                                                     "x_plain_same_end_point = x_plain") if slots else
                                                                    ("the sharded step ends with the byte form in the node-shared host buffer; x_plain_same_end_point divides by the plain step "
                                                                     "that also ends at the byte form (config.matrix_forms.bytes_ms_per_step, 3 steps)"),
                                                          parallelism=sh["config"]["parallelism"], matrix_form=sh["config"]["matrix_form"],
                                                          phases_ms_per_step={k: v for k, v in sh["host_stage_wall_ms_per_step"].items() if k.startswith("shard:")})
        except Exception as e:
            line["secondary"]["c2_sharded_runner"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # What feeds the pass from a file: a synthetic 10x BAM through BamController into the container, by the host reader and by
    # the device path (BGZF inflate + record walk + tag parse on the GPU, include/dropest_bgzf.h), and the inflate kernel alone.
    if (line is not None and world == 1 and not force_sharded and args.config == "c2" and int(args.reads) == 100_000_000 and not args.no_secondary):
        try:
            line["secondary"]["bam_ingest"] = measure_bam_ingest()
        except Exception as e:
            line["secondary"]["bam_ingest"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(finish_line(line)), flush=True)
        if saved_stdout is not None or rccl_came_up:
            os.dup2(2, 1)          # whatever RCCL still prints while the communicator goes down belongs on stderr
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
